"""Workload for the rocprofv3 passes of the graph kernels (tools/prof_graph.sh): k_search_ef at 1M x 768, M = 16, ef = 128, batch 1024
on fp32 rows and on bf16 rows (ZVDB_FLAG_BF16_ROWS); k_search_compat_fast at 100k x 128 with 200k resident queries.  Prints the
in-kernel counters (E distance evaluations, P pops) and the algorithmic gather bytes E d sizeof(elt) + P M0 4 per launch."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, zvdb_amd as zv

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
out = {}


def ef_leg(tag, flags, N=int(os.environ.get("ZVDB_PROFILE_ROWS", "1000000")), d=768, M=16, ef=128, nq=1024, k=10):
    print("building", tag, N, flush=True)
    idx = zv.Index(dim=d, capacity=N, m=M, flags=flags)
    idx.add_synthetic(N, d, 0x5EED)
    idx.build_knn_graph(M, zv.BF_I8)
    width = int(os.environ.get("ZVDB_PROFILE_WIDTH", "1"))  # (a variable of THIS script: the search width it sets through the ABI)
    idx.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev)
    zv.fill_synthetic_dev(q.data_ptr(), nq, d, 0xC0FFEE, 0, stream)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    idx.search_graph_dev(q.data_ptr(), nq, d, k, ef, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), stream)
    torch.cuda.synchronize()
    E, P = int(st[0].item()), int(st[1].item())
    t0 = time.perf_counter()
    for _ in range(5):
        idx.search_graph_dev(q.data_ptr(), nq, d, k, ef, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 5
    rb, _ = idx.storage_info()
    elt = rb // d
    alg = E * d * elt + P * M * 4
    out[tag] = {"rows": N, "dim": d, "M": M, "ef": ef, "search_width": width, "batch": nq, "row_bytes": rb, "E": E, "P": P, "algorithmic_bytes_per_launch": alg,
                "ms_per_launch_wall": el * 1e3, "algorithmic_TBps_wall": alg / el / 1e12, "qps": nq / el}
    print(tag, json.dumps(out[tag]), flush=True)


def compat_leg(P=100_000, d=128, NR=200_000, k=10):
    import oracle  # noqa: F401  (not used: the index is built by the product path)
    h = zv.HNSW(16, 200)
    V = np.random.default_rng(1).random((P, d), dtype=np.float32)
    h.insert_many(V, np.zeros(P, dtype=np.int32))
    idx = h.index
    q = torch.empty((NR, d), dtype=torch.float32, device=dev)
    zv.fill_synthetic_dev(q.data_ptr(), NR, d, 0xC0FFEE, 0, stream)
    ids = torch.empty((NR, k), dtype=torch.int64, device=dev)
    dd = torch.empty((NR, k), dtype=torch.float32, device=dev)
    cnt = torch.empty((NR,), dtype=torch.int32, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    idx.search_graph_dev(q.data_ptr(), NR, d, k, 0, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), stream)
    torch.cuda.synchronize()
    E, Pp = int(st[0].item()), int(st[1].item())
    t0 = time.perf_counter()
    for _ in range(5):
        idx.search_graph_dev(q.data_ptr(), NR, d, k, 0, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 5
    alg = E * d * 4 + Pp * 16 * 4
    out["compat_100kx128"] = {"points": P, "dim": d, "queries": NR, "k": k, "E": E, "P": Pp, "algorithmic_bytes_per_launch": alg,
                              "ms_per_launch_wall": el * 1e3, "algorithmic_TBps_wall": alg / el / 1e12, "qps": NR / el}
    print("compat_100kx128", json.dumps(out["compat_100kx128"]), flush=True)


legs = sys.argv[1:] or ["f32", "bf16", "compat"]
if "f32" in legs:
    ef_leg("ef128_1Mx768_f32_rows", 0)
if "bf16" in legs:
    ef_leg("ef128_1Mx768_bf16_rows", zv.FLAG_BF16_ROWS)
if "batch" in legs:  # the same search at larger batches (more waves per CU): where the bytes per row start to matter
    for nq in (4096, 16384):
        ef_leg(f"ef128_1Mx768_f32_rows_batch{nq}", 0, nq=nq)
        ef_leg(f"ef128_1Mx768_bf16_rows_batch{nq}", zv.FLAG_BF16_ROWS, nq=nq)
if "compat" in legs:
    compat_leg()
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "graph_profile_counters_%s_w%s.json" % ("_".join(legs), os.environ.get("ZVDB_PROFILE_WIDTH", "1"))), "w"), indent=1)
