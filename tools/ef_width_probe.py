# search width of the ef-mode beam search (ZVDB_OPT_EF_SEARCH_WIDTH; spec: orc_hnsw_search_ef_w): kernel time, E / P counters,
# algorithmic gather rate (E d sizeof(elt) + P M0 4 bytes per query over the search kernel's launch time) and recall@k at widths
# 1 / 2 / 4 -- the workload of profiles/r3/graph_kernels_counters.json (1M x 768 uniform rows, exact 16-NN lists, ef = 128, batch
# 1,024, entered at node 0) on fp32 and bf16 rows, and the bulk HNSW graph on structured rows entered by descent.
# usage: python tools/ef_width_probe.py [rows] [dim] [ef ...]
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
EFS = [int(x) for x in sys.argv[3:]] or [128]
B, K, M = 1024, 10, 16
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
out = {}
for tag, structured, hb, bulk in (("uniform_knn16_f32", False, False, False), ("uniform_knn16_bf16rows", False, True, False),
                                  ("rank12_bulk_f32", True, False, True), ("rank12_bulk_bf16rows", True, True, True)):
    stride = 2 * M if bulk else M
    g = zv.Index(dim=D, capacity=N, m=stride, graph_stride=stride, flags=zv.FLAG_BF16_MIRROR | (zv.FLAG_BF16_ROWS if hb else 0))
    q = bench.make_rows(zv, torch, g, N, D, structured, dev, s, B)
    if bulk: g.build_graph(M, 64, zv.BF_I8, 7)
    else: g.build_knn_graph(M, zv.BF_I8)
    entry = zv.ENTRY_DESCEND if bulk else 0
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    wsb = g.bruteforce_workspace(B, K, zv.BF_I8); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    for _ in range(3): g.search_bruteforce_dev(q.data_ptr(), B, D, K, zv.BF_I8, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
    torch.cuda.synchronize(); gt = ids.cpu().numpy().copy(); del ws
    elt = 2 if hb else 4
    for ef in EFS:
        for width in (1, 2, 4):
            g.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
            for _ in range(3): g.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
            st.zero_(); g.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
            torch.cuda.synchronize()
            got = ids.cpu().numpy(); rec = float(np.mean([len(set(got[i]) & set(gt[i])) / K for i in range(B)]))
            E, P = int(st[0].item()), int(st[1].item())
            g.timing_enable(True); t0 = time.perf_counter()
            for _ in range(8): g.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
            kms, kn = g.timing_read(); g.timing_enable(False)
            nbytes = E * D * elt + P * stride * 4
            r = {"ef": ef, "width": width, "qps": B / el, "ms_per_batch": el * 1e3, "kernel_ms": kms / kn, "recall_at_10": rec, "E_per_query": E / B, "P_per_query": P / B,
                 "algorithmic_TBps": nbytes / (kms / kn * 1e-3) / 1e12, "hbm_frac": nbytes / (kms / kn * 1e-3) / 8e12}
            out.setdefault(tag, []).append(r)
            print(tag, json.dumps(r), flush=True)
    del g
json.dump({"rows": N, "dim": D, "batch": B, "k": K, "M": M, "legs": out}, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"ef_width_probe_{N}x{D}.json"), "w"), indent=1)
