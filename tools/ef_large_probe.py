# the large-ef end of the ef-mode beam search (ef = 512 / 1,024: register beam of 8 / 16 blocks, visited table of 40-80 KB => 1-2
# queries per CU): the library's choice of waves per query ("auto") with and without the packed visited table, and 2 / 4 / 8 forced, at search width 4 -- kernel time, E / P counters, algorithmic gather rate.
# usage: python tools/ef_large_probe.py [rows] [dim] [ef ...]
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
EFS = [int(x) for x in sys.argv[3:]] or [512, 1024]
B, K, M = 1024, 10, 16
LEGS = os.environ.get("ZVDB_PROBE_LEGS", "uniform_knn16_bf16rows,uniform_knn16_f32,rank12_bulk_bf16rows").split(",")
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
out = {}
for tag, structured, hb, bulk in (("uniform_knn16_f32", False, False, False), ("uniform_knn16_bf16rows", False, True, False),
                                  ("rank12_bulk_f32", True, False, True), ("rank12_bulk_bf16rows", True, True, True)):
    if tag not in LEGS: continue
    stride = 2 * M if bulk else M
    g = zv.Index(dim=D, capacity=N, m=stride, graph_stride=stride, flags=zv.FLAG_BF16_MIRROR | (zv.FLAG_BF16_ROWS if hb else 0))
    q = bench.make_rows(zv, torch, g, N, D, structured, dev, s, B)
    if bulk: g.build_graph(M, 64, zv.BF_I8, 7)
    else: g.build_knn_graph(M, zv.BF_I8)
    entry = zv.ENTRY_DESCEND if bulk else 0
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    elt = 2 if hb else 4
    for ef in EFS:
        ref = None
        combos = ((1, None, None), (4, None, None), (4, None, "0"), (4, "2", None), (4, "4", None), (4, "8", None))
        if os.environ.get("ZVDB_PROBE_COMBOS") == "auto": combos = ((4, None, None),)  # (the rocprofv3 pass: one kernel per ef)
        for width, wpq, packed in combos:
            g.set_option(zv.OPT_EF_SEARCH_WIDTH, width); zv.set_tuning("ZVDB_EF_WPQ", wpq); zv.set_tuning("ZVDB_EF_PACKED_VISITED", packed)
            for _ in range(3): g.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
            st.zero_(); g.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
            torch.cuda.synchronize()
            E, P, X = int(st[0].item()), int(st[1].item()), int(st[2].item())
            got = (ids.cpu().numpy().copy(), dd.cpu().numpy().copy())
            same = None
            if width == 4:
                if ref is None: ref = got
                else: same = bool((ref[0] == got[0]).all() and (ref[1] == got[1]).all())
            g.timing_enable(True); t0 = time.perf_counter()
            for _ in range(6): g.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 6
            kms, kn = g.timing_read(); g.timing_enable(False)
            nbytes = E * D * elt + P * stride * 4
            r = {"ef": ef, "width": width, "waves_per_query": int(wpq) if wpq else "auto", "packed_visited": packed != "0", "qps": B / el, "ms_per_batch": el * 1e3, "kernel_ms": kms / kn, "E_per_query": E / B, "P_per_query": P / B,
                 "exhausted": X, "same_as_2_waves": same, "algorithmic_TBps": nbytes / (kms / kn * 1e-3) / 1e12, "hbm_frac": nbytes / (kms / kn * 1e-3) / 8e12}
            out.setdefault(tag, []).append(r)
            print(tag, json.dumps(r), flush=True)
    zv.set_tuning("ZVDB_EF_WPQ", None); zv.set_tuning("ZVDB_EF_PACKED_VISITED", None)
    del g
json.dump({"rows": N, "dim": D, "batch": B, "k": K, "M": M, "legs": out}, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"ef_large_probe_{N}x{D}.json"), "w"), indent=1)
