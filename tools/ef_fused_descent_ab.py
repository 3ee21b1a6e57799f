# the entry descent inside the first pass of the register-beam kernel against the two-launch form (ZVDB_EF_FUSE_DESCENT=0): wall time per
# 1,024-query batch on the bulk graph, results and counters compared.  usage: python tools/ef_fused_descent_ab.py
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, zvdb_amd as zv, bench
N, D, B, K, M = 1_000_000, 768, 1024, 10, 16
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
for hb in (True, False):
    g = zv.Index(dim=D, capacity=N, m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_MIRROR | (zv.FLAG_BF16_ROWS if hb else 0))
    q = bench.make_rows(zv, torch, g, N, D, True, dev, s, B)
    g.build_graph(M, 64, zv.BF_I8, 7)
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    for ef in (32, 64, 128, 256, 512):
        for width in (1, 4):
            g.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
            res = {}
            for rep in range(2):
                for fuse in ("1", "0"):
                    zv.set_tuning("ZVDB_EF_FUSE_DESCENT", fuse)
                    for _ in range(3): g.search_graph_dev(q.data_ptr(), B, D, K, ef, zv.ENTRY_DESCEND, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
                    st.zero_(); g.search_graph_dev(q.data_ptr(), B, D, K, ef, zv.ENTRY_DESCEND, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
                    torch.cuda.synchronize(); sig = (int(ids.sum().item()), float(dd.sum().item()), int(st[0].item()), int(st[1].item()))
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(10): g.search_graph_dev(q.data_ptr(), B, D, K, ef, zv.ENTRY_DESCEND, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
                    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10 * 1e3
                    res.setdefault(fuse, []).append((el, sig))
            same = res["1"][0][1] == res["0"][0][1]
            print(f"{'bf16' if hb else 'f32'} rows ef {ef} width {width}: fused {min(x[0] for x in res['1']):.3f} ms, two launches {min(x[0] for x in res['0']):.3f} ms, same results+counters {same}", flush=True)
    zv.set_tuning("ZVDB_EF_FUSE_DESCENT", None)
    del g
