# soak: many different query batches through the ef-mode beam search (register beam, several waves per query, two-pass visited
# sizing, descent entry, search widths 1 / 2 / 4, the large beams beyond ef = 256) on fp32 and bf16 rows; every batch checked (k results, ascending distances), sampled batches compared
# with the LDS-beam kernel (ZVDB_EF_BEAM=lds) bit for bit.  usage: python tools/soak_graph.py [batches]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
N, D, B, K = 500_000, 128, 1024, 10
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
bad = 0
t0 = time.time()
for flags, tag in ((zv.FLAG_BF16_MIRROR, "fp32 rows"), (zv.FLAG_BF16_ROWS, "bf16 rows")):
    idx = zv.Index(dim=D, capacity=N, m=32, graph_stride=32, flags=flags)
    idx.add_synthetic(N, D, 0x5EED)
    idx.build_graph(16, 64, zv.BF_I8, 7)
    q = torch.empty((B, D), dtype=torch.float32, device=dev)
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev); st = torch.zeros(4, dtype=torch.int64, device=dev)
    rng = np.random.default_rng(3)
    for b in range(NB):
        zv.fill_synthetic_dev(q.data_ptr(), B, D, 0xC0FFEE + 7919 * b, 0, s)
        ef = int(rng.choice([16, 32, 64, 100, 128, 200, 256, 400, 512, 600, 1000]))
        width = int(rng.choice([1, 1, 2, 4]))  # search width; beyond ef = 256 the large-beam path (gather merge, packed visited set, wide workgroups)
        idx.set_option(zv.OPT_EF_SEARCH_WIDTH, width)
        entry = zv.ENTRY_DESCEND if b % 2 else 0
        st.zero_()
        idx.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
        torch.cuda.synchronize()
        i1, d1, c1 = ids.cpu().numpy().copy(), dd.cpu().numpy().copy(), cnt.cpu().numpy().copy()
        ok = bool(np.all(c1 == K) and np.all(np.diff(d1, axis=1) >= 0) and int(st[2].item()) == 0)
        if b % 10 == 0:
            zv.set_tuning("ZVDB_EF_BEAM", "lds")
            idx.search_graph_dev(q.data_ptr(), B, D, K, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
            torch.cuda.synchronize()
            zv.set_tuning("ZVDB_EF_BEAM", None)
            ok = ok and np.array_equal(i1, ids.cpu().numpy()) and np.array_equal(d1, dd.cpu().numpy())
        if not ok:
            bad += 1
            print(f"{tag} batch {b} ef {ef} width {width}: NOT ok", flush=True)
        if b % 100 == 99: print(f"{tag}: {b + 1} batches, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done:", 2 * NB, "batches, bad", bad)
