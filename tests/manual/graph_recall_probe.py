"""Recall-vs-ef of the graph routes on the benchmark's data (i.i.d. U[0,1)): SURVEY 7's control (N = 20k, d = 768, M = 32) and a
larger shard; directed exact-kNN lists entered at node 0 against the bulk graph (heuristic + reverse links, cap 2 M) entered by
descent.  Prints recall@10, distance evaluations per query and the fraction of N they touch, QPS (batched, resident queries).
usage: python tests/manual/graph_recall_probe.py [N] [d] [M] [efc] [intrinsic_rank] [bf16rows]
intrinsic_rank > 0: rows = Z A + 0.01 noise with Z ~ U[0,1)^(N x rank), A a fixed rank x d matrix -- data that HAS neighbourhood
structure (what embeddings look like), unlike the benchmark's i.i.d. uniform coordinates where distances concentrate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, zvdb_amd as zv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
M = int(sys.argv[3]) if len(sys.argv) > 3 else 32
efc = int(sys.argv[4]) if len(sys.argv) > 4 else 96
rank = int(sys.argv[5]) if len(sys.argv) > 5 else 0
hrows = len(sys.argv) > 6 and sys.argv[6] == "bf16rows"  # ZVDB_FLAG_BF16_ROWS: the index keeps bf16 rows only (BASELINE configs 4-5)
nq, k = 1024, int(os.environ.get("ZVDB_PROBE_K", "10"))  # (k = 100 + ZVDB_PROBE_METRIC=cosine: BASELINE configs[4]'s shape)
metric = {"l2": zv.METRIC_L2SQ, "dot": zv.METRIC_DOT, "cosine": zv.METRIC_COSINE}[os.environ.get("ZVDB_PROBE_METRIC", "l2")]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
idx = zv.Index(dim=d, capacity=N, m=2 * M, graph_stride=2 * M, metric=metric, flags=zv.FLAG_BF16_ROWS if hrows else zv.FLAG_BF16_MIRROR)
q = torch.empty((nq, d), dtype=torch.float32, device=dev)
if rank > 0:
    g = torch.Generator(device=dev); g.manual_seed(5)
    A = torch.rand((rank, d), generator=g, device=dev) - 0.5
    for r0 in range(0, N, 200_000):
        n = min(200_000, N - r0)
        rows = torch.rand((n, rank), generator=g, device=dev) @ A + 0.01 * torch.randn((n, d), generator=g, device=dev)
        torch.cuda.synchronize()
        idx.add_dev(rows.contiguous().data_ptr(), n, d, stream)
        torch.cuda.synchronize()
    q.copy_(torch.rand((nq, rank), generator=g, device=dev) @ A + 0.01 * torch.randn((nq, d), generator=g, device=dev))
    torch.cuda.synchronize()
else:
    idx.add_synthetic(N, d, 0x5EED)
    zv.fill_synthetic_dev(q.data_ptr(), nq, d, 0xC0FFEE, 0, stream)
Q = q.cpu().numpy()
gt, _, _ = idx.search_bruteforce(Q, k, zv.BF_I8)
ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
st = torch.zeros(4, dtype=torch.int64, device=dev)


def sweep(tag, entry):
    for ef in (32, 64, 128, 256, 512, 1024):
        if ef < k: continue
        st.zero_()
        try:
            idx.search_graph_dev(q.data_ptr(), nq, d, k, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), stream)
            torch.cuda.synchronize()
        except zv.ZvdbError as e:
            print(f"  {tag} ef={ef}: {e}")
            continue
        if int(st[2].item()):
            print(f"  {tag} ef={ef}: visited set overflowed for {int(st[2].item())} queries (LDS)")
            continue
        got = ids.cpu().numpy()
        rec = float(np.mean([len(set(got[i]) & set(gt[i])) / k for i in range(nq)]))
        ev = int(st[0].item()) / nq
        t0 = time.perf_counter()
        for _ in range(3):
            idx.search_graph_dev(q.data_ptr(), nq, d, k, ef, entry, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 3
        print(f"  {tag} ef={ef:5d}: recall@{k} {rec:.4f}  evals/query {ev:9.1f} ({ev / N * 100:6.2f} % of N)  {nq / el:10.0f} QPS  "
              f"{ev * d * (2 if hrows else 4) * nq / el / 1e12:.2f} TB/s gathered", flush=True)


print(f"N={N} d={d} M={M} k={k} metric={os.environ.get('ZVDB_PROBE_METRIC', 'l2')} ef_construction={efc} intrinsic_rank={rank or 'none (i.i.d. uniform coordinates)'}{' bf16 rows' if hrows else ''}")
only = os.environ.get("ZVDB_PROBE_ONLY", "")  # "bulk": skip the plain kNN-list graphs (the 10M-row run)
if only != "bulk":
    t0 = time.perf_counter(); idx.build_knn_graph(M, zv.BF_I8); t1 = time.perf_counter()
    print(f" directed exact {M}-NN lists, entry node 0 (build {t1 - t0:.2f} s)")
    sweep("knn(M)   ", 0)
    t0 = time.perf_counter(); idx.build_knn_graph(2 * M, zv.BF_I8); t1 = time.perf_counter()
    print(f" directed exact {2 * M}-NN lists, entry node 0 (build {t1 - t0:.2f} s)")
    sweep("knn(2M)  ", 0)
bmode = {"i8": zv.BF_I8, "fp4": zv.BF_FP4, "bf16": zv.BF_BF16, "f32": zv.BF_F32}[os.environ.get("ZVDB_PROBE_BUILD", "i8")]  # screen of the build's kNN pass
t0 = time.perf_counter(); idx.build_graph(M, efc, bmode, 7); t1 = time.perf_counter()
deg = np.array([len(idx.get_neighbors(i)) for i in range(0, N, max(1, N // 2000))])
e, ml = idx.graph_info()
print(f" bulk graph M={M} (cap {2 * M}), build {t1 - t0:.2f} s = {N / (t1 - t0):.0f} rows/s, mean layer-0 degree {deg.mean():.1f}, max level {ml}, entry {e}")
sweep("bulk/desc", zv.ENTRY_DESCEND)
if only != "bulk":
    sweep("bulk/top ", e)
for w in [int(x) for x in os.environ.get("ZVDB_PROBE_WIDTHS", "").split(",") if x]:  # search widths (ZVDB_OPT_EF_SEARCH_WIDTH) on the bulk graph
    idx.set_option(zv.OPT_EF_SEARCH_WIDTH, w)
    sweep(f"bulk/desc width {w}", zv.ENTRY_DESCEND)
