"""Reference-exact bulk insert (zvdb_hip_insert_batch): speculative parallel rounds against the one-wave in-order kernel
(ZVDB_INSERT_SPEC=0) -- throughput of both and equality of the graphs they leave on every layer.
usage: python tests/manual/insert_spec_probe.py [N] [d] [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, zvdb_amd as zv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = int(sys.argv[3]) if len(sys.argv) > 3 else 16
V = np.random.default_rng(1).random((N, d), dtype=np.float32)
levels = np.random.default_rng(2).geometric(0.5, N).astype(np.int32) - 1
levels = np.minimum(levels, 31)
graphs = {}
for tag, env in (("speculative rounds", None), ("one wave, in order", "0")):
    if env is None: zv.set_tuning("ZVDB_INSERT_SPEC", None)
    else: zv.set_tuning("ZVDB_INSERT_SPEC", env)
    h = zv.HNSW(m, 200)
    t0 = time.perf_counter()
    h.insert_many(V, levels)
    el = time.perf_counter() - t0
    print(f"{tag:20s}: {N} x {d}, m = {m}: {el:.3f} s = {N / el:,.0f} points/s", flush=True)
    ml = h.max_level
    g = []
    step = max(1, N // 20000)
    for i in range(0, N, step):
        for L in range(0, min(int(levels[i]), ml) + 1):
            g.append((i, L, tuple(int(x) for x in h.index.get_neighbors_layer(i, L))))
    graphs[tag] = g
zv.set_tuning("ZVDB_INSERT_SPEC", None)
a, b = graphs["speculative rounds"], graphs["one wave, in order"]
print("graphs equal on the sampled nodes, every layer:", a == b, f"({len(a)} lists compared)")
