"""Graph-route probe of round 5 (one process = one build of libzvdb_hip.so; tools/ef_ab.sh runs it in turn with several builds on one box):
 leg A  1M x 768 rows of intrinsic rank 12, bf16 rows, bulk HNSW graph (M = 16), by descent, width 4: ef 128 / 256 / 512 at batch 1,024; ef 128 at batch 4,096
 leg B  2M x 1536 rows, cosine, bf16 rows, k = 100, ef = 256 (the shape of one shard of BASELINE configs[4], fewer rows)
Prints one line per case: kernel ms (HIP events inside the library), QPS, algorithmic TB/s (E d sizeof(elt) + P M0 4 bytes per query)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import zvdb_amd as zv

dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
tag = sys.argv[1] if len(sys.argv) > 1 else "tree"
for kv in os.environ.get("EF_PROBE_TUNING", "").split(","):   # e.g. EF_PROBE_TUNING=ZVDB_EF_WPQ=3 (library switches go through the ABI)
    if "=" in kv:
        zv.set_tuning(*kv.split("=", 1))
        tag += "/" + kv.split("=", 1)[1]
legs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["A", "B"]
M = 16


ENTRY = 0 if os.environ.get("EF_PROBE_ENTRY") == "fixed" else zv.ENTRY_DESCEND   # EF_PROBE_ENTRY=fixed: every search starts at the graph's entry node (no descent)


def run(g, q, B, D, K, ef, elt, reps=8):
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    st = torch.zeros(4, dtype=torch.int64, device=dev)
    g.search_graph_reserve(B, K, ef)
    for _ in range(3):
        g.search_graph_dev(q.data_ptr(), B, D, K, ef, ENTRY, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
    st.zero_()
    g.search_graph_dev(q.data_ptr(), B, D, K, ef, ENTRY, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
    torch.cuda.synchronize()
    E, P = int(st[0].item()), int(st[1].item())
    best = None
    for _ in range(3):
        g.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.search_graph_dev(q.data_ptr(), B, D, K, ef, ENTRY, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        kms, kn = g.timing_read()
        g.timing_enable(False)
        if best is None or kms / kn < best[0]:
            best = (kms / kn, el)
    nbytes = E * D * elt + P * 2 * M * 4
    chk = int(ids.sum().item())
    return {"kernel_ms": best[0], "qps": B / best[1], "TBps": nbytes / (best[0] * 1e-3) / 1e12, "E": E / B, "checksum": chk}


if "A" in legs:
    N, D, K = 1_000_000, 768, 10
    g = zv.Index(dim=D, capacity=N, m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_MIRROR | zv.FLAG_BF16_ROWS)
    q = bench.make_rows(zv, torch, g, N, D, True, dev, s, 4096)
    g.build_graph(M, 64, zv.BF_I8, 7)
    g.set_option(zv.OPT_EF_SEARCH_WIDTH, int(os.environ.get("EF_PROBE_WIDTH", "4")))
    cases = ((1024, 128), (1024, 256), (1024, 512), (4096, 128))
    if os.environ.get("EF_PROBE_SWEEP"):   # the launch's fixed cost: kernel time against evaluations per query, small beams
        cases = tuple((1024, ef) for ef in (10, 16, 24, 32, 48, 64, 96, 128))
    for B, ef in cases:
        r = run(g, q, B, D, K, ef, 2)
        print(f"{tag:>10} A 1Mx768 bf16 rank12 bulk  batch {B:5d} ef {ef:4d}: kernel {r['kernel_ms']:.4f} ms  {r['qps']:9.0f} QPS  {r['TBps']:.3f} TB/s = {r['TBps'] / 8:.3f}  E {r['E']:.0f}  chk {r['checksum']}", flush=True)
    del g, q
if "B" in legs:
    N, D, K = 2_000_000, 1536, 100
    g = zv.Index(dim=D, capacity=N, metric=zv.METRIC_COSINE, m=2 * M, graph_stride=2 * M, flags=zv.FLAG_BF16_MIRROR | zv.FLAG_BF16_ROWS)
    q = bench.make_rows(zv, torch, g, N, D, True, dev, s, 1024)
    g.build_graph(M, 64, zv.BF_I8, 7)
    g.set_option(zv.OPT_EF_SEARCH_WIDTH, int(os.environ.get("EF_PROBE_WIDTH", "4")))
    for B, ef in ((1024, 256), (1024, 512)):
        r = run(g, q, B, D, K, ef, 2, reps=5)
        print(f"{tag:>10} B 2Mx1536 cos bf16 k100     batch {B:5d} ef {ef:4d}: kernel {r['kernel_ms']:.4f} ms  {r['qps']:9.0f} QPS  {r['TBps']:.3f} TB/s = {r['TBps'] / 8:.3f}  E {r['E']:.0f}  chk {r['checksum']}", flush=True)
