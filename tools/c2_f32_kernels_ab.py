"""A/B of the three fp32-screen kernels on BASELINE configs[1] (1M x 128 fp32, batch 256) and neighbouring shapes, inside ONE process on
one box (tuning ZVDB_BF_F32_KERNEL): regq = register-resident queries (bf_f32_regq.hip, default from round 6), stream = the 128-query
kernel with queries in LDS (k_i8_stream<0>), v6 = the staged 256 x 256 tile (k_bf_scan5<0>).  Results must be identical.
  python tools/c2_f32_kernels_ab.py [rows dim batch]..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import zvdb_amd as zv

dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream


def leg(N, D, B, reps):
    idx = zv.Index(dim=D, capacity=N)
    idx.add_synthetic(N, D, 0x5EED)
    K, mode = 10, zv.BF_F32
    q = torch.empty((B, D), dtype=torch.float32, device=dev)
    zv.fill_synthetic_dev(q.data_ptr(), B, D, 0xC0FFEE, 0, s)
    wsb = idx.bruteforce_workspace(B, K, mode)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)

    def run(n):
        for _ in range(n):
            idx.search_bruteforce_dev(q.data_ptr(), B, D, K, mode, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
        torch.cuda.synchronize()

    run(8)
    ref = (ids.clone(), dd.clone())
    res = {}
    settings = [("regq", {}), ("regq_sampled", {"ZVDB_BF_F32_GUESS": "sample"}), ("regq_nogated", {"ZVDB_BF_DEFER_MAX": "1000000"}), ("stream", {"ZVDB_BF_F32_KERNEL": "stream"}), ("v6_staged", {"ZVDB_BF_F32_KERNEL": "v6"})]
    if D > 128:
        settings = [("regq", {}), ("v6_staged", {"ZVDB_BF_F32_KERNEL": "v6"})]
    for rnd in range(3):
        for name, sw in settings:
            with zv.tuned(**sw):
                run(3)
                idx.timing_enable(1)
                t0 = time.perf_counter()
                run(reps)
                el = (time.perf_counter() - t0) / reps
                kms, kn = idx.timing_read()
                idx.timing_enable(0)
                assert os.environ.get("C2_AB_NOCHECK") or (torch.equal(ids, ref[0]) and torch.equal(dd, ref[1])), name   # (NOCHECK: the timing-only ablation builds)
            res.setdefault(name, []).append((el * 1e3, kms / max(kn, 1)))
    fl = 2.0 * B * N * D
    for name, v in res.items():
        best = min(x[0] for x in v)
        reg = min(x[1] for x in v)
        print(f"  {name:12s} ms per batch {best:.4f} (runs {' '.join(f'{x[0]:.4f}' for x in v)}) -> {B / (best * 1e-3):.0f} QPS; scan region {reg:.4f} ms = "
              f"{fl / (reg * 1e-3) / 1e12:.1f} TFLOP/s = {fl / (reg * 1e-3) / 157.3e12:.3f} of the fp32-MFMA peak", flush=True)


shapes = [(1_000_000, 128, 256), (1_000_000, 64, 256), (4_000_000, 128, 256), (1_000_000, 128, 1024)]
if len(sys.argv) > 3:
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for N, D, B in shapes:
    print(f"{N} x {D} fp32 screen, batch {B}", flush=True)
    leg(N, D, B, 100 if N * B <= 1_000_000 * 256 else 30)
