"""A/B of the launch-count changes of round 5 on BASELINE configs[1] (1M x 128 fp32, batch 256, fp32 screen) and on the headline's
step shape, inside ONE process on one box: tuning switches select the old launch sequences.
  ZVDB_BF_CERT_DEFER=0  gated repair scan + repair merge behind the verifying merge (two no-op launches per batch)
  ZVDB_BF_FOLD_PREP=0   k_bf_prep_queries as its own launch"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import zvdb_amd as zv

dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream


def leg(N, D, B, mode, reps):
    idx = zv.Index(dim=D, capacity=N, flags=zv.FLAG_BF16_MIRROR)
    idx.add_synthetic(N, D, 0x5EED)
    K = 10
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
    settings = [("new", {}), ("old_gated_pair", {"ZVDB_BF_CERT_DEFER": "0"}), ("old_prep", {"ZVDB_BF_FOLD_PREP": "0"}),
                ("old_both", {"ZVDB_BF_CERT_DEFER": "0", "ZVDB_BF_FOLD_PREP": "0"})]
    for rnd in range(3):          # interleaved rounds: drift of the box shows as spread inside a setting
        for name, sw in settings:
            with zv.tuned(**sw):
                run(3)
                t0 = time.perf_counter()
                run(reps)
                el = (time.perf_counter() - t0) / reps
                assert torch.equal(ids, ref[0]) and torch.equal(dd, ref[1]), name
            res.setdefault(name, []).append(el * 1e3)
    for name, v in res.items():
        print(f"  {name:16s} ms per batch {min(v):.4f} (runs {' '.join(f'{x:.4f}' for x in v)})  -> {B / (min(v) * 1e-3):.0f} QPS", flush=True)
    print("  certificate:", idx.cert_info(), flush=True)


print("configs[1]: 1M x 128 fp32 screen, batch 256", flush=True)
leg(1_000_000, 128, 256, zv.BF_F32, 100)
print("configs[1] on the certified bf16 screen", flush=True)
leg(1_000_000, 128, 256, zv.BF_BF16, 100)
print("headline shape: 10M x 768 bf16 screen, batch 1024", flush=True)
leg(10_000_000, 768, 1024, zv.BF_BF16, 10)
