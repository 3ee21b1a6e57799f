"""Cost of the certificate's repair pass (k_exact_scan + merge) at the headline shape: a batch of nq queries, every one of them listed
(ZVDB_OPT_CERT_FORCE_REPAIR), against the same batch unforced.  python tools/exact_scan_probe.py [rows] [dim] [row_dtype f32|bf16]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import zvdb_amd as zv

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
idx = zv.Index(dim=dim, capacity=rows, flags=zv.FLAG_BF16_MIRROR | (zv.FLAG_BF16_ROWS if bf16 else 0))
idx.add_synthetic(rows, dim, 0x5EED)
out = {"rows": rows, "dim": dim, "row_dtype": "bf16" if bf16 else "f32", "cases": {}}
CASES = ((1, 10), (8, 10), (16, 10), (32, 10), (64, 10), (256, 10)) if os.environ.get("XS_PROBE_FULL") else ((1, 10), (8, 10), (16, 10), (32, 10))
for nq, k in CASES:
    q = torch.empty((nq, dim), dtype=torch.float32, device=dev)
    zv.fill_synthetic_dev(q.data_ptr(), nq, dim, 0xC0FFEE, 0, s)
    wsb = idx.bruteforce_workspace(nq, k, zv.BF_BF16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    res = {}
    keep = None
    for forced in (0, 1):
        idx.set_option(zv.OPT_CERT_FORCE_REPAIR, forced)
        for _ in range(3):
            idx.search_bruteforce_dev(q.data_ptr(), nq, dim, k, zv.BF_BF16, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 5
        for _ in range(R):
            idx.search_bruteforce_dev(q.data_ptr(), nq, dim, k, zv.BF_BF16, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
        torch.cuda.synchronize()
        res["forced_ms" if forced else "plain_ms"] = (time.perf_counter() - t0) / R * 1e3
        if forced:
            res["same_ids"] = bool(torch.equal(keep[0], ids)) and bool(torch.equal(keep[1], dd))
        else:
            keep = (ids.clone(), dd.clone())
    res["repair_ms"] = res["forced_ms"] - res["plain_ms"]
    res["repair_ms_per_query"] = res["repair_ms"] / nq
    res["rows_GBps"] = rows * dim * (2 if bf16 else 4) / (res["repair_ms"] * 1e-3) / 1e9 * max(1, -(-nq // 32))
    out["cases"][f"nq{nq}_k{k}"] = res
    print(f"nq={nq} k={k}: repair {res['repair_ms']:.3f} ms ({res['repair_ms_per_query']:.3f} per query), same results {res['same_ids']}", flush=True)
idx.set_option(zv.OPT_CERT_FORCE_REPAIR, 0)
if os.environ.get("XS_PROBE_JSON"):
    print(json.dumps(out))
