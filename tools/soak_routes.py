# soak: many different query batches through EVERY screen of the brute-force route (bf16, int8, fp4 streams; fp32) on shapes that pick each
# stream-kernel geometry; the bf16 / int8 / fp32 routes return the certified exact top-k, so every batch must agree bit for bit across them
# (the fp4 screen carries no certificate: its recall against the exact answer is printed, nothing asserted).  (The screens'
# operand rings are inline-asm reads on compiler-allocated registers with tied waits -- the form that, in bf_f32_regq.hip's first version,
# lost a neighbour once in ~1e5 queries.  tools/check_asm_reads.py audits their ISA; this is the empirical side.)
# usage: python tools/soak_routes.py [batches per shape [headline batches]]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zvdb_amd as zv
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
ROUTES = [(n, getattr(zv, n)) for n in ("BF_BF16", "BF_I8", "BF_FP4", "BF_F32") if hasattr(zv, n)]
bad = 0
t0 = time.time()
for N, D, B, K, nb in ((1_000_000, 768, 1024, 10, NB), (1_000_000, 512, 512, 10, NB), (2_000_000, 256, 600, 10, NB), (2_000_000, 128, 512, 32, NB),
                       (1_000_000, 1536, 256, 10, NB // 2), (10_000_000, 768, 1024, 10, NH)):
    routes = [r for r in ROUTES if not (r[0] == "BF_F32" and N * D > 2_000_000 * 768)]   # (the fp32 mirror of the largest shape does not pay its build)
    idx = zv.Index(dim=D, capacity=N, flags=zv.FLAG_BF16_MIRROR)
    idx.add_synthetic(N, D, 0x5EED)
    wsb = max(idx.bruteforce_workspace(B, K, m) for _, m in routes)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    q = torch.empty((B, D), dtype=torch.float32, device=dev)
    out = {n: (torch.empty((B, K), dtype=torch.int64, device=dev), torch.empty((B, K), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev))
           for n, _ in routes}
    fp4_hit = fp4_all = 0
    for b in range(nb):
        zv.fill_synthetic_dev(q.data_ptr(), B, D, 0xC0FFEE + 7919 * b, 0, s)
        for n, m in routes:
            i, d, c = out[n]
            idx.search_bruteforce_dev(q.data_ptr(), B, D, K, m, 0, ws.data_ptr(), wsb, i.data_ptr(), d.data_ptr(), c.data_ptr(), s)
        torch.cuda.synchronize()
        r0 = out[routes[0][0]]
        for n, _ in routes[1:]:
            a = out[n]
            if n == "BF_FP4":
                fp4_hit += int((a[0].unsqueeze(2) == r0[0].unsqueeze(1)).any(dim=2).sum())
                fp4_all += B * K
                continue
            if not (torch.equal(a[0], r0[0]) and torch.equal(a[1], r0[1]) and torch.equal(a[2], r0[2]) and bool((a[2] == K).all())):
                bad += 1
                nq = int((a[0] != r0[0]).any(dim=1).sum())
                print(f"{N}x{D} batch {b}: {n} != {routes[0][0]} in {nq} queries", flush=True)
        if b % 250 == 249 or b == nb - 1: print(f"{N}x{D} b{B} k{K}: {b + 1} batches x {len(routes)} routes, {bad} bad, fp4 recall {fp4_hit / max(fp4_all, 1):.3f}, {time.time() - t0:.0f} s", flush=True)
    print(f"{N}x{D}: certificate {idx.cert_info()}", flush=True)
    del idx, ws
    torch.cuda.empty_cache()
print("done:", bad, "bad")
