# the bf16 stream route (guessed + verified thresholds) on rows WITH structure (intrinsic rank 12 + noise, and tight clusters): the guess
# assumes near-Gaussian keys, so this is where it can be off -- results must still equal the fp32 route's, and the time per batch
# shows what repairs cost.  usage: python tools/bf16_structured_probe.py [rows]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zvdb_amd as zv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
D, B, K = 768, 1024, 10
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(11)
def make(kind, n):
    if kind == "rank12":
        return torch.rand((n, 12), generator=g, device=dev) @ A + 0.01 * torch.randn((n, D), generator=g, device=dev)
    c = torch.randint(0, 256, (n,), generator=g, device=dev)
    return C[c] + 0.02 * torch.randn((n, D), generator=g, device=dev)
A = torch.rand((12, D), generator=g, device=dev) - 0.5
C = torch.rand((256, D), generator=g, device=dev)
for kind in ("rank12", "clusters256"):
    idx = zv.Index(dim=D, capacity=N, flags=zv.FLAG_BF16_MIRROR)
    for r0 in range(0, N, 250_000):
        rows = make(kind, min(250_000, N - r0)).contiguous(); torch.cuda.synchronize()
        idx.add_dev(rows.data_ptr(), rows.shape[0], D, s); torch.cuda.synchronize()
    wsb = max(idx.bruteforce_workspace(B, K, m) for m in (zv.BF_F32, zv.BF_BF16))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = {}
    bad = 0
    for b in range(6):
        q = make(kind, B).contiguous(); torch.cuda.synchronize()
        for mode, name in ((zv.BF_F32, "f32"), (zv.BF_BF16, "bf16")):
            ids = torch.empty((B, K), dtype=torch.int64, device=dev); dd = torch.empty((B, K), dtype=torch.float32, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
            t0 = time.perf_counter()
            idx.search_bruteforce_dev(q.data_ptr(), B, D, K, mode, 0, ws.data_ptr(), wsb, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
            torch.cuda.synchronize()
            out.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
            if name == "f32": ref = (ids.cpu().numpy(), dd.cpu().numpy())
            else:
                got = (ids.cpu().numpy(), dd.cpu().numpy())
                same = np.mean([len(set(got[0][i]) & set(ref[0][i])) / K for i in range(B)])
                if same < 1.0 or not np.array_equal(np.sort(got[1], axis=1), np.sort(ref[1], axis=1)): bad += 1; print(f"  {kind} batch {b}: bf16 route differs from the fp32 route (ids agree {same:.4f})")
    print(f"{kind}: {N} x {D}, 6 batches: {bad} differing; ms per batch f32 {[round(t, 2) for t in out['f32']]} bf16 {[round(t, 2) for t in out['bf16']]}", flush=True)
    del idx, ws
