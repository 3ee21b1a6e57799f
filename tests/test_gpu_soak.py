"""A short soak inside the suite: many DIFFERENT query batches through the certified screens of the brute-force route; every batch must come back
bit-equal across routes (ids, distances, counts).  Rare-event bugs (round 6: an inline-asm operand ring that lost one neighbour in ~10^5 queries) do
not show in parity tests of a few batches; tools/soak_routes.py / tools/soak_f32_regq.py run the long version (thousands of batches, the headline
shape), this keeps a few hundred batches over every register-query geometry up to 128 dimensions in every run of the suite."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,b,k,nb", [(600_000, 128, 300, 10, 60), (500_000, 64, 257, 100, 60), (400_000, 96, 300, 10, 60), (300_000, 200, 1024, 32, 30)])
def test_routes_agree_over_many_batches(n, d, b, k, nb):
    import torch
    import zvdb_amd as zv
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    idx = zv.Index(dim=d, capacity=n, flags=zv.FLAG_BF16_MIRROR)
    idx.add_synthetic(n, d, 0x5EED)
    routes = [zv.BF_F32, zv.BF_I8, zv.BF_BF16]
    wsb = max(idx.bruteforce_workspace(b, k, m) for m in routes)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    q = torch.empty((b, d), dtype=torch.float32, device=dev)
    out = {m: (torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), dtype=torch.float32, device=dev), torch.empty((b,), dtype=torch.int32, device=dev))
           for m in routes}
    for bi in range(nb):
        zv.fill_synthetic_dev(q.data_ptr(), b, d, 0xC0FFEE + 7919 * bi, 0, s)
        for m in routes:
            i, dd, c = out[m]
            idx.search_bruteforce_dev(q.data_ptr(), b, d, k, m, 0, ws.data_ptr(), wsb, i.data_ptr(), dd.data_ptr(), c.data_ptr(), s)
        torch.cuda.synchronize()
        r0 = out[routes[0]]
        assert bool((r0[2] == k).all()), bi
        for m in routes[1:]:
            a = out[m]
            assert torch.equal(a[0], r0[0]) and torch.equal(a[1], r0[1]) and torch.equal(a[2], r0[2]), (bi, m)
