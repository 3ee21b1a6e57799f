import sys, time
sys.path.insert(0, '/root/repo')
import torch, numpy as np, zvdb_amd as zv
NG, DG, BG, K = 1_000_000, 768, 1024, 10
dev = torch.device("cuda", 0)
g = zv.Index(dim=DG, capacity=NG, m=16, flags=zv.FLAG_BF16_MIRROR)
g.add_synthetic(NG, DG, 0x5EED)
g.build_knn_graph(16, zv.BF_I8)
s = torch.cuda.current_stream().cuda_stream
qg = torch.empty((BG, DG), dtype=torch.float32, device=dev)
zv.fill_synthetic_dev(qg.data_ptr(), BG, DG, 0xC0FFEE, 0, s)
ids = torch.empty((BG, K), dtype=torch.int64, device=dev); dd = torch.empty((BG, K), dtype=torch.float32, device=dev); cnt = torch.empty((BG,), dtype=torch.int32, device=dev)
st = torch.zeros(8, dtype=torch.int64, device=dev)
for ef in (64, 128, 256):
    st.zero_()
    g.search_graph_dev(qg.data_ptr(), BG, DG, K, ef, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), st.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.search_graph_dev(qg.data_ptr(), BG, DG, K, ef, 0, 0, ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 3
    ev, pp = st[0].item(), st[1].item()
    print(f"ef {ef}: {el*1e3:.2f} ms/batch = {BG/el:,.0f} QPS, evals/q {ev/BG:.0f}, pops/q {pp/BG:.0f}, visited-set overflows {st[2].item()}, "
          f"{ev*DG*4/el/1e12:.2f} TB/s algorithmic ({ev*DG*4/el/8e12:.2f} of 8 TB/s), {el*1e6/(pp/BG):.2f} us per pop")
