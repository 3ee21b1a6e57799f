"""GPU tests of the multi-GPU data path on ONE GPU: two id-range shards as two device indexes,
per-shard brute-force search with global id offsets (`_dev` entry points on torch-owned memory),
the [G][nq][k] gathered layout, and the HIP k-way merge (`zvdb_hip_merge_topk_dev`) -- against
the oracle's exact top-k over the whole corpus.  (The collective itself is covered on CPU by
tests/test_sharded_gloo.py; RCCL needs one GPU per rank.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["f32", "i8"])
def test_two_shards_merge_equals_global_topk(oracle, mode):
    import torch
    import zvdb_amd as zv
    from zvdb_amd.sharded import shard_range
    assert zv.device_count() > 0
    n, d, nq, k, G = 7001, 64, 37, 10, 2
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    dev = torch.device("cuda", 0)
    m = zv.BF_F32 if mode == "f32" else zv.BF_I8
    q_t = torch.from_numpy(Q).to(dev)
    g_ids = torch.empty((G * nq, k), dtype=torch.int64, device=dev)
    g_dist = torch.empty((G * nq, k), dtype=torch.float32, device=dev)
    g_cnt = torch.empty((G * nq,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    keep = []
    for r in range(G):
        lo, hi = shard_range(n, G, r)
        idx = zv.Index(dim=d, capacity=hi - lo, flags=zv.FLAG_BF16_MIRROR)
        idx.add_synthetic(hi - lo, d, 0x5EED, first_row=lo)       # same global stream, shard rows
        assert np.array_equal(idx.get_vector(0), V[lo])
        wsb = idx.bruteforce_workspace(nq, k, m)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        idx.search_bruteforce_dev(q_t.data_ptr(), nq, d, k, m, lo, ws.data_ptr(), wsb,
                                  g_ids[r * nq:(r + 1) * nq].data_ptr(), g_dist[r * nq:(r + 1) * nq].data_ptr(),
                                  g_cnt[r * nq:(r + 1) * nq].data_ptr(), s)
        keep.append((idx, ws))
    o_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    o_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    zv.merge_topk_dev(g_ids.data_ptr(), g_dist.data_ptr(), g_cnt.data_ptr(), G, nq, k, o_ids.data_ptr(),
                      o_dist.data_ptr(), o_cnt.data_ptr(), s)
    torch.cuda.synchronize()
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    assert np.array_equal(o_ids.cpu().numpy().astype(np.uint64), ri)
    assert np.array_equal(o_dist.cpu().numpy(), rd)
    assert np.array_equal(o_cnt.cpu().numpy().astype(np.uint32), rc)
    # packed single-buffer variant (what bench.py all-gathers): rank chunks = ids | dist | count
    pb = zv.packed_result_bytes(nq, k)
    pack = torch.zeros(G * pb, dtype=torch.uint8, device=dev)
    for r in range(G):
        c = pack[r * pb:(r + 1) * pb]
        c[:nq * k * 8].view(torch.int64).copy_(g_ids[r * nq:(r + 1) * nq].reshape(-1))
        c[nq * k * 8:nq * k * 12].view(torch.float32).copy_(g_dist[r * nq:(r + 1) * nq].reshape(-1))
        c[nq * k * 12:nq * k * 12 + nq * 4].view(torch.int32).copy_(g_cnt[r * nq:(r + 1) * nq])
    o_ids.zero_(); o_dist.zero_(); o_cnt.zero_()
    zv.merge_topk_packed_dev(pack.data_ptr(), pb, G, nq, k, o_ids.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr(), s)
    torch.cuda.synchronize()
    assert np.array_equal(o_ids.cpu().numpy().astype(np.uint64), ri) and np.array_equal(o_dist.cpu().numpy(), rd)
    # the HIP merge equals the oracle merge on the same gathered lists (incl. a short list)
    gi = g_ids.cpu().numpy().astype(np.uint64).reshape(G, nq, k)
    gd = g_dist.cpu().numpy().reshape(G, nq, k)
    gc = g_cnt.cpu().numpy().astype(np.uint32).reshape(G, nq)
    mi, md, mc = oracle.merge_topk(gi, gd, gc, k)
    assert np.array_equal(mi, ri) and np.array_equal(md, rd)


def test_fp4_shards_keep_their_share_of_the_survivors(oracle):
    """zvdb_hip_index_set_shard_count: 8 id-range shards of one corpus, each keeping its share (24) of the 64 fp4
    survivors, merged -> the same neighbours as ONE index keeping all 64 (a neighbour is lost only if a shard holds more
    than 24 of the global 64 best screen keys), and every returned distance is the exact fp32 distance of its id."""
    import torch
    import zvdb_amd as zv
    from zvdb_amd.sharded import shard_range
    n, d, nq, k, G = 8 * 262_144, 768, 256, 10, 8   # shards just large enough for the guessed-threshold pipeline
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    q_t = torch.empty((nq, d), dtype=torch.float32, device=dev)
    zv.fill_synthetic_dev(q_t.data_ptr(), nq, d, 0xC0FFEE, 0, s)

    def search(idx, lo, mode, ids, dist, cnt):
        wsb = idx.bruteforce_workspace(nq, k, mode)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        idx.search_bruteforce_dev(q_t.data_ptr(), nq, d, k, mode, lo, ws.data_ptr(), wsb, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(), s)
        torch.cuda.synchronize()

    whole = zv.Index(dim=d, capacity=n, flags=zv.FLAG_BF16_MIRROR)
    whole.add_synthetic(n, d, 0x5EED)
    w_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    w_dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    w_cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    e_ids, e_dist = torch.empty_like(w_ids), torch.empty_like(w_dist)
    search(whole, 0, zv.BF_I8, e_ids, e_dist, w_cnt)      # exact top-k (oracle-parity-tested route)
    search(whole, 0, zv.BF_FP4, w_ids, w_dist, w_cnt)
    del whole
    g_ids = torch.empty((G * nq, k), dtype=torch.int64, device=dev)
    g_dist = torch.empty((G * nq, k), dtype=torch.float32, device=dev)
    g_cnt = torch.empty((G * nq,), dtype=torch.int32, device=dev)
    for r in range(G):
        lo, hi = shard_range(n, G, r)
        idx = zv.Index(dim=d, capacity=hi - lo, flags=zv.FLAG_BF16_MIRROR)
        idx.add_synthetic(hi - lo, d, 0x5EED, first_row=lo)
        idx.set_shard_count(G)
        search(idx, lo, zv.BF_FP4, g_ids[r * nq:(r + 1) * nq], g_dist[r * nq:(r + 1) * nq], g_cnt[r * nq:(r + 1) * nq])
        del idx
    o_ids, o_dist, o_cnt = torch.empty_like(w_ids), torch.empty_like(w_dist), torch.empty_like(w_cnt)
    zv.merge_topk_dev(g_ids.data_ptr(), g_dist.data_ptr(), g_cnt.data_ptr(), G, nq, k, o_ids.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr(), s)
    torch.cuda.synchronize()
    exact, one, sharded = e_ids.cpu().numpy(), w_ids.cpu().numpy(), o_ids.cpu().numpy()
    rec_one = np.mean([len(set(one[i]) & set(exact[i])) / k for i in range(nq)])
    rec_sh = np.mean([len(set(sharded[i]) & set(exact[i])) / k for i in range(nq)])
    # (the shards quantise around their own mean / sigma, so single-index and sharded screens differ in the last bit of a
    # few keys: compare recalls, not id lists)
    assert rec_one >= 0.95 and rec_sh >= rec_one - 0.01, (rec_one, rec_sh)
    assert (o_cnt.cpu().numpy() == k).all()
    # exactness of what is returned: ids found by both routes carry bit-identical fp32 distances
    ed, sd = e_dist.cpu().numpy(), o_dist.cpu().numpy()
    for i in range(nq):
        pos = {int(v): j for j, v in enumerate(exact[i])}
        for j, v in enumerate(sharded[i]):
            if int(v) in pos:
                assert sd[i, j] == ed[i, pos[int(v)]]


def _hip_rank(rank, world, port, n, d, nq, k, ret):
    """One rank of the product path: HipShard (HIP search + HIP merge) under ShardedSearch, collectives over gloo (RCCL wants
    one GPU per rank; the box has one, so both ranks share cuda:0)."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import zvdb_amd as zv
    from zvdb_amd.sharded import HipShard, ShardedSearch
    shard = HipShard(n, d, rank, world, 0)
    shard.add_synthetic(0x5EED)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    q_t = torch.from_numpy(Q).to(shard.device)

    class Gloo(ShardedSearch):  # gloo moves host memory: stage the rank's byte buffer through the CPU
        def _gather(self, mine):
            g = torch.empty((self.world, mine.numel()), dtype=torch.uint8)
            dist.all_gather_into_tensor(g.view(-1), mine.cpu())
            return g.to(mine.device)

    ss = Gloo(shard)
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    ok = True
    for mode in (zv.BF_F32, zv.BF_I8):
        ids, dd, cnt = ss.search(q_t, k, mode)
        torch.cuda.synchronize()
        ok = ok and np.array_equal(ids.cpu().numpy().astype(np.uint64), ri) and np.array_equal(dd.cpu().numpy(), rd)
    ids, dd, cnt = ss.search(q_t, k, zv.BF_FP4)  # approximate screen with the shard's survivor share
    torch.cuda.synchronize()
    a = ids.cpu().numpy().astype(np.uint64)
    rec = np.mean([len(set(a[i]) & set(ri[i])) / k for i in range(nq)])
    # the graph route, sharded: each rank builds the bulk HNSW graph of ITS id range on the device, searches it by descent + beam,
    # and the merged result equals the oracle's merge of the per-shard oracle searches (the per-shard graph and search are
    # bit-exact against the oracle: tests/test_gpu_graph.py) -- on a prefix of the corpus small enough for the CPU build
    ng, M, efc, ef = 3001, 8, 32, 48
    gshard = HipShard(ng, d, rank, world, 0, m=2 * M, graph_stride=2 * M)
    gshard.add(V[gshard.lo:gshard.hi])
    gshard.build_graph(M, efc, zv.BF_I8, 5)
    gids, gdd, gcnt = Gloo(gshard).search(q_t, k, route="graph", ef=ef)
    torch.cuda.synchronize()
    from zvdb_amd.sharded import shard_range
    per_i = np.zeros((world, nq, k), dtype=np.uint64); per_d = np.full((world, nq, k), np.inf, dtype=np.float32); per_c = np.zeros((world, nq), dtype=np.uint32)
    for r in range(world):
        lo, hi = shard_range(ng, world, r)
        ref, levels, top = oracle.build_bulk_graph(V[lo:hi], M, efc, 5)
        for qi in range(nq):
            oi, od = ref.search_ef(Q[qi], k, ef, 0, ref.descend(Q[qi], 0, top))
            per_i[r, qi, :len(oi)] = oi + np.uint64(lo); per_d[r, qi, :len(oi)] = od; per_c[r, qi] = len(oi)
    mi, md, mc = oracle.merge_topk(per_i, per_d, per_c, k)
    gok = (np.array_equal(gids.cpu().numpy().astype(np.uint64), mi) and np.array_equal(gdd.cpu().numpy(), md)
           and np.array_equal(gcnt.cpu().numpy().astype(np.uint32), mc))
    ret[rank] = bool(ok) and rec >= 0.95 and bool(gok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_processes_on_the_hip_backend(oracle):
    """world_size 2, one PROCESS per rank (both on cuda:0): HipShard + ShardedSearch's gather layout + the HIP merge give
    the oracle's global top-k on every rank."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hip_rank, args=(2, port, 20_001, 64, 33, 10, ret), nprocs=2, join=True)
    assert all(ret.get(r, False) for r in range(2)), dict(ret)


def test_rccl_behind_the_c_abi_world_size_1(oracle):
    """zvdb_hip_comm_* + zvdb_hip_search_{bruteforce,graph}_sharded_dev through ctypes at the one world size a one-GPU box allows:
    unique id -> communicator -> local search + ncclAllGather + merge inside libzvdb_hip, on the caller's stream; results equal
    the unsharded calls' (and therefore the oracle's), ids carry the shard's offset."""
    import torch
    import zvdb_amd as zv
    from zvdb_amd.sharded import Comm, HipShard
    n, d, nq, k, M = 30_001, 48, 65, 10, 8
    comm = Comm(0, 1, Comm.unique_id(), 0)
    assert (comm.rank, comm.world) == (0, 1)
    shard = HipShard(n, d, 0, 1, 0, m=2 * M, graph_stride=2 * M, comm=comm)
    shard.add_synthetic(0x5EED)
    shard.lo = 1000  # (a shard in the middle of the id space: results carry the offset)
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    q_t = torch.from_numpy(Q).to(shard.device)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    for mode in (zv.BF_F32, zv.BF_BF16, zv.BF_I8):
        ids, dd, cnt = shard.search(q_t, k, mode)
        torch.cuda.synchronize()
        assert np.array_equal(ids.cpu().numpy().astype(np.uint64), ri + np.uint64(1000)) and np.array_equal(dd.cpu().numpy(), rd), mode
        assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), rc)
    shard.build_graph(M, 32, zv.BF_I8, 9)
    ids, dd, cnt = shard.search(q_t, k, route="graph", ef=64)
    li, ld, lc = shard.local_search(q_t, k, route="graph", ef=64)
    torch.cuda.synchronize()
    assert torch.equal(ids, li) and torch.equal(dd, ld) and torch.equal(cnt, lc)
    assert int(ids.min()) >= 1000
    comm.close()


def test_bench_graph_route_sharded_rehearsal(oracle):
    """`bench.py --route graph --row-dtype bf16` (BASELINE configs[3]'s route: bf16 rows, HNSW M = 16, sharded) launched as the driver
    launches N > 1 runs, at world size 1 with the RCCL step forced on, on a small shard: one JSON line with the graph route's
    recall against the sharded brute-force route and its HBM roofline object."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZVDB_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(root, "bench.py"), "--gpus", "1", "--rows", "200000", "--dim", "128", "--steps", "3", "--warmup", "1",
           "--route", "graph", "--row-dtype", "bf16", "--ef", "64", "--no-cpu-baseline", "--no-paths"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["route"] == "graph" and out["config"]["row_dtype"] == "bf16" and out["n_gpus"] == 1
    assert out["value"] > 0 and 0.0 < out["recall_at_10"] <= 1.0 and out["roofline"]["bound"] == "hbm" and out["roofline"]["frac"] > 0


def test_bench_under_torchrun_with_the_rccl_backend_world_size_1(oracle):
    """bench.py exactly as the driver launches it for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), at the
    one world size a one-GPU box allows: the process group is RCCL ("nccl"), the step is search -> ONE packed
    all_gather_into_tensor -> zvdb_hip_merge_topk_packed_dev, the timing is barrier-bracketed and all-reduced (MAX).
    ZVDB_BENCH_FORCE_DIST=1 keeps that path on at world size 1."""
    import json, os, subprocess, sys
    import zvdb_amd as zv
    assert zv.device_count() > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZVDB_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "1", "--rows", "400000", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-paths"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["dtype"] == "bf16" and out["recall_at_10"] == 1.0
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
