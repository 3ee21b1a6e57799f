"""World-size-2 (and 3) gloo tests of the multi-GPU host logic on CPU: id-range shards, global id
offsets, all-gather layout, k-way merge.  The compute inside each rank is the ORACLE here (the HIP
backend needs a GPU); what is under test is zvdb_amd/sharded.py's collective + layout logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleShard:
    def __init__(self, V_all, rank, world):
        from zvdb_amd.sharded import shard_range
        self.lo, self.hi = shard_range(len(V_all), world, rank)
        self.V = V_all[self.lo:self.hi]

    def local_search(self, queries, k, mode):
        import oracle
        ids, dd, cnt = oracle.bruteforce_topk(self.V, queries.numpy(), k)
        pad = np.arange(k)[None, :] >= cnt[:, None]
        ids = ids.astype(np.int64) + self.lo
        ids[pad] = -1
        dd[pad] = np.inf
        return torch.from_numpy(ids), torch.from_numpy(dd), torch.from_numpy(cnt.astype(np.int32))

    def merge(self, g_ids, g_dist, g_cnt, k):
        import oracle
        oi, od, oc = oracle.merge_topk(g_ids.numpy().astype(np.uint64), g_dist.numpy(),
                                       g_cnt.numpy().astype(np.uint32), k)
        return torch.from_numpy(oi.astype(np.int64)), torch.from_numpy(od), torch.from_numpy(oc.astype(np.int32))


def _worker(rank, world, port, n, d, nq, k, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from zvdb_amd.sharded import ShardedSearch
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    ss = ShardedSearch(OracleShard(V, rank, world))
    ids, dd, cnt = ss.search(torch.from_numpy(Q), k)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    ok = (np.array_equal(ids.numpy().astype(np.uint64)[:, :min(k, n)], ri[:, :min(k, n)])
          and np.array_equal(dd.numpy()[:, :min(k, n)], rd[:, :min(k, n)])
          and np.array_equal(cnt.numpy().astype(np.uint32), rc))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,d,nq,k", [(2, 1001, 16, 9, 10), (3, 64, 8, 4, 10), (2, 5, 4, 3, 10)])
def test_sharded_topk_allgather_merge(world, n, d, nq, k):
    import oracle
    oracle.build()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, d, nq, k, ret), nprocs=world, join=True)
    assert all(ret.get(r, False) for r in range(world)), dict(ret)


def test_shard_ranges_partition_the_id_space():
    from zvdb_amd.sharded import shard_range
    for n in (0, 1, 7, 10_000_000, 100_000_003):
        for g in (1, 2, 4, 8):
            rs = [shard_range(n, g, r) for r in range(g)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(g - 1))
            assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= 1
