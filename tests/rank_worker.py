"""One rank of a multi-rank test, as its own process:  python tests/rank_worker.py KIND RANK WORLD PORT OUT.json ARGS...
(started by tests/_ranks.py; never imported by the pytest process for its worker functions).  The rank writes
{"ok": bool, ...details} to OUT.json as its last act; a crash leaves no file and the launcher reports the log tail.

KINDs
  gloo_bruteforce  n d nq k        id-range shards, oracle compute per rank, zvdb_amd/sharded.py's gather + merge (CPU)
  gloo_graph       n d nq k ef     the graph route of the same host logic (CPU)
  hip_two_rank     n d nq k        HipShard + ShardedSearch on cuda:0, collectives over gloo (GPU box)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np


class OracleShard:
    """A shard whose compute is the oracle: what is under test is the collective + layout logic around it."""

    def __init__(self, V_all, rank, world):
        from zvdb_amd.sharded import shard_range
        self.lo, self.hi = shard_range(len(V_all), world, rank)
        self.V = V_all[self.lo:self.hi]

    def local_search(self, queries, k, mode):
        import oracle
        import torch
        ids, dd, cnt = oracle.bruteforce_topk(self.V, queries.numpy(), k)
        pad = np.arange(k)[None, :] >= cnt[:, None]
        ids = ids.astype(np.int64) + self.lo
        ids[pad] = -1
        dd[pad] = np.inf
        return torch.from_numpy(ids), torch.from_numpy(dd), torch.from_numpy(cnt.astype(np.int32))

    def merge(self, g_ids, g_dist, g_cnt, k):
        import oracle
        import torch
        oi, od, oc = oracle.merge_topk(g_ids.numpy().astype(np.uint64), g_dist.numpy(), g_cnt.numpy().astype(np.uint32), k)
        return torch.from_numpy(oi.astype(np.int64)), torch.from_numpy(od), torch.from_numpy(oc.astype(np.int32))


class OracleGraphShard(OracleShard):
    """The graph route of a shard on the oracle: the shard's own bulk HNSW graph (no cross-shard edges), descent + ef search."""

    def __init__(self, V_all, rank, world, M, efc, seed):
        import oracle
        super().__init__(V_all, rank, world)
        self.ref, self.levels, self.top = oracle.build_bulk_graph(self.V, M, efc, seed)

    def local_search(self, queries, k, mode, route="bruteforce", ef=64, entry=None):
        import torch
        if route != "graph":
            return super().local_search(queries, k, mode)
        Q = queries.numpy()
        ids = np.full((len(Q), k), -1, dtype=np.int64)
        dd = np.full((len(Q), k), np.inf, dtype=np.float32)
        cnt = np.zeros(len(Q), dtype=np.int32)
        for i, q in enumerate(Q):
            a, b = self.ref.search_ef(q, k, ef, 0, self.ref.descend(q, 0, self.top))
            ids[i, :len(a)] = a.astype(np.int64) + self.lo
            dd[i, :len(a)] = b
            cnt[i] = len(a)
        return torch.from_numpy(ids), torch.from_numpy(dd), torch.from_numpy(cnt)


def _init_gloo(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def gloo_bruteforce(rank, world, port, n, d, nq, k):
    import torch
    dist = _init_gloo(rank, world, port)
    import oracle
    from zvdb_amd.sharded import ShardedSearch
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    ss = ShardedSearch(OracleShard(V, rank, world))
    ids, dd, cnt = ss.search(torch.from_numpy(Q), k)
    ri, rd, rc = oracle.bruteforce_topk(V, Q, k)
    kk = min(k, n)
    ok = (np.array_equal(ids.numpy().astype(np.uint64)[:, :kk], ri[:, :kk]) and np.array_equal(dd.numpy()[:, :kk], rd[:, :kk])
          and np.array_equal(cnt.numpy().astype(np.uint32), rc))
    dist.barrier()
    dist.destroy_process_group()
    return {"ok": bool(ok)}


def gloo_graph(rank, world, port, n, d, nq, k, ef):
    import torch
    dist = _init_gloo(rank, world, port)
    import oracle
    from zvdb_amd.sharded import ShardedSearch
    V = oracle.uniform_rows(0x5EED, 0, n, d)
    Q = oracle.uniform_rows(0xC0FFEE, 0, nq, d)
    ss = ShardedSearch(OracleGraphShard(V, rank, world, 8, 32, 3))
    ids, dd, cnt = ss.search(torch.from_numpy(Q), k, route="graph", ef=ef)
    # expected: every shard's list computed here, merged by the oracle
    shards = [OracleGraphShard(V, r, world, 8, 32, 3) for r in range(world)]
    per = [sh.local_search(torch.from_numpy(Q), k, 0, route="graph", ef=ef) for sh in shards]
    gi = np.stack([p[0].numpy().astype(np.uint64) for p in per])
    gd = np.stack([p[1].numpy() for p in per])
    gc = np.stack([p[2].numpy().astype(np.uint32) for p in per])
    mi, md, mc = oracle.merge_topk(gi, gd, gc, k)
    ok = (np.array_equal(ids.numpy().astype(np.uint64), mi) and np.array_equal(dd.numpy(), md)
          and np.array_equal(cnt.numpy().astype(np.uint32), mc))
    # sanity of the route itself: the merged graph result finds most of the exact neighbours on this small corpus
    ri, _, _ = oracle.bruteforce_topk(V, Q, k)
    rec = float(np.mean([len(set(mi[i][:mc[i]]) & set(ri[i])) / k for i in range(nq)]))
    dist.barrier()
    dist.destroy_process_group()
    return {"ok": bool(ok) and rec >= 0.8, "merge_equal": bool(ok), "recall": rec}


def hip_two_rank(rank, world, port, n, d, nq, k):
    """One rank of the product path: HipShard (HIP search + HIP merge) under ShardedSearch, collectives over gloo (RCCL wants
    one GPU per rank; the box has one, so both ranks share cuda:0)."""
    import torch
    dist = _init_gloo(rank, world, port)
    import oracle
    import zvdb_amd as zv
    from zvdb_amd.sharded import HipShard, ShardedSearch, shard_range
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
    exact = {}
    for name, mode in (("f32", zv.BF_F32), ("i8", zv.BF_I8)):
        ids, dd, cnt = ss.search(q_t, k, mode)
        torch.cuda.synchronize()
        exact[name] = bool(np.array_equal(ids.cpu().numpy().astype(np.uint64), ri) and np.array_equal(dd.cpu().numpy(), rd))
    ids, dd, cnt = ss.search(q_t, k, zv.BF_FP4)  # approximate screen with the shard's survivor share
    torch.cuda.synchronize()
    a = ids.cpu().numpy().astype(np.uint64)
    rec = float(np.mean([len(set(a[i]) & set(ri[i])) / k for i in range(nq)]))
    # the graph route, sharded: each rank builds the bulk HNSW graph of ITS id range on the device, searches it by descent +
    # beam, and the merged result equals the oracle's merge of the per-shard oracle searches (the per-shard graph and search
    # are bit-exact against the oracle: tests/test_gpu_graph.py) -- on a prefix of the corpus small enough for the CPU build
    ng, M, efc, ef = 3001, 8, 32, 48
    gshard = HipShard(ng, d, rank, world, 0, m=2 * M, graph_stride=2 * M)
    gshard.add(V[gshard.lo:gshard.hi])
    gshard.build_graph(M, efc, zv.BF_I8, 5)
    gids, gdd, gcnt = Gloo(gshard).search(q_t, k, route="graph", ef=ef)
    torch.cuda.synchronize()
    per_i = np.zeros((world, nq, k), dtype=np.uint64)
    per_d = np.full((world, nq, k), np.inf, dtype=np.float32)
    per_c = np.zeros((world, nq), dtype=np.uint32)
    for r in range(world):
        lo, hi = shard_range(ng, world, r)
        ref, levels, top = oracle.build_bulk_graph(V[lo:hi], M, efc, 5)
        for qi in range(nq):
            oi, od = ref.search_ef(Q[qi], k, ef, 0, ref.descend(Q[qi], 0, top))
            per_i[r, qi, :len(oi)] = oi + np.uint64(lo)
            per_d[r, qi, :len(oi)] = od
            per_c[r, qi] = len(oi)
    mi, md, mc = oracle.merge_topk(per_i, per_d, per_c, k)
    gok = bool(np.array_equal(gids.cpu().numpy().astype(np.uint64), mi) and np.array_equal(gdd.cpu().numpy(), md)
               and np.array_equal(gcnt.cpu().numpy().astype(np.uint32), mc))
    dist.barrier()
    dist.destroy_process_group()
    return {"ok": all(exact.values()) and rec >= 0.95 and gok, "exact": exact, "fp4_recall": rec, "graph_equal": gok}


KINDS = {"gloo_bruteforce": gloo_bruteforce, "gloo_graph": gloo_graph, "hip_two_rank": hip_two_rank}

if __name__ == "__main__":
    kind, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    res = KINDS[kind](rank, world, port, *[int(a) for a in sys.argv[6:]])
    with open(out + ".tmp", "w") as f:
        json.dump(res, f)
    os.replace(out + ".tmp", out)
    sys.exit(0 if res.get("ok") else 1)
