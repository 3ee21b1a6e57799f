"""Id-range sharded search across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Shard s of G owns the
contiguous id range [s*N/G, (s+1)*N/G) (reference ids are dense insertion order, hnsw.zig:77).
Every rank answers every query batch on its shard (ids already offset to global), then ONE
all-gather of the per-shard top-k -- nq*k*(8+4) B + nq*4 B per rank, latency-bound -- and a local
k-way merge by (distance, id) give the identical global top-k on every rank.  There is no other
data-path collective.

The collective + layout logic is backend-agnostic so that it is testable on CPU with gloo
(tests/test_sharded_gloo.py injects an oracle backend); the product backend is `HipShard`, whose
search and merge are HIP kernels and which raises without a GPU."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    return rank * total_rows // world, (rank + 1) * total_rows // world


class HipShard:
    """Product backend: one zvdb_hip index holding rows [lo, hi) on this rank's GPU."""

    def __init__(self, total_rows: int, dim: int, rank: int, world: int, device_index: int, flags: int = 1):
        from . import Index
        self.lo, self.hi = shard_range(total_rows, world, rank)
        self.dim = dim
        self.device = torch.device("cuda", device_index)
        self.index = Index(dim=dim, capacity=self.hi - self.lo, device=device_index, flags=flags)
        self.index.set_shard_count(world)  # the approximate screen keeps this shard's share of the survivor budget
        self._ws = None

    def add_synthetic(self, seed: int) -> None:
        self.index.add_synthetic(self.hi - self.lo, self.dim, seed, first_row=self.lo)

    def add(self, rows_local) -> None:
        self.index.add(rows_local)

    def local_search(self, queries: torch.Tensor, k: int, mode: int):
        from . import _lib  # noqa: F401
        nq = queries.shape[0]
        need = self.index.bruteforce_workspace(nq, k, mode)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        dd = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        cnt = torch.empty((nq,), dtype=torch.int32, device=self.device)
        s = torch.cuda.current_stream().cuda_stream
        self.index.search_bruteforce_dev(queries.data_ptr(), nq, self.dim, k, mode, self.lo, self._ws.data_ptr(),
                                         self._ws.numel(), ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s)
        return ids, dd, cnt

    def merge(self, g_ids, g_dist, g_cnt, k: int):
        from . import merge_topk_dev
        G, nq, _ = g_ids.shape
        o_ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        o_dist = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        o_cnt = torch.empty((nq,), dtype=torch.int32, device=self.device)
        merge_topk_dev(g_ids.data_ptr(), g_dist.data_ptr(), g_cnt.data_ptr(), G, nq, k, o_ids.data_ptr(),
                       o_dist.data_ptr(), o_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return o_ids, o_dist, o_cnt


class ShardedSearch:
    """search(queries, k) = local top-k on this rank's shard -> all-gather -> k-way merge."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def _gather(self, mine: torch.Tensor) -> torch.Tensor:
        """[world][bytes] <- every rank's byte buffer (RCCL on device memory; gloo on host memory)."""
        gathered = torch.empty((self.world, mine.numel()), dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(gathered.view(-1), mine, group=self.group)
        return gathered

    def search(self, queries: torch.Tensor, k: int, mode: int = 0):
        ids, dd, cnt = self.backend.local_search(queries, k, mode)
        if self.world == 1:
            return ids, dd, cnt
        nq = ids.shape[0]
        # ONE collective per batch: the three result arrays travel as one byte buffer per rank (ids | dist | count)
        parts = [t.contiguous().view(torch.uint8).reshape(-1) for t in (ids, dd, cnt)]
        mine = torch.cat(parts)
        gathered = self._gather(mine)
        b0, b1 = parts[0].numel(), parts[0].numel() + parts[1].numel()
        # [G][nq][k] row-major (the layout merge_topk consumes)
        g_ids = gathered[:, :b0].contiguous().view(ids.dtype).view(self.world, nq, k)
        g_dist = gathered[:, b0:b1].contiguous().view(dd.dtype).view(self.world, nq, k)
        g_cnt = gathered[:, b1:].contiguous().view(cnt.dtype).view(self.world, nq)
        return self.backend.merge(g_ids, g_dist, g_cnt, k)
