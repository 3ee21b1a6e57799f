"""Id-range sharded search across the GPUs of one node.

One process per GPU.  Shard s of G owns the contiguous id range [s*N/G, (s+1)*N/G) (reference ids are dense insertion
order, hnsw.zig:77).  Every rank answers every query batch on its shard (ids already offset to global), then ONE
all-gather of the per-shard top-k -- nq*k*(8+4) B + nq*4 B per rank, latency-bound -- and a local k-way merge by
(distance, id) give the identical global top-k on every rank.  There is no other data-path collective.

The product path is the C ABI: `Comm` wraps a `zvdb_hip_comm` (one RCCL rank, created from a unique id that travels over
whatever channel the host has -- here torch.distributed) and `HipShard.search` is ONE call of
`zvdb_hip_search_{bruteforce,graph}_sharded_dev`: local search -> ncclAllGather -> merge, all on the caller's stream
(include/zvdb_hip.h; a Zig / C host shards through the same entry points).  Both routes shard: brute force (exact, certified)
and the graph route (`route="graph"`: per-shard bulk HNSW graph over the shard's own rows, ef-mode beam search entered by
descent; recall = per-shard recall).

`ShardedSearch` keeps the same step expressed over torch.distributed so that the layout + collective logic is testable on
CPU with gloo (tests/test_sharded_gloo.py injects an oracle backend) and on a one-GPU box (two ranks sharing cuda:0 cannot
form an RCCL communicator)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    return rank * total_rows // world, (rank + 1) * total_rows // world


class Comm:
    """A zvdb_hip_comm: this process's RCCL rank behind the C ABI."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device_index: int = -1):
        from . import _lib
        self._L = _lib.hip()
        assert len(unique_id) == _lib.COMM_ID_BYTES
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(self._L.zvdb_hip_comm_create(rank, world, C.cast(buf, C.c_void_p), device_index, C.byref(h)))
        self._h = h
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.hip().zvdb_hip_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device_index: int, group=None) -> "Comm":
        """Rank 0 draws the unique id; torch.distributed (any backend) carries its 128 bytes to the other ranks."""
        from . import _lib
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        # (1) what a rank can fail on alone is checked on every rank and agreed on BEFORE the collective create: a rank that
        # returned early from zvdb_hip_comm_create would leave the others waiting in ncclCommInitRank
        mine = None
        try:
            _lib.check(_lib.hip().zvdb_hip_comm_precheck(device_index))
        except Exception as e:
            mine = f"rank {rank}: {type(e).__name__}: {e}"
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, mine, group=group)
        else:
            every = [mine]
        bad = [m for m in every if m]
        if bad:
            raise RuntimeError("zvdb_hip_comm_precheck failed: " + "; ".join(bad))
        # (2) the unique id: rank 0 draws it, the host channel carries its 128 bytes
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as e:  # (the other ranks wait in the broadcast below: tell them instead of leaving them there)
                box[0] = e
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        if isinstance(box[0], Exception):
            raise box[0]
        return cls(rank, world, box[0], device_index)

    def query(self):
        """(rank, ranks, bus id) as RCCL and HIP report them for this communicator (zvdb_hip_comm_query)."""
        from . import _lib
        r, n = C.c_uint32(0), C.c_uint32(0)
        bus = C.create_string_buffer(64)
        _lib.check(self._L.zvdb_hip_comm_query(self._h, C.byref(r), C.byref(n), bus, 64))
        return r.value, n.value, bus.value.decode()

    def timing_enable(self, on: int) -> None:
        from . import _lib
        _lib.check(self._L.zvdb_hip_comm_timing_enable(self._h, int(on)))

    def timing_read(self):
        """(summed microseconds, samples) of the timed ncclAllGather launches."""
        from . import _lib
        us, n = C.c_double(0.0), C.c_uint32(0)
        _lib.check(self._L.zvdb_hip_comm_timing_read(self._h, C.byref(us), C.byref(n)))
        return us.value, n.value

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.zvdb_hip_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipShard:
    """Product backend: one zvdb_hip index holding rows [lo, hi) on this rank's GPU."""

    def __init__(self, total_rows: int, dim: int, rank: int, world: int, device_index: int, flags: int = 1, metric: int = 0,
                 m: int = 16, graph_stride: int = 0, comm: Optional[Comm] = None):
        from . import Index
        self.lo, self.hi = shard_range(total_rows, world, rank)
        self.dim = dim
        self.device = torch.device("cuda", device_index)
        self.index = Index(dim=dim, metric=metric, m=m, graph_stride=graph_stride, capacity=self.hi - self.lo, device=device_index, flags=flags)
        self.index.set_shard_count(world)  # the approximate screen keeps this shard's share of the survivor budget
        self.comm = comm
        self._ws = None

    def add_synthetic(self, seed: int) -> None:
        self.index.add_synthetic(self.hi - self.lo, self.dim, seed, first_row=self.lo)

    def add(self, rows_local) -> None:
        self.index.add(rows_local)

    def build_graph(self, M: int = 16, ef_construction: int = 64, mode: int = 2, level_seed: int = 7) -> None:
        """The shard's own bulk HNSW graph over its id range (no cross-shard edges: SURVEY 8e)."""
        self.index.build_graph(M, ef_construction, mode, level_seed)

    def _out(self, nq, k):
        return (torch.empty((nq, k), dtype=torch.int64, device=self.device), torch.empty((nq, k), dtype=torch.float32, device=self.device),
                torch.empty((nq,), dtype=torch.int32, device=self.device))

    def _workspace(self, need: int):
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def local_search(self, queries: torch.Tensor, k: int, mode: int = 0, route: str = "bruteforce", ef: int = 128, entry: Optional[int] = None):
        from . import ENTRY_DESCEND
        nq = queries.shape[0]
        ids, dd, cnt = self._out(nq, k)
        s = torch.cuda.current_stream().cuda_stream
        if route == "graph":
            self.index.search_graph_dev(queries.data_ptr(), nq, self.dim, k, ef, ENTRY_DESCEND if entry is None else entry, self.lo,
                                        ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), 0, s)
        else:
            ws = self._workspace(self.index.bruteforce_workspace(nq, k, mode))
            self.index.search_bruteforce_dev(queries.data_ptr(), nq, self.dim, k, mode, self.lo, ws.data_ptr(), ws.numel(), ids.data_ptr(),
                                             dd.data_ptr(), cnt.data_ptr(), s)
        return ids, dd, cnt

    def search(self, queries: torch.Tensor, k: int, mode: int = 0, route: str = "bruteforce", ef: int = 128, entry: Optional[int] = None):
        """The whole sharded step behind the C ABI (needs `comm`): local search -> RCCL all-gather -> merge, one call, on the
        current stream; returns the global top-k (the same on every rank)."""
        from . import ENTRY_DESCEND, _lib
        assert self.comm is not None, "HipShard.search needs a Comm (zvdb_hip_comm); use ShardedSearch for torch.distributed collectives"
        L = _lib.hip()
        nq = queries.shape[0]
        ids, dd, cnt = self._out(nq, k)
        s = torch.cuda.current_stream().cuda_stream
        need = C.c_uint64(0)
        _lib.check(L.zvdb_hip_sharded_workspace(self.index._h, self.comm._h, nq, k, _lib.SHARDED_GRAPH if route == "graph" else mode, C.byref(need)))
        ws = self._workspace(need.value)
        if route == "graph":
            _lib.check(L.zvdb_hip_search_graph_sharded_dev(self.index._h, self.comm._h, queries.data_ptr(), nq, self.dim, k, ef,
                                                           ENTRY_DESCEND if entry is None else entry, self.lo, ws.data_ptr(), ws.numel(),
                                                           ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), None, s))
        else:
            _lib.check(L.zvdb_hip_search_bruteforce_sharded_dev(self.index._h, self.comm._h, queries.data_ptr(), nq, self.dim, k, mode, self.lo,
                                                                ws.data_ptr(), ws.numel(), ids.data_ptr(), dd.data_ptr(), cnt.data_ptr(), s))
        return ids, dd, cnt

    def merge(self, g_ids, g_dist, g_cnt, k: int):
        from . import merge_topk_dev
        G, nq, _ = g_ids.shape
        o_ids, o_dist, o_cnt = self._out(nq, k)
        merge_topk_dev(g_ids.data_ptr(), g_dist.data_ptr(), g_cnt.data_ptr(), G, nq, k, o_ids.data_ptr(),
                       o_dist.data_ptr(), o_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return o_ids, o_dist, o_cnt


class ShardedSearch:
    """search(queries, k) = local top-k on this rank's shard -> all-gather -> k-way merge, the collective over torch.distributed
    (a backend with a `comm` takes the C-ABI call instead: one RCCL all-gather inside libzvdb_hip)."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def _gather(self, mine: torch.Tensor) -> torch.Tensor:
        """[world][bytes] <- every rank's byte buffer (RCCL on device memory; gloo on host memory)."""
        gathered = torch.empty((self.world, mine.numel()), dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(gathered.view(-1), mine, group=self.group)
        return gathered

    def search(self, queries: torch.Tensor, k: int, mode: int = 0, **route_kw):
        if getattr(self.backend, "comm", None) is not None:
            return self.backend.search(queries, k, mode, **route_kw)
        ids, dd, cnt = self.backend.local_search(queries, k, mode, **route_kw)
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
