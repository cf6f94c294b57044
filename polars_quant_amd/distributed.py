"""Symbol sharding across the GPUs of one node.

The path has no cross-symbol dependency (independent capital pools): rank r of G owns the contiguous symbol range
[floor(N*r/G), floor(N*(r+1)/G)) -- with the symbol-major layout a contiguous byte range of every column -- and
computes on it with no communication.  The only exchange is the per-symbol summary table ([n, 8] f64, 64 B per
symbol): one all_gather over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_symbols: int, rank: int, world: int) -> tuple[int, int]:
    return (n_symbols * rank) // world, (n_symbols * (rank + 1)) // world


def gather_summaries(local: torch.Tensor, n_symbols: int, group=None) -> torch.Tensor:
    """local: [n_local, 8] summary rows of this rank's shard -> [n_symbols, 8] on every rank (symbol order)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(n_symbols, r, world)[1] - shard_range(n_symbols, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((n_symbols, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged shards: pad to the largest, gather, trim
    m = max(sizes)
    pad = torch.zeros((m, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:k] for p, k in zip(parts, sizes)], dim=0)


# ---- BASELINE config 4 (cross-sectional factor IC / Rank-IC): the parallel axis is the DAY, not the symbol (SURVEY 8e exception)
# A day's IC needs every symbol of that day, so the symbol-sharded layout of the rest of the path does not fit: rank r owns
# the days [shard_range(T, r, G)) with ALL symbols.  If the columns arrive symbol-sharded (the layout every other call
# uses), one all-to-all re-shards them -- rank r sends rank q the day-slice q of its symbol rows, 1/G of its data per peer:
# at 10 000 x 5 040 that is 50 MB per column and GPU in total, ~6.3 MB per peer, spread over the 7 xGMI links at once -- and
# one all_gather of the [T_local] IC series puts the full series on every rank.
def days_all_to_all(local_cols: torch.Tensor, n_symbols: int, group=None) -> torch.Tensor:
    """local_cols: [n_local, T] (this rank's symbol shard, all days) -> [n_symbols, T_local] (all symbols, this rank's days)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    T = local_cols.shape[1]
    if world == 1:
        return local_cols
    t_lo, t_hi = shard_range(T, rank, world)
    send = [local_cols[:, shard_range(T, q, world)[0]:shard_range(T, q, world)[1]].contiguous() for q in range(world)]
    recv = [torch.empty((shard_range(n_symbols, q, world)[1] - shard_range(n_symbols, q, world)[0], t_hi - t_lo),
                        dtype=local_cols.dtype, device=local_cols.device) for q in range(world)]
    # point-to-point pairs (xGMI is point-to-point: every peer has its own link) rather than the all_to_all collective, which
    # the gloo backend used by the CPU tests does not implement
    recv[rank].copy_(send[rank])
    ops = []
    for q in range(world):
        if q != rank:
            ops.append(dist.P2POp(dist.isend, send[q], q, group))
            ops.append(dist.P2POp(dist.irecv, recv[q], q, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    return torch.cat(recv, dim=0)


def gather_day_series(local: torch.Tensor, T: int, group=None) -> torch.Tensor:
    """local: [T_local] values of this rank's days -> [T] on every rank (ragged shards padded to the longest)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(T, r, world)[1] - shard_range(T, r, world)[0] for r in range(world)]
    m = max(sizes)
    pad = torch.zeros(m, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:k] for p, k in zip(parts, sizes)])


def factor_ic_day_sharded(factor_local, fwd_local, n_symbols: int, method: int = 0, compute=None, group=None):
    """Day-sharded IC / Rank-IC of symbol-sharded inputs [n_local, T]: all-to-all -> per-day IC on this rank's days -> gather.
    `compute(factor [N, T_local], fwd [N, T_local], method) -> (ic [T_local], n_valid [T_local])`; default: the HIP path."""
    if compute is None:
        from . import api
        compute = lambda f, r, m: api.factor_ic(f, r, method=m)
    T = factor_local.shape[1]
    f = days_all_to_all(factor_local, n_symbols, group)
    r = days_all_to_all(fwd_local, n_symbols, group)
    ic, nv = compute(f, r, method)
    return gather_day_series(ic, T, group), gather_day_series(nv, T, group)


class CabiComm:
    """The C ABI's own communicator (pq_comm_init / pq_gather_summaries, csrc/comm.hip): what a non-Python host uses.  The
    128-byte rendezvous id is made by rank 0 and shipped here through torch.distributed (a Rust host would use its own channel)."""

    def __init__(self, device, rank: int, world: int, group=None):
        import ctypes as C

        from . import api
        from ._lib import check, lib
        self._C, self._check, self._lib = C, check, lib()
        self.device = torch.device(device)
        self.h = api.ctx(self.device.index)
        self.stream = torch.cuda.current_stream(self.device)   # the communicator lives on this context = this stream
        idbuf = torch.zeros(129, dtype=torch.uint8)   # [ok flag, 128 id bytes]: every rank learns whether rank 0 could make the id
        if rank == 0:
            raw = (C.c_ubyte * 128)()
            if self._lib.pq_comm_unique_id(raw) == 0:
                idbuf = torch.tensor([1] + list(raw), dtype=torch.uint8)
        if world > 1:
            on = idbuf.to(self.device) if dist.get_backend(group) == "nccl" else idbuf
            dist.broadcast(on, src=0, group=group)
            idbuf = on.cpu()
        if int(idbuf[0]) != 1:
            raise RuntimeError("pq_comm_unique_id failed on rank 0 (is RCCL loadable?)")
        idbuf = idbuf[1:]
        raw = (C.c_ubyte * 128)(*idbuf.tolist())
        with torch.cuda.device(self.device):
            check(self._lib.pq_comm_init(self.h, rank, world, raw))
        self.rank, self.world = rank, world

    def gather_summaries(self, local: torch.Tensor, n_symbols: int) -> torch.Tensor:
        C = self._C
        lo, hi = shard_range(n_symbols, self.rank, self.world)
        if local.shape[0] != hi - lo:
            raise ValueError(f"rank {self.rank} of {self.world} owns {hi - lo} of {n_symbols} symbols, got {local.shape[0]} summary rows")
        cur = torch.cuda.current_stream(self.device)
        out = torch.empty((n_symbols, local.shape[1]), dtype=local.dtype, device=local.device)
        loc = local.contiguous()
        if cur != self.stream:        # the collective runs on the communicator's stream: order it behind the producer of `local` ...
            self.stream.wait_stream(cur)
            out.record_stream(self.stream); loc.record_stream(self.stream)
        with torch.cuda.device(self.device):
            self._check(self._lib.pq_gather_summaries(self.h, C.c_void_p(loc.data_ptr()), n_symbols, C.c_void_p(out.data_ptr())))
        if cur != self.stream:        # ... and the consumers of `out` behind the collective
            cur.wait_stream(self.stream)
        return out

    def gather_begin(self, local: torch.Tensor, n_symbols: int, out: torch.Tensor, slot: int) -> None:
        """pq_gather_summaries_begin: the exchange of `local` ([n_local, 8], contiguous, written by work already enqueued on the
        context's stream) into `out` ([n_symbols, 8]) on the communicator's own stream; returns at once"""
        C = self._C
        lo, hi = shard_range(n_symbols, self.rank, self.world)
        if local.shape[0] != hi - lo or not local.is_contiguous() or not out.is_contiguous() or out.shape[0] != n_symbols:
            raise ValueError(f"rank {self.rank} of {self.world}: local must be the contiguous [{hi - lo}, 8] shard, out [{n_symbols}, 8]")
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:        # the context lives on another stream than the step: bridge (device-side waits only)
            self.stream.wait_stream(cur)
        with torch.cuda.device(self.device):
            self._check(self._lib.pq_gather_summaries_begin(self.h, C.c_void_p(local.data_ptr()), n_symbols, C.c_void_p(out.data_ptr()), slot))

    def gather_end(self, slot: int) -> None:
        """pq_gather_summaries_end: the context's stream -- and torch's current stream -- wait (on the device) for that slot's exchange"""
        with torch.cuda.device(self.device):
            self._check(self._lib.pq_gather_summaries_end(self.h, slot))
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_stream(self.stream)

    def sync(self) -> None:
        """pq_comm_sync: block the HOST until the communicator's own stream is idle"""
        with torch.cuda.device(self.device):
            self._check(self._lib.pq_comm_sync(self.h))

    def trial(self, timeout_note: str = "") -> torch.Tensor:
        """one exchange of [1, 8] rows per rank on the communicator's OWN stream, waited for on the host: the stream the steps run on is
        not touched by a collective that never returns (call from a helper thread with a deadline) -> the gathered [world, 8] table"""
        loc = torch.full((1, 8), float(self.rank), dtype=torch.float64, device=self.device)
        out = torch.empty((self.world, 8), dtype=torch.float64, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()
        self.gather_begin(loc, self.world, out, 0)
        self.sync()
        self.gather_end(0)
        return out

    def close(self):
        with torch.cuda.device(self.device):
            self._check(self._lib.pq_comm_destroy(self.h))


class OverlappedGather:
    """The per-step exchange taken off the step's critical path: two slots of (`local`, `all`) buffers; the exchange of step k runs
    beside the kernels of step k + 1 and is waited for only when its slot comes round again (or at `drain`).

        og = OverlappedGather(n_symbols, n_local, device, comm=CabiComm(...))     # GPU: pq_gather_summaries_begin / _end
        og = OverlappedGather(n_symbols, n_local, "cpu", group=None)              # any torch.distributed backend (tests: gloo)
        for k in range(steps):
            slot = og.acquire()          # waits (device-side / handle) for the exchange that last used this slot
            ... run the step that writes og.local[slot] ...
            og.begin(slot)               # returns at once
        og.drain()                       # all[0], all[1] hold the tables of the last two steps

    `all[slot]` may be read after the next acquire() of that slot or after drain()."""

    def __init__(self, n_symbols: int, n_local: int, device, comm: "CabiComm | None" = None, group=None, local=None):
        self.n, self.comm, self.group = n_symbols, comm, group
        dev = torch.device(device)
        self.local = list(local) if local is not None else [torch.zeros((n_local, 8), dtype=torch.float64, device=dev) for _ in range(2)]
        self.all = [torch.zeros((n_symbols, 8), dtype=torch.float64, device=dev) for _ in range(2)]
        self._work = [None, None]      # torch.distributed work handles (the non-C-ABI backend)
        self._next = 0
        self.world = comm.world if comm is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = comm.rank if comm is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self._sizes = [shard_range(n_symbols, r, self.world)[1] - shard_range(n_symbols, r, self.world)[0] for r in range(self.world)]

    def acquire(self) -> int:
        slot = self._next
        self._next ^= 1
        self.wait(slot)
        return slot

    def wait(self, slot: int) -> None:
        if self.comm is not None:
            self.comm.gather_end(slot)
        elif self._work[slot] is not None:
            for w in self._work[slot]:
                w.wait()
            self._work[slot] = None

    def begin(self, slot: int) -> None:
        if self.comm is not None:
            self.comm.gather_begin(self.local[slot], self.n, self.all[slot], slot)
            return
        if self.world == 1:
            self.all[slot].copy_(self.local[slot])
            return
        if len(set(self._sizes)) == 1:
            self._work[slot] = [dist.all_gather_into_tensor(self.all[slot], self.local[slot], group=self.group, async_op=True)]
        else:   # ragged shards: every rank's rows are broadcast into their place
            lo, hi = shard_range(self.n, self.rank, self.world)
            self.all[slot][lo:hi].copy_(self.local[slot])
            self._work[slot] = []
            for r in range(self.world):
                a, b = shard_range(self.n, r, self.world)
                if b > a:
                    self._work[slot].append(dist.broadcast(self.all[slot][a:b], src=dist.get_global_rank(self.group, r) if self.group else r,
                                                           group=self.group, async_op=True))

    def drain(self) -> None:
        self.wait(0)
        self.wait(1)
