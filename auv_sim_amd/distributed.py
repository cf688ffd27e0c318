"""Multi-GPU sharding of independent planning episodes.

Episodes share no mutable state, so the only cross-rank step is one gather of result records after the
kernels: rank r plans episodes shard_range(E, r, G) (seed = global episode id, so results do not depend
on G), then the fixed-stride per-episode summary records and the variable-length best paths are gathered
to every rank.  No all-reduce sits on the data path; the payload is a few MB, so the step is latency
bound (7 xGMI links x ~153 GB/s per GPU are nowhere near saturated).

Two transports with one interface (`gather_records`, `gather_paths`):
  RcclGather   libauvplan.so's own RCCL entry points (auvp_comm_init / auvp_gather / auvp_gather_var,
               include/auvplan.h): collectives on the planner handle's HIP stream, device pointers in and
               out, no padding of the variable-length blocks.  What a C/C++ host would call; bench.py uses
               it on the GPUs.
  TorchGather  torch.distributed all-gathers ("nccl" = RCCL on ROCm, "gloo" in the CPU tests).
Shards may be uneven (the first E % G ranks hold one episode more): records are padded to the largest
shard for the equal-size collective and cut back on return.

Round 6: both transports also gather TO ONE RANK (`root_begin` / `root_end`) -- what north_star asks for.  With the
headline's batch the payload is not "a few MB": 12 288 episodes x ~170 path elements x 56 B = ~0.12 GB per rank and step,
which an all-gather delivers seven times over to EVERY rank of an 8-GPU node (~0.84 GB of ingress per rank per 100 ms step);
the gather to a root moves each rank's block once, every peer over its own xGMI link, and RcclGather enqueues it on the
handle's gather stream so that step k's transfer overlaps step k + 1's kernels (root_end(ticket) before the next
root_begin).  Fixed-stride blocks of a block partition need no count exchange (`rows` = the shard sizes).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world_size):
    """block partition of the episode index; the first (n_total % world_size) ranks get one extra"""
    base, extra = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_total, world_size):
    return [b - a for a, b in (shard_range(n_total, r, world_size) for r in range(world_size))]


class TorchGather:
    """all-gathers through torch.distributed (process group already initialised)"""

    name = "torch.distributed"

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.last_ms = None

    def _counts(self, n, device):
        mine = torch.tensor([int(n)], dtype=torch.int64, device=device)
        allc = torch.empty(self.world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        return [int(v) for v in allc.tolist()]

    def gather_records(self, records):
        """records [E_local, stride] (E_local may differ between ranks) -> list over ranks of [E_r, stride]"""
        records = records.contiguous()
        counts = self._counts(records.shape[0], records.device)
        cap = max(max(counts), 1)
        pad = torch.zeros((cap,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
        pad[:records.shape[0]] = records
        out = torch.empty((self.world * cap,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        out = out.view((self.world, cap) + tuple(records.shape[1:]))
        return [out[r, :counts[r]] for r in range(self.world)]

    def gather_paths(self, paths, lengths):
        """paths [n_local_elems, W] (concatenated per-episode blocks), lengths [E_local] int64 ->
        (list over ranks of lengths [E_r], list over ranks of [n_r, W])"""
        lens = self.gather_records(lengths.to(torch.int64).reshape(-1, 1))
        lens = [l.reshape(-1) for l in lens]
        n = int(lengths.sum().item())
        blocks = self.gather_records(paths[:n])
        return lens, blocks

    def take_ms(self):
        return None  # no stream timing on this transport

    # ---- gather to one rank (synchronous on this transport: root_begin does the work, root_end hands it out) ----
    def root_begin(self, tensors, rows=None, root=0):
        """tensors: list of [n_local, ...] tensors; rows: per tensor the list of every rank's row count, or None (one count
        exchange).  Returns a ticket for root_end."""
        rows = rows or [None] * len(tensors)
        out = []
        for t, known in zip(tensors, rows):
            t = t.contiguous()
            counts = [int(k) for k in known] if known is not None else self._counts(t.shape[0], t.device)
            cap = max(max(counts), 1)
            pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:t.shape[0]] = t
            bufs = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == root else None
            dist.gather(pad, bufs, dst=root, group=self.group)
            out.append([bufs[r][:counts[r]] for r in range(self.world)] if self.rank == root else None)
        return {"root": root, "out": out, "bytes": sum(t.numel() * t.element_size() for t in tensors)}

    def root_end(self, ticket):
        """on the root: per tensor the list over ranks of [n_r, ...]; None elsewhere"""
        self.last_root_bytes = ticket["bytes"]
        return ticket["out"] if self.rank == ticket["root"] else None


class RcclGather:
    """the C-ABI gather of libauvplan.so on the planner context's stream.  `exchange_id(id_bytes_or_None)` must
    return rank 0's 128-byte id on every rank (e.g. through torch.distributed.broadcast_object_list, MPI, a file)."""

    name = "rccl (auvp_gather, C-ABI)"

    @staticmethod
    def usable():
        """can this process reach RCCL through the C-ABI at all?  (every rank asks BEFORE the collective initialisation, so
        that a rank without it cannot leave the others waiting inside ncclCommInitRank)"""
        from . import _lib
        return _lib.load().auvp_comm_available() == 1  # binds the library only: no bootstrap root is started

    @staticmethod
    def library():
        """path of the RCCL image the C-ABI bound (or why none could be)"""
        from . import _lib
        L = _lib.load()
        L.auvp_comm_library.restype = C.c_char_p
        return (L.auvp_comm_library() or b"").decode()

    def __init__(self, ctx, rank, world, exchange_id, L=None):
        """`L`: the bound C-ABI (default: libauvplan.so); the CPU tests pass a stand-in with the same entry points"""
        from . import _lib
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        if L is not None:
            self.L = L
            self._join(exchange_id)
            return
        L = _lib.load()
        L.auvp_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
        L.auvp_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
        L.auvp_comm_destroy.argtypes = [C.c_void_p]
        L.auvp_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.auvp_gather_var.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.auvp_gather_counts.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.auvp_gather_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.auvp_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.auvp_last_gather_ms.argtypes = [C.c_void_p]
        L.auvp_last_gather_ms.restype = C.c_double
        L.auvp_gather_blocks_root_async.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.auvp_gather_wait.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        self.L = L
        self._join(exchange_id)

    def _join(self, exchange_id):
        from . import _lib
        L = self.L
        mine = None
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            mine = bytes(buf) if L.auvp_comm_unique_id(buf) == 0 else b""
        uid = exchange_id(mine)  # every rank takes part, also when rank 0 has nothing to hand out
        if not uid or len(uid) != 128:
            raise _lib.AuvpError(-6, "auvp_comm_unique_id failed on rank 0 (RCCL not loadable?)")
        arr = (C.c_uint8 * 128).from_buffer_copy(uid)
        self.ctx._chk(L.auvp_comm_init(self.ctx.h, self.world, self.rank, arr))
        self.last_ms = None

    def info(self):
        """(world size given, rank and rank count as the communicator itself reports them)"""
        w, r, n = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        self.ctx._chk(self.L.auvp_comm_info(self.ctx.h, C.byref(w), C.byref(r), C.byref(n)))
        return int(w.value), int(r.value), int(n.value)

    def close(self):
        if self.ctx is not None and getattr(self.ctx, "h", None):
            self.L.auvp_comm_destroy(self.ctx.h)
        self.ctx = None

    def _var(self, t):
        """variable-size all-gather of a contiguous device tensor's bytes -> (counts, uint8 tensor of all blocks)"""
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        counts = (C.c_int64 * self.world)()
        # one count exchange, then the payload: the caller sizes its buffer in between (auvp_gather_counts / _blocks)
        self.ctx._chk(self.L.auvp_gather_counts(self.ctx.h, nbytes, counts))
        ms = float(self.L.auvp_last_gather_ms(self.ctx.h))
        total = sum(counts)
        out = torch.empty(max(total, 1), dtype=torch.uint8, device=t.device)
        if total > 0:
            self.ctx._chk(self.L.auvp_gather_blocks(self.ctx.h, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), total, counts))
            ms += float(self.L.auvp_last_gather_ms(self.ctx.h))
        self.last_ms = (self.last_ms or 0.0) + ms
        return list(counts), out[:total]

    def gather_records(self, records):
        records = records.contiguous()
        row = int(np.prod(records.shape[1:])) * records.element_size() if records.dim() > 1 else records.element_size()
        counts, raw = self._var(records)
        out, pos = [], 0
        for r in range(self.world):
            blk = raw[pos:pos + counts[r]]
            pos += counts[r]
            out.append(blk.view(records.dtype).view((counts[r] // max(row, 1),) + tuple(records.shape[1:])))
        return out

    def gather_paths(self, paths, lengths):
        lens = [l.reshape(-1) for l in self.gather_records(lengths.to(torch.int64).reshape(-1, 1))]
        n = int(lengths.sum().item())
        blocks = self.gather_records(paths[:n])
        return lens, blocks

    def take_ms(self):
        """HIP-event time of the collectives since the last call (sum), then reset"""
        ms, self.last_ms = self.last_ms, None
        return ms

    # ---- gather to one rank on the handle's gather stream (auvp_gather_blocks_root_async / auvp_gather_wait) ----
    def root_begin(self, tensors, rows=None, root=0):
        """ENQUEUE the gather of `tensors` (contiguous device tensors [n_local, ...]) to `root`, ordered behind what the planner
        stream holds now; returns a ticket for root_end.  rows: per tensor the list of every rank's row count (fixed-stride
        blocks of a block partition: no count exchange) or None (one count exchange, which waits).  The tensors -- and what
        they alias: the planner's result buffers -- must stay untouched until root_end; one ticket at a time."""
        if getattr(self, "_ticket_open", False):
            raise RuntimeError("root_end() the previous ticket first: no other collective may be issued while transfers are pending")
        rows = rows or [None] * len(tensors)
        plan = []
        for t, known in zip(tensors, rows):  # every count exchange BEFORE the first enqueue
            t = t.contiguous()
            row = (int(np.prod(t.shape[1:])) if t.dim() > 1 else 1) * t.element_size()
            counts = (C.c_int64 * self.world)()
            if known is not None:
                for r in range(self.world):
                    counts[r] = int(known[r]) * row
            else:
                self.ctx._chk(self.L.auvp_gather_counts(self.ctx.h, t.numel() * t.element_size(), counts))
            plan.append((t, row, counts))
        items = []
        for t, row, counts in plan:
            total = sum(counts)
            recv = torch.empty(max(total, 1), dtype=torch.uint8, device=t.device) if self.rank == root else None
            self.ctx._chk(self.L.auvp_gather_blocks_root_async(self.ctx.h, int(root), C.c_void_p(t.data_ptr()),
                                                               C.c_void_p(recv.data_ptr()) if recv is not None else None,
                                                               total if recv is not None else 0, counts))
            items.append((t, row, list(counts), recv))
        self._ticket_open = True
        return {"root": int(root), "items": items}

    def root_end(self, ticket):
        """wait for the ticket's transfers.  On the root: per tensor the list over ranks of [n_r, ...] views of the received
        bytes; None elsewhere.  The transfer's stream time and this rank's bytes go to last_root_ms / last_root_bytes."""
        ms, nb = C.c_double(0.0), C.c_int64(0)
        self.ctx._chk(self.L.auvp_gather_wait(self.ctx.h, C.byref(ms), C.byref(nb)))
        self._ticket_open = False
        self.last_root_ms, self.last_root_bytes = float(ms.value), int(nb.value)
        if self.rank != ticket["root"]:
            return None
        out = []
        for t, row, counts, recv in ticket["items"]:
            blocks, pos = [], 0
            for r in range(self.world):
                blk = recv[pos:pos + counts[r]]
                pos += counts[r]
                blocks.append(blk.view(t.dtype).view((counts[r] // max(row, 1),) + tuple(t.shape[1:])))
            out.append(blocks)
        return out


def summaries_to_tensor(summ, device):
    """numpy structured summaries -> uint8 tensor [E, itemsize] on `device`"""
    raw = np.ascontiguousarray(summ).view(np.uint8).reshape(len(summ), summ.dtype.itemsize)
    return torch.from_numpy(raw.copy()).to(device)


def device_records(ptr, n, itemsize, device):
    """uint8 view [n, itemsize] of device-resident records (e.g. auvp_rrt_summaries_dev): no host round trip.
    The memory stays owned by the planner handle; the view is only valid until the next batch."""
    class _Mem:
        pass
    m = _Mem()
    m.__cuda_array_interface__ = {"shape": (int(n), int(itemsize)), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(m, device=device)


def tensor_to_summaries(t, dtype):
    a = t.cpu().numpy()
    return np.ascontiguousarray(a).reshape(-1, a.shape[-1]).view(dtype).reshape(a.shape[:-1])
