"""Multi-GPU sharding of independent planning episodes.

Episodes share no mutable state, so the only cross-rank step is one gather of fixed-stride result
records after the kernels: rank r plans episodes [r*E/G, (r+1)*E/G) (seed = global episode id, so
results do not depend on G), then the per-episode summary records and the best paths are
all-gathered -- `torch.distributed` backend "nccl" (= RCCL over xGMI) on the GPUs, "gloo" in the
CPU tests.  No all-reduce sits on the data path; the payload is a few MB, so the step is latency
bound (7 xGMI links x ~153 GB/s per GPU are nowhere near saturated)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world_size):
    """block partition of the episode index; the first (n_total % world_size) ranks get one extra"""
    base, extra = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_records(records, group=None):
    """all-gather equally sized per-rank record blocks.  records: uint8/float tensor [E_local, stride]
    (same shape on every rank).  Returns [world, E_local, stride]."""
    world = dist.get_world_size(group)
    shape = tuple(records.shape)
    out = torch.empty((world * shape[0],) + shape[1:], dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(out, records.contiguous(), group=group)  # concatenates along dim 0
    return out.view((world,) + shape)


def gather_paths(paths, lengths, group=None):
    """Two-phase gather of variable-length best paths.
    paths [n_local_elems, 7] f64 (concatenated root->leaf courses), lengths [E_local] int64.
    Phase 1 gathers the per-episode lengths, phase 2 the payload padded to the largest rank total.
    Returns (all_lengths [world, E_local], list over ranks of [n_r, 7] tensors)."""
    world = dist.get_world_size(group)
    lengths = lengths.to(torch.int64).contiguous()
    all_len = torch.empty(world * lengths.numel(), dtype=torch.int64, device=lengths.device)
    dist.all_gather_into_tensor(all_len, lengths, group=group)
    all_len = all_len.view(world, lengths.numel())
    totals = all_len.sum(dim=1)
    cap = int(totals.max().item())
    pad = torch.zeros((max(cap, 1), 7), dtype=paths.dtype, device=paths.device)
    n = int(lengths.sum().item())
    if n:
        pad[:n] = paths[:n]
    allp = torch.empty((world * pad.shape[0], 7), dtype=paths.dtype, device=paths.device)
    dist.all_gather_into_tensor(allp, pad, group=group)
    allp = allp.view(world, pad.shape[0], 7)
    return all_len, [allp[r, :int(totals[r].item())] for r in range(world)]


def summaries_to_tensor(summ, device):
    """numpy structured summaries -> uint8 tensor [E, itemsize] on `device`"""
    raw = np.ascontiguousarray(summ).view(np.uint8).reshape(len(summ), summ.dtype.itemsize)
    return torch.from_numpy(raw.copy()).to(device)


def tensor_to_summaries(t, dtype):
    a = t.cpu().numpy()
    return np.ascontiguousarray(a).reshape(-1, a.shape[-1]).view(dtype).reshape(a.shape[:-1])
