"""The N>1 path on CPU: gloo runs of the shard + gather logic bench.py uses, at world size 2 and at the target's world size 8
(eight processes: uneven and empty shards, config 4's 512 episodes, config 5's 12 500 episodes per rank as lengths)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E_total, q):
    import sys
    sys.path.insert(0, REPO)
    from auv_sim_amd import _lib, _prrt_lib, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = D.TorchGather()
    lo, hi = D.shard_range(E_total, rank, world)
    n = hi - lo  # uneven when E_total % world != 0: rank 0 holds one episode more
    summ = np.zeros(n, dtype=_lib.SUMMARY_DTYPE)
    psumm = np.zeros(n, dtype=_prrt_lib.PRRT_SUMMARY_DTYPE)  # the Planner_RRT record goes through the same helpers
    for i in range(n):
        e = lo + i  # global episode id
        summ[i]["best_leaf"] = e
        summ[i]["best_path_len"] = 3 + (e % 5)
        summ[i]["best_cost"] = [-(e + 0.5), 0, 0, 0]
        psumm[i]["steps"] = 100 + e
        psumm[i]["done"] = e % 2
        psumm[i]["path_len"] = (2 + e % 3) if e % 2 else 0
        psumm[i]["arc"] = [e, 0, 0, 0, 0, e + 0.25]
    lens = torch.from_numpy(summ["best_path_len"].astype(np.int64))
    paths = torch.zeros((int(lens.sum()), 7), dtype=torch.float64)
    pos = 0
    for i in range(n):
        L = int(lens[i])
        paths[pos:pos + L, 0] = lo + i
        paths[pos:pos + L, 1] = torch.arange(L, dtype=torch.float64)
        pos += L
    plens = torch.from_numpy(psumm["path_len"].astype(np.int64))
    ppaths = torch.zeros((int(plens.sum()), 5), dtype=torch.float64)
    pos = 0
    for i in range(n):
        L = int(plens[i])
        ppaths[pos:pos + L, 0] = lo + i
        pos += L
    rec = G.gather_records(D.summaries_to_tensor(summ, "cpu"))
    all_len, all_paths = G.gather_paths(paths, lens)
    prec = G.gather_records(D.summaries_to_tensor(psumm, "cpu"))
    pall_len, pall_paths = G.gather_paths(ppaths, plens)
    ok = True
    seen = []
    for r in range(world):
        rlo, rhi = D.shard_range(E_total, r, world)
        rs = D.tensor_to_summaries(rec[r], _lib.SUMMARY_DTYPE)
        ps = D.tensor_to_summaries(prec[r], _prrt_lib.PRRT_SUMMARY_DTYPE)
        ok &= len(rs) == rhi - rlo and len(ps) == rhi - rlo and len(all_len[r]) == rhi - rlo
        pos = ppos = 0
        for i in range(rhi - rlo):
            e = rlo + i
            ok &= int(rs[i]["best_leaf"]) == e and float(rs[i]["best_cost"][0]) == -(e + 0.5)
            ok &= int(ps[i]["steps"]) == 100 + e and float(ps[i]["arc"][5]) == e + 0.25
            L = int(all_len[r][i])
            ok &= L == 3 + (e % 5)
            seg = all_paths[r][pos:pos + L]
            ok &= bool((seg[:, 0] == e).all()) and bool((seg[:, 1] == torch.arange(L, dtype=torch.float64)).all())
            pos += L
            PL = int(pall_len[r][i])
            ok &= PL == ((2 + e % 3) if e % 2 else 0)
            ok &= bool((pall_paths[r][ppos:ppos + PL, 0] == e).all())
            ppos += PL
            seen.append(e)
        ok &= pos == all_paths[r].shape[0] and ppos == pall_paths[r].shape[0]
    ok &= seen == list(range(E_total))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    from auv_sim_amd import distributed as D
    for n in (0, 1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


import pytest


def _run_ranks(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)], sorted(res)


@pytest.mark.parametrize("E_total", [11, 12, 1])  # uneven split, even split, one rank with nothing
def test_two_rank_gather_gloo(E_total):
    _run_ranks(_worker, 2, E_total)


@pytest.mark.parametrize("E_total", [512, 9, 1])  # config 4 (64 per rank), one rank with two, seven ranks with nothing
def test_eight_rank_gather_gloo(E_total):
    """the target's world size: every rank rebuilds the whole result in global episode order (== the one-rank order)"""
    _run_ranks(_worker, 8, E_total)


# ---------------------------------------------------------------------------------------------------------------------
# RcclGather's own Python (count exchange, buffer sizing, slicing of the gathered blocks, path gather) with two ranks:
# the C-ABI entry points it calls are replaced by a stand-in with the same signatures and semantics that moves the bytes
# over gloo, so everything above the C boundary runs exactly as on the GPUs.
# ---------------------------------------------------------------------------------------------------------------------
class _FakeCtx:
    h = 1

    def _chk(self, rc):
        assert rc == 0, rc


class _FakeAbi:
    """auvp_comm_* / auvp_gather_* of include/auvplan.h over torch.distributed (gloo), host pointers"""

    def __init__(self):
        self.world = self.rank = None
        self.calls = []

    def auvp_comm_unique_id(self, buf):
        for i in range(128):
            buf[i] = (i * 7 + 3) % 256
        return 0

    def auvp_comm_init(self, h, world, rank, arr):
        assert bytes(arr) == bytes((i * 7 + 3) % 256 for i in range(128))  # rank 0's id reached this rank
        self.world, self.rank = world, rank
        return 0

    def auvp_comm_destroy(self, h):
        return 0

    def auvp_comm_info(self, h, w, r, n):
        import ctypes as C
        C.cast(w, C.POINTER(C.c_int32))[0] = self.world
        C.cast(r, C.POINTER(C.c_int32))[0] = self.rank
        C.cast(n, C.POINTER(C.c_int32))[0] = dist.get_world_size()
        return 0

    def auvp_last_gather_ms(self, h):
        return 0.25

    def auvp_gather_counts(self, h, nbytes, counts):
        self.calls.append("counts")
        mine = torch.tensor([int(nbytes)], dtype=torch.int64)
        allc = torch.empty(self.world, dtype=torch.int64)
        dist.all_gather_into_tensor(allc, mine)
        for r in range(self.world):
            counts[r] = int(allc[r])
        return 0

    def auvp_gather_blocks(self, h, send, recv, cap, counts):
        import ctypes as C
        self.calls.append("blocks")
        cs = [int(counts[r]) for r in range(self.world)]
        if sum(cs) > cap:
            return -3
        off = 0
        for r in range(self.world):  # one broadcast per rank, like the grouped ncclBroadcast of the library
            if cs[r] > 0:
                t = torch.empty(cs[r], dtype=torch.uint8)
                if r == self.rank:
                    C.memmove(t.data_ptr(), send.value, cs[r])
                dist.broadcast(t, src=r)
                C.memmove(recv.value + off, t.data_ptr(), cs[r])
            off += cs[r]
        return 0


    # ---- gather to a root: the stand-in moves the bytes over gloo send / recv when the call is made (the library enqueues
    # on its gather stream); auvp_gather_wait reports what the library would
    def auvp_gather_blocks_root_async(self, h, root, send, recv, cap, counts):
        import ctypes as C
        self.calls.append("root")
        cs = [int(counts[r]) for r in range(self.world)]
        if self.rank == root:
            if sum(cs) > cap:
                return -2
            off = 0
            for r in range(self.world):
                if cs[r] > 0:
                    t = torch.empty(cs[r], dtype=torch.uint8)
                    if r == root:
                        C.memmove(t.data_ptr(), send.value, cs[r])
                    else:
                        dist.recv(t, src=r)
                    C.memmove(recv.value + off, t.data_ptr(), cs[r])
                off += cs[r]
            self.pending_bytes = getattr(self, "pending_bytes", 0) + sum(cs)
        else:
            assert recv is None or not recv.value
            if cs[self.rank] > 0:
                t = torch.empty(cs[self.rank], dtype=torch.uint8)
                C.memmove(t.data_ptr(), send.value, cs[self.rank])
                dist.send(t, dst=root)
            self.pending_bytes = getattr(self, "pending_bytes", 0) + cs[self.rank]
        return 0

    def auvp_gather_wait(self, h, ms, nbytes):
        import ctypes as C
        self.calls.append("wait")
        C.cast(ms, C.POINTER(C.c_double))[0] = 0.5
        C.cast(nbytes, C.POINTER(C.c_int64))[0] = getattr(self, "pending_bytes", 0)
        self.pending_bytes = 0
        return 0


def _root_worker(rank, world, port, E_total, transport, q):
    """gather TO ONE RANK (root_begin / root_end) through either transport: fixed-stride records and lengths with static counts
    (no count exchange), variable-length paths with one exchange; the root rebuilds everything in global episode order, the
    other ranks receive nothing; a second ticket only after the first was ended"""
    import sys
    sys.path.insert(0, REPO)
    from auv_sim_amd import _lib, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def exchange(mine):
        box = [mine]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    abi = None
    if transport == "rccl":
        abi = _FakeAbi()
        G = D.RcclGather(_FakeCtx(), rank, world, exchange, L=abi)
    else:
        G = D.TorchGather()
    ok = True
    sizes = D.shard_sizes(E_total, world)
    lo, hi = D.shard_range(E_total, rank, world)
    n = hi - lo
    for root in (0, world - 1):
        summ = np.zeros(n, dtype=_lib.SUMMARY_DTYPE)
        for i in range(n):
            summ[i]["best_leaf"] = lo + i
            summ[i]["best_path_len"] = 1 + ((lo + i) % 6)
        rec = D.summaries_to_tensor(summ, "cpu")
        lens = torch.from_numpy(summ["best_path_len"].astype(np.int64))
        paths = torch.zeros((int(lens.sum()), 7), dtype=torch.float64)
        pos = 0
        for i in range(n):
            L = int(lens[i])
            paths[pos:pos + L, 0] = lo + i
            paths[pos:pos + L, 3] = torch.arange(L, dtype=torch.float64)
            pos += L
        calls0 = len(abi.calls) if abi else 0
        ticket = G.root_begin([rec, lens.reshape(-1, 1), paths], rows=[sizes, sizes, None], root=root)
        if abi:  # one count exchange (the paths), three enqueues, nothing waited for yet
            ok &= abi.calls[calls0:] == ["counts", "root", "root", "root"]
            try:
                G.root_begin([rec], rows=[sizes], root=root)
                ok = False
            except RuntimeError:
                pass
        got = G.root_end(ticket)
        if rank != root:
            ok &= got is None
            ok &= G.last_root_bytes == rec.numel() + 8 * n + paths.numel() * 8
            continue
        recs, lns, pths = got
        total_bytes = 0
        seen = []
        for r in range(world):
            rlo, rhi = D.shard_range(E_total, r, world)
            rs = D.tensor_to_summaries(recs[r], _lib.SUMMARY_DTYPE)
            ok &= len(rs) == rhi - rlo and lns[r].shape == (rhi - rlo, 1)
            pos = 0
            for i in range(rhi - rlo):
                e = rlo + i
                L = int(lns[r][i, 0])
                ok &= int(rs[i]["best_leaf"]) == e and L == 1 + (e % 6)
                seg = pths[r][pos:pos + L]
                ok &= bool((seg[:, 0] == e).all()) and bool((seg[:, 3] == torch.arange(L, dtype=torch.float64)).all())
                pos += L
                seen.append(e)
            ok &= pos == pths[r].shape[0]
            total_bytes += recs[r].numel() + 8 * (rhi - rlo) + pths[r].numel() * 8
        ok &= seen == list(range(E_total))
        if abi:
            ok &= G.last_root_bytes == total_bytes and G.last_root_ms == 0.5
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["rccl", "torch"])
@pytest.mark.parametrize("E_total", [9, 8, 1])
def test_gather_to_root_two_ranks(E_total, transport):
    _run_ranks(_root_worker, 2, E_total, transport)


@pytest.mark.parametrize("transport", ["rccl", "torch"])
@pytest.mark.parametrize("E_total", [512, 13, 3])
def test_gather_to_root_eight_ranks(E_total, transport):
    _run_ranks(_root_worker, 8, E_total, transport)


def _rccl_python_worker(rank, world, port, E_total, q):
    import sys
    sys.path.insert(0, REPO)
    from auv_sim_amd import _lib, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def exchange(mine):
        box = [mine]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    abi = _FakeAbi()
    G = D.RcclGather(_FakeCtx(), rank, world, exchange, L=abi)
    ok = G.info() == (world, rank, world)
    lo, hi = D.shard_range(E_total, rank, world)
    n = hi - lo
    summ = np.zeros(n, dtype=_lib.SUMMARY_DTYPE)
    for i in range(n):
        summ[i]["best_leaf"] = lo + i
        summ[i]["best_path_len"] = 2 + ((lo + i) % 4)
        summ[i]["nn_scanned"] = 1000 + lo + i
    rec = G.gather_records(D.summaries_to_tensor(summ, "cpu"))
    lens = torch.from_numpy(summ["best_path_len"].astype(np.int64))
    paths = torch.zeros((int(lens.sum()) + 3, 7), dtype=torch.float64)  # slack rows past sum(lengths) are not sent
    pos = 0
    for i in range(n):
        L = int(lens[i])
        paths[pos:pos + L, 0] = lo + i
        paths[pos:pos + L, 6] = torch.arange(L, dtype=torch.float64)
        pos += L
    all_len, blocks = G.gather_paths(paths, lens)
    seen = []
    for r in range(world):
        rlo, rhi = D.shard_range(E_total, r, world)
        rs = D.tensor_to_summaries(rec[r], _lib.SUMMARY_DTYPE)
        ok &= rec[r].shape == (rhi - rlo, _lib.SUMMARY_DTYPE.itemsize) and len(all_len[r]) == rhi - rlo
        pos = 0
        for i in range(rhi - rlo):
            e = rlo + i
            ok &= int(rs[i]["best_leaf"]) == e and int(rs[i]["nn_scanned"]) == 1000 + e
            L = int(all_len[r][i])
            ok &= L == 2 + (e % 4)
            seg = blocks[r][pos:pos + L]
            ok &= bool((seg[:, 0] == e).all()) and bool((seg[:, 6] == torch.arange(L, dtype=torch.float64)).all())
            pos += L
            seen.append(e)
        ok &= pos == blocks[r].shape[0]
    ok &= seen == list(range(E_total))
    # one count exchange per gather (not two), and no payload call when nobody has anything to send
    ok &= abi.calls.count("counts") == 3
    empty = G.gather_records(torch.zeros((0, 16), dtype=torch.uint8))
    ok &= all(e.shape == (0, 16) for e in empty) and abi.calls[-1] == "counts"
    ok &= G.take_ms() is not None and G.take_ms() is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("E_total", [9, 8, 1])  # uneven, even, one rank with an empty shard
def test_rccl_gather_python_two_ranks(E_total):
    _run_ranks(_rccl_python_worker, 2, E_total)


@pytest.mark.parametrize("E_total", [512, 13, 3])  # config 4; uneven; five ranks with an empty shard
def test_rccl_gather_python_eight_ranks(E_total):
    """RcclGather's Python at world size 8 (the C-ABI stand-in broadcasts from every root in turn, like the library's grouped
    ncclBroadcast): counts, buffer sizing, slicing, order"""
    _run_ranks(_rccl_python_worker, 8, E_total)


def _lengths_worker(rank, world, port, per_rank, q):
    """config 5's shape as LENGTHS only: 12 500 episodes per rank, variable path lengths, payload of one float per point"""
    import sys
    sys.path.insert(0, REPO)
    from auv_sim_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = D.TorchGather()
    E_total = per_rank * world
    lo, hi = D.shard_range(E_total, rank, world)
    ok = (hi - lo) == per_rank
    e = torch.arange(lo, hi, dtype=torch.int64)
    lens = (e * 7919) % 5                          # 0..4 points, a function of the global episode id
    paths = torch.repeat_interleave(e.to(torch.float64), lens).reshape(-1, 1)
    all_len, all_paths = G.gather_paths(paths, lens)
    for r in range(world):
        rlo, rhi = D.shard_range(E_total, r, world)
        ee = torch.arange(rlo, rhi, dtype=torch.int64)
        want_len = (ee * 7919) % 5
        ok &= bool((all_len[r].cpu() == want_len).all())
        ok &= bool((all_paths[r].cpu().reshape(-1) == torch.repeat_interleave(ee.to(torch.float64), want_len)).all())
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gather_of_config5_sized_shards():
    _run_ranks(_lengths_worker, 8, 12500)


def test_rccl_unloadable_is_an_error_code_not_a_crash():
    """AUVP_RCCL_LIBRARY pointing nowhere: auvp_comm_available() == 0, auvp_comm_unique_id -> AUVP_ERR_COMM, and the reason is
    readable (the round-2 build dereferenced a null dlerror() here)"""
    import subprocess
    import sys
    code = (
        "import ctypes as C, os, sys\n"
        "L = C.CDLL(os.path.join(%r, 'auv_sim_amd', 'libauvplan.so'))\n"
        "L.auvp_comm_library.restype = C.c_char_p\n"
        "assert L.auvp_comm_available() == 0\n"
        "assert L.auvp_comm_unique_id((C.c_uint8 * 128)()) == -6\n"
        "msg = L.auvp_comm_library().decode()\n"
        "assert 'cannot load RCCL' in msg and 'no_such_rccl' in msg, msg\n"
        "print('ok')\n" % REPO)
    env = dict(os.environ, AUVP_RCCL_LIBRARY="/nonexistent/libno_such_rccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)
