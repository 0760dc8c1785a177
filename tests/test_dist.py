"""CPU tests of the N > 1 path (gloo, world_size 2): point-range sharding + all-gather + fold of the MSM partials.
The kernel is injected; on CPU the oracle stands in for it (tests may call the oracle, the product never does)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from simpleworks_amd.dist import shard_range, sharded_msm


def test_shard_range_partitions():
    for n, w in ((10, 3), (1 << 20, 8), (7, 8), (0, 2)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle_lib import Oracle, golden, h2i, p64
    from pyref.prng import fr_array
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    G = orc.points_to_mont([tuple(h2i(v) for v in golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    sc = fr_array(n, 77)
    lo, hi = shard_range(n, world, rank)

    def add(a, b):
        out = np.zeros(18, dtype=np.uint64)
        orc.lib.oracle_g1_add(p64(a), p64(b), p64(out))
        return out

    total = sharded_msm(lambda: orc.msm(np.ascontiguousarray(bases[lo:hi]), np.ascontiguousarray(sc[lo:hi])), add)
    ref = orc.msm(bases, sc)
    q.put((rank, orc.jac_to_affine_int(total) == orc.jac_to_affine_int(ref)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [257, 1024])
def test_sharded_msm_world2_gloo(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


def _worker_bytes(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from simpleworks_amd.dist import make_byte_allgather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ag = make_byte_allgather()
    ok = True
    for nbytes in (192, 1, 4096):  # 192 = one XYZZ partial, the size swm_set_msm_sharding exchanges
        mine = bytes((rank * 31 + i) & 0xff for i in range(nbytes))
        want = b"".join(bytes((r * 31 + i) & 0xff for i in range(nbytes)) for r in range(world))
        ok = ok and ag(mine) == want
    q.put((rank, ok))
    dist.destroy_process_group()


def test_byte_allgather_world2_gloo():
    """The exchange step of the sharded prover (include/swmarlin.h swm_allgather_fn) over torch.distributed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bytes, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


# ---- one transform over two ranks (gloo): the four-step split with a single all-to-all, the oracle as the kernel
def _ntt_worker(rank, world, port, log_n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    from oracle_lib import Oracle, p64
    from pyref.prng import fr_array
    from simpleworks_amd.dist import sharded_ntt, blocks_rows, cyclic_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    n = 1 << log_n
    x = orc.fr_to_mont(fr_array(n, 31 + log_n))

    def alltoall(chunks):  # gloo has no all_to_all: every rank publishes its chunks, each picks its own (what the library's
        blk = chunks[0].shape[0]                                   # all-gather callback path does as well)
        mine = torch.from_numpy(np.concatenate(chunks).view(np.int64).copy())
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        return [e.numpy().view(np.uint64).reshape(-1, 4)[rank * blk:(rank + 1) * blk] for e in everyone]

    def mul_scalars(arr, ints):
        out = np.empty_like(arr)
        orc.lib.oracle_fr_mul(p64(np.ascontiguousarray(arr)), p64(orc.fr_mont_from_ints(ints)), p64(out), arr.shape[0])
        return out

    def add(a, b):
        out = np.empty_like(a)
        orc.lib.oracle_fr_add(p64(np.ascontiguousarray(a)), p64(np.ascontiguousarray(b)), p64(out), a.shape[0])
        return out

    local_ntt = lambda arr, lg, inv: orc.ntt(arr, lg, int(inv), 0, 1)
    ok = True
    for inverse in (False, True):
        ref = orc.ntt(x, log_n, int(inverse), 0, 2)
        cyc, blo = cyclic_rows(log_n, world, rank), blocks_rows(log_n, world, rank)
        got = sharded_ntt(np.ascontiguousarray(x[cyc]), log_n, rank, world, inverse, False, local_ntt, alltoall, mul_scalars, add)
        ok = ok and np.array_equal(got, ref[blo])            # CYCLIC in -> BLOCKS out
        got = sharded_ntt(np.ascontiguousarray(x[blo]), log_n, rank, world, inverse, True, local_ntt, alltoall, mul_scalars, add)
        ok = ok and np.array_equal(got, ref[cyc])            # BLOCKS in -> CYCLIC out
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("log_n", [4, 9])
def test_sharded_ntt_world2_gloo(log_n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ntt_worker, args=(r, 2, port, log_n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_sharded_ntt_single_process_worlds():
    """The same algorithm for 2, 4 and 8 ranks in one process (the all-to-all is a list transpose): layouts and twiddles."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle_lib import Oracle, p64
    from pyref.prng import fr_array
    from simpleworks_amd.dist import sharded_ntt, blocks_rows, cyclic_rows
    import threading
    orc = Oracle()

    def mul_scalars(arr, ints):
        out = np.empty_like(arr)
        orc.lib.oracle_fr_mul(p64(np.ascontiguousarray(arr)), p64(orc.fr_mont_from_ints(ints)), p64(out), arr.shape[0])
        return out

    def add(a, b):
        out = np.empty_like(a)
        orc.lib.oracle_fr_add(p64(np.ascontiguousarray(a)), p64(np.ascontiguousarray(b)), p64(out), a.shape[0])
        return out

    local_ntt = lambda arr, lg, inv: orc.ntt(arr, lg, int(inv), 0, 1)
    for world, log_n in ((2, 5), (4, 6), (8, 7)):
        x = orc.fr_to_mont(fr_array(1 << log_n, 5 + world))
        for inverse in (False, True):
            ref = orc.ntt(x, log_n, int(inverse), 0, 1)
            for blocks_in in (False, True):
                mailbox = [[None] * world for _ in range(world)]
                barrier = threading.Barrier(world)
                out = [None] * world

                def run(rank):
                    def alltoall(chunks):
                        for c in range(world):
                            mailbox[c][rank] = chunks[c]
                        barrier.wait(timeout=60)
                        return list(mailbox[rank])
                    rows_in = (blocks_rows if blocks_in else cyclic_rows)(log_n, world, rank)
                    out[rank] = sharded_ntt(np.ascontiguousarray(x[rows_in]), log_n, rank, world, inverse, blocks_in, local_ntt,
                                            alltoall, mul_scalars, add)
                ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join(timeout=120)
                for rank in range(world):
                    rows_out = (cyclic_rows if blocks_in else blocks_rows)(log_n, world, rank)
                    assert np.array_equal(out[rank], ref[rows_out]), (world, log_n, inverse, blocks_in, rank)
