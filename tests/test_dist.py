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
