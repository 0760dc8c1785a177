"""The Pedersen Merkle tree of BASELINE config #5 (2^log_n u8 leaves) on the GPU, beside the CPU oracle on a bounded sample.
Prints one JSON line.  usage: merkle_build.py [log_n=18] [reps=5] [cpu_log_n=12]
  gpu_ms_resident   swm_merkle_tree_build_dev: leaves and nodes stay in HBM (19 kernel launches)
  gpu_ms_host       swm_merkle_tree_build: host leaves in, (2 n - 1) x 32 bytes of nodes out over PCIe
  cpu               oracle.c (ark-crypto-primitives' per-bit evaluation restated) on 2^cpu_log_n leaves, all threads and one"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import simpleworks_amd as swm
from simpleworks_amd import hash as H, marlin as M

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cpu_lg = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ctx = M.default_context()
rng = M.generate_rand()
leaf = H.PedersenCRH.setup(rng, H.LEAF_WINDOWS)
inner = H.PedersenCRH.setup(rng, H.TWO_TO_ONE_WINDOWS)
n = 1 << lg
leaves = np.random.default_rng(1).integers(0, 256, size=(n, 1), dtype=np.uint8)
d_leaves = ctx.to_device(leaves)
d_nodes = ctx.alloc((2 * n - 1) * 32)
for _ in range(2):
    ctx.merkle_tree_build_dev(leaf.h, inner.h, d_leaves, 1, n, d_nodes)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.merkle_tree_build_dev(leaf.h, inner.h, d_leaves, 1, n, d_nodes)
ctx.synchronize()
res = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for _ in range(reps):
    nodes = ctx.merkle_tree_build(leaf.h, inner.h, leaves)
host = (time.perf_counter() - t0) / reps
out = {"workload": "Pedersen Merkle tree, 2^%d u8 leaves (144 / 128 windows of 4 bits, ed-on-BLS12-377)" % lg, "hashes": 2 * n - 1,
       "gpu_ms_resident": res * 1e3, "gpu_ms_host": host * 1e3, "hashes_per_s_resident": (2 * n - 1) / res,
       "root": nodes[-1].tobytes()[::-1].hex()}
if cpu_lg:
    import oracle_lib as OL
    lib = OL.load()
    m = 1 << cpu_lg
    cpu = {}
    for th in (lib.oracle_max_threads(), 1):
        t0 = time.perf_counter()
        want = OL.merkle_tree(lib, leaf.generators, inner.generators, leaves[:m], threads=th)
        dt = time.perf_counter() - t0
        cpu["threads_%d" % th] = {"sample": "2^%d leaves" % cpu_lg, "s": dt, "hashes_per_s": (2 * m - 1) / dt,
                                  "full_size_estimate_s": dt * (2 * n - 1) / (2 * m - 1)}
    got = ctx.merkle_tree_build(leaf.h, inner.h, leaves[:m])
    out["cpu_oracle"] = cpu
    out["sample_tree_equal"] = bool(np.array_equal(got, want))
print(json.dumps(out))
