"""simpleworks_amd — MI355X-native (gfx950 HIP) Marlin prover hot path behind simpleworks' src/marlin API.

Layout (only what the path needs):
  csrc/            hand-written HIP kernels (G1 MSM, Fr NTT, R1CS mat-vec, support kernels) + the C ABI
  libswmarlin.so   built in-tree by __graft_entry__.build() / `make -C simpleworks_amd/csrc`
  _lib.py          ctypes binding of include/swmarlin.h (fails loudly when the library or the GPU is missing)
  marlin.py        mirror of src/marlin/mod.rs (generate_universal_srs / ..._keys / generate_proof / verify_proof)
  serialization.py mirror of src/marlin/serialization.rs
  workloads.py     the BASELINE.json circuits; dist.py: one proof / one MSM over several GPUs

There is no CPU fallback anywhere in this package; the CPU oracle lives in oracle/ and is test infrastructure.
"""
from ._lib import SwmError, load_library, Context  # noqa: F401
