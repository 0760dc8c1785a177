"""ctypes wrapper of tests/native/dropin_harness.cpp: the per-proof call sequence of the Rust binding (swmarlin-sys/src/marlin.rs
`prove`, behind /root/reference/src/marlin/mod.rs:70-77) replayed against the C ABI from T threads that share ONE resident key.
Test infrastructure and a measurement leg of bench.py; built by __graft_entry__.build() (g++, no HIP) into tests/native/_build/."""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "dropin_harness.cpp")
OUT = os.path.join(ROOT, "tests", "native", "_build", "libdropin.so")
_lib = None


def build(force=False):
    """g++ -shared tests/native/dropin_harness.cpp against the in-tree libswmarlin.so (rpath relative to the harness)."""
    lib = os.path.join(ROOT, "simpleworks_amd", "libswmarlin.so")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", os.path.join(ROOT, "simpleworks_amd"), "-lswmarlin",
                           "-Wl,-rpath,$ORIGIN/../../../simpleworks_amd", "-lpthread", "-o", OUT])
    return OUT


def load():
    global _lib
    if _lib is None:
        import simpleworks_amd._lib as L
        L.load_library()  # the product library first: the harness binds to the copy the process already uses
        if not os.path.exists(OUT):
            build()
        lib = ctypes.CDLL(OUT)
        lib.dropin_run.restype = ctypes.c_int
        lib.dropin_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                   ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                   ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        _lib = lib
    return _lib


def run(pk, vk, packed, threads, proofs_per_thread, rng_key, rng_word_pos, pack_mode="view", device=None):
    """-> (report dict, [last proof bytes of every thread]).  `packed`: anything with .instance / .witness (n x 4 uint64
    Montgomery limbs) and .num_constraints (simpleworks_amd.marlin.PackedR1cs / AssignmentOnly)."""
    lib = load()
    inst = np.ascontiguousarray(packed.instance, dtype=np.uint64)
    wit = np.ascontiguousarray(packed.witness, dtype=np.uint64)
    proofs = np.zeros((threads, 1024), dtype=np.uint8)
    lens = (ctypes.c_size_t * threads)()
    buf = ctypes.create_string_buffer(4096)
    rc = lib.dropin_run(pk.ctx.device if device is None else device, pk.h, vk.h, inst.ctypes.data, inst.shape[0],
                        wit.ctypes.data if wit.size else None, wit.shape[0], packed.num_constraints, threads, proofs_per_thread,
                        1 if pack_mode == "copy" else 0, bytes(rng_key), rng_word_pos, proofs.ctypes.data, lens, buf, len(buf))
    report = json.loads(buf.value.decode() or "{}") if buf.value else {}
    if rc != 0:
        raise RuntimeError("dropin_run failed: status %d %s" % (rc, report.get("error", "")))
    return report, [bytes(proofs[t, : lens[t]]) for t in range(threads)]
