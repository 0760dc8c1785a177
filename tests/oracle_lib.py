"""ctypes binding of oracle/_build/liboracle.so (the CPU checker).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY — never by simpleworks_amd/."""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

R = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
Q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
M64 = (1 << 64) - 1

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load(path=None):
    if path is None:
        if not os.path.exists(LIB_PATH):
            build()
        path = LIB_PATH
    lib = ctypes.CDLL(path)
    sz = ctypes.c_size_t
    for name, args in {
        "oracle_fr_to_mont": [_u64p, _u64p, sz], "oracle_fr_from_mont": [_u64p, _u64p, sz],
        "oracle_fq_to_mont": [_u64p, _u64p, sz], "oracle_fq_from_mont": [_u64p, _u64p, sz],
        "oracle_fr_mul": [_u64p, _u64p, _u64p, sz], "oracle_fr_add": [_u64p, _u64p, _u64p, sz],
        "oracle_fr_sub": [_u64p, _u64p, _u64p, sz], "oracle_fq_mul": [_u64p, _u64p, _u64p, sz],
        "oracle_fr_inv": [_u64p, _u64p, sz], "oracle_batch_inverse_fr": [_u64p, sz],
        "oracle_g1_add_mixed": [_u64p, _u64p, _u64p], "oracle_g1_double": [_u64p, _u64p],
        "oracle_g1_add": [_u64p, _u64p, _u64p],
        "oracle_g1_fixed_base_mul": [_u64p, _u64p, sz, _u64p, ctypes.c_int],
        "oracle_msm_g1": [_u64p, _u64p, sz, _u64p, ctypes.c_int],
        "oracle_ntt_fr": [_u64p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int],
        "oracle_spmv_fr": [_u32p, _u32p, _u64p, _u64p, _u64p, sz],
        "oracle_spmv_fr_mt": [_u32p, _u32p, _u64p, _u64p, _u64p, sz, ctypes.c_int],
        "oracle_pedersen_hash": [_u64p, sz, sz, ctypes.c_void_p, sz, sz, ctypes.c_void_p, ctypes.c_int],
        "oracle_merkle_tree": [_u64p, sz, _u64p, sz, sz, ctypes.c_void_p, sz, sz, ctypes.c_void_p, ctypes.c_int],
    }.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None
    lib.oracle_g1_to_affine.argtypes = [_u64p, _u64p]
    lib.oracle_g1_to_affine.restype = ctypes.c_int
    lib.oracle_g1_is_on_curve.argtypes = [_u64p]
    lib.oracle_g1_is_on_curve.restype = ctypes.c_int
    lib.oracle_msm_window.argtypes = [sz]
    lib.oracle_msm_window.restype = ctypes.c_uint
    lib.oracle_max_threads.restype = ctypes.c_int
    return lib


def p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def p32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u32p)


# ---- int <-> limb arrays
def ints_to_limbs(vals, nlimbs):
    out = np.empty((len(vals), nlimbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(nlimbs):
            out[i, k] = (v >> (64 * k)) & M64
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, arr.shape[-1])
    return [sum(int(arr[i, k]) << (64 * k) for k in range(arr.shape[1])) for i in range(arr.shape[0])]


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


# ---- what arkworks ITSELF produced (tests/golden/arkworks/, written by swmarlin-sys/tests/pin_golden.rs when someone with a Rust
# toolchain runs it: `cargo test --features pin --test pin_golden`).  Absent in this repository until then: the oracle is unpinned.
ARKWORKS_DIR = os.environ.get("SWM_ARKWORKS_GOLDEN") or os.path.join(GOLDEN, "arkworks")


def arkworks_schema():
    with open(os.path.join(GOLDEN, "arkworks_schema.json")) as f:
        return {k: v for k, v in json.load(f).items() if not k.startswith("_")}


def arkworks_golden(name, directory=None):
    """The arkworks-generated twin of tests/golden/<name>, or None when nobody has run the pin kit yet."""
    path = os.path.join(directory or ARKWORKS_DIR, name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return {k: v for k, v in json.load(f).items() if not k.startswith("_")}


def _dotted(d, key):
    for part in key.split("."):
        d = d[part]
    return d


def arkworks_project(name, model):
    """The part of the model's golden file `name` that the pin kit records: the arkworks schema applied to our own data (what a
    fixture file produced by arkworks has to equal, key for key)."""
    spec = arkworks_schema()[name]
    if "per_case" in spec:
        keys = spec["per_case"]["required"] + spec["per_case"].get("optional", [])
        return {case: {k: v[k] for k in keys if k in v} for case, v in model.items() if isinstance(v, dict) and
                all(k in v for k in spec["per_case"]["required"])}
    out = {}
    for key in spec["keys"]:
        cur = out
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = _dotted(model, key)
    return out


def arkworks_diff(name, model, theirs):
    """Differences between the model's golden file and an arkworks fixture of the same name, as 'file: case.key' strings; a
    fixture may hold a subset of the cases, but what it holds must carry every required key and agree."""
    spec = arkworks_schema()[name]
    diffs = []
    if "per_case" in spec:
        for case, rec in theirs.items():
            if case not in model:
                diffs.append("%s: case %s is not in the model's file" % (name, case))
                continue
            for k in spec["per_case"]["required"]:
                if k not in rec:
                    diffs.append("%s: %s.%s missing from the arkworks fixture" % (name, case, k))
            for k, v in rec.items():
                if k in model[case] and model[case][k] != v:
                    diffs.append("%s: %s.%s differs (arkworks %s..., model %s...)" % (name, case, k, str(v)[:24], str(model[case][k])[:24]))
    else:
        for key in spec["keys"]:
            try:
                v = _dotted(theirs, key)
            except (KeyError, TypeError):
                continue
            if _dotted(model, key) != v:
                diffs.append("%s: %s differs (arkworks %s..., model %s...)" % (name, key, str(v)[:24], str(_dotted(model, key))[:24]))
    return diffs


def expected_bytes(name, case, key):
    """[(source, hex)]: what a GPU golden-bytes test compares the HIP path with — the model's value, and arkworks' own when the
    pin kit's fixtures are present."""
    out = [("model", golden(name)[case][key])]
    ark = arkworks_golden(name)
    if ark is not None and case in ark and key in ark[case]:
        out.append(("arkworks", ark[case][key]))
    return out



def h2i(s):
    return int(s, 16)


class Oracle:
    """numpy-level convenience API over the C oracle."""

    def __init__(self, path=None):
        self.lib = load(path)

    # Fr / Fq conversions
    def fr_to_mont(self, std):
        out = np.empty_like(std)
        self.lib.oracle_fr_to_mont(p64(std), p64(out), std.shape[0])
        return out

    def fr_from_mont(self, m):
        out = np.empty_like(m)
        self.lib.oracle_fr_from_mont(p64(m), p64(out), m.shape[0])
        return out

    def fq_to_mont(self, std):
        out = np.empty_like(std)
        self.lib.oracle_fq_to_mont(p64(std), p64(out), std.shape[0])
        return out

    def fq_from_mont(self, m):
        out = np.empty_like(m)
        self.lib.oracle_fq_from_mont(p64(m), p64(out), m.shape[0])
        return out

    def fr_mont_from_ints(self, vals):
        return self.fr_to_mont(ints_to_limbs(vals, 4))

    def fr_ints_from_mont(self, m):
        return limbs_to_ints(self.fr_from_mont(np.ascontiguousarray(m)))

    # points: list of (x, y) ints or None -> (n, 12) Montgomery
    def points_to_mont(self, pts):
        std = np.zeros((len(pts) * 2, 6), dtype=np.uint64)
        for i, P in enumerate(pts):
            if P is not None:
                std[2 * i] = ints_to_limbs([P[0]], 6)[0]
                std[2 * i + 1] = ints_to_limbs([P[1]], 6)[0]
        m = self.fq_to_mont(std)
        return np.ascontiguousarray(m.reshape(len(pts), 12))

    def points_from_mont(self, m):
        m = np.ascontiguousarray(m).reshape(-1, 12)
        std = self.fq_from_mont(np.ascontiguousarray(m.reshape(-1, 6)))
        v = limbs_to_ints(std)
        out = []
        for i in range(m.shape[0]):
            x, y = v[2 * i], v[2 * i + 1]
            out.append(None if x == 0 and y == 0 else (x, y))
        return out

    def jac_to_affine_int(self, jac18):
        aff = np.zeros(12, dtype=np.uint64)
        inf = self.lib.oracle_g1_to_affine(p64(np.ascontiguousarray(jac18)), p64(aff))
        if inf:
            return None
        return self.points_from_mont(aff.reshape(1, 12))[0]

    def msm(self, bases_mont, scalars_std, threads=1):
        n = scalars_std.shape[0]
        out = np.zeros(18, dtype=np.uint64)
        self.lib.oracle_msm_g1(p64(bases_mont), p64(scalars_std), n, p64(out), threads)
        return out

    def ntt(self, data_mont, log_n, inverse=0, coset=0, threads=1):
        d = np.ascontiguousarray(data_mont).copy()
        self.lib.oracle_ntt_fr(p64(d), log_n, inverse, coset, threads)
        return d

    def spmv(self, rowptr, col, val_mont, z_mont, threads=1):
        rows = len(rowptr) - 1
        out = np.zeros((rows, 4), dtype=np.uint64)
        self.lib.oracle_spmv_fr_mt(p32(rowptr), p32(col), p64(val_mont), p64(z_mont), p64(out), rows, threads)
        return out

    def fixed_base_mul(self, base_mont12, scalars_std, threads=None):
        n = scalars_std.shape[0]
        out = np.zeros((n, 12), dtype=np.uint64)
        if threads is None:
            threads = self.lib.oracle_max_threads()
        self.lib.oracle_g1_fixed_base_mul(p64(base_mont12), p64(scalars_std), n, p64(out), threads)
        return out

    def srs_bases(self, n, tau, gen_mont12):
        """P_i = [tau^i] G, i < n — the SRS shape KZG10::setup produces (bench/test inputs)."""
        pw = np.empty((n, 4), dtype=np.uint64)
        # powers of tau via the oracle's Fr arithmetic (Montgomery), then back to standard form
        cur = self.fr_mont_from_ints([1])
        t = self.fr_mont_from_ints([tau])
        # doubling trick: pw[0:k] known -> pw[k:2k] = pw[0:k] * tau^k
        pwm = np.empty((n, 4), dtype=np.uint64)
        pwm[0] = cur[0]
        k = 1
        tk = t.copy()
        while k < n:
            m = min(k, n - k)
            rep = np.ascontiguousarray(np.repeat(tk, m, axis=0))
            seg = np.ascontiguousarray(pwm[:m])
            out = np.empty((m, 4), dtype=np.uint64)
            self.lib.oracle_fr_mul(p64(seg), p64(rep), p64(out), m)
            pwm[k:k + m] = out
            k += m
            tk2 = np.empty_like(tk)
            self.lib.oracle_fr_mul(p64(tk), p64(tk), p64(tk2), 1)
            tk = tk2
        pw = self.fr_from_mont(pwm)
        return self.fixed_base_mul(gen_mont12, pw)


# ---- Pedersen CRH + Merkle tree (oracle.c; generators as [[(x, y)]] integers in standard form)
def _gens_limbs(gens):
    flat = [c for row in gens for pt in row for c in pt]
    return np.array([[(v >> (64 * k)) & M64 for k in range(4)] for v in flat], dtype=np.uint64).reshape(-1)


def pedersen_hash(lib, gens, inputs, threads=1):
    """inputs: uint8 [count, len] -> uint8 [count, 32]"""
    a = np.ascontiguousarray(inputs, dtype=np.uint8)
    g = _gens_limbs(gens)
    out = np.empty((a.shape[0], 32), dtype=np.uint8)
    lib.oracle_pedersen_hash(p64(g), len(gens), len(gens[0]), a.ctypes.data, a.shape[1], a.shape[0], out.ctypes.data, threads)
    return out


def merkle_tree(lib, leaf_gens, inner_gens, leaves, threads=1):
    """leaves: uint8 [n, leaf_len] -> uint8 [2 n - 1, 32]"""
    a = np.ascontiguousarray(leaves, dtype=np.uint8)
    gl, gi = _gens_limbs(leaf_gens), _gens_limbs(inner_gens)
    out = np.empty((2 * a.shape[0] - 1, 32), dtype=np.uint8)
    lib.oracle_merkle_tree(p64(gl), len(leaf_gens), p64(gi), len(inner_gens), len(leaf_gens[0]), a.ctypes.data, a.shape[1],
                           a.shape[0], out.ctypes.data, threads)
    return out
