"""Mirror of /root/reference/src/marlin/mod.rs: the same free functions, argument meaning and error behaviour,
implemented on top of libswmarlin.so (HIP kernels on MI355X).  No CPU fallback: setup, indexing and proving need
the GPU; verify_proof is host arithmetic in the reference as well and runs without one.

    reference (Rust)                                            here (Python over the C ABI)
    generate_rand() -> StdRng                                   generate_rand() -> Rng
    generate_universal_srs(nc, nv, nnz, &mut rng)               generate_universal_srs(nc, nv, nnz, rng)
    generate_proving_and_verifying_keys(&srs, cs)               generate_proving_and_verifying_keys(srs, cs)
    generate_proof(cs, proving_key, &mut rng) -> MarlinProof    generate_proof(cs, proving_key, rng) -> MarlinProof
    verify_proof(vk, &public_inputs, &proof, &mut rng)          verify_proof(vk, public_inputs, proof, rng) -> bool
Errors surface as MarlinError (the reference returns anyhow::Error built from the arkworks error's Debug string).
"""
import ctypes

import numpy as np

from ._lib import SwmError, load_library, _p64, _p32, _vp, Context

R_MODULUS = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
_MONT_R = (1 << 256) % R_MODULUS
_M64 = (1 << 64) - 1


class MarlinError(SwmError):
    pass


def _to_mont_limbs(values):
    """ints (standard form) -> (n, 4) uint64 Montgomery limbs (ark-ff Fp256 layout)."""
    out = np.empty((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        m = (int(v) % R_MODULUS) * _MONT_R % R_MODULUS
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & _M64
    return out


_default_ctx = None


def default_context():
    """The process-wide GPU context (one GPU, one stream).  Raises SwmError when no MI355X is usable."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


class Rng:
    """rand::rngs::StdRng handle (ChaCha12)."""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        try:
            if self.h:
                load_library().swm_rng_free(self.h)
                self.h = None
        except Exception:
            pass

    def next_u64(self):
        v = ctypes.c_uint64(0)
        _check(load_library().swm_rng_next_u64(self.h, ctypes.byref(v)), "swm_rng_next_u64")
        return v.value

    def rand_fr_mont(self):
        out = np.zeros(4, dtype=np.uint64)
        _check(load_library().swm_rng_rand_fr(self.h, _p64(out)), "swm_rng_rand_fr")
        return out

    def fill_bytes(self, n):
        """RngCore::fill_bytes: the next n bytes of the stream (whole 32-bit words are consumed)."""
        buf = (ctypes.c_uint8 * n)()
        _check(load_library().swm_rng_fill_bytes(self.h, buf, n), "swm_rng_fill_bytes")
        return bytes(buf)

    def word_pos(self):
        """rand_chacha's get_word_pos of a built-in / adopted generator: 32-bit keystream words consumed so far."""
        v = ctypes.c_uint64(0)
        _check(load_library().swm_rng_word_pos(self.h, ctypes.byref(v)), "swm_rng_word_pos")
        return v.value


def _check(rc, what, ctx=None):
    if rc != 0:
        detail = (ctx.lib.swm_last_error(ctx.h) if ctx is not None else load_library().swm_last_error(None)).decode(errors="replace")
        raise MarlinError(rc, what, detail)


def generate_rand():
    """src/marlin/mod.rs:33-35 — ark_std::test_rng()."""
    h = _vp()
    _check(load_library().swm_rng_test_new(ctypes.byref(h)), "swm_rng_test_new")
    return Rng(h)


def rng_from_seed(seed32):
    h = _vp()
    buf = (ctypes.c_uint8 * 32)(*bytes(seed32))
    _check(load_library().swm_rng_from_seed(buf, ctypes.byref(h)), "swm_rng_from_seed")
    return Rng(h)


TEST_RNG_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)  # ark_std::test_rng()


def rng_from_chacha(key32, word_pos=0, rounds=12):
    """swm_rng_from_chacha: adopt the STATE of a caller's ChaCha generator (rand 0.8 StdRng = ChaCha12: get_seed /
    get_word_pos) instead of calling back into it; read the position back with Rng.word_pos() and set_word_pos it."""
    h = _vp()
    buf = (ctypes.c_uint8 * 32)(*bytes(key32))
    _check(load_library().swm_rng_from_chacha(buf, word_pos, rounds, ctypes.byref(h)), "swm_rng_from_chacha")
    return Rng(h)


_FILL_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)


def rng_from_fill_bytes(fill_bytes):
    """swm_rng_from_callback: a generator owned by the CALLER behind the Rng handle.  fill_bytes(n) -> n bytes, the next
    n bytes of the caller's stream (rand's RngCore::fill_bytes).  This is how a binding keeps the reference's
    `&mut StdRng` parameters (src/marlin/mod.rs:49,73,83) and the draw stream at the same time."""
    def _cb(_user, dest, n):
        data = fill_bytes(n)
        assert len(data) == n
        ctypes.memmove(dest, data, n)
    cb = _FILL_FN(_cb)
    h = _vp()
    _check(load_library().swm_rng_from_callback(ctypes.cast(cb, ctypes.c_void_p), None, ctypes.byref(h)),
           "swm_rng_from_callback")
    r = Rng(h)
    r._cb = cb  # the trampoline lives as long as the handle
    return r


def rng_behind_callback(caller_rng):
    """A handle whose every draw goes, through swm_rng_from_callback, to ANOTHER library generator (`caller_rng`, which
    stands for the caller's StdRng): the callback is the library's own native trampoline, so what is measured is the
    callback path itself, not a Python function.  `caller_rng` must outlive the handle."""
    lib = load_library()
    h = _vp()
    _check(lib.swm_rng_from_callback(ctypes.cast(lib.swm_rng_fill_bytes_cb, ctypes.c_void_p), caller_rng.h, ctypes.byref(h)),
           "swm_rng_from_callback")
    r = Rng(h)
    r._caller = caller_rng
    return r


class ConstraintSystem:
    """What a ConstraintSystemRef<Fr> (src/marlin/mod.rs:16) holds once a ConstraintSynthesizer has run:
    variables with their assignment and the rows a * b = c.  Same builder vocabulary as ark-relations:
    new_input_variable / new_witness_variable / enforce_constraint(a, b, c) with linear combinations given as
    lists of (coefficient, variable); ConstraintSystem.one() is the constant."""

    def __init__(self):
        self.instance = [1]
        self.witness = []
        self.rows = ([], [], [])

    @staticmethod
    def one():
        return ("i", 0)

    def new_input_variable(self, value):
        self.instance.append(int(value) % R_MODULUS)
        return ("i", len(self.instance) - 1)

    def new_witness_variable(self, value):
        self.witness.append(int(value) % R_MODULUS)
        return ("w", len(self.witness) - 1)

    def enforce_constraint(self, a, b, c):
        for dst, lc in zip(self.rows, (a, b, c)):
            dst.append(list(lc))

    @property
    def num_constraints(self):
        return len(self.rows[0])

    def num_instance_variables(self):
        return len(self.instance)

    def num_witness_variables(self):
        return len(self.witness)

    def _csr(self, rows):
        ninst = len(self.instance)
        rowptr = np.zeros(len(rows) + 1, dtype=np.uint32)
        cols, vals = [], []
        for r, lc in enumerate(rows):
            acc = {}
            for coeff, (kind, k) in lc:
                col = k if kind == "i" else ninst + k
                acc[col] = (acc.get(col, 0) + int(coeff)) % R_MODULUS
            for col in sorted(acc):
                if acc[col]:
                    cols.append(col)
                    vals.append(acc[col])
            rowptr[r + 1] = len(cols)
        return rowptr, np.array(cols, dtype=np.uint32), _to_mont_limbs(vals)

    def pack(self):
        """Flat arrays in the layout of struct swm_r1cs (kept alive by the returned object)."""
        return PackedR1cs(_to_mont_limbs(self.instance), _to_mont_limbs(self.witness), *[self._csr(r) for r in self.rows])

    def pack_assignment(self):
        """What swm_generate_proof reads: the two assignment vectors and the shape, no matrices (NULL pointers)."""
        return AssignmentOnly(_to_mont_limbs(self.instance), _to_mont_limbs(self.witness), self.num_constraints)

    def is_satisfied(self, ctx=None):
        """ConstraintSystem::is_satisfied on the GPU (K3): A z o B z == C z."""
        return self.pack().is_satisfied(ctx)


class _R1csStruct(ctypes.Structure):
    _fields_ = [("num_instance", ctypes.c_size_t), ("num_witness", ctypes.c_size_t), ("num_constraints", ctypes.c_size_t),
                ("instance", ctypes.c_void_p), ("witness", ctypes.c_void_p),
                ("a_rowptr", ctypes.c_void_p), ("a_col", ctypes.c_void_p), ("a_val", ctypes.c_void_p),
                ("b_rowptr", ctypes.c_void_p), ("b_col", ctypes.c_void_p), ("b_val", ctypes.c_void_p),
                ("c_rowptr", ctypes.c_void_p), ("c_col", ctypes.c_void_p), ("c_val", ctypes.c_void_p)]


class PackedR1cs:
    """A synthesised constraint system as flat numpy arrays (instance/witness Montgomery limbs + CSR of A, B, C)."""

    def __init__(self, instance, witness, a, b, c):
        self.instance = np.ascontiguousarray(instance, dtype=np.uint64).reshape(-1, 4)
        self.witness = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
        self.mats = []
        for rowptr, col, val in (a, b, c):
            self.mats.append((np.ascontiguousarray(rowptr, dtype=np.uint32), np.ascontiguousarray(col, dtype=np.uint32),
                              np.ascontiguousarray(val, dtype=np.uint64).reshape(-1, 4)))
        self.num_constraints = self.mats[0][0].shape[0] - 1

    def struct(self):
        s = _R1csStruct()
        s.num_instance = self.instance.shape[0]
        s.num_witness = self.witness.shape[0]
        s.num_constraints = self.num_constraints
        s.instance = self.instance.ctypes.data
        s.witness = self.witness.ctypes.data if self.witness.size else None
        for name, (rowptr, col, val) in zip("abc", self.mats):
            setattr(s, name + "_rowptr", rowptr.ctypes.data)
            setattr(s, name + "_col", col.ctypes.data if col.size else None)
            setattr(s, name + "_val", val.ctypes.data if val.size else None)
        return s

    def pack(self):
        return self

    def pack_assignment(self):
        return AssignmentOnly(self.instance, self.witness, self.num_constraints)

    def is_satisfied(self, ctx=None):
        ctx = ctx or default_context()
        ok = ctypes.c_int(0)
        bad = ctypes.c_size_t(0)
        s = self.struct()
        _check(ctx.lib.swm_r1cs_is_satisfied(ctx.h, ctypes.byref(s), ctypes.byref(ok), ctypes.byref(bad)),
               "swm_r1cs_is_satisfied", ctx)
        return bool(ok.value)


class AssignmentOnly:
    """instance + witness assignment and the number of constraints: everything swm_generate_proof reads from a constraint
    system (include/swmarlin.h).  The nine matrix pointers of struct swm_r1cs stay NULL."""

    def __init__(self, instance, witness, num_constraints):
        self.instance = np.ascontiguousarray(instance, dtype=np.uint64).reshape(-1, 4)
        self.witness = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
        self.num_constraints = int(num_constraints)

    def struct(self):
        s = _R1csStruct()
        s.num_instance = self.instance.shape[0]
        s.num_witness = self.witness.shape[0]
        s.num_constraints = self.num_constraints
        s.instance = self.instance.ctypes.data
        s.witness = self.witness.ctypes.data if self.witness.size else None
        return s

    def pack(self):
        return self

    def pack_assignment(self):
        return self


class UniversalSRS:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @property
    def max_degree(self):
        return self.ctx.lib.swm_srs_max_degree(self.h)

    def power_of_g(self, i):
        out = np.zeros(12, dtype=np.uint64)
        _check(self.ctx.lib.swm_srs_power_of_g(self.ctx.h, self.h, i, _p64(out)), "swm_srs_power_of_g", self.ctx)
        return out

    def export(self):
        """swm_srs_export: (powers (n, 12), gamma powers (3, 12), h (24,), beta_h (24,)) — the fields of arkworks'
        kzg10::UniversalParams as Montgomery limbs."""
        n = self.max_degree + 1
        powers = np.zeros((n, 12), dtype=np.uint64)
        gamma = np.zeros((3, 12), dtype=np.uint64)
        h, bh = np.zeros(24, dtype=np.uint64), np.zeros(24, dtype=np.uint64)
        _check(self.ctx.lib.swm_srs_export(self.ctx.h, self.h, 0, n, _p64(powers), _p64(gamma), _p64(h), _p64(bh)),
               "swm_srs_export", self.ctx)
        return powers, gamma, h, bh

    @staticmethod
    def from_parts(powers, gamma, h, beta_h, ctx=None):
        """swm_srs_import: a UniversalSRS built elsewhere (arkworks' UniversalParams flattened by the binding)."""
        ctx = ctx or default_context()
        powers = np.ascontiguousarray(powers, dtype=np.uint64).reshape(-1, 12)
        gamma = np.ascontiguousarray(gamma, dtype=np.uint64).reshape(3, 12)
        hd = _vp()
        _check(ctx.lib.swm_srs_import(ctx.h, _p64(powers), powers.shape[0], _p64(gamma),
                                      _p64(np.ascontiguousarray(h, dtype=np.uint64)),
                                      _p64(np.ascontiguousarray(beta_h, dtype=np.uint64)), ctypes.byref(hd)),
               "swm_srs_import", ctx)
        return UniversalSRS(ctx, hd)

    def free(self):
        if self.h:
            self.ctx.lib.swm_srs_destroy(self.ctx.h, self.h)
            self.h = None


class ProvingKey:
    """A device-resident proving key (swm_pk): resident per DEVICE, read-only and reference-counted.  `ctx` is the context
    this holder proves on; attach(other_ctx) gives another holder of the SAME resident key for a context of another host
    thread on that device (swm_pk_attach) — one copy of the tables serves every proving thread."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def attach(self, ctx):
        _check(ctx.lib.swm_pk_attach(ctx.h, self.h), "swm_pk_attach", ctx)
        return ProvingKey(ctx, self.h)

    @property
    def refcount(self):
        return self.ctx.lib.swm_pk_refcount(self.h)

    def free(self):
        """Drops this holder's reference; the last one frees the key."""
        if self.h:
            self.ctx.lib.swm_pk_destroy(self.ctx.h, self.h)
            self.h = None


class VerifyingKey:
    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        try:
            if self.h:
                load_library().swm_vk_destroy(self.h)
                self.h = None
        except Exception:
            pass


class MarlinProof:
    """ark_marlin::Proof, held in its CanonicalSerialize byte form."""

    def __init__(self, data):
        self.data = bytes(data)


def generate_universal_srs(num_constraints, num_variables, num_non_zero, rng, ctx=None):
    """src/marlin/mod.rs:45-55."""
    ctx = ctx or default_context()
    h = _vp()
    _check(ctx.lib.swm_generate_universal_srs(ctx.h, num_constraints, num_variables, num_non_zero, rng.h, ctypes.byref(h)),
           "swm_generate_universal_srs", ctx)
    return UniversalSRS(ctx, h)


def generate_proving_and_verifying_keys(universal_srs, constraint_system):
    """src/marlin/mod.rs:88-94."""
    ctx = universal_srs.ctx
    packed = constraint_system.pack()
    s = packed.struct()
    pk, vk = _vp(), _vp()
    _check(ctx.lib.swm_generate_proving_and_verifying_keys(ctx.h, universal_srs.h, ctypes.byref(s), ctypes.byref(pk),
                                                           ctypes.byref(vk)), "swm_generate_proving_and_verifying_keys", ctx)
    return ProvingKey(ctx, pk), VerifyingKey(vk)


def generate_proof(constraint_system, proving_key, rng):
    """src/marlin/mod.rs:70-77.  Raises MarlinError(SWM_ERR_UNSATISFIED) for an unsatisfied witness.  The prover reads only
    the assignment and the shape of the constraint system (the matrices are the key's), so a constraint system that offers
    pack_assignment() — an AssignmentOnly, or a ConstraintSystem — is not flattened into CSR per proof."""
    ctx = proving_key.ctx
    packed = constraint_system.pack_assignment() if hasattr(constraint_system, "pack_assignment") else constraint_system.pack()
    s = packed.struct()
    buf = (ctypes.c_uint8 * 2048)()
    n = ctypes.c_size_t(0)
    _check(ctx.lib.swm_generate_proof(ctx.h, proving_key.h, ctypes.byref(s), rng.h, buf, len(buf), ctypes.byref(n)),
           "swm_generate_proof", ctx)
    return MarlinProof(bytes(buf[: n.value]))


def generate_proof_uncompressed(constraint_system, proving_key, rng):
    """swm_generate_proof_ex(SWM_PROOF_UNCOMPRESSED): the proof as serialize_uncompressed bytes — what a binding that rebuilds an
    arkworks `Proof` in the same process reads with deserialize_unchecked (no square roots, no subgroup checks)."""
    ctx = proving_key.ctx
    packed = constraint_system.pack_assignment() if hasattr(constraint_system, "pack_assignment") else constraint_system.pack()
    s = packed.struct()
    buf = (ctypes.c_uint8 * 4096)()
    n = ctypes.c_size_t(0)
    _check(ctx.lib.swm_generate_proof_ex(ctx.h, proving_key.h, ctypes.byref(s), rng.h, 1, buf, len(buf), ctypes.byref(n)),
           "swm_generate_proof_ex", ctx)
    return bytes(buf[: n.value])


def verify_proof(verifying_key, public_inputs, proof, rng):
    """src/marlin/mod.rs:79-86.  public_inputs: field elements as ints (e.g. the bit-expanded inputs of
    src/merkle_tree/simple_merkle_tree.rs:129-143)."""
    lib = load_library()
    pi = _to_mont_limbs(list(public_inputs))
    data = (ctypes.c_uint8 * len(proof.data)).from_buffer_copy(proof.data)
    ok = ctypes.c_int(0)
    _check(lib.swm_verify_proof(verifying_key.h, _p64(pi) if len(pi) else None, len(pi), data, len(proof.data), rng.h,
                                ctypes.byref(ok)), "swm_verify_proof")
    return bool(ok.value)


class MarlinInst:
    """`MarlinInst` (= ark_marlin::Marlin<Fr, MultiPC, FS>, src/marlin/mod.rs:14) as the reference's in-tree callers use
    it: associated functions that take a ConstraintSynthesizer — SimpleMerkleTree::{new, prove, verify}
    (src/merkle_tree/simple_merkle_tree.rs:39,83,119,148), examples/manual-constraints.rs:89-99,
    examples/merkle-tree/main.rs:212-257, examples/simple-payments/transaction.rs:96-125.  A synthesizer here is any
    object with generate_constraints(cs) (ark_relations::r1cs::ConstraintSynthesizer); index / prove run it into a fresh
    ConstraintSystem, as ark-marlin does, and forward to the five functions below.  Mirrors swmarlin-sys'
    `pub struct MarlinInst` (swmarlin-sys/src/marlin.rs) method for method."""

    @staticmethod
    def universal_setup(num_constraints, num_variables, num_non_zero, rng, ctx=None):
        return generate_universal_srs(num_constraints, num_variables, num_non_zero, rng, ctx)

    @staticmethod
    def _synthesize(circuit):
        cs = ConstraintSystem()
        circuit.generate_constraints(cs)
        return cs

    @staticmethod
    def index(universal_srs, circuit):
        return generate_proving_and_verifying_keys(universal_srs, MarlinInst._synthesize(circuit))

    @staticmethod
    def index_from_constraint_system(universal_srs, constraint_system):
        return generate_proving_and_verifying_keys(universal_srs, constraint_system)

    @staticmethod
    def prove(index_pk, circuit, zk_rng):
        return generate_proof(MarlinInst._synthesize(circuit), index_pk, zk_rng)

    @staticmethod
    def prove_from_constraint_system(index_pk, constraint_system, zk_rng):
        return generate_proof(constraint_system, index_pk, zk_rng)

    @staticmethod
    def verify(index_vk, public_input, proof, rng):
        return verify_proof(index_vk, public_input, proof, rng)
