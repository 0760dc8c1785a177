"""ctypes binding of libswmarlin.so (include/swmarlin.h).  The product path: every call here ends in a
hand-written gfx950 kernel.  Missing library or missing GPU is an error, never a silent fallback."""
import ctypes
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWM_LIB_PATH") or os.path.join(_HERE, "libswmarlin.so")  # override: A/B runs of two builds

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_int = ctypes.c_int

# name -> (restype, argtypes): every symbol include/swmarlin.h declares
ABI = {
    "swm_version": (_int, []),
    "swm_strerror": (ctypes.c_char_p, [_int]),
    "swm_init": (_int, [_int, ctypes.POINTER(_vp)]),
    "swm_destroy": (None, [_vp]),
    "swm_last_error": (ctypes.c_char_p, [_vp]),
    "swm_set_stream": (_int, [_vp, _vp]),
    "swm_set_msm_sharding": (_int, [_vp, ctypes.c_uint, ctypes.c_uint, _vp, _vp]),
    "swm_rccl_unique_id": (_int, [ctypes.c_void_p]),
    "swm_rccl_init": (_int, [_vp, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]),
    "swm_set_rccl_comm": (_int, [_vp, _vp, ctypes.c_uint, ctypes.c_uint]),
    "swm_selftest_exchange": (_int, [_vp, ctypes.c_void_p, ctypes.c_void_p, _sz, _int]),
    "swm_rccl_info": (_int, [ctypes.c_char_p, _sz]),
    "swm_exchange_stats": (_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "swm_synchronize": (_int, [_vp]),
    "swm_malloc": (_int, [_vp, _sz, ctypes.POINTER(_vp)]),
    "swm_free": (_int, [_vp, _vp]),
    "swm_memcpy_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "swm_memcpy_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "swm_srs_upload": (_int, [_vp, _u64p, _sz, ctypes.POINTER(_vp)]),
    "swm_srs_free": (_int, [_vp, _vp]),
    "swm_srs_len": (_sz, [_vp]),
    "swm_msm_g1": (_int, [_vp, _vp, _sz, _u64p, _sz, _u64p]),
    "swm_msm_g1_dev": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _u64p]),
    "swm_g1_normalize": (_int, [_u64p, _u64p, ctypes.POINTER(_int)]),
    "swm_g1_add_jac": (_int, [_u64p, _u64p, _u64p]),
    "swm_ntt_fr": (_int, [_vp, _u64p, ctypes.c_uint, _int, _int]),
    "swm_ntt_fr_dev": (_int, [_vp, _vp, ctypes.c_uint, _int, _int]),
    "swm_ntt_fr_sharded_dev": (_int, [_vp, ctypes.c_void_p, ctypes.c_uint, _int, _int]),
    "swm_spmv_fr": (_int, [_vp, _u32p, _u32p, _u64p, _u64p, _sz, _u64p, _sz, _sz]),
    "swm_spmv_fr_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "swm_batch_inverse_fr": (_int, [_vp, _u64p, _sz]),
    "swm_batch_inverse_fr_dev": (_int, [_vp, _vp, _sz]),
    "swm_vec_mul_fr": (_int, [_vp, _u64p, _u64p, _u64p, _sz]),
    "swm_vec_mul_fr_dev": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "swm_rng_test_new": (_int, [ctypes.POINTER(_vp)]),
    "swm_rng_from_seed": (_int, [ctypes.c_void_p, ctypes.POINTER(_vp)]),
    "swm_rng_from_callback": (_int, [_vp, _vp, ctypes.POINTER(_vp)]),
    "swm_rng_from_chacha": (_int, [ctypes.c_void_p, ctypes.c_uint64, _int, ctypes.POINTER(_vp)]),
    "swm_rng_word_pos": (_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "swm_rng_fill_bytes": (_int, [_vp, ctypes.c_void_p, ctypes.c_size_t]),
    "swm_rng_fill_bytes_cb": (None, [_vp, ctypes.c_void_p, ctypes.c_size_t]),
    "swm_rng_free": (None, [_vp]),
    "swm_rng_next_u64": (_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "swm_rng_rand_fr": (_int, [_vp, _u64p]),
    "swm_generate_universal_srs": (_int, [_vp, _sz, _sz, _sz, _vp, ctypes.POINTER(_vp)]),
    "swm_srs_destroy": (None, [_vp, _vp]),
    "swm_srs_max_degree": (_sz, [_vp]),
    "swm_srs_power_of_g": (_int, [_vp, _vp, _sz, _u64p]),
    "swm_srs_export": (_int, [_vp, _vp, _sz, _sz, _u64p, _u64p, _u64p, _u64p]),
    "swm_srs_import": (_int, [_vp, _u64p, _sz, _u64p, _u64p, _u64p, ctypes.POINTER(_vp)]),
    "swm_generate_proving_and_verifying_keys": (_int, [_vp, _vp, ctypes.c_void_p, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "swm_pk_destroy": (None, [_vp, _vp]),
    "swm_pk_retain": (_int, [_vp]),
    "swm_pk_attach": (_int, [_vp, _vp]),
    "swm_pk_device": (_int, [_vp]),
    "swm_pk_refcount": (_int, [_vp]),
    "swm_device_mem_info": (_int, [_vp, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "swm_vk_destroy": (None, [_vp]),
    "swm_generate_proof": (_int, [_vp, _vp, ctypes.c_void_p, _vp, ctypes.c_void_p, _sz, ctypes.POINTER(_sz)]),
    "swm_generate_proof_ex": (_int, [_vp, _vp, ctypes.c_void_p, _vp, ctypes.c_uint, ctypes.c_void_p, _sz, ctypes.POINTER(_sz)]),
    "swm_proof_recode": (_int, [ctypes.c_void_p, _sz, _int, ctypes.c_void_p, _sz, ctypes.POINTER(_sz)]),
    "swm_verify_proof": (_int, [_vp, _u64p, _sz, ctypes.c_void_p, _sz, _vp, ctypes.POINTER(_int)]),
    "swm_vk_serialize": (_int, [_vp, ctypes.c_void_p, _sz, ctypes.POINTER(_sz)]),
    "swm_vk_deserialize": (_int, [ctypes.c_void_p, _sz, ctypes.POINTER(_vp)]),
    "swm_proof_validate": (_int, [ctypes.c_void_p, _sz]),
    "swm_pk_serialize": (_int, [_vp, _vp, ctypes.c_void_p, _sz, ctypes.POINTER(_sz)]),
    "swm_pk_deserialize": (_int, [_vp, ctypes.c_void_p, _sz, ctypes.POINTER(_vp)]),
    "swm_r1cs_is_satisfied": (_int, [_vp, ctypes.c_void_p, ctypes.POINTER(_int), ctypes.POINTER(_sz)]),
    "swm_blake2s": (_int, [ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "swm_chacha_block": (_int, [ctypes.c_void_p, ctypes.c_uint64, _int, ctypes.c_void_p]),
    "swm_pedersen_create": (_int, [_vp, ctypes.c_void_p, _sz, _sz, ctypes.POINTER(_vp)]),
    "swm_pedersen_destroy": (None, [_vp, _vp]),
    "swm_pedersen_hash": (_int, [_vp, _vp, ctypes.c_void_p, _sz, _sz, ctypes.c_void_p]),
    "swm_pedersen_hash_dev": (_int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "swm_merkle_tree_build": (_int, [_vp, _vp, _vp, ctypes.c_void_p, _sz, _sz, ctypes.c_void_p]),
    "swm_merkle_tree_build_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp]),
    "swm_profile_enable": (_int, [_vp, _int]),
    "swm_profile_reset": (_int, [_vp]),
    "swm_profile_json": (_int, [_vp, ctypes.c_char_p, _sz]),
    "swm_selftest_mul": (_int, [_vp, _int, _u64p, _u64p, _u64p, _sz]),
    "swm_selftest_g1_add": (_int, [_vp, _u64p, _u64p, _u64p, _sz]),
    "swm_selftest_mul_throughput": (_int, [_vp, _int, _sz, _int, ctypes.POINTER(ctypes.c_float)]),
    "swm_selftest_pairing": (_int, [ctypes.POINTER(ctypes.c_uint)]),
    "swm_selftest_fr_inv": (_int, [_u64p, _u64p, _sz, ctypes.POINTER(ctypes.c_uint)]),
}


def rccl_info():
    """swm_rccl_info: (usable, description) of the RCCL this process would use / uses for the library's exchanges."""
    buf = ctypes.create_string_buffer(1024)
    rc = load_library().swm_rccl_info(buf, len(buf))
    return rc == 0, buf.value.decode(errors="replace")


def rccl_unique_id():
    """swm_rccl_unique_id: 128 bytes created on rank 0, to be handed to every rank (swm_rccl_init)."""
    buf = (ctypes.c_uint8 * 128)()
    rc = load_library().swm_rccl_unique_id(buf)
    if rc != 0:
        raise SwmError(rc, "swm_rccl_unique_id", load_library().swm_last_error(None).decode(errors="replace"))
    return bytes(buf)


class SwmError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__("%s failed: %s (%d)%s" % (what, _strerror(code), code, (": " + detail) if detail else ""))


_lib = None


def load_library():
    """Loads libswmarlin.so and types every exported symbol.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libswmarlin.so is not built (run __graft_entry__.build() or `make -C simpleworks_amd/csrc`); "
                               "simpleworks_amd has no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _strerror(code):
    try:
        return load_library().swm_strerror(code).decode()
    except Exception:
        return "error"


def _p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "expected a C-contiguous uint64 array"
    return a.ctypes.data_as(_u64p)


def _p32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"], "expected a C-contiguous uint32 array"
    return a.ctypes.data_as(_u32p)


class DeviceBuffer:
    """An HBM allocation owned by a Context (freed with it or explicitly)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = nbytes
        p = _vp()
        ctx._check(ctx.lib.swm_malloc(ctx.h, nbytes, ctypes.byref(p)), "swm_malloc")
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.swm_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes), "swm_memcpy_h2d")
        return self

    def download(self, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.swm_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes), "swm_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.swm_free(self.ctx.h, self.ptr)
            self.ptr = None


class Bases:
    def __init__(self, ctx, handle, n):
        self.ctx, self.h, self.n = ctx, handle, n

    def free(self):
        if self.h:
            self.ctx.lib.swm_srs_free(self.ctx.h, self.h)
            self.h = None


class Context:
    """One GPU + one HIP stream (swm_ctx).  Raises SwmError(SWM_ERR_NO_DEVICE) when no MI355X is usable."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.swm_init(device, ctypes.byref(h))
        if rc != 0:
            raise SwmError(rc, "swm_init")
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.lib.swm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SwmError(rc, what, self.lib.swm_last_error(self.h).decode(errors="replace"))

    def set_stream(self, stream_ptr):
        self._check(self.lib.swm_set_stream(self.h, stream_ptr), "swm_set_stream")

    def synchronize(self):
        self._check(self.lib.swm_synchronize(self.h), "swm_synchronize")

    def mem_info(self):
        """swm_device_mem_info: (free, total) bytes of HBM on the context's device."""
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self.lib.swm_device_mem_info(self.h, ctypes.byref(free), ctypes.byref(total)), "swm_device_mem_info")
        return free.value, total.value

    ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

    def set_msm_sharding(self, rank, world, allgather=None):
        """swm_set_msm_sharding: split every commitment MSM of the prover / indexer by point range over `world`
        contexts.  `allgather(send: bytes) -> bytes` must return the concatenation of every rank's `send` in rank
        order (simpleworks_amd.dist.enable_sharded_prover supplies one over torch.distributed).  world <= 1 or
        allgather None switches sharding off."""
        if world <= 1 or allgather is None:
            self._check(self.lib.swm_set_msm_sharding(self.h, 0, 1, None, None), "swm_set_msm_sharding")
            self._shard_cb = None
            self.shard_rank, self.shard_world = 0, 1
            return
        self.shard_rank, self.shard_world = rank, world

        def _cb(_user, send, nbytes, recv):
            try:
                out = allgather(ctypes.string_at(send, nbytes))
                if len(out) != nbytes * world:
                    return 1
                ctypes.memmove(recv, out, len(out))
                return 0
            except Exception:  # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        cb = Context.ALLGATHER_FN(_cb)
        self._check(self.lib.swm_set_msm_sharding(self.h, rank, world, ctypes.cast(cb, ctypes.c_void_p), None),
                    "swm_set_msm_sharding")
        self._shard_cb = cb  # keep the trampoline alive as long as the library may call it

    def rccl_init(self, unique_id, rank, world):
        """swm_rccl_init: the library's own RCCL communicator for this context (one process per GPU); the sharded
        prover then exchanges its partial sums with one ncclAllGather per round."""
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.swm_rccl_init(self.h, buf, rank, world), "swm_rccl_init")

    def selftest_exchange(self, send_buf, recv_buf, bytes_per_peer, alltoall=True):
        self._check(self.lib.swm_selftest_exchange(self.h, send_buf.ptr, recv_buf.ptr, bytes_per_peer, int(alltoall)), "swm_selftest_exchange")

    def exchange_stats(self):
        calls, nbytes = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.lib.swm_exchange_stats(self.h, ctypes.byref(calls), ctypes.byref(nbytes)), "swm_exchange_stats")
        return calls.value, nbytes.value

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 64)).upload(arr)

    # ---- K1
    def srs_upload(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 12)
        h = _vp()
        self._check(self.lib.swm_srs_upload(self.h, _p64(xy), xy.shape[0], ctypes.byref(h)), "swm_srs_upload")
        return Bases(self, h, xy.shape[0])

    def msm_g1(self, bases, scalars_std, offset=0):
        sc = np.ascontiguousarray(scalars_std, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(18, dtype=np.uint64)
        self._check(self.lib.swm_msm_g1(self.h, bases.h, offset, _p64(sc), sc.shape[0], _p64(out)), "swm_msm_g1")
        return out

    def msm_g1_dev(self, bases, d_scalars, n, montgomery, offset=0):
        out = np.zeros(18, dtype=np.uint64)
        ptr = d_scalars.ptr if isinstance(d_scalars, DeviceBuffer) else int(d_scalars)
        self._check(self.lib.swm_msm_g1_dev(self.h, bases.h, offset, ptr, n, 1 if montgomery else 0, _p64(out)),
                    "swm_msm_g1_dev")
        return out

    def g1_normalize(self, jac):
        jac = np.ascontiguousarray(jac, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        inf = _int(0)
        self._check(self.lib.swm_g1_normalize(_p64(jac), _p64(out), ctypes.byref(inf)), "swm_g1_normalize")
        return out, bool(inf.value)

    def g1_add_jac(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros(18, dtype=np.uint64)
        self._check(self.lib.swm_g1_add_jac(_p64(a), _p64(b), _p64(out)), "swm_g1_add_jac")
        return out

    # ---- K2
    def ntt_fr(self, data_mont, log_n, inverse=False, coset=False):
        d = np.ascontiguousarray(data_mont, dtype=np.uint64).copy()
        assert d.size == 4 << log_n
        self._check(self.lib.swm_ntt_fr(self.h, _p64(d), log_n, int(inverse), int(coset)), "swm_ntt_fr")
        return d

    def ntt_fr_dev(self, dbuf, log_n, inverse=False, coset=False):
        ptr = dbuf.ptr if isinstance(dbuf, DeviceBuffer) else int(dbuf)
        self._check(self.lib.swm_ntt_fr_dev(self.h, ptr, log_n, int(inverse), int(coset)), "swm_ntt_fr_dev")

    def ntt_fr_sharded_dev(self, dbuf, log_n, inverse=False, blocks_in=False):
        """One transform over the ranks of this context's sharding: in place on the rank's n / G elements,
        CYCLIC -> BLOCKS layout (blocks_in = False) or BLOCKS -> CYCLIC (include/swmarlin.h: swm_ntt_fr_sharded_dev)."""
        ptr = dbuf.ptr if isinstance(dbuf, DeviceBuffer) else int(dbuf)
        self._check(self.lib.swm_ntt_fr_sharded_dev(self.h, ptr, log_n, int(inverse), int(blocks_in)), "swm_ntt_fr_sharded_dev")

    # ---- K3
    def spmv_fr(self, rowptr, col, val_mont, z_mont):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        val = np.ascontiguousarray(val_mont, dtype=np.uint64).reshape(-1, 4)
        z = np.ascontiguousarray(z_mont, dtype=np.uint64).reshape(-1, 4)
        rows = rowptr.shape[0] - 1
        out = np.zeros((rows, 4), dtype=np.uint64)
        self._check(self.lib.swm_spmv_fr(self.h, _p32(rowptr), _p32(col) if col.size else _p32(np.zeros(1, np.uint32)),
                                         _p64(val) if val.size else _p64(np.zeros(4, np.uint64)), _p64(z), z.shape[0],
                                         _p64(out), rows, col.shape[0]), "swm_spmv_fr")
        return out

    def spmv_fr_dev(self, d_rowptr, d_col, d_val, d_z, d_out, rows):
        self._check(self.lib.swm_spmv_fr_dev(self.h, d_rowptr.ptr, d_col.ptr, d_val.ptr, d_z.ptr, d_out.ptr, rows),
                    "swm_spmv_fr_dev")

    # ---- K4
    def batch_inverse_fr(self, data_mont):
        d = np.ascontiguousarray(data_mont, dtype=np.uint64).reshape(-1, 4).copy()
        self._check(self.lib.swm_batch_inverse_fr(self.h, _p64(d), d.shape[0]), "swm_batch_inverse_fr")
        return d

    def batch_inverse_fr_dev(self, dbuf, n):
        self._check(self.lib.swm_batch_inverse_fr_dev(self.h, dbuf.ptr, n), "swm_batch_inverse_fr_dev")

    def vec_mul_fr(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        out = np.empty_like(a)
        self._check(self.lib.swm_vec_mul_fr(self.h, _p64(a), _p64(b), _p64(out), a.shape[0]), "swm_vec_mul_fr")
        return out

    # ---- Pedersen CRH + Merkle tree (include/swmarlin.h; simpleworks_amd/hash.py is the caller-facing mirror)
    def pedersen_create(self, generators_xy, num_windows, window_size):
        """generators_xy: num_windows * window_size affine points, 64 bytes each (x || y, little-endian standard form)."""
        buf = np.ascontiguousarray(np.frombuffer(bytes(generators_xy), dtype=np.uint8))
        assert buf.size == 64 * num_windows * window_size
        h = _vp()
        self._check(self.lib.swm_pedersen_create(self.h, buf.ctypes.data, num_windows, window_size, ctypes.byref(h)),
                    "swm_pedersen_create")
        return h

    def pedersen_destroy(self, handle):
        if self.h and handle:
            self.lib.swm_pedersen_destroy(self.h, handle)

    def pedersen_hash(self, handle, inputs):
        """inputs: uint8 array [count, input_len] -> uint8 [count, 32] digests (x coordinate, little-endian)."""
        a = np.ascontiguousarray(inputs, dtype=np.uint8)
        assert a.ndim == 2
        out = np.empty((a.shape[0], 32), dtype=np.uint8)
        self._check(self.lib.swm_pedersen_hash(self.h, handle, a.ctypes.data, a.shape[1], a.shape[0], out.ctypes.data),
                    "swm_pedersen_hash")
        return out

    def merkle_tree_build(self, leaf_handle, two_to_one_handle, leaves):
        """leaves: uint8 [n, leaf_len] -> uint8 [2 n - 1, 32]: n leaf digests | n / 2 | ... | root."""
        a = np.ascontiguousarray(leaves, dtype=np.uint8)
        assert a.ndim == 2
        out = np.empty((2 * a.shape[0] - 1, 32), dtype=np.uint8)
        self._check(self.lib.swm_merkle_tree_build(self.h, leaf_handle, two_to_one_handle, a.ctypes.data, a.shape[1], a.shape[0],
                                                   out.ctypes.data), "swm_merkle_tree_build")
        return out

    def merkle_tree_build_dev(self, leaf_handle, two_to_one_handle, d_leaves, leaf_len, n_leaves, d_nodes):
        self._check(self.lib.swm_merkle_tree_build_dev(self.h, leaf_handle, two_to_one_handle, d_leaves.ptr, leaf_len, n_leaves,
                                                       d_nodes.ptr), "swm_merkle_tree_build_dev")

    # ---- measurement
    def profile_enable(self, on=True):
        self._check(self.lib.swm_profile_enable(self.h, int(on)), "swm_profile_enable")

    def profile_reset(self):
        self._check(self.lib.swm_profile_reset(self.h), "swm_profile_reset")

    def profile(self):
        buf = ctypes.create_string_buffer(1 << 20)
        self._check(self.lib.swm_profile_json(self.h, buf, len(buf)), "swm_profile_json")
        doc = json.loads(buf.value.decode())
        self.last_work = doc.get("work", {})
        self.last_calls = doc.get("calls", [])
        return {k["name"]: k for k in doc["kernels"]}

    # ---- self tests
    def selftest_mul(self, which, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.empty_like(a)
        self._check(self.lib.swm_selftest_mul(self.h, which, _p64(a), _p64(b), _p64(out), a.shape[0]), "swm_selftest_mul")
        return out

    def selftest_g1_add(self, a_xy, b_xy):
        a = np.ascontiguousarray(a_xy, dtype=np.uint64).reshape(-1, 12)
        b = np.ascontiguousarray(b_xy, dtype=np.uint64).reshape(-1, 12)
        out = np.zeros((a.shape[0], 18), dtype=np.uint64)
        self._check(self.lib.swm_selftest_g1_add(self.h, _p64(a), _p64(b), _p64(out), a.shape[0]), "swm_selftest_g1_add")
        return out

    def selftest_mul_throughput(self, which, threads, iters):
        ms = ctypes.c_float(0)
        self._check(self.lib.swm_selftest_mul_throughput(self.h, which, threads, iters, ctypes.byref(ms)),
                    "swm_selftest_mul_throughput")
        return ms.value
