"""Mirror of /root/reference/src/marlin/serialization.rs:5-45 (ark-serialize CanonicalSerialize byte strings)."""
import ctypes

from ._lib import load_library, _vp
from .marlin import MarlinError, MarlinProof, ProvingKey, VerifyingKey, _check, default_context


def serialize_proof(proof):
    return bytes(proof.data)


def deserialize_proof(bytes_proof):
    data = bytes(bytes_proof)
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    _check(load_library().swm_proof_validate(buf, len(data)), "Error deserializing proof")
    return MarlinProof(data)


def proof_recode(data, to_uncompressed):
    """swm_proof_recode: the proof between the compressed (CanonicalSerialize::serialize, what serialization.rs:5-17 moves) and
    the uncompressed (serialize_uncompressed, what swm_generate_proof_ex(SWM_PROOF_UNCOMPRESSED) writes) forms; checked parse."""
    data = bytes(data)
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    out = (ctypes.c_uint8 * 4096)()
    n = ctypes.c_size_t(0)
    _check(load_library().swm_proof_recode(buf, len(data), 1 if to_uncompressed else 0, out, len(out), ctypes.byref(n)),
           "Error recoding proof")
    return bytes(out[: n.value])


def serialize_verifying_key(verifying_key):
    lib = load_library()
    n = ctypes.c_size_t(0)
    _check(lib.swm_vk_serialize(verifying_key.h, None, 0, ctypes.byref(n)), "Error serializing verifying key")
    buf = (ctypes.c_uint8 * n.value)()
    _check(lib.swm_vk_serialize(verifying_key.h, buf, n.value, ctypes.byref(n)), "Error serializing verifying key")
    return bytes(buf)


def deserialize_verifying_key(bytes_verifying_key):
    data = bytes(bytes_verifying_key)
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    h = _vp()
    rc = load_library().swm_vk_deserialize(buf, len(data), ctypes.byref(h))
    if rc != 0:
        raise MarlinError(rc, "Error deserializing verifying key")
    return VerifyingKey(h)


def serialize_proving_key(proving_key):
    ctx = proving_key.ctx
    n = ctypes.c_size_t(0)
    _check(ctx.lib.swm_pk_serialize(ctx.h, proving_key.h, None, 0, ctypes.byref(n)), "Error serializing proving key", ctx)
    buf = (ctypes.c_uint8 * n.value)()
    _check(ctx.lib.swm_pk_serialize(ctx.h, proving_key.h, buf, n.value, ctypes.byref(n)), "Error serializing proving key", ctx)
    return bytes(buf)


def deserialize_proving_key(bytes_proving_key, ctx=None):
    ctx = ctx or default_context()
    data = bytes(bytes_proving_key)
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    h = _vp()
    _check(ctx.lib.swm_pk_deserialize(ctx.h, buf, len(data), ctypes.byref(h)), "Error deserializing proving key", ctx)
    return ProvingKey(ctx, h)
