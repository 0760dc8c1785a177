"""Multi-GPU sharding of the hot path — one process per GPU, torch.distributed over RCCL/xGMI.

MSM shards by POINT RANGE: rank g owns bases/scalars [g*n, (g+1)*n) of the global problem (the SRS shard is uploaded
once and stays resident), runs the full Pippenger on its shard and contributes ONE Jacobian point (144 bytes).
EC-point addition is not an RCCL reduction op, so the "all-reduce" is an all-gather of 144 B per rank followed by a
local (G-1)-term fold on every rank: latency-bound (microseconds), never xGMI-bandwidth-bound.  The result is the same
group element for every G (EC addition is associative and commutative); compare after affine normalisation.

The reference has no counterpart (single process, no collectives).
"""
import numpy as np


def shard_range(n_total, world, rank):
    """Contiguous point range of `rank`: sizes differ by at most one."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def fold_partials(partials, add_fn):
    """Left fold of Jacobian partials (each 18 x uint64) with the group law."""
    acc = np.ascontiguousarray(partials[0], dtype=np.uint64)
    for p in partials[1:]:
        acc = add_fn(acc, np.ascontiguousarray(p, dtype=np.uint64))
    return acc


def all_gather_partials(part, group=None):
    """All-gather one 18-limb Jacobian point per rank.  Uses the process group's device (cuda for RCCL, cpu for gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(part, dtype=np.uint64).view(np.int64).copy())
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [o.cpu().numpy().view(np.uint64) for o in out]


def sharded_msm(local_msm, add_fn, group=None):
    """local_msm() -> this rank's Jacobian partial (e.g. lambda: ctx.msm_g1_dev(bases_shard, d_scalars, n, True));
    add_fn(a, b) -> a + b on Jacobian limbs (ctx.g1_add_jac).  Returns the global sum on every rank."""
    import torch.distributed as dist
    part = local_msm()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return part
    return fold_partials(all_gather_partials(part, group), add_fn)


def make_byte_allgather(group=None):
    """allgather(send: bytes) -> bytes of every rank in rank order, over torch.distributed (device tensors with the
    nccl backend = RCCL over xGMI, host tensors with gloo).  This is the callback swm_set_msm_sharding asks for."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"

    def allgather(send):
        t = torch.frombuffer(bytearray(send), dtype=torch.uint8)
        if on_gpu:
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        return b"".join(o.cpu().numpy().tobytes() for o in out)

    return allgather


def enable_sharded_prover(ctx, group=None):
    """One proof over all ranks of `group` (include/swmarlin.h swm_set_msm_sharding): every rank calls
    generate_proof with the SAME constraint system, key and rng state; each commitment MSM is computed by point range
    and the 192-byte partial sums are all-gathered.  All ranks return the same proof bytes as a single-GPU run."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        ctx.set_msm_sharding(0, 1, None)
        return
    ctx.set_msm_sharding(dist.get_rank(group), world, make_byte_allgather(group))


# ---------------------------------------------------------------------------------------------------------------------
# ONE transform over G ranks with a single all-to-all (SURVEY.md §8e "NTT partitioning (ii)") — the host-side statement
# of what csrc/ntt.hip `ntt_sharded_run` does on the GPU, with the kernels injected (the CPU tests inject the oracle).
#   n = 2^log_n elements, m = n / G per rank, blk = m / G.
#   CYCLIC layout  local[j] = v[rank + G j]               BLOCKS layout  local[k1 blk + t] = v[m k1 + rank blk + t]
#   blocks_in = False: CYCLIC -> BLOCKS;  True: BLOCKS -> CYCLIC.  With i = i1 + G i2 and k = m k1 + k2:
#   w^(ik) = w_G^(i1 k1) w_n^(i1 k2) w_m^(i2 k2): local length-m transforms, a twiddle, one all-to-all, a length-G transform.
# ---------------------------------------------------------------------------------------------------------------------
FR_MODULUS = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
FR_TWO_ADIC_ROOT = pow(22, (FR_MODULUS - 1) >> 47, FR_MODULUS)


def fr_root_of_unity(log_n, inverse=False):
    """ark-ff: TWO_ADIC_ROOT_OF_UNITY^(2^(47 - log_n)), or its inverse."""
    w = pow(FR_TWO_ADIC_ROOT, 1 << (47 - log_n), FR_MODULUS)
    return pow(w, FR_MODULUS - 2, FR_MODULUS) if inverse else w


def blocks_rows(log_n, world, rank):
    """Global indices, in local order, of the BLOCKS layout of `rank`: what a row-sharded mat-vec computes."""
    m = (1 << log_n) // world
    blk = m // world
    return [m * k1 + rank * blk + t for k1 in range(world) for t in range(blk)]


def cyclic_rows(log_n, world, rank):
    return list(range(rank, 1 << log_n, world))


def sharded_ntt(local, log_n, rank, world, inverse, blocks_in, local_ntt, alltoall, fr_mul_scalars, fr_add):
    """local: (m, 4) uint64 Montgomery limbs of this rank.  Injected kernels:
         local_ntt(arr, log_m, inverse) -> arr          the length-m transform (inverse scales by 1 / m)
         alltoall(list of G (blk, 4) arrays) -> list    chunk c goes to rank c; entry i of the result came from rank i
         fr_mul_scalars(arr, ints) -> arr               arr[x] * ints[x] (ints: standard-form integers)
         fr_add(a, b) -> a + b                          elementwise
    Returns the rank's part of the transformed vector in the OTHER layout."""
    G = world
    log_g = G.bit_length() - 1
    assert 1 << log_g == G and log_n >= 2 * log_g
    m = (1 << log_n) >> log_g
    blk = m >> log_g
    w = fr_root_of_unity(log_n, inverse)
    wg = pow(w, m, FR_MODULUS)
    g_inv = pow(G, FR_MODULUS - 2, FR_MODULUS) if inverse else 1

    def cross(parts):  # out[k] = scale * sum_i wg^(i k) parts[i]
        out = []
        for k in range(G):
            acc = parts[0]
            for i in range(1, G):
                e = (i * k) % G
                acc = fr_add(acc, fr_mul_scalars(parts[i], [pow(wg, e, FR_MODULUS)] * blk) if e else parts[i])
            out.append(fr_mul_scalars(acc, [g_inv] * blk) if inverse else acc)
        return out

    if not blocks_in:
        y = local_ntt(np.ascontiguousarray(local), log_n - log_g, inverse)
        if rank:
            y = fr_mul_scalars(y, [pow(w, rank * k2, FR_MODULUS) for k2 in range(m)])
        got = alltoall([np.ascontiguousarray(y[c * blk:(c + 1) * blk]) for c in range(G)])
        return np.ascontiguousarray(np.concatenate(cross(got)))
    t = cross([np.ascontiguousarray(local[i * blk:(i + 1) * blk]) for i in range(G)])
    t = [fr_mul_scalars(t[k1], [pow(w, (rank * blk + x) * k1, FR_MODULUS) for x in range(blk)]) if k1 else t[k1] for k1 in range(G)]
    u = np.ascontiguousarray(np.concatenate(alltoall(t)))
    return local_ntt(u, log_n - log_g, inverse)
