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
