#!/usr/bin/env python3
"""bench.py — headline measurement of the hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload msm|prove] [--log-n L]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM.
  workload msm   : one G1 MSM over 2^L SRS-shaped bases [tau^i]G with uniform Fr scalars (Montgomery form in HBM,
                   i.e. polynomial coefficients as the prover hands them to KZG commit).  value = points/s.
With N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) every rank owns a 2^L-point shard of a
N*2^L-point MSM; the per-rank Jacobian partials (144 B) are all-gathered and folded on every rank ("weak" scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "Marlin prove() constraints/sec at 2^20 R1CS; G1 MSM points/sec"


def cpu_baseline_msm(orc, log_n_sample, tau, G):
    """arkworks-algorithm CPU restatement (oracle/oracle.c: VariableBaseMSM, one task per window) on a bounded
    sample of the same workload, timed on this host."""
    from pyref.prng import fr_array
    n = 1 << log_n_sample
    bases = orc.srs_bases(n, tau, G)
    sc = fr_array(n, 7)
    nwin = (253 + orc.lib.oracle_msm_window(n) - 1) // orc.lib.oracle_msm_window(n)
    threads = max(1, min(nwin, orc.lib.oracle_max_threads(), os.cpu_count() or 1))
    t0 = time.perf_counter()
    orc.msm(bases, sc, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": "one 2^%d-point G1 MSM, arkworks Pippenger (c=%d, %d windows, one thread per window), %.2f s"
                      % (log_n_sample, orc.lib.oracle_msm_window(n), nwin, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="msm", choices=["msm"])
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cpu-log-n", type=int, default=18)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import simpleworks_amd as swm
    from oracle_lib import Oracle, golden, h2i  # oracle: input generation + cpu_baseline leg only
    from pyref.prng import fr_array

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    ctx = swm.Context(local_rank)
    orc = Oracle()
    n = 1 << args.log_n
    tau = h2i(golden("msm.json")["tau"])
    G = orc.points_to_mont([tuple(h2i(v) for v in golden("g1.json")["generator"])])
    # rank r owns bases [tau^(r n) .. tau^((r+1) n)) of the global SRS: shift the generator by tau^(r n)
    shift = pow(tau, rank * n, int("12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001", 16))
    from oracle_lib import ints_to_limbs
    G_r = orc.fixed_base_mul(G, ints_to_limbs([shift], 4), threads=1) if rank else G
    bases = orc.srs_bases(n, tau, np.ascontiguousarray(G_r.reshape(1, 12)))
    bh = ctx.srs_upload(bases)
    d_sc = ctx.to_device(fr_array(n, 1000 + rank))  # any reduced limbs are valid Montgomery residues
    del bases

    def step():
        part = ctx.msm_g1_dev(bh, d_sc, n, True)
        if world > 1:
            t = torch.from_numpy(part.view(np.int64)).cuda()
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            acc = out[0].cpu().numpy().view(np.uint64)
            for o in out[1:]:
                acc = ctx.g1_add_jac(acc, o.cpu().numpy().view(np.uint64))
            return acc
        return part

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    ctx.profile_enable(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof = ctx.profile()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_points = n * world * args.steps
        dom = prof["msm_accumulate"]
        alg_bytes = 128.0 * n  # 96 B affine base + 32 B scalar per point, n points per launch
        achieved = alg_bytes / (dom["avg_ms"] * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": total_points / dt, "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32x12 (384-bit Montgomery Fq) / u32x8 (Fr)", "data": "synthetic",
            "config": {"workload": "g1_msm: 2^%d SRS-shaped bases [tau^i]G per GPU, uniform Fr scalars resident in HBM"
                                   % args.log_n, "points_per_gpu": n, "sharding": "point-range, all-gather of 144-B partials"},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": dom["avg_ms"], "algorithmic_bytes_per_launch": alg_bytes},
            "kernels_ms_per_step": {k: v["total_ms"] / args.steps for k, v in sorted(prof.items())},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_msm(orc, args.cpu_log_n, tau, G)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
