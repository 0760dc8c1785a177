#!/usr/bin/env python3
"""bench.py — headline measurement of the hot path on MI355X (contract: task statement; notes in DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload prove|msm|prove_sharded] [--log-n L]

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM.
  workload prove (default) : one Marlin prove() of the 2^L-constraint synthetic R1CS (BASELINE.json configs[2]; L = 20):
                             universal SRS and proving key are built once, untimed, and stay device resident; the
                             timed region is generate_proof (witness upload included).  value = constraints/s.
  workload msm             : one G1 MSM over 2^L SRS-shaped bases [tau^i]G with uniform Montgomery scalars in HBM.
                             value = points/s.
With N > 1 (torch.distributed.run, one rank per GPU, RCCL): `prove` runs one independent proof per rank (replicas,
weak scaling, no data-path collective); `msm` gives each rank a 2^L-point shard of an N*2^L-point MSM and folds the
144-byte Jacobian partials after an all-gather; `prove_sharded` (strong scaling, not the default) runs ONE proof over
all ranks: every commitment MSM split by point range, partial sums all-gathered, everything else replicated
(simpleworks_amd.dist.enable_sharded_prover).  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "Marlin prove() constraints/sec at 2^20 R1CS; G1 MSM points/sec"
FR_R = int("12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001", 16)


def _oracle_inputs(native=False):
    """The CPU checker.  native=True (cpu_baseline leg only): a copy compiled on THIS host with -O3 -march=native
    (oracle/Makefile `native`) when gcc is present, so that the CPU column is not handicapped by portable code."""
    from oracle_lib import Oracle, golden, h2i, ORACLE_DIR
    path, flags = None, "-O3 (portable build shipped with the repo)"
    if native:
        import subprocess
        try:
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            path = os.path.join(ORACLE_DIR, "_build", "liboracle_native.so")
            flags = "-O3 -march=native -fopenmp, gcc on the bench host"
        except Exception:
            path = None
    orc = Oracle(path)
    tau = h2i(golden("msm.json")["tau"])
    G = orc.points_to_mont([tuple(h2i(v) for v in golden("g1.json")["generator"])])
    return orc, tau, G, flags


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline_msm(log_n_sample):
    """arkworks-algorithm CPU restatement (oracle/oracle.c: VariableBaseMSM, one thread per window) on a bounded
    sample of the MSM workload, timed on this host: all window threads, and one thread."""
    from pyref.prng import fr_array
    orc, tau, G, flags = _oracle_inputs(native=True)
    n = 1 << log_n_sample
    bases = orc.srs_bases(n, tau, G)
    sc = fr_array(n, 7)
    c = orc.lib.oracle_msm_window(n)
    nwin = (253 + c - 1) // c
    threads = max(1, min(nwin, orc.lib.oracle_max_threads(), os.cpu_count() or 1))
    t0 = time.perf_counter()
    orc.msm(bases, sc, threads=threads)
    dt = time.perf_counter() - t0
    n1 = n >> 2
    t0 = time.perf_counter()
    orc.msm(np.ascontiguousarray(bases[:n1]), np.ascontiguousarray(sc[:n1]), threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n / dt, "unit": "points/s", "cores": threads, "kind": "port", "cpu_model": _cpu_model(),
            "host_threads": os.cpu_count(), "build": flags,
            "sample": "one 2^%d-point G1 MSM, arkworks Pippenger (c=%d, %d windows, one thread per window as ark-ec's "
                      "`parallel` feature), %.2f s" % (log_n_sample, c, nwin, dt),
            "one_thread": {"value": n1 / dt1, "unit": "points/s", "cores": 1,
                           "sample": "one 2^%d-point MSM on one thread, %.2f s" % (log_n_sample - 2, dt1)}}


def cpu_baseline_prove(calls, N, mats, z, log_shift_1t):
    """CPU port of one prove(): the oracle's arkworks-algorithm kernels (Pippenger MSM with arkworks' window rule and
    one thread per window, radix-2 FFT and row-parallel mat-vec with OpenMP where arkworks' `parallel` feature uses
    rayon) replayed over the K1-K3 call list the library LOGGED for one GPU proof of this workload,
    at full size on all host threads, and with every size divided by 2^log_shift_1t on one thread.  The pointwise work
    between the kernels is not replayed (a few % of a CPU proof).  constraints/s = N / total time."""
    from pyref.prng import fr_array
    orc, tau, G, flags = _oracle_inputs(native=True)
    all_threads = max(1, min(orc.lib.oracle_max_threads(), os.cpu_count() or 1))
    msm_sizes = [v for k, v in calls if k == "m"]
    ntt_logs = [v for k, v in calls if k == "n"]
    spmv = [v for k, v in calls if k == "s"]
    max_m = max(msm_sizes)
    bases = orc.srs_bases(max_m, tau, G)
    sc = fr_array(max_m, 8)
    x = orc.fr_to_mont(fr_array(1 << max(ntt_logs), 9))

    def replay(shift, threads):
        t = {"msm": 0.0, "ntt": 0.0, "spmv": 0.0}
        t0 = time.perf_counter()
        for m in msm_sizes:
            m = max(m >> shift, 1)
            orc.msm(np.ascontiguousarray(bases[:m]), np.ascontiguousarray(sc[:m]), threads=threads)
        t["msm"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        for lg in ntt_logs:
            lg = max(lg - shift, 1)
            orc.ntt(np.ascontiguousarray(x[: 1 << lg]), lg, 0, 0, threads)
        t["ntt"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(len(spmv)):
            rowptr, col, val = mats[i % len(mats)]
            rows = max((len(rowptr) - 1) >> shift, 1)
            rp = np.ascontiguousarray(rowptr[: rows + 1])
            orc.spmv(rp, col, val, z, threads=threads)
        t["spmv"] = time.perf_counter() - t0
        return t
    ta = replay(0, all_threads)
    t1 = replay(log_shift_1t, 1)
    tot_a, tot_1 = sum(ta.values()), sum(t1.values())
    desc = "%d MSMs (%.1f N points), %d NTTs (%.1f N elements), %d mat-vecs" % (
        len(msm_sizes), sum(msm_sizes) / N, len(ntt_logs), sum(1 << l for l in ntt_logs) / N, len(spmv))
    return {"value": N / tot_a, "unit": "constraints/s", "cores": all_threads, "kind": "port",
            "cpu_model": _cpu_model(), "host_threads": os.cpu_count(), "build": flags,
            "sample": "the K1-K3 call list of ONE proof at full size (N = %d: %s) replayed on the arkworks-algorithm CPU "
                      "restatement with %d threads: MSM %.1f s (window-parallel: at most %d busy), NTT %.1f s, mat-vec %.2f s"
                      % (N, desc, all_threads, ta["msm"], (253 + orc.lib.oracle_msm_window(max_m) - 1) // orc.lib.oracle_msm_window(max_m),
                         ta["ntt"], ta["spmv"]),
            "one_thread": {"value": (N >> log_shift_1t) / tot_1, "unit": "constraints/s", "cores": 1,
                           "sample": "the same call list with every size divided by %d (N = %d) on ONE thread: MSM %.1f s, "
                                     "NTT %.1f s, mat-vec %.2f s" % (1 << log_shift_1t, N >> log_shift_1t, t1["msm"], t1["ntt"], t1["spmv"])}}


def _run_sharded_children(args, sizes, rank, world, device_index, dist, torch, coll_dev):
    """Parent side of the sharded leg: broadcast rank 0's ncclUniqueId, start this rank's child, collect one JSON line per size
    under a deadline.  Returns a list (one entry per size) of per-rank result dicts, {"error": ...} from the first failure on."""
    import queue
    import subprocess
    import threading
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SWM_BENCH_FORCE_DIST")}

    def spawn(id_arg):
        cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", id_arg, "--child-rank", str(rank), "--child-world", str(world),
               "--child-device", str(device_index), "--sharded-log-n", ",".join(str(x) for x in sizes), "--circuit", args.circuit]
        if args.r1cs:
            cmd += ["--r1cs", args.r1cs]
        c = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, env=env, text=True)
        q = queue.Queue()

        def reader():  # (reads a pipe: no GPU, no collective — safe to leave behind)
            for line in c.stdout:
                q.put(line)
            q.put(None)
        threading.Thread(target=reader, daemon=True).start()
        return c, q
    # The ncclUniqueId is created by rank 0's CHILD — the process (and the librccl: the children do not import torch, so theirs is
    # the loader's copy, not torch's) that will also call ncclCommInitRank with it — and travels child -> parent -> torch broadcast
    # -> the other parents -> their children's command lines.  (An id made by this process would come from torch's RCCL build.)
    child, lines, id_hex, err = None, None, "", None
    if rank == 0:  # (whatever happens here, rank 0 takes part in the broadcast below: zeros tell the others that there is no id)
        try:
            child, lines = spawn("create")
            first = lines.get(timeout=float(os.environ.get("SWM_BENCH_SHARDED_TIMEOUT", "300")))
            rec = json.loads(first) if first else {}
            id_hex = rec.get("id", "")
            if not id_hex:
                err = rec.get("error", "rank 0's child produced no ncclUniqueId")
        except Exception as e:  # noqa: BLE001
            err = "unique id: %r" % (e,)
    try:
        idt = torch.zeros(128, dtype=torch.uint8, device=coll_dev)
        if rank == 0 and len(id_hex) == 256:
            idt = torch.frombuffer(bytearray(bytes.fromhex(id_hex)), dtype=torch.uint8).to(coll_dev)
        dist.broadcast(idt, 0)
        id_hex = bytes(idt.cpu().numpy().tobytes()).hex()
        if not any(idt.cpu().numpy().tolist()):
            raise RuntimeError("rank 0's child produced no ncclUniqueId")
        if rank != 0:
            child, lines = spawn(id_hex)
    except Exception as e:  # noqa: BLE001
        err = err or "unique id: %r" % (e,)
    if err:
        if child is not None and child.poll() is None:
            child.kill()
        return [{"error": err}]
    out = []
    for lg_s in sizes:
        budget = float(os.environ.get("SWM_BENCH_SHARDED_TIMEOUT", "300" if lg_s <= 20 else "480"))
        try:
            line = lines.get(timeout=budget)
        except queue.Empty:
            line = "TIMEOUT"
        if line is None or line == "TIMEOUT":
            out.append({"error": ("timed out after %.0f s" % budget) if line else "the child exited without a result (status %s)" % child.poll(),
                        "log_n": lg_s})
            break
        try:
            res = json.loads(line)
        except ValueError:
            res = {"error": "unparsable child output: %r" % line[:200]}
        out.append(res)
        if "error" in res:
            break
    if child.poll() is None:
        if out and "error" in out[-1]:
            child.kill()  # exactly the process started above
        try:
            child.wait(timeout=60)
        except subprocess.TimeoutExpired:
            child.kill()
    return out


def sharded_child(args):
    """One rank of the sharded leg, in a process of its own: no torch, no process group — the library's communicator
    (swm_rccl_init with the id the parent handed over) is the only thing that connects the ranks.  One JSON line per size."""
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)  # RCCL's banner goes to stderr
    rank, world = args.child_rank, args.child_world
    import simpleworks_amd as swm
    from simpleworks_amd import marlin as M
    from simpleworks_amd import workloads as W
    from simpleworks_amd._lib import rccl_info

    def emit(obj):
        json_out.write(json.dumps(obj) + "\n")
        json_out.flush()
    try:
        ctx = swm.Context(args.child_device)
        M.set_default_context(ctx)
        if args.sharded_child == "create":  # rank 0: the id comes from THIS process's librccl and goes to the parent first
            from simpleworks_amd._lib import rccl_unique_id
            uid = rccl_unique_id()
            emit({"id": uid.hex()})
        else:
            uid = bytes.fromhex(args.sharded_child)
        ctx.rccl_init(uid, rank, world)
    except Exception as e:  # noqa: BLE001
        emit({"error": "rank %d: %r [%s]" % (rank, e, rccl_info()[1])})
        return 1
    for lg_s in [int(x) for x in args.sharded_log_n.split(",") if x]:
        try:
            rng_s = M.generate_rand()
            if args.r1cs:
                scs, pub_s = W.load_r1cs(args.r1cs)
            elif args.circuit == "merkle":
                mcs_s, pub_s, _ = W.merkle_membership_circuit(leaf_u8=0xA7)
                scs = mcs_s.pack()
            else:
                scs, pub_s = W.synthetic_r1cs(1 << lg_s, 0x1234567, 0x7654321)  # the SAME system on every rank
            ns = scs.num_constraints
            nv = scs.instance.shape[0] + scs.witness.shape[0]
            srs_s = M.generate_universal_srs(ns, nv if (args.r1cs or args.circuit == "merkle") else ns,
                                             max(int(m[0][-1]) for m in scs.mats), rng_s)
            pk_s, vk_s = M.generate_proving_and_verifying_keys(srs_s, scs)
            srs_s.free()
            M.generate_proof(scs, pk_s, M.rng_from_seed(bytes(32)))
            c0, b0 = ctx.exchange_stats()
            reps = 3
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                pr = M.generate_proof(scs, pk_s, M.rng_from_seed(bytes(32)))
            ctx.synchronize()
            dts = time.perf_counter() - t0
            c1, b1 = ctx.exchange_stats()
            emit({"constraints": ns, "seconds": dts, "reps": reps, "rccl": rccl_info()[1], "sha256": hashlib.sha256(pr.data).hexdigest(),
                  "exchanges_per_proof": (c1 - c0) / reps, "bytes_per_rank_per_proof": (b1 - b0) / reps,
                  "verifies": bool(M.verify_proof(vk_s, pub_s, pr, M.generate_rand()))})
            pk_s.free()
        except Exception as e:  # noqa: BLE001
            emit({"error": "rank %d, 2^%d: %r [%s]" % (rank, lg_s, e, rccl_info()[1])})
            return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # (the first two proofs after set-up run ~2 % slower: clocks, first-touch of scratch)
    ap.add_argument("--workload", default="prove", choices=["prove", "msm", "prove_sharded"])
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--circuit", default="synthetic", choices=["synthetic", "merkle"],
                    help="prove workloads: `synthetic` = the 2^L-row a*b=c circuit of BASELINE configs[1-3] (default, the "
                         "headline); `merkle` = the Pedersen-hash Merkle-membership circuit of BASELINE configs[4] "
                         "(tree over 2^18 leaves, + the simpleworks UInt8 gadget block; --log-n is ignored)")
    ap.add_argument("--r1cs", default=None, metavar="FILE",
                    help="prove workloads: the constraint system of an SWMR1CS1 dump (simpleworks_amd/workloads.py; written on the "
                         "Rust side by swmarlin_sys::r1cs_dump::dump_r1cs, INTEGRATION.md) instead of a built-in circuit — how the "
                         "reference's own circuits, e.g. MerkleTreeVerificationU8 at height 19 (BASELINE configs[4]), are timed")
    ap.add_argument("--rng", default="builtin", choices=["builtin", "callback", "adopt"],
                    help="where the prover's randomness comes from in the TIMED loop: `builtin` = the library's ChaCha12 "
                         "(ark_std::test_rng's stream; the headline), `callback` = a caller-owned generator behind "
                         "swm_rng_from_callback (what a binding that keeps `&mut StdRng` gets; the caller here is a host "
                         "ChaCha12 standing for StdRng), `adopt` = the caller's ChaCha state handed over with "
                         "swm_rng_from_chacha and written back after every proof.  Whatever is chosen, the line carries a "
                         "`drop_in_rng` object with the other modes measured over a few proofs.")
    ap.add_argument("--sharded-log-n", default=None,
                    help="N > 1, workload prove: sizes (comma-separated log2 of the constraint count) of the `sharded` leg — ONE proof "
                         "over all ranks through the library's RCCL exchange.  Default: the --log-n size, then 2^22 (BASELINE "
                         "configs[3]) when the headline size is 2^20")
    ap.add_argument("--sharded-child", default=None, metavar="ID_HEX", help=argparse.SUPPRESS)  # internal: one rank of the sharded leg
    ap.add_argument("--child-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--child-world", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--child-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-drop-in", action="store_true", help="skip the drop_in_rng proofs after the timed loop (profiling runs)")
    ap.add_argument("--overlap", type=int, default=0, metavar="K",
                    help="N = 1, workload prove: after the timed loop, K contexts on THIS GPU (all attached to the ONE resident key, one host "
                         "thread each) prove concurrently; reported as the `overlapped` object — what a service that keeps K proofs in "
                         "flight gets out of one GPU.  `value` stays the one-proof-at-a-time figure.")
    ap.add_argument("--cpu-log-n", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true",
                    help="bracket EVERY kernel launch with HIP events (per-kernel table in the JSON line; costs ~2 %% of a "
                         "2^20 proof) instead of the dominant kernel only")
    args = ap.parse_args()
    if args.sharded_child:
        sys.exit(sharded_child(args))

    # The contract is ONE JSON line on stdout.  RCCL prints a banner (hostname, library path, ...) to stdout when its
    # communicator comes up, so everything written to file descriptor 1 from here on is sent to stderr and the JSON line
    # goes to a private duplicate of the original stdout at the very end.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import simpleworks_amd as swm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SWM_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, barriers, collectives) even with one rank —
    # lets a single-GPU box exercise the RCCL calls the N > 1 runs make
    use_dist = world > 1 or bool(os.environ.get("SWM_BENCH_FORCE_DIST"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hook for single-GPU boxes: SWM_BENCH_BACKEND=gloo SWM_BENCH_DEVICE=0 runs all ranks on one device (the
    # collectives then carry host tensors).  The driver's multi-GPU runs use the default: RCCL, one rank per GPU.
    backend = os.environ.get("SWM_BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("SWM_BENCH_DEVICE", local_rank))
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if use_dist:
        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    ctx = swm.Context(device_index)
    n = 1 << args.log_n

    if args.workload == "msm":
        from oracle_lib import ints_to_limbs
        from pyref.prng import fr_array
        orc, tau, G, _ = _oracle_inputs()  # oracle used for INPUT generation only ([tau^i]G bases)
        shift = pow(tau, rank * n, FR_R)
        G_r = orc.fixed_base_mul(G, ints_to_limbs([shift], 4), threads=1) if rank else G
        bases = orc.srs_bases(n, tau, np.ascontiguousarray(G_r.reshape(1, 12)))
        bh = ctx.srs_upload(bases)
        d_sc = ctx.to_device(fr_array(n, 1000 + rank))  # any reduced limbs are valid Montgomery residues
        del bases

        def step():
            part = ctx.msm_g1_dev(bh, d_sc, n, True)
            if use_dist:
                t = torch.from_numpy(part.view(np.int64)).to(coll_dev)
                out = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(out, t)
                acc = out[0].cpu().numpy().view(np.uint64)
                for o in out[1:]:
                    acc = ctx.g1_add_jac(acc, o.cpu().numpy().view(np.uint64))
                return acc
            return part
        dominant, units, unit = "msm_accumulate", n, "points/s"
        alg_bytes = 128.0 * n  # 96 B affine base + 32 B scalar per point, n points per launch
        workload = "g1_msm: 2^%d SRS-shaped bases [tau^i]G per GPU, uniform Fr scalars resident in HBM" % args.log_n
    else:
        from simpleworks_amd import marlin as M
        from simpleworks_amd import workloads as W
        M.set_default_context(ctx)
        sharded = args.workload == "prove_sharded"
        if sharded and use_dist:
            from simpleworks_amd.dist import enable_sharded_prover
            enable_sharded_prover(ctx)
        rng = M.generate_rand()
        if args.r1cs:
            cs, public = W.load_r1cs(args.r1cs)
            n = cs.num_constraints
            nvars = cs.instance.shape[0] + cs.witness.shape[0]
            nnz = max(int(m[0][-1]) for m in cs.mats)
            assert cs.is_satisfied(ctx), "bench: the assignment of %s does not satisfy its constraints" % args.r1cs
            srs = M.generate_universal_srs(n, nvars, nnz, rng)
        elif args.circuit == "merkle":
            # configs[4] for real: the tree over 2^18 u8 leaves is built first (MerkleTree::new,
            # src/merkle_tree/simple_merkle_tree.rs:47-49 — on the GPU, swm_merkle_tree_build), the proof is for one of its paths
            mparams = W.MerkleParams()
            leaves = np.random.default_rng(18).integers(0, 256, size=1 << 18, dtype=np.uint8)
            idx = 0x2A5A7 + (0 if sharded else rank)
            mparams.build_tree(leaves[:2], ctx)                 # parameter tables resident, kernels loaded
            t_tree = time.perf_counter()
            nodes = ctx.merkle_tree_build(mparams.crh(ctx)[0].h, mparams.crh(ctx)[1].h, leaves.reshape(-1, 1))
            merkle_tree_ms = (time.perf_counter() - t_tree) * 1e3
            path, off, cnt = [], 0, 1 << 18
            for lvl in range(18):
                path.append(int.from_bytes(nodes[off + ((idx >> lvl) ^ 1)].tobytes(), "little"))
                off, cnt = off + cnt, cnt >> 1
            mcs, public, _ = W.merkle_membership_circuit(leaf_u8=int(leaves[idx]), leaf_index=idx, params=mparams, siblings=path)
            assert public[0] == int.from_bytes(nodes[-1].tobytes(), "little"), "bench: the circuit's root is not the tree's root"
            cs = mcs.pack()
            n = cs.num_constraints
            nvars = cs.instance.shape[0] + cs.witness.shape[0]
            nnz = max(int(m[0][-1]) for m in cs.mats)
            srs = M.generate_universal_srs(n, nvars, nnz, rng)
        else:
            srs = M.generate_universal_srs(n, n, n, rng)
            cs, public = W.synthetic_r1cs(n, 0x1234567 + (0 if sharded else rank), 0x7654321)
        pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
        srs.free()
        last = {}
        caller = M.generate_rand()  # "the caller's StdRng" of the callback mode (host ChaCha12, AVX2, test_rng seed)
        cb_rng = M.rng_behind_callback(caller)
        adopt_pos = [0]             # "the caller's StdRng" of the adopt mode: (TEST_RNG_SEED, word position)

        def prove_with(mode):
            if mode == "callback":      # every draw goes through fill_bytes of the caller's generator
                return M.generate_proof(cs, pk, cb_rng)
            if mode == "adopt":         # get_seed / get_word_pos in, set_word_pos out: the caller's stream, drawn on the GPU
                r = M.rng_from_chacha(M.TEST_RNG_SEED, adopt_pos[0], 12)
                proof = M.generate_proof(cs, pk, r)
                adopt_pos[0] = r.word_pos()   # what the binding writes back with rng.set_word_pos(..)
                return proof
            return M.generate_proof(cs, pk, rng)

        def step():
            last["proof"] = prove_with(args.rng)
        dominant, units, unit = "msm_accumulate", n, "constraints/s"
        alg_bytes = None
        if args.r1cs:
            workload = ("marlin_prove: constraint system of the SWMR1CS1 dump %s: %d constraints, %d variables, max nnz %d, "
                        "SRS + proving key device resident" % (os.path.basename(args.r1cs), n, nvars, nnz))
        elif args.circuit == "merkle":
            workload = ("marlin_prove: Pedersen-hash Merkle-membership circuit (BASELINE configs[4] stand-in: tree height 19 = "
                        "2^18 leaves, 256-bit digests, + 2400 simpleworks UInt8 gadget ops): %d constraints, %d variables, "
                        "max nnz %d (|H| = 2^17, |K| = 2^18), SRS + proving key device resident; the path is one of a real tree over "
                        "2^18 leaves built on the GPU before the timed region (%.1f ms incl. PCIe, not part of `value`)"
                        % (n, nvars, nnz, merkle_tree_ms))
        else:
            workload = ("marlin_prove: synthetic R1CS, 2^%d constraints = variables = non-zeros per matrix (|H| = |K| = 2^%d), "
                        "SRS + proving key device resident" % (args.log_n, args.log_n))

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    ctx.profile_enable(1 if args.profile_all else 2)  # HIP events around the roofline kernels (msm_accumulate, ntt_pass, spmv_*) or around everything
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof = ctx.profile()
    work_timed, calls_timed = dict(ctx.last_work), list(ctx.last_calls)
    standalone = {}
    if args.workload != "msm":
        verify_ms = []
        for _ in range(3):  # verify_proof runs on the host (one thread): quoted beside the proof it checks, not part of `value`
            tv = time.perf_counter()
            assert M.verify_proof(vk, public, last["proof"], M.generate_rand()), "bench: proof does not verify"
            verify_ms.append((time.perf_counter() - tv) * 1e3)
        verify_ms = sorted(verify_ms)[1]
        if rank == 0:
            # the secondary kernels ALONE on the chip (inside a proof they share it with accumulations): five transforms of
            # the proof's largest size and three mat-vecs with its A matrix, bracketed by the same HIP events
            lg = max(v for k, v in calls_timed if k == "n")
            buf = ctx.to_device(np.random.default_rng(1).integers(0, 1 << 60, size=(1 << lg, 4), dtype=np.uint64))
            ctx.ntt_fr_dev(buf, lg)
            ctx.profile_reset()
            ctx.profile_enable(2)
            for _ in range(5):
                ctx.ntt_fr_dev(buf, lg)
            ctx.synchronize()
            ctx.profile_enable(False)
            p2 = ctx.profile()
            buf.free()
            ms = p2["ntt_pass"]["total_ms"] / 5
            standalone["ntt_pass"] = {"log_n": lg, "ms": ms, "achieved": 64.0 * (1 << lg) / (ms * 1e-3) / 1e9}
            # the mat-vec with everything RESIDENT (as inside a proof: matrix, z and the result in HBM) and only the kernels that do
            # the product — the row-statistics kernel the plan-less entry point runs first (inside a proof the plan comes from the key)
            # is not part of it.  (Until r05 this probe timed the host-pointer form and summed every spmv_* event: 6.3 % of HBM.)
            zvec = np.ascontiguousarray(np.concatenate([cs.instance, cs.witness]))
            rp, cl, vl = cs.mats[0]
            dm = [ctx.to_device(a) for a in (rp, cl, vl, zvec)]
            dout = ctx.alloc((len(rp) - 1) * 32)
            ctx.spmv_fr_dev(dm[0], dm[1], dm[2], dm[3], dout, len(rp) - 1)
            ctx.synchronize()
            ctx.profile_reset()
            ctx.profile_enable(2)
            for _ in range(5):
                ctx.spmv_fr_dev(dm[0], dm[1], dm[2], dm[3], dout, len(rp) - 1)
            ctx.synchronize()
            ctx.profile_enable(False)
            p3 = ctx.profile()
            for x in dm + [dout]:
                x.free()
            ms = sum(v["total_ms"] for k, v in p3.items() if k.startswith("spmv_") and k != "spmv_row_stats") / 5
            b = 68.0 * int(rp[-1]) + 36.0 * (len(rp) - 1)
            standalone["spmv"] = {"rows": len(rp) - 1, "nnz": int(rp[-1]), "ms": ms, "achieved": b / (ms * 1e-3) / 1e9,
                                  "form": "device-resident operands, product kernels only"}
    drop_in = None
    if args.workload == "prove" and rank == 0 and not use_dist and not args.no_drop_in:
        # what a source-compatible caller (its own `&mut StdRng`) sees, next to the headline: a few proofs per mode
        drop_in = {}
        for mode in ("builtin", "callback", "adopt"):
            prove_with(mode)
            ctx.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                pr = prove_with(mode)
            ctx.synchronize()
            drop_in[mode] = {"ms_per_step": (time.perf_counter() - t1) / 3 * 1e3}
            assert M.verify_proof(vk, public, pr, M.generate_rand()), "bench: %s-rng proof does not verify" % mode
        drop_in["note"] = ("builtin = library ChaCha12 (the headline mode); callback = swm_rng_from_callback over a host ChaCha12 "
                           "(AVX2, one thread) standing for the caller's StdRng: the 3|H| mask coefficients travel through "
                           "fill_bytes while the GPU works on the rest of round 1; adopt = swm_rng_from_chacha with the caller's "
                           "(seed, word position), position written back: the caller's stream word for word, drawn on the GPU")
    binding = None
    if args.workload == "prove" and rank == 0 and not use_dist and not args.no_drop_in:
        # What generate_proof costs a caller of the Rust binding ON TOP of swm_generate_proof: tests/native/dropin_harness.cpp
        # replays the shim's per-proof call sequence (key lookup by vk digest in the process-wide cache, assignment-only pack —
        # no to_matrices(), NULL matrix pointers —, swm_generate_proof on a context of the thread's own with the SHARED key,
        # proof bytes out) — here on one thread, with the assignment handed over as a pointer view and as a copy.
        try:
            import dropin_lib
            binding = {}
            for mode in ("view", "copy"):
                rep, _ = dropin_lib.run(pk, vk, cs.pack_assignment(), threads=1, proofs_per_thread=8, rng_key=M.TEST_RNG_SEED,
                                        rng_word_pos=0, pack_mode=mode)
                binding[mode] = {k: rep[k] for k in ("ms_per_proof", "key_lookup_ms", "host_pack_ms", "prove_call_ms",
                                                     "checked_deserialize_proxy_ms", "binding_overhead_ms")}
            binding["host_pack_ms"] = binding["view"]["host_pack_ms"]
            binding["binding_overhead_ms"] = binding["view"]["binding_overhead_ms"]
            binding["note"] = ("per proof, one thread, a context of its own on the key this bench proved with (swm_pk_attach): "
                               "binding_overhead_ms = key lookup (vk bytes -> Blake2s -> cache) + assignment pack; `view` = the "
                               "assignment vectors handed over where they are (Vec<Fr> = Montgomery limbs), `copy` = flattened per "
                               "proof; no matrices cross the boundary at prove time; the proof comes back as serialize_uncompressed bytes "
                               "(swm_generate_proof_ex) for Proof::deserialize_unchecked.  checked_deserialize_proxy_ms (not part of the "
                               "overhead): what the checked Proof::deserialize of the compressed bytes would add per proof, measured on "
                               "the library's own checked reader (swm_proof_validate)")
        except Exception as e:  # noqa: BLE001 — a missing g++ on the bench box must not cost the headline line
            binding = {"error": "%s: %s" % (type(e).__name__, e)}
    overlapped = None
    if args.workload == "prove" and rank == 0 and not use_dist and args.overlap >= 2:
        # K proofs in flight on one GPU: a proof leaves the chip under-used whenever no accumulation is in flight (~13 of 50 ms at
        # 2^20: preludes, sorts, the last bucket stage of a round, host turnarounds) — another proof's accumulations fill that.
        # ONE resident key: every extra context attaches to the key of the timed loop (swm_pk_attach), only scratch is per context.
        import threading
        import simpleworks_amd as swm_pkg
        free_before = ctx.mem_info()[0]
        workers = [(ctx, pk)]
        for _ in range(args.overlap - 1):
            c2 = swm_pkg.Context(device_index)
            workers.append((c2, pk.attach(c2)))
        reps = max(3, min(args.steps, 10))
        proofs = [None] * len(workers)
        errors = []
        packed_a = cs.pack_assignment()

        def run(i, count):
            try:
                r_i = M.generate_rand()
                for _ in range(count):
                    proofs[i] = M.generate_proof(packed_a, workers[i][1], r_i)
            except Exception as e:  # noqa: BLE001
                errors.append((i, e))
        walls = {}
        for phase, count in (("warm", 2), ("timed", reps)):
            ths = [threading.Thread(target=run, args=(i, count)) for i in range(len(workers))]
            for c_i, _ in workers:
                c_i.synchronize()
            t1 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            for c_i, _ in workers:
                c_i.synchronize()
            walls[phase] = time.perf_counter() - t1
            if errors:
                break
        hbm_extra = free_before - ctx.mem_info()[0]
        for c_i, pk_i in workers[1:]:
            pk_i.free()
            c_i.close()
        if errors:
            raise RuntimeError("bench: overlapped proof on worker %d failed: %r" % errors[0])
        for i in range(len(workers)):
            assert M.verify_proof(vk, public, proofs[i], M.generate_rand()), "bench: an overlapped proof does not verify"
        wall = walls["timed"]
        total = reps * len(workers)
        overlapped = {"contexts": len(workers), "proofs": total, "ms_per_proof": wall / total * 1e3,
                      "constraints_per_s": n * total / wall, "latency_ms_per_proof": wall / reps * 1e3,
                      "shared_key": True, "hbm_added_by_extra_contexts_bytes": hbm_extra,
                      "note": "K contexts on one GPU sharing ONE resident key (swm_pk_attach), one host thread each, their proofs "
                              "concurrent; ms_per_proof = wall / proofs"}
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # N > 1, workload prove: besides the replicas figure (value), ONE proof over all ranks through the library's own RCCL
    # exchange (swm_rccl_init; point-range sharded commitment MSMs, one ncclAllGather per prover round) is measured and
    # reported as the `sharded` sub-object.  The leg runs in a FRESH CHILD PROCESS per rank (r06): this process only
    # hands the ncclUniqueId over (a torch broadcast), starts `bench.py --sharded-child ...`, reads its JSON lines under a
    # deadline and kills exactly that child when it is late — a stuck RCCL collective then dies with its process instead of
    # being abandoned by a watchdog thread inside a process that still owns a GPU context and a process group.  The headline
    # measurement above is complete before any of it starts; a failed leg is `sharded: {"error": ...}` in the line (and exit
    # code 3 with SWM_BENCH_SHARDED_STRICT=1).
    sharded_info, sharded_more = None, []
    if use_dist and args.workload == "prove" and not os.environ.get("SWM_BENCH_NO_SHARDED"):
        if args.sharded_log_n:
            sizes = [int(x) for x in args.sharded_log_n.split(",") if x]
        else:
            sizes = [args.log_n] + ([22] if args.circuit == "synthetic" and not args.r1cs and args.log_n == 20 and world > 1 else [])
        mine = _run_sharded_children(args, sizes, rank, world, device_index, dist, torch, coll_dev)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        results = []
        for k in range(len(sizes)):
            per_rank = [g[k] if k < len(g) else {"error": "no result"} for g in gathered]
            errs = [r["error"] for r in per_rank if "error" in r]
            if errs:
                results.append({"error": errs[0], "log_n": sizes[k]})
                break  # (a rank that gave up took no part in the later sizes)
            dts = max(r["seconds"] for r in per_rank)
            r0 = per_rank[0]
            results.append({"constraints": r0["constraints"], "ms_per_proof": dts / r0["reps"] * 1e3,
                            "constraints_per_s": r0["constraints"] * r0["reps"] / dts, "ranks": world,
                            "exchange": "ncclAllGather / grouped ncclSend+ncclRecv inside libswmarlin (swm_rccl_init), one child process per rank",
                            "rccl": r0["rccl"], "rccl_same_on_all_ranks": len({r["rccl"] for r in per_rank}) == 1,
                            "exchanges_per_proof": r0["exchanges_per_proof"], "bytes_per_rank_per_proof": r0["bytes_per_rank_per_proof"],
                            "proof_bytes_identical_on_all_ranks": len({r["sha256"] for r in per_rank}) == 1,
                            "proof_verifies": all(r["verifies"] for r in per_rank)})
        sharded_info, sharded_more = (results[0] if results else None), results[1:]

    if rank == 0:
        dom = prof[dominant]
        launches_per_step = dom["calls"] / args.steps
        work = work_timed
        # dominant kernel: the bucket accumulation behind the KZG commitments, one launch per MSM.  Algorithmic bytes per
        # launch: 128 B (96 B base + 32 B scalar) per point that reaches the kernel; a point with a ZERO scalar
        # (or an identity base) is read as its 32-B scalar only and never reaches it — the synthetic witness of the headline
        # circuit has 3 N such points per proof (z_B is constant) — so those are priced at 32 B.  `frac_all_points_128B` is
        # the figure with every logged point at 128 B (what r01 / r02 reported).
        zero_pts = work.get("msm_zero_points", 0)
        alg_bytes_unpriced = 128.0 * work["msm_points"] / work["msm_calls"]
        alg_bytes = (128.0 * (work["msm_points"] - zero_pts) + 32.0 * zero_pts) / work["msm_calls"]
        achieved = alg_bytes / (dom["avg_ms"] * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: NOT measured in this run (PMC passes need their own rocprofv3 invocation,
        # MI355X_MICROARCH.md); taken from the newest committed PMC summary under profiles/ and scaled per point.
        # identity of a kernel's code = sha256 over every file its translation unit includes (tools/srchash.py; stored with
        # every counter summary by tools/pmc_summaries.py): a summary whose sources have changed since is reported as stale —
        # the figure is from another kernel, or another launch geometry, than the one timed here
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from srchash import unit_sha16

        def stale(summary, kernel):
            return None if summary is None else summary.get("source_sha16") != unit_sha16(kernel)
        traffic, traffic_source, traffic_stale = None, None, None
        try:
            pmcs = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_pmc_msm_accumulate\.json", f)),
                          key=lambda f: int(f[1:f.index("_")]))
            if pmcs:
                pmc = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1])))
                traffic = pmc["hbm_bytes_per_point"] * (alg_bytes / 128.0)
                traffic_stale = stale(pmc, "msm_accumulate")
                traffic_source = "profiles/" + pmcs[-1] + " (separate rocprofv3 --pmc passes; bytes per point scaled to this launch size)"
        except Exception:
            traffic = None
        # the other two kernels of the path against the same roof: NTT 64 B per element per transform
        # (a transform is 2-3 ntt_pass launches), mat-vec 68 B per non-zero + 36 B per row
        def newest_profile(stem):
            try:
                fs = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_" + stem + r"\.json", f)),
                            key=lambda f: int(f[1:f.index("_")]))
                return (json.load(open(os.path.join(ROOT, "profiles", fs[-1]))), "profiles/" + fs[-1]) if fs else (None, None)
            except Exception:
                return None, None
        # the issue ceiling of the dominant kernel, MEASURED: the newest committed SQ pass of the same kernel gives the share
        # of its wave-cycles in which a VALU instruction issues; rate / share = the rate at which every cycle would issue
        sq, sq_src = newest_profile("pmc_sq_msm_accumulate")
        ceiling = sq["mixed_adds_per_s_of_the_pass"] / sq["valu_issue_share_of_wave_cycles"] if sq else None
        ntt_pmc, ntt_src = newest_profile("pmc_ntt_pass")
        spmv_pmc, spmv_src = newest_profile("pmc_spmv")
        secondary = []
        if "ntt_pass" in prof and work.get("ntt_elements"):
            b = 64.0 * work["ntt_elements"]
            a = b / (prof["ntt_pass"]["total_ms"] * 1e-3) / 1e9
            secondary.append({"bound": "hbm", "kernel": "ntt_pass", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": a / HBM_PEAK_GBS,
                              "traffic": ntt_pmc["hbm_bytes_per_element_per_transform"] * work["ntt_elements"] / args.steps if ntt_pmc else None,
                              "traffic_source": ntt_src and ntt_src + " (separate --pmc passes on the stand-alone transform, bytes per element scaled to this step)",
                              "traffic_stale": stale(ntt_pmc, "ntt_pass"),
                              "algorithmic_bytes": "64 B x %d elements over %d transforms (%d launches)"
                                                   % (work["ntt_elements"] / args.steps, work["ntt_calls"] / args.steps,
                                                      prof["ntt_pass"]["calls"] / args.steps),
                              "ms_per_step": prof["ntt_pass"]["total_ms"] / args.steps,
                              "note": "inside the proof, beside accumulations; `standalone`: the same kernel alone on the chip",
                              "standalone": dict(standalone.get("ntt_pass", {}), frac=standalone.get("ntt_pass", {}).get("achieved", 0) / HBM_PEAK_GBS)})
        sp = [k for k in prof if k.startswith("spmv_")]
        if sp and work.get("spmv_nnz"):
            b = 68.0 * work["spmv_nnz"] + 36.0 * work["spmv_rows"]
            ms = sum(prof[k]["total_ms"] for k in sp)
            a = b / (ms * 1e-3) / 1e9
            secondary.append({"bound": "hbm", "kernel": "+".join(sorted(sp)), "achieved": a, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                              "traffic": spmv_pmc["hbm_bytes_per_nnz"] * work["spmv_nnz"] / args.steps if spmv_pmc else None,
                              "traffic_source": spmv_src and spmv_src + " (separate --pmc passes on the stand-alone mat-vec, bytes per non-zero scaled to this step)",
                              "traffic_stale": stale(spmv_pmc, "spmv"),
                              "algorithmic_bytes": "68 B x %d non-zeros + 36 B x %d rows over %d mat-vecs"
                                                   % (work["spmv_nnz"] / args.steps, work["spmv_rows"] / args.steps,
                                                      work["spmv_calls"] / args.steps),
                              "ms_per_step": ms / args.steps,
                              "standalone": dict(standalone.get("spmv", {}), frac=standalone.get("spmv", {}).get("achieved", 0) / HBM_PEAK_GBS)})
        out = {
            "metric": METRIC, "value": units * (1 if args.workload == "prove_sharded" else world) * args.steps / dt, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.workload == "prove_sharded" else "weak",
            "vs_baseline": None, "dtype": "u32 limbs (384-bit Montgomery Fq for G1, 256-bit Fr)", "data": "synthetic",
            "config": {"workload": workload, "per_gpu_units": n,
                       "multi_gpu": {"prove": "replicas (one proof per rank)",
                                     "prove_sharded": "one proof over all ranks: point-range sharded commitment MSMs, "
                                                      "192-B partials all-gathered, transforms replicated",
                                     "msm": "point-range shards, all-gather of 144-B partials"}[args.workload]},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                         "avg_launch_ms": dom["avg_ms"], "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "zero_scalar_points_per_launch": zero_pts / work["msm_calls"],
                         "frac_all_points_128B": alg_bytes_unpriced / (dom["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # explanatory figure: the kernel is bound by integer issue, not by HBM — mixed
                         # additions per second (msm_adds = NON-ZERO digits = bucket entries, counted by the sort on the
                         # device) against the rate at which every wave-cycle would issue a vector instruction, from the
                         # committed SQ counters of the same kernel (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)
                         "mixed_adds_per_s": work["msm_adds"] / (dom["total_ms"] * 1e-3) if dom["total_ms"] else None,
                         "issue_ceiling_mixed_adds_per_s": ceiling, "issue_ceiling_source": sq_src,
                         "issue_ceiling_stale": stale(sq, "msm_accumulate")},
            "roofline_secondary": secondary,
            "sharded": sharded_info, "sharded_more": sharded_more or None,
            "verify_ms_host": verify_ms if args.workload != "msm" else None,
            "rng": args.rng, "drop_in_rng": drop_in, "drop_in": binding, "overlapped": overlapped,
            "work_per_step": {k: v / args.steps for k, v in work.items()},
            "kernels_ms_per_step": {k: round(v["total_ms"] / args.steps, 4) for k, v in sorted(prof.items())},
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
            if args.workload == "msm":
                out["cpu_baseline"] = cpu_baseline_msm(args.cpu_log_n or 18)
            else:
                per_proof = calls_timed[: len(calls_timed) // args.steps]  # the log of ONE proof
                lg = max(n - 1, 1).bit_length()
                shift_all = max(0, lg - args.cpu_log_n) if args.cpu_log_n else 0
                if shift_all:  # --cpu-log-n: bound the all-thread replay too (slow hosts)
                    per_proof = [[k, (v - shift_all if k == "n" else max(v >> shift_all, 1))] for k, v in per_proof]
                z = np.ascontiguousarray(np.concatenate([cs.instance, cs.witness]))
                base = cpu_baseline_prove(per_proof, n >> shift_all, cs.mats, z, max(0, lg - shift_all - 16))
                if shift_all:
                    base["sample"] += " [sizes divided by %d: --cpu-log-n]" % (1 << shift_all)
                out["cpu_baseline"] = base
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if use_dist:
        failed = [r for r in [sharded_info] + sharded_more if r is not None and "error" in r]
        dist.destroy_process_group()  # (this process ran no sharded proof itself: nothing of it is stuck, whatever its children did)
        if failed:
            sys.stderr.write("bench: the sharded leg failed (reported in the JSON line): %s\n" % failed[0]["error"])
            if os.environ.get("SWM_BENCH_SHARDED_STRICT"):
                sys.exit(3)


if __name__ == "__main__":
    main()
