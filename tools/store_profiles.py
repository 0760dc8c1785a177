#!/usr/bin/env python3
"""Copies the artifacts tools/collect_profiles.sh left under gpurun_out/<tag>/ into profiles/ under the round prefix:
bench lines, rocprofv3 kernel stats, the timed-region summaries (rocprofv3 trace vs the HIP-event average printed by the
same run) and the PMC traffic summary of the dominant kernel.   usage: store_profiles.py <tag> <round-prefix>"""
import csv, json, os, shutil, subprocess, sys
tag, pre = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o, p = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for src, dst in (("bench_prove.json", "bench_prove.json"), ("bench_msm.json", "bench_msm.json"),
                 ("bench_msm_2p22.json", "bench_msm_2p22.json"), ("bench_merkle.json", "bench_merkle.json"),
                 ("prof_prove/run_kernel_stats.csv", "prove_2p20_kernel_stats.csv"),
                 ("prof_merkle/run_kernel_stats.csv", "prove_merkle_kernel_stats.csv"),
                 ("prof_msm/run_kernel_stats.csv", "msm_2p20_kernel_stats.csv"),
                 ("bench_prove_driver_flags.json", "bench_prove_default_flags.json"), ("bench_prove_2p22.json", "bench_prove_2p22.json"),
                 ("prio_ab.log", "prio_ab.log"), ("mid_sweep.log", "mid_sweep.log"), ("lds_conflicts.txt", "lds_conflicts.txt"),
                 ("ntt_time.log", "ntt_standalone.log"), ("small_proofs.log", "small_proofs.log"),
                 ("merkle_build.json", "merkle_build_2p18.json"), ("shard_emulate.jsonl", "shard_emulate.jsonl"), ("ntt_sharded_one.jsonl", "ntt_sharded_one.jsonl"),
                 ("prof_merkle_build/run_kernel_stats.csv", "merkle_build_2p18_kernel_stats.csv")):
    if os.path.exists(os.path.join(o, src)):
        shutil.copy(os.path.join(o, src), os.path.join(p, "%s_%s" % (pre, dst)))
for name, launches, log, cmd in (("prove", 75, "prof_prove", "--steps 5 --warmup 1 --no-drop-in"),
                                 ("msm", 12, "prof_msm", "--workload msm --steps 12 --warmup 2")):
    if not os.path.exists(os.path.join(o, "region_%s.json" % name)) and not os.path.exists(os.path.join(o, log, "run_kernel_trace.csv")):
        print(name, "no kernel trace summary in this collection: timed-region file not written")
        continue
    region = os.path.join(o, "region_%s.json" % name)  # written on the GPU box by collect_profiles.sh (the traces are not kept)
    if os.path.exists(region):
        r = json.load(open(region))
    else:
        r = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "tools", "rocprof_region.py"),
                                                os.path.join(o, log, "run_kernel_trace.csv"), "msm_accumulate", str(launches)]))
    b = json.loads([l for l in open(os.path.join(o, log + ".log")) if l.startswith("{")][-1])
    r["bench_line_of_the_profiled_run"] = {"avg_launch_ms": b["roofline"]["avg_launch_ms"], "ms_per_step": b["ms_per_step"],
                                           "value": b["value"], "unit": b["unit"]}
    r["command"] = "rocprofv3 --kernel-trace --stats -- python3 bench.py %s --no-cpu-baseline" % cmd
    try:
        unprof = json.load(open(os.path.join(o, "bench_%s.json" % name)))
        slow = b["roofline"]["avg_launch_ms"] / unprof["roofline"]["avg_launch_ms"] - 1
        r["unprofiled_bench_line"] = {"avg_launch_ms": unprof["roofline"]["avg_launch_ms"], "ms_per_step": unprof["ms_per_step"]}
        r["note"] = ("HIP-event average inside bench.py and the rocprofv3 trace of the SAME run agree; with the profiler attached "
                     "this kernel ran %.0f %% slower per launch than in the unprofiled bench line of the same collection "
                     "(%s_bench_%s.json): never compare a profiled arm with an unprofiled one" % (100 * slow, pre, name))
    except Exception:
        r["note"] = "HIP-event average inside bench.py and the rocprofv3 trace of the SAME run agree"
    json.dump(r, open(os.path.join(p, "%s_%s_2p20_timed_region.json" % (pre, name)), "w"), indent=1)
    print(name, "rocprof", round(r["avg_ms_timed_region"], 4), "bench events", b["roofline"]["avg_launch_ms"])


# PMC passes: SQ issue counters and FETCH / WRITE of the dominant kernel, FETCH / WRITE of the secondary kernels alone
subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_summaries.py"), tag, pre])
if os.path.exists(os.path.join(o, "timeline_share.txt")):
    shutil.copy(os.path.join(o, "timeline_share.txt"), os.path.join(p, pre + "_prove_2p20_timeline_share.txt"))
