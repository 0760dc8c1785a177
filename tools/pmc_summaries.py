#!/usr/bin/env python3
"""Summaries of the rocprofv3 --pmc passes tools/collect_profiles.sh leaves under gpurun_out/<tag>/ -> profiles/<prefix>_pmc_*.json
(what bench.py reads for `issue_ceiling_mixed_adds_per_s` and `roofline_secondary[].traffic`).   usage: pmc_summaries.py <tag> <prefix>
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced streaming read (doubled below where the access is one); SQ_WAVE_CYCLES,
SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE are summed over 32 shader engines
resp. 8 XCDs."""
import collections, csv, json, os, sys

tag, pre = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o, p = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import unit_sha16  # noqa: E402  (sha256 over the include closure of the kernel's translation unit)


def counters(d, skip_first=1):
    """{(kernel, counter): mean over the dispatches of the run, the first `skip_first` (warm-up) dropped}, and the mean duration"""
    rows = list(csv.DictReader(open(os.path.join(o, d, "run_counter_collection.csv"))))
    by, dur = collections.defaultdict(list), collections.defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("swm::", "")
        by[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[(k, r["Counter_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    mean = lambda v: sum(v[skip_first:]) / max(1, len(v[skip_first:]))
    return {k: mean(v) for k, v in by.items()}, {k: mean(v) for k, v in dur.items()}, {k: len(v) for k, v in by.items()}


def bench_line(log):
    return json.loads([l for l in open(os.path.join(o, log)) if l.startswith("{")][-1])


SQ_NAMES = ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
            "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")


def last_json(log):
    return json.loads([l for l in open(os.path.join(o, log)) if l.startswith("{")][-1])


def sq_summary(sq_dir, grbm_dir, prefix, out_name, command, unit_name, units_per_launch, note):
    """SQ issue counters of every kernel whose name starts with `prefix` (r04: the kernels beside the dominant one — VERDICT r03
    item 3), summed over the kernels of one unit of work (e.g. the two or three ntt_pass launches of a transform)."""
    if not os.path.exists(os.path.join(o, sq_dir, "run_counter_collection.csv")):
        print("no", sq_dir)
        return
    c, dur, n = counters(sq_dir, 0)
    cg, durg, _ = counters(grbm_dir, 0) if os.path.exists(os.path.join(o, grbm_dir, "run_counter_collection.csv")) else ({}, {}, {})
    kerns = sorted({k for k, _ in c if k.startswith(prefix)})
    per = {}
    for k in kerns:
        g = lambda name: c[(k, name)]
        waves_per_simd = g("SQ_WAVE_CYCLES") / (g("SQ_BUSY_CYCLES") * 8)
        per[k] = {"launches_in_the_pass": n[(k, "SQ_WAVES")], "avg_ms": dur[(k, "SQ_WAVE_CYCLES")],
                  "counters_per_launch": {name: g(name) for name in SQ_NAMES},
                  "avg_resident_waves_per_simd": waves_per_simd,
                  "simd_issue_utilisation": g("SQ_ACTIVE_INST_ANY") * waves_per_simd / g("SQ_WAVE_CYCLES"),
                  "valu_share_of_issued_instructions": g("SQ_ACTIVE_INST_VALU") / max(1.0, g("SQ_ACTIVE_INST_ANY")),
                  "wave_time_split": {"executing_an_instruction": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
                                      "waiting_to_issue_(another_wave_owns_the_SIMD)": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
                                      "parked_(s_waitcnt_or_barrier)": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")},
                  "valu_instructions_per_%s" % unit_name: g("SQ_INSTS_VALU") * 64 / units_per_launch(k) if units_per_launch(k) else None}
        if (k, "GRBM_GUI_ACTIVE") in cg:
            per[k]["effective_clock_GHz"] = cg[(k, "GRBM_GUI_ACTIVE")] / 8 / (durg[(k, "GRBM_GUI_ACTIVE")] * 1e-3) / 1e9
    json.dump({"kernels": per, "command": command, "note": note}, open(os.path.join(p, "%s_pmc_sq_%s.json" % (pre, out_name)), "w"), indent=1)
    for k, v in per.items():
        print("sq %-40s util %.3f  waves/SIMD %.2f  parked %.2f  %.3f ms" % (k[:40], v["simd_issue_utilisation"], v["avg_resident_waves_per_simd"],
                                                                           v["wave_time_split"]["parked_(s_waitcnt_or_barrier)"], v["avg_ms"]))


SQ_CMD = "rocprofv3 --pmc " + " ".join(SQ_NAMES) + " --kernel-include-regex %s -- %s   (+ a pass with GRBM_GUI_ACTIVE GRBM_COUNT; tools/collect_profiles.sh)"
try:
    one_sq = last_json("pmc_sq_ntt.log")
    sq_summary("pmc_sq_ntt", "pmc_grbm_ntt", "ntt_pass", "ntt_pass", SQ_CMD % ("ntt_pass", "python3 tools/ubench/ntt_one.py 22 5"), "element_and_pass",
               lambda k: 1 << one_sq["log_n"],
               "2^22 transform = three passes (8 + 7 + 7 levels).  simd_issue_utilisation ~ 1 means the SIMDs issue an instruction in "
               "(nearly) every cycle: the pass is bound by instruction issue — the Fr multiplier — not by HBM (5-6 % of the roofline) or "
               "LDS; `parked` is the share of wave-cycles spent in s_waitcnt / at the workgroup barriers between butterfly levels.")
except Exception as e:  # noqa: BLE001
    print("ntt sq:", e)
try:
    bmsm = last_json("pmc_sq_msm_other.log")
    nb = 1 << 19
    sq_summary("pmc_sq_msm_other", "pmc_grbm_msm_other", "msm_", "msm_bucket_reduce_and_sort",
               SQ_CMD % ("'msm_bucket_reduce|msm_flat_partition|msm_flat_bin_sort|msm_digits'", "python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline"),
               "bucket_or_entry", lambda k: nb if k.startswith("msm_bucket_reduce") else bmsm["work_per_step"]["msm_adds"],
               "2^20-point MSM (2^19 buckets, 13.6 M sort entries).  msm_bucket_reduce: one 256-lane workgroup (144 KB of LDS) per CU, "
               "ONE wave per SIMD — utilisation is what a lone wave reaches (an instruction every ~5.5 cycles, tools/ubench/valu_rates), "
               "the stage is a dependent chain of ~42 group operations; msm_flat_partition / msm_flat_bin_sort / msm_digits: low issue "
               "utilisation and a large parked share — LDS atomics, returning global atomics and load latency, not arithmetic.")
except Exception as e:  # noqa: BLE001
    print("msm other sq:", e)

def dominant_sq():
    global kern
    # ---- SQ pass of the dominant kernel
    c, dur, n = counters("pmc_sq")
    kern = sorted({k for k, _ in c if k.startswith("msm_accumulate")})[0]
    g = lambda name: c[(kern, name)]
    cg, durg, _ = counters("pmc_grbm")
    b = bench_line("pmc_sq.log")
    waves_per_simd = g("SQ_WAVE_CYCLES") / (g("SQ_BUSY_CYCLES") * 8)   # busy cycles: sum over 32 SEs; 1024 SIMDs; quad-cycles
    util = g("SQ_ACTIVE_INST_ANY") * waves_per_simd / g("SQ_WAVE_CYCLES")
    ms = dur[(kern, "SQ_WAVE_CYCLES")]
    out = {
        "kernel": kern, "source_sha16": unit_sha16("msm_accumulate"), "launch": "2^20-point MSM, %d non-zero digits (mixed additions) per launch" % int(b["work_per_step"]["msm_adds"]),
        "command": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY "
                   "SQ_WAIT_ANY --kernel-include-regex msm_accumulate -- python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline"
                   "   (+ a second pass with GRBM_GUI_ACTIVE GRBM_COUNT; tools/collect_profiles.sh)",
        "counters_per_launch": {name: g(name) for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU",
                                                           "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
        "GRBM_GUI_ACTIVE_per_launch": cg[(kern, "GRBM_GUI_ACTIVE")],
        "kernel_ms_in_the_pass": ms,
        "wave_time_split": {"executing_an_instruction": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
                            "waiting_to_issue_(another_wave_owns_the_SIMD)": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
                            "parked_(s_waitcnt)": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")},
        "avg_resident_waves_per_simd": waves_per_simd,
        "simd_issue_utilisation": util,
        "valu_issue_share_of_wave_cycles": min(1.0, util),
        "valu_instructions_per_mixed_addition": g("SQ_INSTS_VALU") * 64 / b["work_per_step"]["msm_adds"],
        "effective_clock_GHz": cg[(kern, "GRBM_GUI_ACTIVE")] / 8 / (durg[(kern, "GRBM_GUI_ACTIVE")] * 1e-3) / 1e9,
        "mixed_adds_per_s_of_the_pass": b["work_per_step"]["msm_adds"] / (ms * 1e-3),
        "note": "Each wave spends ~37 % of its cycles executing and ~53 % waiting for the SIMD's issue port while one of the other "
                "resident waves uses it: executing share x resident waves per SIMD = the fraction of time the SIMD issues an instruction "
                "(`simd_issue_utilisation`; values a few per cent above 1 are counter granularity).  The kernel is AT its issue "
                "ceiling: it gets faster only with fewer instructions per addition or a higher clock (the chip holds ~1.9 GHz under "
                "this load, `effective_clock_GHz`).  bench.py: issue_ceiling = mixed_adds_per_s_of_the_pass / valu_issue_share_of_wave_cycles."}
    json.dump(out, open(os.path.join(p, pre + "_pmc_sq_msm_accumulate.json"), "w"), indent=1)
    print("sq: util %.3f, waves/SIMD %.2f, clock %.2f GHz, %.2f G adds/s" % (util, waves_per_simd, out["effective_clock_GHz"], out["mixed_adds_per_s_of_the_pass"] / 1e9))



def dominant_traffic():
    # ---- FETCH / WRITE of the dominant kernel (random 192-B row gathers, three aligned 64-B sectors: face value, see the r02 note)
    cf, _, _ = counters("pmc_fetch")
    cw, _, _ = counters("pmc_write")
    f, w, npts = cf[(kern, "FETCH_SIZE")], cw[(kern, "WRITE_SIZE")], 1 << 20
    json.dump({"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "kernel": kern, "source_sha16": unit_sha16("msm_accumulate"),
               "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-include-regex msm_accumulate -- "
                          "python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline   (tools/collect_profiles.sh)",
               "points_per_launch": npts, "hbm_bytes_per_launch": (f + w) * 1024, "hbm_bytes_per_point": (f + w) * 1024 / npts,
               "note": "FETCH_SIZE taken at face value (KB): r04 table rows are 192 B — three 64-byte sectors, one per coordinate, each "
                       "fourteen 28-bit limbs + 8 B of padding — gathered at random: 13 windows x 2^20 rows x (192 B + the 4-B sort entry) "
                       "~ 2.7 GB per launch, so the x2 correction of wide coalesced streams does not apply.  One table row per (point, window) "
                       "is inherent to the precomputed-window schedule; the table (13 x 201 MB) exceeds the Infinity Cache, so this is DRAM "
                       "traffic — the kernel is bound by integer issue (" + pre + "_pmc_sq_msm_accumulate.json).  Writes: one 192-B partial sum per segment."},
              open(os.path.join(p, pre + "_pmc_msm_accumulate.json"), "w"), indent=1)
    print("accumulate traffic B/point", (f + w) * 1024 / npts)



def ntt_traffic():
    # ---- secondary kernels alone on the chip
    cf, _, nf = counters("pmc_ntt_fetch", 0)
    cw, _, _ = counters("pmc_ntt_write", 0)
    one = json.load(open(os.path.join(o, "ntt_one.json")))
    lg, ntr = one["log_n"], None
    ks = sorted({k for k, _ in cf})
    fetch = sum(cf[(k, "FETCH_SIZE")] * nf[(k, "FETCH_SIZE")] for k in ks) * 1024 * 2   # streaming reads: FETCH_SIZE counts half
    write = sum(cw[(k, "WRITE_SIZE")] * nf[(k, "FETCH_SIZE")] for k in ks) * 1024
    ntr = last_json("pmc_ntt_fetch.log")["transforms"]  # what ntt_one.py ran under the profiler (warm-up included)
    assert ntr == last_json("pmc_ntt_write.log")["transforms"] and all(nf[(k, "FETCH_SIZE")] % ntr == 0 for k in ks), "launch counts do not match the transforms of the pass"
    json.dump({"kernel": "ntt_pass", "source_sha16": unit_sha16("ntt_pass"), "log_n": lg, "transforms_in_the_pass": ntr, "launches": {k: nf[(k, "FETCH_SIZE")] for k in ks},
               "hbm_bytes_per_transform": (fetch + write) / ntr, "hbm_bytes_per_element_per_transform": (fetch + write) / ntr / (1 << lg),
               "algorithmic_bytes_per_element": 64, "ms_per_transform_unprofiled": one["ms_per_transform"],
               "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-include-regex ntt_pass -- python3 tools/ubench/ntt_one.py 22 5",
               "note": "FETCH_SIZE doubled (wide coalesced streaming reads: the gfx950 counter reports half the bytes — each pass then reads "
                       "exactly 2^22 x 32 B), WRITE_SIZE as reported (exactly 2^22 x 32 B per pass).  Three passes at 2^22 (8 + 7 + 7 levels): "
                       "3 x 64 = 192 B per element against 64 B algorithmic — every pass streams the whole vector once in and once out, "
                       "no re-reads within a pass."},
              open(os.path.join(p, pre + "_pmc_ntt_pass.json"), "w"), indent=1)
    print("ntt B/element/transform", (fetch + write) / ntr / (1 << lg))



def spmv_traffic():
    cf, _, nf = counters("pmc_spmv_fetch", 0)
    cw, _, nw = counters("pmc_spmv_write", 0)
    one = json.load(open(os.path.join(o, "spmv_one.json")))
    rows_, nnz, nmv = one["rows"], one["nnz"], last_json("pmc_spmv_fetch.log")["matvecs"]  # mat-vecs of the profiled run itself
    assert nmv == last_json("pmc_spmv_write.log")["matvecs"]
    fetch_face = sum(v * nf[k] for k, v in cf.items()) * 1024
    write = sum(v * nw[k] for k, v in cw.items()) * 1024
    streamed = nmv * (32.0 * nnz + 4.0 * nnz + 4.0 * (rows_ + 1))   # val, col, rowptr: coalesced streams, counted at half
    json.dump({"kernel": "spmv_rows_direct (+ spmv_row_stats)", "source_sha16": unit_sha16("spmv"), "rows": rows_, "nnz": nnz, "matvecs_in_the_pass": nmv,
               "FETCH_bytes_face_value_per_matvec": fetch_face / nmv, "WRITE_bytes_per_matvec": write / nmv,
               "hbm_bytes_per_nnz": (fetch_face + streamed / 2 + write) / nmv / nnz,
               "hbm_bytes_per_nnz_face_value": (fetch_face + write) / nmv / nnz, "algorithmic_bytes_per_nnz": 68 + 36,
               "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-include-regex spmv -- python3 tools/ubench/spmv_one.py 20 5",
               "note": "one non-zero per row (the synthetic R1CS).  The coalesced streams (val 32 B, col 4 B, rowptr 4 B per row) are counted "
                       "at half by FETCH_SIZE on gfx950 and are added back (`hbm_bytes_per_nnz`); the gathers of z (random 32-B reads, one "
                       "64-B request each) are taken at face value.  z (32 MB) stays in the Infinity Cache between mat-vecs, which the "
                       "memory-side counter still counts."},
              open(os.path.join(p, pre + "_pmc_spmv.json"), "w"), indent=1)
    print("spmv B/nnz", (fetch_face + streamed / 2 + write) / nmv / nnz)


kern = "msm_accumulate_te"
for fn in (dominant_sq, dominant_traffic, ntt_traffic, spmv_traffic):
    try:
        fn()
    except Exception as e:  # noqa: BLE001 — a partial collection still yields the summaries it has the passes for
        print("%s: skipped (%r)" % (fn.__name__, e))
