#!/usr/bin/env python3
"""Wall-time attribution from a rocprofv3 kernel trace: at every instant the wall time is split equally between the
kernels in flight, so the per-kernel totals add up to the GPU-busy time of the window (unlike summed durations, which
count overlapped time twice).  Window = the last proof of a `bench.py --workload prove` run.
usage: trace_share.py <kernel_trace.csv> [n_last_proofs]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (lambda k: (re.search(r"ew_kernel<.*?(\w+)[:(]", k) or [None, k.split("(")[0].replace("swm::", "").replace("void ", "").split("<")[0]])[1])(r["Kernel_Name"]))
            for r in rows)
# one bulk mask sampling (sample_candidates) per proof: consecutive launches delimit one full proof period
def name(k):
    m = re.search(r"ew_kernel<swm::(\w+)", k) or re.search(r"ew_kernel<.*?::(\w+)\(", k)
    return m.group(1) if m else k
import os
_mk = os.environ.get("TRACE_MARK") or ("swm_proof_begin" if any(x[2].startswith("swm_proof_begin") for x in ev) else "sample_candidates")   # (TRACE_MARK: see trace_dump.py)
marks = []
for s, e, k in ev:
    if k.startswith(_mk) and (not marks or s - marks[-1] > 10e6):
        marks.append(s)
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0, t1 = marks[-nlast - 1], marks[-nlast]
sel = [(max(s, t0), min(e, t1), k) for s, e, k in ev if e > t0 and s < t1]
pts = []
for s, e, k in sel:
    pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort(key=lambda x: (x[0], x[1]))
share = collections.Counter(); alone = collections.Counter(); live = collections.Counter(); last = pts[0][0]; busy = 0
for t, d, k in pts:
    n = sum(live.values())
    if n and t > last:
        busy += t - last
        for kk, c in live.items():
            if c:
                share[kk] += (t - last) * c / n
                if n == c: alone[kk] += t - last
    live[k] += d; last = t
print("window %.2f ms, GPU busy %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6))
# the dominant kernel saturates the chip: what runs while NO msm_accumulate is in flight is the exposed part of a proof
live2 = collections.Counter(); last2 = pts[0][0]; noacc = collections.Counter(); noacc_t = 0; acc_t = 0
for t, d, k in pts:
    if t > last2:
        n = sum(live2.values())
        if sum(c for kk, c in live2.items() if kk.startswith("msm_accumulate")) > 0:
            acc_t += t - last2
        else:
            noacc_t += t - last2
            for kk, c in live2.items():
                if c: noacc[kk] += (t - last2) * c / n
            if n == 0: noacc["(idle)"] += t - last2
    live2[k] += d; last2 = t
print("msm_accumulate in flight %.2f ms, not in flight %.2f ms; what runs then (share ms):" % (acc_t / 1e6, noacc_t / 1e6))
for k, v in noacc.most_common(14):
    print("   %-28s %7.2f" % (k[:28], v / 1e6))
print("%-28s %9s %9s" % ("kernel", "share ms", "alone ms"))
for k, v in share.most_common(25):
    print("%-28s %9.2f %9.2f" % (k[:28], v / 1e6, alone[k] / 1e6))
# idle gaps: (length, offset in window, kernel that ended before, kernel that starts after)
iv = sorted((s, e, k) for s, e, k in sel)
gaps = []; cur_e, cur_k = iv[0][1], iv[0][2]
for s, e, k in iv[1:]:
    if s > cur_e: gaps.append((s - cur_e, cur_e - t0, cur_k, k))
    if e > cur_e: cur_e, cur_k = e, k
print("idle gaps > 0.15 ms (ms, at ms, after kernel, before kernel):")
for g in sorted(gaps, key=lambda g: g[1]):
    if g[0] > 150_000: print("  %.3f at %6.2f  %s -> %s" % (g[0] / 1e6, g[1] / 1e6, g[2][:24], g[3][:24]))
