#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace: GPU busy/idle time and per-kernel exclusive time over the last
proof of a bench.py --workload prove run.  usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("swm::", "")) for r in rows))
# last proof = from the last "z_pad"/first kernel after the last open_scale of previous... use last occurrence of w_evals
starts = [s for s, e, k in ev if "w_evals" in k or ("ew_kernel" in k)]
# find last spmv_products pair start: proofs begin with spmv products; take the start of the 2nd-to-last group
idx = [i for i, (s, e, k) in enumerate(ev) if "sample_candidates" in k]
t0 = ev[idx[-1]][0] - 15_000_000 if idx else ev[0][0]
sel = [(s, e, k) for s, e, k in ev if s >= t0]
# crude: begin at first kernel after a >1 ms idle gap before t0+...
t_begin, t_end = sel[0][0], max(e for s, e, k in sel)
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
gaps = []
for s, e, k in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t_begin, k))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.2f ms  busy %.2f ms  idle %.2f ms" % ((t_end - t_begin) / 1e6, busy / 1e6, (t_end - t_begin - busy) / 1e6))
print("largest gaps (ms, at ms, next kernel):")
for g in sorted(gaps, reverse=True)[:15]:
    print("  %.3f at %.2f before %s" % (g[0] / 1e6, g[1] / 1e6, g[2][:50]))
tot = collections.Counter()
for s, e, k in sel:
    tot[k[:40]] += e - s
print("kernel time (sum of durations, overlapping counted twice):")
for k, v in tot.most_common(12):
    print("  %-40s %.2f ms" % (k, v / 1e6))
