#!/usr/bin/env python3
"""Kernels around the last twin pair of a rocprofv3 kernel trace (csv): usage trace_twin.py <kernel_trace.csv> [before] [after]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 30
na = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("swm::", "").replace("void ", "")[:34],
             r["Queue_Id"]) for r in rows)
key = sys.argv[4] if len(sys.argv) > 4 else "msm_twin_copy"
idx = [i for i, e in enumerate(ev) if key in e[2]]
if not idx:
    print("no", key); sys.exit(0)
i0 = idx[-1]
t0 = ev[max(0, i0 - nb)][0]
for s, e, k, q in ev[max(0, i0 - nb): i0 + na]:
    print("%9.1f -> %9.1f (%7.1f us) %-34s q=%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, q))
