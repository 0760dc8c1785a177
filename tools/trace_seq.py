#!/usr/bin/env python3
"""Start / end of the MSM stage kernels over the last proof of a rocprofv3 kernel trace (which hardware queue they ran on,
whether stages of consecutive MSMs overlap).  usage: trace_seq.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("swm::", "").replace("void ", "")[:28],
             r["Queue_Id"]) for r in rows)
marks = [s for s, e, k, q in ev if "sample_candidates" in k]
t0, t1 = marks[-2], marks[-1]
for s, e, k, q in ev:
    if t0 <= s < t1 and (k.startswith("msm_accumulate") or k.startswith("msm_bucket_reduce") or k.startswith("msm_flat_partition")):
        print("%8.3f -> %8.3f  (%6.3f) %-26s q=%s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k, q))
print("proof window %.2f ms" % ((t1 - t0) / 1e6))
