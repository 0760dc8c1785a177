#!/usr/bin/env python3
"""Every kernel of the last proof of a rocprofv3 kernel trace in start order: start, end, duration (ms from the start of the
proof), hardware queue, name; '*' marks kernels that start while no msm_accumulate is in flight (the exposed part of a proof).
usage: trace_dump.py <kernel_trace.csv> [min_us]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
def name(k):
    m = re.search(r"ew_kernel<.*?(\w+)\(.*?lambda.*?#(\d+)", k)
    if m: return "ew:%s#%s" % (m.group(1), m.group(2))
    return k.split("(")[0].replace("swm::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:40]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
# one mark per proof: the bulk mask sampling (sample_candidates); TRACE_MARK=<kernel name prefix> picks another kernel (a proof with a
# caller-owned generator has none) — launches of it closer than 10 ms to the previous mark belong to the same proof
import os
_mk = os.environ.get("TRACE_MARK") or ("swm_proof_begin" if any(x[2].startswith("swm_proof_begin") for x in ev) else "sample_candidates")
marks = []
for s, e, k, q in ev:
    if k.startswith(_mk) and (not marks or s - marks[-1] > float(os.environ.get("TRACE_MARK_GAP_MS", "10")) * 1e6):  # (small proofs: 2)
        marks.append(s)
t0, t1 = marks[-2], marks[-1]
acc = [(s, e) for s, e, k, q in ev if k.startswith("msm_accumulate")]
for s, e, k, q in ev:
    if not (t0 <= s < t1) or (e - s) / 1e3 < min_us: continue
    inflight = any(a <= s < b for a, b in acc)
    print("%s %8.3f %8.3f %7.3f q%-2s %s" % (" " if inflight else "*", (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, k))
print("proof window %.3f ms" % ((t1 - t0) / 1e6))
