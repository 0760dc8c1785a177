#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV over the TIMED REGION of bench.py: the last `launches` dispatches of a
kernel (bench.py runs setup/index/warm-up first, which the whole-process --stats summary averages in).
usage: rocprof_region.py <kernel_trace.csv> <kernel-substring> <launches>"""
import csv
import json
import sys

path, needle, launches = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if needle in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-launches:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in sel]
print(json.dumps({"kernel": needle, "launches_in_process": len(rows), "launches_in_timed_region": len(sel),
                  "avg_ms_timed_region": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
                  "avg_ms_whole_process": sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows) / len(rows)}))
