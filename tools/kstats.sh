#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) of the bucket stage and the accumulation inside proofs of one size under a
# list of environment settings:   [KSTATS_MORE=batch_inverse,ntt_pass] bash tools/kstats.sh <outdir> <log_n> "<ENV A>" "<ENV B>" ...
o=$1; lg=$2; shift 2
mkdir -p $o
i=0
for e in "$@"; do
  i=$((i+1))
  env $e REPS=7 rocprofv3 --kernel-trace --stats -d $o/k$i -o run --output-format csv -- python3 tools/small_proofs.py $lg > $o/k$i.log 2>&1
  echo "[$e] $(grep 'prove 2' $o/k$i.log | tail -1)" >> $o/kstats.txt
  python3 - $o/k$i/run_kernel_stats.csv >> $o/kstats.txt <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("bucket_reduce", "msm_accumulate", "big_bucket") + tuple(filter(None, __import__("os").environ.get("KSTATS_MORE", "").split(",")))):
        print("    %-46s calls %4s  avg %8.1f us  total %8.2f ms" % (n.split("(")[0].replace("swm::", "").replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
  rm -rf $o/k$i
done
