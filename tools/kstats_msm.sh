#!/bin/bash
# Per-kernel durations of the stand-alone 2^L-point MSM (bench.py --workload msm) for a list of library builds:
#   bash tools/kstats_msm.sh <outdir> <log_n> <lib.so> [<lib.so> ...]       ("" = the in-tree library)
o=$1; lg=$2; shift 2
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for lib in "$@"; do
  i=$((i+1))
  if [ -n "$lib" ]; then export SWM_LIB_PATH=$PWD/$lib; else unset SWM_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats -d $o/k$i -o run --output-format csv -- python3 bench.py --workload msm --log-n $lg --steps 12 --warmup 2 --no-cpu-baseline > $o/k$i.log 2>&1
  echo "== ${lib:-in-tree} $(grep -o '"ms_per_step": [0-9.]*' $o/k$i.log | tail -1)" >> $o/kstats.txt
  python3 - $o/k$i/run_kernel_stats.csv >> $o/kstats.txt <<'P'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if int(r["Calls"]) < 12 or "table" in n or "te_convert" in n:
        continue
    tot += float(r["AverageNs"]) / 1e3
    print("    %-44s calls %4s  avg %8.1f us" % (n.split("(")[0].replace("swm::", "").replace("void ", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
print("    sum of averages %.1f us" % tot)
P
  rm -rf $o/k$i
done
cat $o/kstats.txt
