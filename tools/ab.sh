#!/bin/bash
# A/B of two builds of the library on ONE box, alternating (box-to-box spread is +-3 %, more than most changes):
#   bash tools/ab.sh <libA.so> <libB.so> [rounds] [bench.py arguments ...]      prints ms per step of every run
a=$1; b=$2; shift 2; rounds=${1:-2}; [ $# -gt 0 ] && shift
args=${@:---steps 10 --warmup 3 --no-cpu-baseline --no-drop-in}
for r in $(seq $rounds); do
  for lib in $a $b; do
    SWM_LIB_PATH=$PWD/$lib python3 bench.py $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', round(d['ms_per_step'],3), {k: round(v,3) for k,v in (d.get('kernels_ms_per_step') or {}).items() if 'spmv' not in k}, d['roofline'].get('avg_launch_ms'))"
  done
done
