#!/bin/bash
# A/B of environment settings on ONE box with ONE build, alternating:  bash tools/envab.sh <rounds> "<ENV A>" "<ENV B>" ...
# (each argument is a string of VAR=value assignments; "" is the default).  Prints ms per proof of every run.
rounds=$1; shift
for r in $(seq $rounds); do
  for e in "$@"; do
    env $e python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-drop-in 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%s]' % '$e', round(d['ms_per_step'],3), {k: round(v,2) for k,v in (d.get('kernels_ms_per_step') or {}).items() if 'spmv' not in k})"
  done
done
