#!/bin/bash
# alternating bench runs under a list of environment settings on ONE box:  bash tools/envab_quick.sh <rounds> "<ENV A>" "<ENV B>" ...
rounds=$1; shift
for r in $(seq $rounds); do
  for e in "$@"; do
    v=$(env $e python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-drop-in 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    echo "[$e] $v"
  done
done
