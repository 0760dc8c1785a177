#!/bin/bash
# Mid-size proofs (2^16 / 2^18 constraints, the Merkle circuit) under a list of environment settings, ONE box, one build:
#   [ROUNDS=2] bash tools/sweep_mid.sh <out.log> "<sizes>" "<ENV A>" "<ENV B>" ...     sizes: e.g. "16 18 merkle"
# Every setting builds its own keys (the table width is a property of the key).  One line per (round, setting, size); the rounds
# alternate over the settings, so that drift of the box hits all of them alike.
out=$1; sizes=$2; shift 2
for r in $(seq ${ROUNDS:-1}); do
for e in "$@"; do
  for s in $sizes; do
    if [ "$s" = merkle ]; then
      v=$(env $e python3 bench.py --circuit merkle --steps 7 --warmup 2 --no-cpu-baseline --no-drop-in 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
      echo "[$e] merkle: $v ms" | tee -a $out
    else
      v=$(env $e REPS=${REPS:-9} python3 tools/small_proofs.py $s 2>/dev/null | tail -1)
      echo "[$e] $v" | tee -a $out
    fi
  done
done
done
