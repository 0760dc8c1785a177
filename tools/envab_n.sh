#!/bin/bash
# like envab.sh with the bench arguments given:  bash tools/envab_n.sh <rounds> "<bench args>" "<ENV A>" "<ENV B>" ...
rounds=$1; args=$2; shift 2
for r in $(seq $rounds); do
  for e in "$@"; do
    env $e python3 bench.py $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%s]' % '$e', round(d['ms_per_step'],3), {k: round(v,2) for k,v in (d.get('kernels_ms_per_step') or {}).items() if 'spmv' not in k})"
  done
done
