#!/bin/bash
# A/B of environment switches on prove(): usage ab_prove.sh "<log_n list>" "ENV1=.. ENV2=.." "ENV.." ...   ("-" = no switch)
sizes=$1; shift
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  for ln in $sizes; do
    env $v python bench.py --log-n $ln --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prove 2^$ln [$v]', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,2), 'M/s  acc', round(d['roofline']['avg_launch_ms'],3))"
  done
done
