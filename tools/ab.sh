# A/B of a compile-time variant of msm.hip on the GPU box: usage ab.sh "<flags A>" "<flags B>"
for flag in "$1" "$2" "$1" "$2"; do
  touch simpleworks_amd/csrc/msm.hip
  make -C simpleworks_amd/csrc EXTRA="$flag" > /dev/null 2>&1
  python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "msm" 2>&1 | tail -1
  for ln in 20 22; do
  python bench.py --workload msm --log-n $ln --steps 10 --warmup 2 --no-cpu-baseline --profile-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('flag [$flag] msm 2^$ln', round(d['ms_per_step'],3), 'acc', k['msm_accumulate'], 'red', k['msm_bucket_reduce'])"
  done
done
