set -x
timeout 900 python -m pytest tests/test_gpu_switches.py -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/twin_tests.txt
timeout 600 python -m pytest tests/test_gpu_marlin.py -x -q -k "2p20 or golden or 2p17" 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 >> gpurun_out/twin_tests.txt
for r in 1 2 3; do
  for t in 1 0; do
    SWM_MSM_TWIN=$t python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-drop-in 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('twin=$t', round(d['ms_per_step'],3))" >> gpurun_out/twin_ab.txt
  done
done
cat gpurun_out/twin_tests.txt gpurun_out/twin_ab.txt
