timeout 900 python3 -m pytest tests/test_gpu_marlin.py -x -q -m gpu -k "golden_proof_bytes or roundtrip or rng_modes" 2>&1 | tail -3
for e in "SWM_MSM_QUAD=0 SWM_BINV_SMALL=0" "SWM_MSM_QUAD=1" "SWM_MSM_QUAD_BLOCKS=32" "SWM_MSM_QUAD_RB=64" "SWM_MSM_QUAD_RB=64 SWM_MSM_QUAD_BLOCKS=128" "SWM_MSM_QUAD_MAXB=8192" "SWM_MSM_QUAD=0 SWM_BINV_SMALL=0"; do
  echo "== $e"; env $e REPS=21 python3 tools/small_proofs.py 10 12 14 2>&1 | tail -3
done
