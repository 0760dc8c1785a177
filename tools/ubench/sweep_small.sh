run() { echo "== $1 :: $2"; env $1 REPS=9 python tools/small_proofs.py $2 2>&1 | grep -v amdgpu.ids; }
run "" "16 17"
run "SWM_MSM_BIG_NSEG=6" "16 17"
run "SWM_MSM_BIG_NSEG=8" "16 17"
run "SWM_MSM_BIG_NSEG=16" "16 17"
run "" "16 17"
run "SWM_MSM_BIG_NSEG=6" "16 17"
