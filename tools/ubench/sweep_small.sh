run() { echo "== $1 :: $2"; env $1 REPS=9 python tools/small_proofs.py $2 2>&1 | grep -v amdgpu.ids; }
run "SWM_MSM_ZERO_COPY=0" "16 18 20"
run "SWM_MSM_ZERO_COPY=1" "16 18 20"
run "SWM_MSM_ZERO_COPY=0" "16 18 20"
run "SWM_MSM_ZERO_COPY=1" "16 18 20"
