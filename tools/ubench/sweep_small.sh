run() { echo "== $1 :: $2"; env $1 REPS=7 python tools/small_proofs.py $2 2>&1 | grep -v amdgpu.ids; }
run "" "10 12 14 16 18"
run "SWM_MSM_BATCH_BELOW=1000000" "18"
run "SWM_MSM_BATCH_BELOW=1000000 SWM_MSM_LAT_BELOW=1000000" "18"
run "SWM_MSM_BATCH_BELOW=400000" "18"
run "" "10 12 14 16 18"
