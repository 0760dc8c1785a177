run() { echo "== $1 :: $2"; env $1 REPS=9 python tools/small_proofs.py $2 2>&1 | grep -v amdgpu.ids; }
run "" "10 12 14 16 18 20"
run "" "10 12 14 16 18 20"
