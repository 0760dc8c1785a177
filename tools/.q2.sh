cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
REPS=4 SWM_MSM_QUAD_RB=128 rocprofv3 --kernel-trace -d gpurun_out/tr12 -o run --output-format csv -- python3 tools/small_proofs.py 12 > gpurun_out/tr12.log 2>&1
python3 tools/trace_dump.py gpurun_out/tr12/run_kernel_trace.csv > gpurun_out/tr12_dump.txt 2>&1
rm -rf gpurun_out/tr12
SWM_TRACE=1 SWM_MSM_QUAD_RB=128 REPS=3 python3 tools/small_proofs.py 12 > gpurun_out/tr12_phase.log 2>&1
