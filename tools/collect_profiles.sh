#!/bin/bash
# Regenerates the measurement artifacts kept under profiles/ (run on the GPU box from the repo root):
#   bench lines (prove 2^20, MSM 2^20 / 2^22), rocprofv3 --kernel-trace --stats of the same commands,
#   and the separate --pmc FETCH_SIZE / WRITE_SIZE passes for the dominant kernel (MI355X_MICROARCH.md recipe).
# usage: bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 10 --warmup 2 > $out/bench_prove.json 2> $out/bench_prove.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drop-in > $out/bench_prove_driver_flags.json 2>> $out/bench_prove.err   # the driver's flags
python3 bench.py --log-n 22 --steps 4 --warmup 1 --no-cpu-baseline --no-drop-in > $out/bench_prove_2p22.json 2>> $out/bench_prove.err
python3 bench.py --circuit merkle --steps 10 --warmup 2 --overlap 4 > $out/bench_merkle.json 2> $out/bench_merkle.err   # (+ four contexts on ONE shared key)
python3 bench.py --workload msm --steps 12 --warmup 2 > $out/bench_msm.json 2> $out/bench_msm.err
python3 bench.py --workload msm --log-n 22 --steps 8 --warmup 2 --cpu-log-n 18 > $out/bench_msm_2p22.json 2> $out/bench_msm22.err
SWM_PROOF_MARKS=1 rocprofv3 --kernel-trace --stats -d $out/prof_prove -o run --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-drop-in > $out/prof_prove.log 2>&1
python3 tools/trace_share.py $out/prof_prove/run_kernel_trace.csv > $out/timeline_share.txt 2>&1
python3 tools/trace_dump.py $out/prof_prove/run_kernel_trace.csv 15 > $out/trace_dump.txt 2>&1
python3 tools/rocprof_region.py $out/prof_prove/run_kernel_trace.csv msm_accumulate 75 > $out/region_prove.json
rocprofv3 --kernel-trace --stats -d $out/prof_merkle -o run --output-format csv -- python3 bench.py --circuit merkle --steps 5 --warmup 1 --no-cpu-baseline --no-drop-in > $out/prof_merkle.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/prof_msm -o run --output-format csv -- python3 bench.py --workload msm --steps 12 --warmup 2 --no-cpu-baseline > $out/prof_msm.log 2>&1
python3 tools/rocprof_region.py $out/prof_msm/run_kernel_trace.csv msm_accumulate 12 > $out/region_msm.json
# (the kernel traces themselves are tens of MB: summarised above, not kept)
rm -f $out/prof_prove/run_kernel_trace.csv $out/prof_merkle/run_kernel_trace.csv $out/prof_msm/run_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex msm_accumulate -d $out/pmc_fetch -o run --output-format csv -- python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex msm_accumulate -d $out/pmc_write -o run --output-format csv -- python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1
# SQ issue counters of the dominant kernel (8 SQ slots = one pass), then FETCH / WRITE of the two secondary kernels alone
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-include-regex msm_accumulate -d $out/pmc_sq -o run --output-format csv -- python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-include-regex msm_accumulate -d $out/pmc_grbm -o run --output-format csv -- python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_grbm.log 2>&1
# r04: the SQ issue counters of the kernels beside the dominant one — the transform alone on the chip, and the
# bucket stage + sort kernels of a stand-alone MSM — each with a GRBM pass for the clock
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
rocprofv3 --pmc $SQ --kernel-include-regex ntt_pass -d $out/pmc_sq_ntt -o run --output-format csv -- python3 tools/ubench/ntt_one.py 22 5 > $out/pmc_sq_ntt.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-include-regex ntt_pass -d $out/pmc_grbm_ntt -o run --output-format csv -- python3 tools/ubench/ntt_one.py 22 5 > $out/pmc_grbm_ntt.log 2>&1
rocprofv3 --pmc $SQ --kernel-include-regex 'msm_bucket_reduce|msm_flat_partition|msm_flat_bin_sort|msm_digits' -d $out/pmc_sq_msm_other -o run --output-format csv -- python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_sq_msm_other.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-include-regex 'msm_bucket_reduce|msm_flat_partition|msm_flat_bin_sort|msm_digits' -d $out/pmc_grbm_msm_other -o run --output-format csv -- python3 bench.py --workload msm --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_grbm_msm_other.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex ntt_pass -d $out/pmc_ntt_fetch -o run --output-format csv -- python3 tools/ubench/ntt_one.py 22 5 > $out/pmc_ntt_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex ntt_pass -d $out/pmc_ntt_write -o run --output-format csv -- python3 tools/ubench/ntt_one.py 22 5 > $out/pmc_ntt_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex spmv -d $out/pmc_spmv_fetch -o run --output-format csv -- python3 tools/ubench/spmv_one.py 20 5 > $out/pmc_spmv_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex spmv -d $out/pmc_spmv_write -o run --output-format csv -- python3 tools/ubench/spmv_one.py 20 5 > $out/pmc_spmv_write.log 2>&1
# r06: LDS bank conflicts and wait shares of every kernel of a proof (what found the 192-byte slots of the bucket stage and the
# LDS-resident scalar of msm_digits)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d $out/pmc_lds -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-drop-in > $out/pmc_lds.log 2>&1
python3 tools/pmc_lds.py $(ls $out/pmc_lds/*counter_collection.csv | head -1) > $out/lds_conflicts.txt 2>&1
rm -rf $out/pmc_lds
python3 tools/ubench/ntt_one.py 22 10 > $out/ntt_one.json 2> $out/ntt_one.err
python3 tools/ubench/spmv_one.py 20 10 > $out/spmv_one.json 2> $out/spmv_one.err
python3 tools/ubench/ntt_time.py > $out/ntt_time.log 2>&1
python3 tools/small_proofs.py 10 12 14 16 17 18 19 > $out/small_proofs.log 2>&1
# the Pedersen Merkle tree of config #5 (2^18 leaves) beside the CPU oracle on 2^14 leaves, and its kernel stats
python3 tools/ubench/merkle_build.py 18 5 14 > $out/merkle_build.json 2> $out/merkle_build.err
rocprofv3 --kernel-trace --stats -d $out/prof_merkle_build -o run --output-format csv -- python3 tools/ubench/merkle_build.py 18 3 0 > $out/prof_merkle_build.log 2>&1
rm -f $out/prof_merkle_build/run_kernel_trace.csv
# what ONE rank of a proof split over G = 1, 2, 4, 8 ranks computes (exchange emulated on the host): strong-scaling bound, DESIGN.md section 6
for lg in "20 8" "22 3"; do
  SWM_SHARD_R1_OFF=1 python3 tools/ubench/shard_emulate.py $lg 2>> $out/shard_emulate.err | grep "^{" >> $out/shard_emulate.jsonl
  SWM_SHARD_R1_OFF=1 SWM_SHARD_RANGE=1 python3 tools/ubench/shard_emulate.py $lg 2>> $out/shard_emulate.err | grep "^{" >> $out/shard_emulate.jsonl
  SWM_SHARD_R1_OFF=1 SWM_SHARD_BUCKETS=1 python3 tools/ubench/shard_emulate.py $lg 2>> $out/shard_emulate.err | grep "^{" >> $out/shard_emulate.jsonl
  # rounds 1 - 3 sharded as well, the library's device exchanges handing back the rank's own chunks ("rounds": "sharded"): the
  # transforms and pointwise work of a rank as in a real run, but on wrong values — the polynomials that come out have fewer
  # non-zero coefficients and the MSMs 25 - 40 % less work than in a real run: a LOWER bound beside the upper bound above
  SWM_SHARD_EMULATE=1 python3 tools/ubench/shard_emulate.py $lg 2 4 8 2>> $out/shard_emulate.err | grep "^{" >> $out/shard_emulate.jsonl
done
SWM_SHARD_EMULATE=1 python3 tools/ubench/ntt_sharded_one.py 22 24 > $out/ntt_sharded_one.jsonl 2> $out/ntt_sharded_one.err
ls -R $out | head -40
# r05: what the issue priorities (csrc/ff.cuh SWM_LIGHT_PRIO / SWM_TAIL_PRIO) are worth — the same sources built with both at 0
# (build/libswmarlin_p0.so: bash tools/buildvar.sh p0 -DSWM_LIGHT_PRIO=0 -DSWM_TAIL_PRIO=0), alternating on this box; and
# the joint bucket stage / the low-LDS kernel at the mid sizes
if [ -f build/libswmarlin_p0.so ]; then
  bash tools/envab.sh 3 "" "SWM_LIB_PATH=$PWD/build/libswmarlin_p0.so" 2>&1 | sed "s|$PWD/||" > $out/prio_ab.log
  ROUNDS=2 bash tools/sweep_mid.sh $out/mid_sweep.log "16 18 merkle" "" "SWM_PROVE_ONE_STREAM_LOG=0" "SWM_MSM_LOW=0" "SWM_MSM_BATCH_BELOW=200000" "SWM_LIB_PATH=$PWD/build/libswmarlin_p0.so" > /dev/null 2>&1
  sed -i "s|$PWD/||" $out/mid_sweep.log
fi
