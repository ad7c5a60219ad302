#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):  bash tools/collect_profiles.sh r02
# Kernel trace of the default bench command, an isolated (single-stream) trace, and the two separate PMC passes for HBM
# traffic -- for the fp32 contract workload and for --dtype bf16.  Summaries land in gpurun_out/prof_<round>/.
set -u
R=${1:-r05}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-others --no-probe"
for mode in fp32 bf16; do
  extra=""; [ $mode = bf16 ] && extra="--dtype bf16"
  # steps in the traced process: 2 warm-up + 5 timed (+ 5 of bench.py's isolated pass where the wgrad side stream is on = bf16 mode)
  KS=7; [ $mode = bf16 ] && KS=12
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$mode -o k -- $BENCH $extra > $OUT/kt_$mode.log 2>&1
  python tools/kstats.py $OUT/kt_$mode/k_kernel_stats.csv $KS 80 > $OUT/${R}_bench_b64_${mode}_kernel_summary.txt
  cp $OUT/kt_$mode/k_kernel_stats.csv $OUT/${R}_bench_b64_${mode}_kernel_stats.csv
  grep '^{"metric"' $OUT/kt_$mode.log > $OUT/${R}_bench_b64_${mode}_under_rocprof.json
  TAG_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iso_$mode -o k -- $BENCH $extra > $OUT/iso_$mode.log 2>&1
  python tools/kstats.py $OUT/iso_$mode/k_kernel_stats.csv 7 80 > $OUT/${R}_bench_b64_${mode}_isolated_kernel_summary.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    TAG_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${mode}_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-others --no-probe $extra > $OUT/pmc_${mode}_$c.log 2>&1
  done
  python tools/pmc_traffic.py $OUT/pmc_${mode}_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_${mode}_WRITE_SIZE/p_counter_collection.csv \
      $OUT/${R}_pmc_hbm_traffic_${mode}.json 3 "bench.py --steps 2 --warmup 1 $extra, TAG_WGRAD_STREAM=0, batch 64" > $OUT/${R}_pmc_hbm_traffic_${mode}.txt
  # issue-side SQ counters (separate passes, no tracing): MFMA busy / waits / LDS for the conv kernels, VALU issue for the passes
  TAG_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_$mode -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-others --no-probe $extra > $OUT/sq_$mode.log 2>&1
  python tools/pmc_sq.py $(find $OUT/sq_$mode -name "*counter_collection.csv" | head -1) conv3x3 > $OUT/${R}_pmc_sq_counters_${mode}.txt 2>&1
  python tools/pmc_sq.py $(find $OUT/sq_$mode -name "*counter_collection.csv" | head -1) gru >> $OUT/${R}_pmc_sq_counters_${mode}.txt 2>&1
  # (round 5: the batched Winograd-domain products run on gemm_kernel; the transform passes beside them)
  python tools/pmc_sq.py $(find $OUT/sq_$mode -name "*counter_collection.csv" | head -1) gemm_kernel >> $OUT/${R}_pmc_sq_counters_${mode}.txt 2>&1
  python tools/pmc_sq.py $(find $OUT/sq_$mode -name "*counter_collection.csv" | head -1) wino_fused >> $OUT/${R}_pmc_sq_counters_${mode}.txt 2>&1
  TAG_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/valu_$mode -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-others --no-probe $extra > $OUT/valu_$mode.log 2>&1
  python tools/pmc_valu.py $(find $OUT/valu_$mode -name "*counter_collection.csv" | head -1) > $OUT/${R}_pmc_valu_issue_${mode}.txt 2>&1
  rm -rf $OUT/sq_$mode $OUT/valu_$mode
  rm -f $OUT/kt_$mode/k_kernel_trace.csv $OUT/iso_$mode/k_kernel_trace.csv
  rm -rf $OUT/pmc_${mode}_FETCH_SIZE $OUT/pmc_${mode}_WRITE_SIZE
done
ls -la $OUT
