mkdir -p gpurun_out/r05g
python -m pytest tests -x -q -m gpu > gpurun_out/r05g/t_all.txt 2>&1; echo "all rc=$?"
grep -E "passed|failed" gpurun_out/r05g/t_all.txt | tail -2
rm -rf gpurun_out/prof_r05
bash tools/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1
head -2 gpurun_out/prof_r05/r05_bench_b64_fp32_kernel_summary.txt | cut -c1-120
python bench.py --steps 20 --warmup 3 > gpurun_out/prof_r05/bench_default.json 2> gpurun_out/prof_r05/bench_default.err; tail -2 gpurun_out/prof_r05/bench_default.err
