#!/bin/bash
# Everything the round's profile record needs, in one GPU call: tools/r03_profiles.sh <tag>
TAG=$1
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-330 $O/bench_default.json
for i in 1 2; do python bench.py --cpu-budget 0 --small-sweep 0 --steps 20 --warmup 5 2>/dev/null >> $O/bench_steps20.json; done; cut -c60-200 $O/bench_steps20.json
bash tools/run_rocprof_stats.sh $TAG > $O/rocprof_stats.log 2>&1; cp gpurun_out/prof_$TAG/kernel_stats.csv $O/kernel_stats_bench_default.csv; cp gpurun_out/prof_$TAG/bench.json $O/bench_under_rocprof.json; head -3 $O/kernel_stats_bench_default.csv | cut -c1-200
bash tools/run_pmc.sh $TAG --steps 32 --warmup 4 --steps-per-launch 32 > $O/pmc.log 2>&1; cp gpurun_out/pmc_$TAG/pmc_hbm_traffic.json $O/; tail -1 $O/pmc.log | cut -c1-400
bash tools/run_pmc_sq.sh $TAG > $O/pmc_sq.txt 2>&1; tail -22 $O/pmc_sq.txt
python tools/bench_configs.py 2>&1 | grep -v amdgpu > $O/all_configs_256_jobs.txt; cat $O/all_configs_256_jobs.txt
python tools/trace_profile.py --jobs 256 --procedure SE-gPoE 2>/dev/null > $O/wave_trace_256_jobs_SE.txt
python tools/trace_profile.py --jobs 1 --procedure SE-gPoE 2>/dev/null > $O/wave_trace_single_job_SE.txt
python tools/trace_profile.py --jobs 256 --head regression --steps 8 2>/dev/null > $O/wave_trace_256_jobs_regression_head.txt
python tools/trace_profile.py --jobs 256 --head endtoend --steps 8 2>/dev/null > $O/wave_trace_256_jobs_endtoend_head.txt
python tools/trace_profile.py --jobs 256 --procedure SM-T1w_sMRI --forward 2>/dev/null > $O/wave_trace_256_jobs_forward_only.txt
python tools/eager_facade_rate.py 2>/dev/null > $O/eager_facade_rate.txt; cat $O/eager_facade_rate.txt
