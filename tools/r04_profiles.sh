#!/bin/bash
# Everything the round's profile record needs, in one GPU call: tools/r04_profiles.sh <tag>  ->  gpurun_out/<tag>/
# (every step keeps its stderr in <step>.err; a failing step is reported and the script goes on to the next one)
TAG=$1
O=gpurun_out/$TAG
mkdir -p $O
step() { name=$1; shift; "$@" 2> $O/$name.err || echo "!! step $name failed (see $O/$name.err)"; }
step bench_default bash -c "python bench.py > $O/bench_default.json"; cut -c1-330 $O/bench_default.json
for i in 1 2; do step bench_steps20_$i bash -c "python bench.py --cpu-budget 0 --small-sweep 0 --steps 20 --warmup 5 >> $O/bench_steps20.json"; done; cut -c60-200 $O/bench_steps20.json
step rocprof_stats bash -c "bash tools/run_rocprof_stats.sh $TAG > $O/rocprof_stats.log"; cp gpurun_out/prof_$TAG/kernel_stats.csv $O/kernel_stats_bench_default.csv; cp gpurun_out/prof_$TAG/bench.json $O/bench_under_rocprof.json; head -3 $O/kernel_stats_bench_default.csv | cut -c1-200
step pmc bash -c "bash tools/run_pmc.sh $TAG --steps 32 --warmup 4 --steps-per-launch 32 > $O/pmc.log"; cp gpurun_out/pmc_$TAG/pmc_hbm_traffic.json $O/; tail -1 $O/pmc.log | cut -c1-400
step pmc_sq bash -c "bash tools/run_pmc_sq.sh $TAG > $O/pmc_sq.txt"; tail -34 $O/pmc_sq.txt
step all_configs bash -c "python tools/bench_configs.py | grep -v amdgpu > $O/all_configs_256_jobs.txt"; cat $O/all_configs_256_jobs.txt
step trace256 bash -c "python tools/trace_profile.py --jobs 256 --procedure SE-gPoE > $O/wave_trace_256_jobs_SE.txt"
step trace1 bash -c "python tools/trace_profile.py --jobs 1 --procedure SE-gPoE > $O/wave_trace_single_job_SE.txt"
step small bash -c "python tools/bench_small.py --trace --modes wg,split,rs2h0,rs2,rs4h0,rs4 | grep -v amdgpu > $O/small_sweeps_rowsplit.txt"; grep models $O/small_sweeps_rowsplit.txt
step smallprof bash -c "cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/prof_small_$TAG -- python3 \$GRAFT_REPO_ROOT/tools/bench_small.py --modes rs4 > /dev/null"; find gpurun_out/prof_small_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_small_sweeps_rowsplit.csv \; ; rm -rf gpurun_out/prof_small_$TAG; head -3 $O/kernel_stats_small_sweeps_rowsplit.csv | cut -c1-200
step dev bash -c "python tools/bench_deviation.py | grep -v amdgpu > $O/deviation_pass_nm_devpass.txt"; tail -1 $O/deviation_pass_nm_devpass.txt
step devg bash -c "python tools/bench_deviation.py --general | grep -v amdgpu > $O/deviation_pass_general_kernel.txt"; tail -1 $O/deviation_pass_general_kernel.txt
step devprof bash -c "cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/prof_dev_$TAG -- python3 \$GRAFT_REPO_ROOT/tools/bench_deviation.py > /dev/null"; find gpurun_out/prof_dev_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_deviation_pass.csv \; ; rm -rf gpurun_out/prof_dev_$TAG; head -3 $O/kernel_stats_deviation_pass.csv | cut -c1-200
step facade bash -c "python tools/eager_facade_rate.py > $O/eager_facade_rate.txt"; cat $O/eager_facade_rate.txt
