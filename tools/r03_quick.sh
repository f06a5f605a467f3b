#!/bin/bash
# Round-3 quick look: bench (default + driver form + small sweeps) and the wave traces.  tools/r03_quick.sh <tag>
TAG=${1:-q}
O=gpurun_out/$TAG
mkdir -p $O
python bench.py --cpu-budget 0 > $O/bench_default.json 2> $O/bench_default.err; cut -c60-200 $O/bench_default.json
python bench.py --cpu-budget 0 --steps 20 --warmup 5 2>/dev/null > $O/bench_steps20.json; cut -c60-200 $O/bench_steps20.json
for j in 5 20; do python bench.py --cpu-budget 0 --jobs $j 2>/dev/null > $O/bench_jobs$j.json; cut -c60-200 $O/bench_jobs$j.json; done
python tools/trace_profile.py --jobs 1 --procedure SE-gPoE 2>/dev/null > $O/wave_trace_single_job_SE.txt; cat $O/wave_trace_single_job_SE.txt
python tools/trace_profile.py --jobs 256 --procedure SE-gPoE 2>/dev/null > $O/wave_trace_256_jobs_SE.txt; cat $O/wave_trace_256_jobs_SE.txt
