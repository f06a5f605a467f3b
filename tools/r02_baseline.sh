set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r02a_tests.log 2>&1
tail -3 gpurun_out/r02a_tests.log
python bench.py --cpu-budget 0 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
python bench.py --cpu-budget 0 --steps 20 --warmup 5 > gpurun_out/r02a_bench20.json 2>> gpurun_out/r02a_bench.err
python bench.py --cpu-budget 0 --steps 20 --warmup 5 >> gpurun_out/r02a_bench20.json 2>> gpurun_out/r02a_bench.err
cut -c1-200 gpurun_out/r02a_bench.json gpurun_out/r02a_bench20.json
python tools/bench_configs.py > gpurun_out/r02a_bench_configs.txt 2>&1
cat gpurun_out/r02a_bench_configs.txt
python tools/trace_profile.py --jobs 256 --procedure SE-gPoE > gpurun_out/r02a_trace256_se.txt 2>&1
python tools/ablate.py --procedure SE-gPoE > gpurun_out/r02a_ablate.txt 2>&1
cat gpurun_out/r02a_ablate.txt
