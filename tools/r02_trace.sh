python tools/trace_profile.py --jobs 256 --procedure SE-gPoE 2>/dev/null > gpurun_out/v2_trace256_se.txt
python tools/trace_profile.py --jobs 1 --procedure SE-gPoE 2>/dev/null > gpurun_out/v2_trace1_se.txt
cat gpurun_out/v2_trace256_se.txt gpurun_out/v2_trace1_se.txt
