python tools/ablate.py --procedure SE-gPoE 2>/dev/null | tail -1
python tools/ablate.py --procedure SM-T1w_sMRI 2>/dev/null | tail -1
python tools/ablate.py --procedure SE-gPoE 2>/dev/null | tail -1
python bench.py --cpu-budget 0 2>/dev/null | cut -c1-260
python tools/bench_configs.py 2>&1 | grep -v amdgpu
