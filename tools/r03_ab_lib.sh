#!/bin/bash
# A/B of two library builds on one box (bench --lean, 128-step launches, 3 interleaved repeats): tools/r03_ab_lib.sh libA libB
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "
    NMHIP_LIB_NAME=$lib python bench.py --cpu-budget 0 --lean 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_min'])"
  done
done
