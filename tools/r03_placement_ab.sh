#!/bin/bash
# A/B on one box: fold placement x start-offset scheme (bench --lean, 128-step launches)
for rep in 1 2 3; do
  for arm in "none cu" "rank cu" "rank xcd" "rank 0" "none xcd"; do
    set -- $arm
    echo -n "placement=$1 dephase=$2: "
    NMHIP_DEPHASE=$2 python bench.py --cpu-budget 0 --lean --placement $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_min'])"
  done
done
