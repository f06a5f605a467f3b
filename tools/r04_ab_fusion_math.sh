#!/bin/bash
# A/B of the fusion arithmetic (hardware transcendentals vs the library's sequences) on the bench workload: three
# interleaved repeats of `bench.py --lean` per library on one box.  usage: tools/r04_ab_fusion_math.sh <alt library name>
ALT=$1
for i in 1 2 3; do
  for lib in libnmhip.so $ALT; do
    v=$(NMHIP_LIB_NAME=$lib python bench.py --lean --cpu-budget 0 --repeats 5 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o['value'], o['roofline']['frac'])")
    echo "$lib $v"
  done
done
