#!/bin/bash
# 20-step launches (the driver's bench form): with / without start offsets, two repeats each
O=gpurun_out/ab_dephase20.txt; : > $O
for rep in 1 2; do
  echo -n "base            " >> $O; python bench.py --cpu-budget 0 --steps 20 --warmup 5 2>/dev/null | cut -c60-130 >> $O
  for sc in 1 0.5 0.25; do
    echo -n "dephase x$sc     " >> $O; NMHIP_LIB_NAME=libnmhip_dp16.so NMHIP_DEPHASE_SCALE=$sc python bench.py --cpu-budget 0 --steps 20 --warmup 5 2>/dev/null | cut -c60-130 >> $O
  done
done
cat $O
