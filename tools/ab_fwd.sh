#!/bin/bash
O=gpurun_out/ab_fwd.txt; : > $O
for rep in 1 2; do for v in fd0 fd32 fd16 fd8 fd4 fd2; do
  echo -n "$v rep$rep " >> $O
  NMHIP_LIB_NAME=libnmhip_$v.so python tools/bench_deviation.py 2>/dev/null | tail -1 >> $O
done; done
cat $O
