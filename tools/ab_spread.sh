#!/bin/bash
# launch time (last workgroup's end) of 20- and 128-step launches for several start-offset scales
O=gpurun_out/ab_spread.txt; : > $O
for st in 20 128; do for sc in 0 0.5 1 2; do
  echo "== steps $st, NMHIP_DEPHASE_SCALE=$sc" >> $O
  NMHIP_DEPHASE_SCALE=$sc python tools/wg_spread.py --steps $st 2>/dev/null | grep "^launch" | cut -c1-220 >> $O
done; done
cat $O
