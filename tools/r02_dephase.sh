for mode in cu 0 xcd cu; do
  echo -n "dephase=$mode " ; NMHIP_DEPHASE=$mode python bench.py --cpu-budget 0 2>/dev/null | cut -c70-125
done
for mode in cu 0; do
  echo -n "SM dephase=$mode " ; NMHIP_DEPHASE=$mode python bench.py --cpu-budget 0 --procedure SM-T1w_sMRI 2>/dev/null | cut -c70-125
done
