#!/bin/bash
# A/B of library variants on one box: tools/ab.sh <tag> <libA> <libB> ...  (ablate SE-gPoE + SM per variant, interleaved, 2 reps)
TAG=$1; shift
OUT=gpurun_out/ab_$TAG.txt
: > $OUT
for rep in 1 2; do
  for lib in "$@"; do
    for proc in SE-gPoE SM-T1w_sMRI; do
      echo -n "$lib rep$rep " >> $OUT
      NMHIP_LIB_NAME=$lib python tools/ablate.py --procedure $proc 2>/dev/null | tail -1 >> $OUT
    done
  done
done
cat $OUT
