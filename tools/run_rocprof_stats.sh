#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command; summary lands in gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-budget 0 --lean "$@" > $OUT/bench.json 2> $OUT/bench.err \
  || { echo "rocprofv3 --kernel-trace --stats failed:"; tail -5 $OUT/bench.err; exit 1; }
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
test -s $OUT/kernel_stats.csv || { echo "no kernel_stats.csv was written"; exit 1; }
head -5 $OUT/kernel_stats.csv
tail -1 $OUT/bench.json | cut -c1-400
rm -rf $OUT/raw
