#!/bin/bash
# rocprofv3 --kernel-trace --stats of a tools/*.py diagnostic; summary lands in gpurun_out/prof_<tag>/kernel_stats.csv
set -e
TAG=$1; shift
SCRIPT=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > $OUT/out.txt 2> $OUT/err.txt || true
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cut -c1-160 $OUT/kernel_stats.csv | head -8
rm -rf $OUT/raw
