#!/bin/bash
# HBM traffic of the step kernel from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes).
# usage: tools/run_pmc.sh <tag> [bench args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-budget 0 "$@" > $OUT/$ctr.log 2>&1 || true
done
python3 - <<PY
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % ctr, recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if row.get("Counter_Name") == ctr:
                agg[k][0] += 1
                agg[k][1] += float(row["Counter_Value"])
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:4]:
        print(ctr, k[:60], "dispatches", n, "sum", v, "per-dispatch", v / max(n, 1))
PY
