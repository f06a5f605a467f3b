#!/bin/bash
# SQ counter passes over the deviation-pass kernel (tools/bench_deviation.py): one rocprofv3 --pmc pass per group.
# usage: tools/run_pmc_sq_devpass.sh <tag>   ->  stdout: per-counter sums over the nm_devpass_kernel dispatches
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcdv_$TAG
mkdir -p $OUT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/tools/bench_deviation.py --reps 1 > $OUT/g$i.log 2>&1 \
    || { echo "rocprofv3 --pmc group $i failed:"; tail -5 $OUT/g$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "nm_devpass_kernel" in row.get("Kernel_Name", ""):
            k = row["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
want = "$G1 $G2".split()
missing = [k for k in want if agg[k][0] == 0]
for k, (n, v) in sorted(agg.items()):
    print(f"{k:34s} dispatches {n:3d}  sum {v:.6g}  per-dispatch {v / max(n, 1):.6g}")
if missing:
    sys.exit(f"counters without a single nm_devpass_kernel dispatch: {missing}")
PY
rm -rf $OUT/g1 $OUT/g2
