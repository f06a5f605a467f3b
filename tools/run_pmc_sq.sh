#!/bin/bash
# SQ / TA counter passes over the bench command (one rocprofv3 --pmc pass per group, kernel-trace only).
# usage: tools/run_pmc_sq.sh <tag>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$TAG
mkdir -p $OUT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
G3="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
# instruction cache (VERDICT r2 #1b: the step's loop body is ~280 KB of code)
G4="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ SQ_INSTS_SMEM SQ_WAVES"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-budget 0 --lean --steps 32 --warmup 4 --repeats 1 --min-warm-s 0 --steps-per-launch 32 > $OUT/g$i.log 2>&1 \
    || { echo "rocprofv3 --pmc group $i failed:"; tail -5 $OUT/g$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "nm_step_kernel" in row.get("Kernel_Name", ""):
            k = row["Counter_Name"]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
import sys
want = "$G1 $G2 $G3 $G4".split()
missing = [k for k in want if agg[k][0] == 0]
for k, (n, v) in sorted(agg.items()):
    print(f"{k:34s} dispatches {n:3d}  sum {v:.6g}  per-dispatch {v / max(n, 1):.6g}")
if missing:
    sys.exit(f"counters without a single nm_step_kernel dispatch: {missing} -- the record is partial")
PY
rm -rf $OUT/g1 $OUT/g2 $OUT/g3 $OUT/g4
