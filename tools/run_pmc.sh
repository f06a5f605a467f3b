#!/bin/bash
# HBM traffic of the step kernel from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes).
# usage: tools/run_pmc.sh <tag> [bench args...]   ->  gpurun_out/pmc_<tag>/pmc_hbm_traffic.json
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-budget 0 --lean --repeats 1 --min-warm-s 0 "$@" > $OUT/$ctr.log 2>&1 \
    || { echo "rocprofv3 --pmc $ctr failed:"; tail -5 $OUT/$ctr.log; exit 1; }
done
python3 - "$OUT" "$TAG" "$@" <<'PY'
import csv, glob, collections, json, sys, hashlib
from pathlib import Path
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3:]
root = Path(out).resolve().parents[1]
def arg(name, default):
    return type(default)(args[args.index(name) + 1]) if name in args else default
steps, warm = arg("--steps", 128), arg("--warmup", 16)
jobs, proc = arg("--jobs", 256), arg("--procedure", "SE-gPoE")
spl = min(arg("--steps-per-launch", 128), steps)
tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    s = n = 0
    for f in glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == ctr and "nm_step_kernel" in row.get("Kernel_Name", ""):
                n += 1; s += float(row["Counter_Value"])
    tot[ctr] = (n, s)
    print(ctr, "nm_step_kernel dispatches", n, "sum (KB)", s)
if tot["FETCH_SIZE"][0] == 0 or tot["WRITE_SIZE"][0] == 0 or tot["FETCH_SIZE"][0] != tot["WRITE_SIZE"][0]:
    sys.exit(f"no / unequal nm_step_kernel dispatches in the counter passes ({tot}): no record written")
job_steps = jobs * (steps + warm)                 # every dispatch of the run: warm-up + one timed region
h = hashlib.sha256()
csrc = root / "multi_modal_normative_modeling_amd" / "csrc"
for p in (csrc / "nm_core.inc", csrc / "nmhip.hip", csrc / "nm_rowsplit.hip", csrc / "nm_devpass.hip", csrc / "nm_wide.inc", root / "include" / "nmhip.h"):
    h.update(p.read_bytes())
fetch = tot["FETCH_SIZE"][1] * 1024 / job_steps
write = tot["WRITE_SIZE"][1] * 1024 / job_steps
rec = {"tag": tag, "procedure": proc, "jobs": jobs, "steps": steps, "warmup": warm, "steps_per_launch": spl,
       "kernel_src_sha16": h.hexdigest()[:16],
       "fetch_bytes_per_job_step_raw": fetch, "write_bytes_per_job_step": write,
       "hbm_bytes_per_job_step_raw": fetch + write,
       "hbm_bytes_per_job_step_corrected": 2 * fetch + write,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; counters in KB; FETCH_SIZE doubled "
               "(gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included"}
Path(out, "pmc_hbm_traffic.json").write_text(json.dumps(rec, indent=1))
print(json.dumps(rec))
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
