#!/usr/bin/env python3
"""GPU diagnostic: per-wave interval timers (NM_F_TRACE) of one workgroup."""
import argparse, ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload, _lib
TAGS = {0: "enc L0 fwd", 1: "enc hidden fwd", 2: "enc heads", 3: "latent + KL", 4: "zc build", 5: "dec hidden fwd",
        6: "out: wait+x req+mfma", 7: "out: epilogue", 8: "out: dlv + dgrad", 9: "out: wgrad+adam", 10: "dec bwd dgrad+act",
        11: "dec bwd wgrad+adam", 12: "dz + fusion bwd", 13: "enc bwd heads dgrad+act", 14: "enc bwd heads/hidden wgrad+...",
        15: "enc L0 wgrad+adam", 62: "tail", 63: "step barrier"}
ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=1)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--procedure", default="SM-T1w_sMRI")
a = ap.parse_args()
cohort = prep.synthetic_cohort(n=1280, d=379)
jobs = workload.build_sweep_jobs(cohort, a.procedure, 5, a.jobs, "cuda:0")
js = nm.JobSet(jobs)
js.train(4); torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 512)()
lib.nm_trace_read(buf, 1)
js._launch(js.jobs[0].step, a.steps, 1, _lib.NM_F_BACKWARD | _lib.NM_F_ADAM | _lib.NM_F_TRACE)
torch.cuda.synchronize()
lib.nm_trace_read(buf, 1)
tot = [sum(buf[w * 64 + t] for t in range(64)) / a.steps for w in range(8)]
print("cycles/step per wave:", [int(x) for x in tot])
print(f"{'tag':22s}" + "".join(f"{'w%d' % w:>9s}" for w in range(8)))
for t in range(64):
    row = [buf[w * 64 + t] / a.steps for w in range(8)]
    if max(row) > 0:
        print(f"{TAGS.get(t, str(t)):22s}" + "".join(f"{int(x):9d}" for x in row))
