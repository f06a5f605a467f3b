#!/usr/bin/env python3
"""GPU diagnostic: the deviation pass alone (256 unimodal models x 1064 subjects x 379 ROI, forward-only launch)."""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep
ap = argparse.ArgumentParser(); ap.add_argument("--jobs", type=int, default=256); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
DEV = "cuda:0"
cohort = prep.synthetic_cohort(n=1280, d=379)
N = 1064
x32 = cohort.x["T1w_sMRI"][:N].astype(np.float32)
xall = prep.robust_scaler_transform(x32, *prep.robust_scaler_fit(x32)).astype(np.float32)
call = prep.one_hot_covariates(cohort.age[:N], cohort.gender[:N])
tab = nm.Table(xall, call, DEV)
djobs = []
for j in range(a.jobs):
    job = nm.Job(nm.ModelSpec([379], [110, 110], 10, 29), [tab], combine="poe", seed=j, init_seed=42 + j, n_tiles_ws=tab.n_tiles)
    job.enable_exports(loc=False, sqerr=True, rowdev=True, latent=False)
    djobs.append(job)
djs = nm.JobSet(djobs)
djs.forward(); torch.cuda.synchronize()
out = []
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        djs.forward()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 8 * 1e3)
byt = N * a.jobs * 379 * 8.0
print("deviation pass us/pass:", " ".join(f"{v:.1f}" for v in out), f" frac of 8 TB/s (8 N D): {byt / min(out) / 8e6:.3f}")
