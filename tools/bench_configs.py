#!/usr/bin/env python3
"""GPU diagnostic: train-step throughput of every BASELINE.json model shape with the chip full of jobs
(256 independent models, one workgroup each).  Prints one line per shape: us per sweep step, steps/s,
algorithmic bytes per job-step and the HBM-roofline fraction they imply."""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload

ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=256)
ap.add_argument("--steps", type=int, default=32)
a = ap.parse_args()
DEV = "cuda:0"
cohort = prep.synthetic_cohort(n=1280, d=379)
folds = prep.kfold_indices(len(cohort.iid), 5, 42)


def timed(fn, steps):
    fn(4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(steps); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3          # us per sweep step


def report(name, us, bytes_js, n_jobs):
    sps = n_jobs / us * 1e6
    print(f"{name:44s} {us:8.1f} us/sweep-step  {sps:10.0f} steps/s  {bytes_js / 1e6:6.2f} MB/job-step  "
          f"frac {sps * bytes_js / 8e12:5.3f}", flush=True)


for proc in ("SM-T1w_sMRI", "SE-gPoE", "SM-" + prep.EARLY_FUSION, "UCA-gPoE"):
    jobs = workload.build_sweep_jobs(cohort, proc, 5, a.jobs, DEV)
    js = nm.JobSet(jobs)
    us = timed(js.train, a.steps)
    report(proc, us, workload.step_work(jobs[0].spec.input_dims)["bytes"], a.jobs)
    del js, jobs
    torch.cuda.empty_cache()

# regression model (3 x 379, c = 2 raw covariates): export launch + head kernel + trunk launch per step
tabs = {}
def reg_jobs():
    out = []
    cov = np.stack([cohort.age, cohort.gender], axis=1).astype(np.float32)
    for j in range(a.jobs):
        k = j % 5
        if k not in tabs:
            xs, _ = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[k][0])
            tabs[k] = [nm.Table(x, cov[folds[k][0]], DEV) for x in xs]
        spec = nm.ModelSpec([379] * 3, [110, 110], 10, 2, True, "regression")
        job = nm.Job(spec, tabs[k], combine="gpoe", seed=j, init_seed=42 + j, loss_cap=8)
        job.set_fi(cohort.fi[folds[k][0]].astype(np.float32))
        out.append(job)
    return out
jobs = reg_jobs()
js = nm.JobSet(jobs)
us = timed(js.train_regression, a.steps)
n_reg = 1137 * 128 + 128 + 128 * 64 + 64 + 64 + 1
report("regression 3x379 (trunk + head kernels)", us, workload.step_work([379] * 3, c_dim=2)["bytes"] + 24 * n_reg, a.jobs)
del js, jobs
tabs.clear()
torch.cuda.empty_cache()

# config 5: end-to-end, Z = 64, classifier [128, 64, 32]
def e2e_jobs():
    out = []
    lab = (cohort.dia != 1).astype(np.int32)
    for j in range(a.jobs):
        k = j % 5
        if k not in tabs:
            xs, c = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[k][0])
            tabs[k] = [nm.Table(x, c, DEV) for x in xs]
        spec = nm.ModelSpec([379] * 3, [110, 110], 64, 29, True, "endtoend", (128, 64, 32), 2)
        job = nm.Job(spec, tabs[k], combine="poe", kl_weight=0.1, ll_weight=0.1, seed=j, init_seed=42 + j, loss_cap=8,
                     single_bypass=False)
        job.cls_dropout = 0.5
        job.set_labels(lab[folds[k][0]])
        out.append(job)
    return out
jobs = e2e_jobs()
js = nm.JobSet(jobs)
us = timed(js.train_endtoend, a.steps)
n_par = jobs[0].layout.n_params
report("config 5 end-to-end Z=64 cls[128,64,32]", us, 4.0 * 256 * (1137 + 29 + 64) + 24.0 * n_par, a.jobs)

# deviation pass (..._regression.py:163-192): forward-only, one workgroup per (model, 256-row tile), squared residuals out
del js, jobs
tabs.clear()
torch.cuda.empty_cache()
N = 1064
xall = prep.robust_scaler_transform(cohort.x["T1w_sMRI"][:N].astype(np.float32), *prep.robust_scaler_fit(cohort.x["T1w_sMRI"][:N].astype(np.float32))).astype(np.float32)
call = prep.one_hot_covariates(cohort.age[:N], cohort.gender[:N])
tab = nm.Table(xall, call, DEV)
djobs = []
for j in range(a.jobs):
    spec = nm.ModelSpec([379], [110, 110], 10, 29)
    job = nm.Job(spec, [tab], combine="poe", seed=j, init_seed=42 + j, n_tiles_ws=tab.n_tiles)
    job.enable_exports(loc=False, sqerr=True, rowdev=True, latent=False)
    djobs.append(job)
djs = nm.JobSet(djobs)
djs.forward(loss=False); torch.cuda.synchronize()          # (the compact deviation-pass kernel: nm_devpass)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8):
    djs.forward(loss=False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 8
rows = N * a.jobs
byt = rows * 379 * 8.0                                   # 4 N D read + 4 N D write
print(f"{'deviation pass 379 ROI, N=1064 x %d models' % a.jobs:44s} {ms * 1e3:8.1f} us/pass  {rows / ms * 1e3:12.0f} rows/s  "
      f"{byt / ms / 1e6:8.1f} GB/s algorithmic  frac {byt / ms / 1e6 / 8000:5.3f}", flush=True)
