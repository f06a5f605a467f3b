#!/usr/bin/env python3
"""GPU diagnostic: long single-launch training of head models (256 regression / end-to-end jobs): every loss stays
finite, the regression MSE and the classifier's cross entropy fall, parameters stay finite."""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=768); ap.add_argument("--jobs", type=int, default=256)
a = ap.parse_args()
DEV = "cuda:0"
cohort = prep.synthetic_cohort(n=1280, d=379)
cohort.fi[:] = (cohort.fi - cohort.fi.mean()) / cohort.fi.std()
folds = prep.kfold_indices(len(cohort.iid), 5)
for kind in ("regression", "endtoend"):
    tabs, jobs = {}, []
    for j in range(a.jobs):
        k = j % 5
        if kind == "regression":
            if k not in tabs:
                xs, _ = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[k][0])
                cov = np.stack([cohort.age, cohort.gender], axis=1).astype(np.float32)
                tabs[k] = [nm.Table(x, cov[folds[k][0]], DEV) for x in xs]
            job = nm.Job(nm.ModelSpec([379] * 3, [110, 110], 10, 2, True, "regression"), tabs[k], combine="gpoe", seed=j,
                         init_seed=42 + j, loss_cap=16, lr=1e-3)
            job.set_fi(cohort.fi[folds[k][0]].astype(np.float32))
        else:
            if k not in tabs:
                xs, c = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[k][0])
                tabs[k] = [nm.Table(x, c, DEV) for x in xs]
            job = nm.Job(nm.ModelSpec([379] * 3, [110, 110], 64, 29, True, "endtoend", (128, 64, 32), 2), tabs[k], combine="poe",
                         kl_weight=0.1, ll_weight=0.1, seed=j, init_seed=42 + j, loss_cap=16, single_bypass=False, lr=1e-3)
            job.cls_dropout = 0.5
            job.set_labels((cohort.dia != 1).astype(np.int32)[folds[k][0]])
        jobs.append(job)
    js = nm.JobSet(jobs)
    run = js.train_regression if kind == "regression" else js.train_endtoend
    col = _lib.NM_LOSS_REG if kind == "regression" else _lib.NM_LOSS_CE
    first, done = None, 0
    while done < a.steps:
        run(128); done += 128
        torch.cuda.synchronize()
        js.assert_finite()
        last = torch.stack([j.loss_log[(j.step - 1) % j.loss_cap] for j in jobs]).cpu()
        if first is None:
            first = torch.stack([j.loss_log[0] for j in jobs]).cpu()
        print(f"{kind:10s} step {done:5d}: total mean {float(last[:, 0].mean()):10.3f}  head loss mean {float(last[:, col].mean()):8.4f}", flush=True)
    assert torch.isfinite(torch.stack([j.params for j in jobs])).all()
    assert float(last[:, col].mean()) < 0.9 * float(first[:, col].mean()), (float(first[:, col].mean()), float(last[:, col].mean()))
    del js, jobs
    torch.cuda.empty_cache()
print("ok")
