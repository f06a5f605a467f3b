#!/usr/bin/env python3
"""GPU diagnostic: long fused training of a full sweep shard (256 models) -- every loss stays finite and falls."""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload

ap = argparse.ArgumentParser()
ap.add_argument("--procedure", default="SE-gPoE")
ap.add_argument("--steps", type=int, default=2048)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--wide", default="", help="general-shape path: '-H'-style list, e.g. '300 300 30'")
a = ap.parse_args()
cohort = prep.synthetic_cohort(n=1280, d=379)
if a.wide:
    hz = [int(v) for v in a.wide.split()]
    mods, combine = workload.procedure_modalities(a.procedure)
    folds = prep.kfold_indices(len(cohort.iid), 5, 42)
    tabs = {}
    jobs = []
    for j in range(256):
        k = j % 5
        if k not in tabs:
            xs, cc = prep.fold_train_tables(cohort, mods, folds[k][0])
            tabs[k] = [nm.Table(x, cc, "cuda:0") for x in xs]
        spec = nm.ModelSpec([t.D for t in tabs[k]], hz[:-1], hz[-1], 29)
        assert spec.wide
        jobs.append(nm.Job(spec, tabs[k], combine=combine, lr=a.lr, seed=j, init_seed=42 + j, loss_cap=64))
else:
    jobs = workload.build_sweep_jobs(cohort, a.procedure, 5, 256, "cuda:0", lr=a.lr)
js = nm.JobSet(jobs)
first = None
done = 0
while done < a.steps:
    js.train(64)
    done += 64
    torch.cuda.synchronize()
    last = torch.stack([j.loss_log[(j.step - 1) % j.loss_cap] for j in jobs]).cpu()
    if first is None:
        first = torch.stack([j.loss_log[0] for j in jobs]).cpu()
    assert torch.isfinite(last).all(), f"non-finite loss after {done} steps"
    if done % 512 == 0:
        print(f"step {done:5d}: total mean {float(last[:, 0].mean()):10.2f}  min {float(last[:, 0].min()):10.2f}  max {float(last[:, 0].max()):10.2f}", flush=True)
print("first-step total mean %.2f -> last %.2f" % (float(first[:, 0].mean()), float(last[:, 0].mean())))
assert float(last[:, 0].mean()) < 0.8 * float(first[:, 0].mean())
params = torch.stack([j.params for j in jobs])
assert torch.isfinite(params).all()
print("ok")
