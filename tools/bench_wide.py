#!/usr/bin/env python3
"""GPU diagnostic: train-step throughput of the general-shape path (nm_launch_wide) at shapes of the reference's sweeps, next to
the fused kernel at the default shape (256 models each, one workgroup per model)."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload
DEV = "cuda:0"
cohort = prep.synthetic_cohort(n=1280, d=379)
folds = prep.kfold_indices(len(cohort.iid), 5, 42)
xs, c = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[0][0])
tabs = [nm.Table(x, c, DEV) for x in xs]
for name, mods, hidden, Z, jobs in (("fused  SE 3x379 [110,110]/10", 3, [110, 110], 10, 256),
                                    ("wide   SE 3x379 [110,110]/70", 3, [110, 110], 70, 256),
                                    ("wide   SE 3x379 [110,110]/100", 3, [110, 110], 100, 256),
                                    ("wide   SE 3x379 [300,300]/30", 3, [300, 300], 30, 256),
                                    ("wide   SM 379 [1024,512,256]/32", 1, [1024, 512, 256], 32, 256),
                                    ("wide   SM 379 [2048]/10", 1, [2048], 10, 256)):
    spec = nm.ModelSpec([379] * mods, hidden, Z, 29)
    js = nm.JobSet([nm.Job(spec, tabs[:mods], combine="gpoe", seed=j, init_seed=42 + j, loss_cap=8) for j in range(jobs)])
    js.train(2); torch.cuda.synchronize()
    n = int(os.environ.get("NM_WIDE_STEPS", "64"))
    t0 = time.perf_counter(); js.train(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    js.assert_finite()
    w = workload.step_work(spec.input_dims, hidden=hidden, latent=Z)
    print(f"{name:34s} {dt / n * 1e6:9.1f} us/sweep-step  {jobs * n / dt:10.0f} steps/s  {w['bytes'] / 1e6:7.2f} MB/job-step  "
          f"frac {jobs * n / dt * w['bytes'] / 8e12:5.3f}  MFMA {jobs * n / dt * w['flop'] / 1e12:6.1f} TF/s", flush=True)
    del js
    torch.cuda.empty_cache()
