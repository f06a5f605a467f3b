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
        15: "enc L0 wgrad+adam", 62: "tail", 63: "step barrier",
        20: "pass-1 hand-off", 21: "reg head: layer 1 fwd", 22: "reg head: layers 2-3 + MSE", 23: "reg head: bwd layers 3-2",
        24: "reg head: L1 wgrad+adam", 25: "reg head: L1 d x_hat (dgrad)", 26: "cls head: hidden fwd", 27: "cls head: out + CE + hinge",
        29: "head tail (cls: backward) + hand-off", 30: "cls fwd: input save + bias + GEMM", 31: "cls fwd: batch statistics",
        32: "cls fwd: BN / ReLU / dropout epilogue",
        # general-shape path (nm_wide.inc)
        40: "wide enc L0 fwd", 41: "wide enc hidden fwd", 42: "wide enc heads", 43: "wide fusion + KL", 44: "wide z|c tiles: hand-off",
        45: "wide dec hidden fwd", 46: "wide out fwd + NLL + delta", 48: "wide dgrad parts (all layers)", 49: "wide out wgrad+adam",
        51: "wide dec hidden wgrad+adam", 52: "wide fusion bwd", 53: "wide enc heads dgrad", 54: "wide enc heads wgrad+adam",
        55: "wide enc hidden wgrad+adam", 56: "wide enc L0 wgrad+adam", 57: "wide z|c tiles: set-up", 58: "wide z|c tiles: loop"}
ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=1)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--procedure", default="SM-T1w_sMRI")
ap.add_argument("--forward", action="store_true", help="trace the forward-only export launch over all row tiles (deviation pass)")
ap.add_argument("--head", choices=["", "regression", "endtoend"], default="",
                help="trace the head-model launch (nm_train_steps_head) of 3 x 379 regression / config-5 end-to-end jobs")
ap.add_argument("--wide", default="", help="general-shape path: '-H'-style list, e.g. '110 110 100' (hidden widths, then the latent)")
a = ap.parse_args()
cohort = prep.synthetic_cohort(n=1280, d=379)
lib = _lib.load()
buf = (C.c_ulonglong * 512)()
if a.head:
    import numpy as np
    folds = prep.kfold_indices(len(cohort.iid), 5)
    tabs, jobs = {}, []
    for j in range(a.jobs):
        k = j % 5
        if a.head == "regression":
            if k not in tabs:
                xs, _ = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[k][0])
                cov = np.stack([cohort.age, cohort.gender], axis=1).astype(np.float32)
                tabs[k] = [nm.Table(x, cov[folds[k][0]], "cuda:0") for x in xs]
            job = nm.Job(nm.ModelSpec([379] * 3, [110, 110], 10, 2, True, "regression"), tabs[k], combine="gpoe", seed=j,
                         init_seed=42 + j, loss_cap=8)
            job.set_fi(cohort.fi[folds[k][0]].astype(np.float32))
        else:
            if k not in tabs:
                xs, c = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[k][0])
                tabs[k] = [nm.Table(x, c, "cuda:0") for x in xs]
            job = nm.Job(nm.ModelSpec([379] * 3, [110, 110], 64, 29, True, "endtoend", (128, 64, 32), 2), tabs[k], combine="poe",
                         kl_weight=0.1, ll_weight=0.1, seed=j, init_seed=42 + j, loss_cap=8, single_bypass=False)
            job.cls_dropout = 0.5
            job.set_labels((cohort.dia != 1).astype(np.int32)[folds[k][0]])
        jobs.append(job)
    js = nm.JobSet(jobs)
    run = js.train_regression if a.head == "regression" else js.train_endtoend
    run(4); torch.cuda.synchronize()
    lib.nm_trace_read(buf, 1)
    js._train_head(js.jobs[0].step, a.steps, _lib.NM_F_TRACE)
else:
    if a.wide:
        hz = [int(v) for v in a.wide.split()]
        mods, _ = workload.procedure_modalities(a.procedure)
        folds = prep.kfold_indices(len(cohort.iid), 5, 42)
        xs, cc = prep.fold_train_tables(cohort, mods, folds[0][0])
        tabs = [nm.Table(x, cc, "cuda:0") for x in xs]
        spec = nm.ModelSpec([t.D for t in tabs], hz[:-1], hz[-1], 29)
        jobs = [nm.Job(spec, tabs, combine="gpoe", seed=j, init_seed=42 + j, loss_cap=8) for j in range(a.jobs)]
    else:
        jobs = workload.build_sweep_jobs(cohort, a.procedure, 5, a.jobs, "cuda:0")
    js = nm.JobSet(jobs)
    js.train(4); torch.cuda.synchronize()
    if a.forward:
        for j in jobs:
            j.enable_exports(loc=False, sqerr=True, rowdev=True, latent=False)
        nt = jobs[0].tables[0].n_tiles
        js.forward(); torch.cuda.synchronize()
        lib.nm_trace_read(buf, 1)
        a.steps = 4
        for _ in range(a.steps):
            js._launch(0, 1, nt, _lib.NM_F_EXPORT | _lib.NM_F_TRACE)
    else:
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
