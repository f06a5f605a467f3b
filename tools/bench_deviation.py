#!/usr/bin/env python3
"""GPU diagnostic: the deviation pass alone (256 unimodal models x 1064 subjects x 379 ROI, forward-only launch)."""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep
ap = argparse.ArgumentParser(); ap.add_argument("--jobs", type=int, default=256); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--general", action="store_true", help="the general forward-only kernel (256-row tiles, loss log) instead of nm_devpass")
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
djs.forward(loss=a.general); torch.cuda.synchronize()
out = []
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        djs.forward(loss=a.general)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 8 * 1e3)
byt = N * a.jobs * 379 * 8.0
if not a.general:
    import ctypes as C
    from multi_modal_normative_modeling_amd import _lib
    lib = _lib.load(); buf = (C.c_ulonglong * 512)()
    lib.nm_trace_read_dv(buf, 1)
    djs._dv_flags = _lib.NM_F_TRACE
    djs.forward(loss=False); torch.cuda.synchronize()
    djs._dv_flags = 0
    lib.nm_trace_read_dv(buf, 1)
    tags = {0: "enc first layer", 1: "enc hidden", 2: "heads + draw", 4: "z|c + request", 5: "dec hidden", 6: "out: wait + GEMM", 7: "out: epilogue", 8: "row sums"}
    print("trace of workgroup (0,0), tile 0: cycles per wave 0 / mean; total", sum(buf[t] for t in range(64)))
    for t in range(64):
        v = [buf[w * 64 + t] for w in range(8)]
        if max(v) > 0:
            print(f"  [{t:2d}] {tags.get(t, ''):20s} {v[0]:9d} {sum(v) // 8:9d}")
print("deviation pass" + (" (general kernel)" if a.general else " (nm_devpass)") + " us/pass:", " ".join(f"{v:.1f}" for v in out), f" frac of 8 TB/s (8 N D): {byt / min(out) / 8e6:.3f}")
