#!/usr/bin/env python3
"""GPU diagnostic: time per sweep step for kernel flag combinations (forward only / +backward / +Adam)."""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--procedure", default="SM-T1w_sMRI")
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--xcd", action="store_true")
a = ap.parse_args()
cohort = prep.synthetic_cohort(n=1280, d=379)
for nj in (256,):
    jobs = workload.build_sweep_jobs(cohort, a.procedure, 5, nj, "cuda:0", xcd_affinity=a.xcd)
    js = nm.JobSet(jobs)
    out = []
    for name, flags in (("fwd", 0), ("fwd+bwd", _lib.NM_F_BACKWARD), ("fwd+bwd+adam", _lib.NM_F_BACKWARD | _lib.NM_F_ADAM)):
        js._launch(0, 4, 1, flags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); js._launch(0, a.steps, 1, flags); e1.record()
        torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1) / a.steps * 1e3:8.1f} us/step")
    print(f"{a.procedure} jobs={nj:4d}: " + " | ".join(out), flush=True)
    del js, jobs
    torch.cuda.empty_cache()
