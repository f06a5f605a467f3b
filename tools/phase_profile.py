#!/usr/bin/env python3
"""GPU diagnostic: per-phase shader-clock cycles of one workgroup of the step kernel.
Usage: python tools/phase_profile.py [--jobs J] [--procedure SM-T1w_sMRI] [--steps S]"""
import argparse, ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload, _lib

PH = ["ENC_L0", "ENC_REST", "HEADS", "LATENT", "DEC_ZC", "DEC_HID", "OUT_GEMM", "OUT_DLV", "OUT_DGRAD", "OUT_WGRAD",
      "NLL_RED", "DEC_FINISH", "DEC_LOAD", "DEC_DGRAD", "DEC_WGRAD", "DEC_DELTA", "ALPHA", "ENCB_PREP",
      "ENCB_HEADS_DGRAD", "ENCB_HEADS_WGRAD", "ENCB_LOAD", "ENCB_DGRAD", "ENCB_WGRAD", "ENCB_DELTA", "ENCB_L0_WGRAD",
      "X_LOADS", "X_MFMA", "X_EPI"]
ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=1)
ap.add_argument("--procedure", default="SM-T1w_sMRI")
ap.add_argument("--steps", type=int, default=16)
a = ap.parse_args()
t0 = time.time()
cohort = prep.synthetic_cohort(n=1280, d=379)
jobs = workload.build_sweep_jobs(cohort, a.procedure, 5, a.jobs, "cuda:0")
print(f"setup {time.time() - t0:.1f}s", flush=True)
js = nm.JobSet(jobs)
js.train(4)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 32)()
lib.nm_prof_read(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); js.train(a.steps, profile=True); e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
lib.nm_prof_read(buf, 1)
tot = sum(buf[i] for i in range(len(PH)))
print(f"{a.procedure} jobs={a.jobs}: {ms / a.steps * 1e3:.1f} us/step (events); cycles/step {tot / a.steps:.0f}")
for i, n in enumerate(PH):
    print(f"  {n:20s} {buf[i] / a.steps:10.0f} cyc  {100.0 * buf[i] / max(tot, 1):5.1f}%")
