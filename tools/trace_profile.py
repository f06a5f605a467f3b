#!/usr/bin/env python3
"""GPU diagnostic: per-wave interval timers (NM_F_TRACE) of one workgroup."""
import argparse, ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload, _lib
TAGS = {0: "FL issue w", 1: "FL wait+mfma", 2: "FL barrier1", 3: "FL epilogue", 4: "FL barrier2", 5: "FL store_act",
        6: "F0 prologue", 7: "F0 xstore", 8: "F0 barrier a", 9: "F0 issue next", 10: "F0 mfma+cvt next", 11: "F0 barrier b", 12: "F0 epi",
        13: "HD compute", 14: "HD full barrier", 15: "LAT loop", 22: "LAT blocksum", 23: "ZC build", 24: "ZC barrier+save",
        16: "XC prev->top", 17: "XC zero+barrier", 18: "XC issue loads", 19: "XC wait+mfma", 20: "XC epilogue", 21: "XC barrier",
        33: "DG entry(gap)", 34: "DG compute", 35: "DG barrier", 26: "WG request", 27: "WG tiles", 28: "WG barrier", 29: "WG sweep",
        30: "WG barrier2", 31: "DB gap", 32: "DB load_act", 36: "DB barrier", 37: "EB prep", 38: "EB load_act", 39: "EB barrier",
        40: "L0W gap", 41: "L0W xstore", 42: "L0W barrier", 43: "BW gap", 44: "BW finish_delta", 45: "BW barrier", 62: "tail", 63: "step barrier"}
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
