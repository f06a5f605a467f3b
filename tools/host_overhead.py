#!/usr/bin/env python3
"""GPU diagnostic: host time of JobSet.train (everything before the asynchronous launch returns) vs the launch's GPU time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload
cohort = prep.synthetic_cohort(n=1280, d=379)
jobs = workload.build_sweep_jobs(cohort, "SE-gPoE", 5, 256, "cuda:0")
js = nm.JobSet(jobs)
js.train(8); torch.cuda.synchronize()
for k in (20, 20, 20, 128):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); js.train(k); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"train({k}): host {1e3 * (t1 - t0):.3f} ms before the call returns, {1e3 * (t2 - t0):.3f} ms until the GPU is done "
          f"-> {1e3 * (t2 - t0) / k:.4f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); js.train(20); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
