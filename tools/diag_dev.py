#!/usr/bin/env python3
"""GPU diagnostic: forward-only (deviation) pass timing by export set and tile count."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, _lib
cohort = prep.synthetic_cohort(n=1280, d=379)
for N in (256, 1064):
    x = cohort.x["T1w_sMRI"][:N].astype(np.float32)
    call = prep.one_hot_covariates(cohort.age[:N], cohort.gender[:N])
    tab = nm.Table(x, call, "cuda:0")
    for exports in ((False, False, False, False), (False, True, True, False), (True, True, True, True)):
        jobs = []
        for j in range(256):
            job = nm.Job(nm.ModelSpec([379], [110, 110], 10, 29), [tab], combine="poe", seed=j, init_seed=42 + j, n_tiles_ws=tab.n_tiles)
            job.enable_exports(*exports)
            jobs.append(job)
        js = nm.JobSet(jobs)
        for flags in (0, _lib.NM_F_EXPORT):
            js._launch(0, 1, tab.n_tiles, flags); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                js._launch(0, 1, tab.n_tiles, flags)
            e1.record(); torch.cuda.synchronize()
            print(f"N={N} tiles={tab.n_tiles} exports(loc,sqerr,rowdev,latent)={exports} flags={flags}: {e0.elapsed_time(e1) / 4 * 1e3:9.1f} us", flush=True)
        del js, jobs
