#!/usr/bin/env python3
"""GPU diagnostic: when do the 256 workgroups of a K-step launch finish?  (100 MHz shared counter.)"""
import argparse, ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--jobs", type=int, default=256)
a = ap.parse_args()
cohort = prep.synthetic_cohort(n=1280, d=379)
jobs = workload.build_sweep_jobs(cohort, "SE-gPoE", 5, a.jobs, "cuda:0")
js = nm.JobSet(jobs); js.train(64); torch.cuda.synchronize()
lib = _lib.load(); buf = (C.c_ulonglong * 1024)()
for rep in range(3):
    js._launch(js.jobs[0].step, a.steps, 1, _lib.NM_F_BACKWARD | _lib.NM_F_ADAM | _lib.NM_F_TRACE)
    torch.cuda.synchronize()
    for j in js.jobs: j.step += a.steps; j.t += a.steps
    lib.nm_wgtimes_read(buf)
    t = np.array(buf[:], dtype=np.int64).reshape(512, 2)[:a.jobs] / 100.0      # us
    t0 = t[:, 0].min(); st, en = t[:, 0] - t0, t[:, 1] - t0
    dur = en - st
    dur[0] = np.median(dur)          # workgroup 0 also carries the per-wave tracer of NM_F_TRACE (+5..8 %): not a real straggler
    en[0] = st[0] + dur[0]
    print(f"launch {rep}: {a.steps} steps; start spread {st.max():.1f} us; end: min {en.min():.0f} median {np.median(en):.0f} max {en.max():.0f} us; "
          f"busy time per workgroup: min {dur.min():.0f} median {np.median(dur):.0f} max {dur.max():.0f} us ({100 * (en.max() / np.median(dur) - 1):.1f} % over the median)")
    by_xcd = [np.median(dur[x::8]) for x in range(8)]
    print("   median busy time by XCD (workgroup index mod 8):", " ".join(f"{v:.0f}" for v in by_xcd))
    order = np.argsort(dur)
    print("   slowest workgroups:", [(int(i), int(jobs[i].tables[0].N), round(float(dur[i]))) for i in order[-6:]], " fastest:", [(int(i), round(float(dur[i]))) for i in order[:4]])
    fold = np.array([i % 5 for i in range(a.jobs)])
    print("   median busy time by fold (job index mod 5):", " ".join(f"{np.median(dur[fold == k]):.0f}" for k in range(5)))
    qs = np.percentile(dur, [50, 75, 90, 95, 99, 100])
    print("   busy time percentiles 50/75/90/95/99/100:", " ".join(f"{v:.0f}" for v in qs), "; workgroups > median + 1 %:", int((dur > 1.01 * np.median(dur)).sum()),
          "their indices:", [int(i) for i in np.nonzero(dur > 1.01 * np.median(dur))[0]][:40])
