#!/usr/bin/env python3
"""GPU diagnostic: steps/s of the reference's own train loop on the drop-in class (one model, one launch per call:
forward_multimodal -> loss_function_multimodal -> backward -> optimizer1.step), next to the fused JobSet path."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep, workload
DEV = "cuda:0"
cohort = prep.synthetic_cohort(n=1280, d=379)
folds = prep.kfold_indices(len(cohort.iid), 5)
xs, c = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[0][0])
xs = [torch.from_numpy(x).to(DEV) for x in xs]
c = torch.from_numpy(c).to(DEV)
model = nm.cVAE_multimodal([379] * 3, [110, 110], 10, 29, learning_rate=1e-4, modalities=3, non_linear=True)
model.to(DEV)
B = 256
def epoch():
    n = 0
    for b0 in range(0, xs[0].shape[0] - B + 1, B):                      # the loop of multimodal_kfold_train_cvae_supervised.py:186-199
        xb = [x[b0:b0 + B] for x in xs]
        cb = [c[b0:b0 + B]] * 3
        fwd = model.forward_multimodal(xb, cb, "gpoe")
        loss = model.loss_function_multimodal(xb, fwd)
        model.optimizer1.zero_grad(); loss["total"].backward(); model.optimizer1.step()
        n += 1
    return n
epoch(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = sum(epoch() for _ in range(25)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"eager drop-in class, 1 model (3 x 379, batch 256): {n / dt:8.1f} steps/s ({1e3 * dt / n:.3f} ms per step, host-bound)")
jobs = workload.build_sweep_jobs(cohort, "SE-gPoE", 5, 1, DEV)
js = nm.JobSet(jobs); js.train(8, split=False); torch.cuda.synchronize()
t0 = time.perf_counter(); js.train(400, split=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"fused launch,        1 model, one workgroup:          {400 / dt:8.1f} steps/s")
js.train(8); torch.cuda.synchronize()
t0 = time.perf_counter(); js.train(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"fused launch,        1 model, one workgroup / modality: {400 / dt:8.1f} steps/s")
