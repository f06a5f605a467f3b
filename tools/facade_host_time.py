#!/usr/bin/env python3
"""GPU diagnostic: host time per call of the reference's train loop on the drop-in class (perf_counter around each call, no
synchronisation inside the loop: what the Python side costs, not the kernels)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import prep
DEV = "cuda:0"
cohort = prep.synthetic_cohort(n=1280, d=379)
folds = prep.kfold_indices(len(cohort.iid), 5)
xs, c = prep.fold_train_tables(cohort, prep.HCP_MODALITIES, folds[0][0])
xs = [torch.from_numpy(x).to(DEV) for x in xs]
c = torch.from_numpy(c).to(DEV)
model = nm.cVAE_multimodal([379] * 3, [110, 110], 10, 29, learning_rate=1e-4, modalities=3, non_linear=True)
model.to(DEV)
B = 256
acc = {"slice": 0.0, "forward_multimodal": 0.0, "loss_function": 0.0, "zero_grad": 0.0, "backward": 0.0, "step": 0.0}
def epoch(rec):
    for b0 in range(0, xs[0].shape[0] - B + 1, B):
        t0 = time.perf_counter()
        xb = [x[b0:b0 + B] for x in xs]; cb = [c[b0:b0 + B]] * 3
        t1 = time.perf_counter()
        fwd = model.forward_multimodal(xb, cb, "gpoe")
        t2 = time.perf_counter()
        loss = model.loss_function_multimodal(xb, fwd)
        t3 = time.perf_counter()
        model.optimizer1.zero_grad()
        t4 = time.perf_counter()
        loss["total"].backward()
        t5 = time.perf_counter()
        model.optimizer1.step()
        t6 = time.perf_counter()
        if rec:
            for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                acc[k] += d
for _ in range(5):
    epoch(False)
torch.cuda.synchronize()
n = 0
t0 = time.perf_counter()
for _ in range(50):
    epoch(True); n += (xs[0].shape[0] // B)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"{n} steps, {tot / n * 1e6:.1f} us per step wall")
for k, v in acc.items():
    print(f"  {k:22s} {v / n * 1e6:8.1f} us host")
