#!/usr/bin/env python3
"""GPU diagnostic: cProfile of the reference's train loop on the drop-in class (host time per call)."""
import cProfile, pstats, sys, io
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
def epoch():
    for b0 in range(0, xs[0].shape[0] - B + 1, B):
        xb = [x[b0:b0 + B] for x in xs]
        cb = [c[b0:b0 + B]] * 3
        fwd = model.forward_multimodal(xb, cb, "gpoe")
        loss = model.loss_function_multimodal(xb, fwd)
        model.optimizer1.zero_grad(); loss["total"].backward(); model.optimizer1.step()
epoch(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(25):
    epoch()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
