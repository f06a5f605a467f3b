"""Timed CPU leg of bench.py (TEST INFRASTRUCTURE): the oracle's train step driven the way the
reference drives its own (nn.Parameter leaves + torch.optim.Adam, eager CPU fp32), so that the
number is comparable with running the reference itself on the same host cores.
multimodal_kfold_train_cvae_supervised.py:193-199 over cVAE.py:1166-1196."""
from __future__ import annotations

import time
from typing import Dict, List

import torch

from . import cvae_ref as R


class CpuStepper:
    def __init__(self, spec: R.Spec, params: Dict[str, torch.Tensor], combine: str, lr: float = 1e-4):
        self.spec, self.combine = spec, combine
        self.leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        self.opt = torch.optim.Adam(list(self.leaves.values()), lr=lr)

    def step(self, xes: List[torch.Tensor], cs: List[torch.Tensor], eps=None):
        if eps is None:
            eps = torch.randn(xes[0].shape[0], self.spec.latent)        # torch.randn_like(mu), cVAE.py:1132
        fwd = R.forward_multimodal(self.leaves, self.spec, xes, cs, self.combine, eps)
        loss = R.loss_multimodal(self.spec, xes, fwd)
        self.opt.zero_grad()
        loss["total"].sum().backward()
        self.opt.step()
        return loss


def time_cpu_steps(stepper: CpuStepper, batches, budget_s: float = 12.0, warmup: int = 10, min_steps: int = 30):
    """Run train steps over `batches` (list of (xes, cs)) round-robin for about `budget_s` seconds.
    Returns (steps_per_s, steps_done)."""
    nb = len(batches)
    for i in range(warmup):
        stepper.step(*batches[i % nb])
    t0 = time.perf_counter()
    n = 0
    while True:
        stepper.step(*batches[n % nb])
        n += 1
        if n >= min_steps and time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    return n / dt, n
