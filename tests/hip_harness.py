"""Shared helpers for the GPU parity tests: run the HIP path on a golden case and collect
its outputs next to the oracle's (tests only; the oracle is the checker, never the product)."""
from __future__ import annotations

import torch

import multi_modal_normative_modeling_amd as nm
from oracle import cvae_ref as R
from tests.golden_util import Golden

DEV = "cuda:0"


def make_job(g: Golden, step: int, state=None, combine=None, kind="multimodal"):
    spec = nm.ModelSpec(g.dims, g.hidden, g.Z, g.c_dim, kind=kind)
    xes = g.xs(step)
    c = g.t("c")[step] if g.t("c").dim() == 3 else g.t("c")
    tables = [nm.Table(xes[m], c, DEV) for m in range(g.M)]
    job = nm.Job(spec, tables, combine=combine or g.combine, state=state if state is not None else g.weights("w0"))
    eps = g.t("eps")
    job.set_eps(eps[step] if eps.dim() == 3 else eps)
    return job


def swap_batch(job: nm.Job, g: Golden, step: int):
    """Point the job at the rows of golden step `step` (the golden cases use a fresh batch per
    step; the kernel's batch index is always 0 here)."""
    xes = g.xs(step)
    c = g.t("c")[step]
    job.tables = [nm.Table(xes[m], c, DEV) for m in range(g.M)]
    job.set_eps(g.t("eps")[step])
    job.step = 0
    job.touch()


def oracle_step0(g: Golden, kind="multimodal"):
    rs = R.Spec(g.dims, g.hidden, g.Z, g.c_dim, kind=kind)
    P = g.weights("w0")
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xes = g.xs(0)
    c = g.t("c")[0] if g.t("c").dim() == 3 else g.t("c")
    eps = g.t("eps")[0] if g.t("eps").dim() == 3 else g.t("eps")
    fwd = R.forward_multimodal(leaves, rs, xes, [c.long()] * g.M, g.combine, eps)
    loss = R.loss_multimodal(rs, xes, fwd)
    loss["total"].sum().backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return fwd, loss, grads


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|"""
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
