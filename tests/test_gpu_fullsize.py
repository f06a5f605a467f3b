"""Parity at BASELINE.json's full sizes (configs 2-5) and size-independent properties.

No golden files at these sizes: the oracle (pinned by tests/test_oracle_golden.py) is evaluated on
the same seeded inputs, in the reference's fp32 arithmetic (1e-4 bound on the reconstruction loss)
and in bf16-operand mode (tight elementwise agreement)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import multi_modal_normative_modeling_amd as nm
from oracle import cvae_ref as R

DEV = "cuda:0"


def onehot(g, B, c_dim=29):
    a = torch.randint(0, c_dim - 2, (B,), generator=g)
    s = torch.randint(0, 2, (B,), generator=g)
    c = torch.zeros(B, c_dim)
    c[torch.arange(B), a] = 1
    c[torch.arange(B), c_dim - 2 + s] = 1
    return c


def run_case(dims, Z, combine, B, seed, hidden=(110, 110), c_dim=29, non_linear=True):
    g = torch.Generator().manual_seed(seed)
    spec = nm.ModelSpec(list(dims), list(hidden), Z, c_dim, non_linear)
    lay = nm.ParamLayout(spec)
    P = lay.init_reference_rule(seed)
    xs = [torch.randn(B, d, generator=g) * 1.2 for d in dims]
    c = onehot(g, B, c_dim)
    eps = torch.randn(B, Z, generator=g)
    job = nm.Job(spec, [nm.Table(x, c, DEV) for x in xs], combine=combine, state=P)
    job.set_eps(eps)
    job.enable_exports(sqerr=False, rowdev=False)
    nm.JobSet([job]).grads(0)
    torch.cuda.synchronize()
    rs = R.Spec(list(dims), list(hidden), Z, c_dim, non_linear)
    res = {}
    for mode in ("fp32", "bf16"):
        R.set_operand_rounding(mode)
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            fwd = R.forward_multimodal(leaves, rs, xs, [c.long()] * len(dims), combine, eps)
            loss = R.loss_multimodal(rs, xs, fwd)
            loss["total"].sum().backward()
            res[mode] = (fwd, loss, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()})
        finally:
            R.set_operand_rounding("fp32")
    row = job.loss_log[0].cpu()
    ll32, ll16 = float(res["fp32"][1]["ll"]), float(res["bf16"][1]["ll"])
    assert abs(float(row[2]) - ll32) <= 1e-4 * abs(ll32), (float(row[2]), ll32)          # north-star bound
    assert abs(float(row[2]) - ll16) <= 5e-6 * abs(ll16)
    tot32 = float(res["fp32"][1]["total"])
    assert abs(float(row[0]) - tot32) <= 1e-4 * abs(tot32)
    mu16 = res["bf16"][0]["mu"].detach()
    assert float((job.out_mu[:B].cpu() - mu16).abs().max()) <= 3e-3 * float(mu16.abs().max())
    for m in range(len(dims)):
        l16 = res["bf16"][0]["locs"][m].detach()
        assert float((job.out_loc[m][:B].cpu() - l16).abs().max()) <= 3e-3 * float(l16.abs().max()), m
    got = job.grads_dict()
    for k, r32 in res["fp32"][2].items():
        if float(r32.abs().max()) == 0:
            continue
        a = got[k].flatten()
        if a.numel() >= 8:
            cos = float(torch.nn.functional.cosine_similarity(a, r32.flatten(), dim=0))
            assert cos > 0.985, (k, cos)
        r16 = res["bf16"][2][k].flatten()
        assert float((a - r16).norm()) <= 4e-2 * float(r16.norm()) + 1e-9, k
    return job


def test_config2_single_modality_T1w():
    run_case([379], 10, "gpoe", 256, seed=11)


def test_config3_three_modalities_gpoe():
    run_case([379, 379, 379], 10, "gpoe", 256, seed=12)


def test_config4_early_fusion_1137_and_uca4():
    run_case([1137], 10, "poe", 256, seed=13)
    run_case([379, 379, 379, 1137], 10, "gpoe", 256, seed=14)


def test_config5_trunk_latent64_poe_ragged():
    # end-to-end model's trunk shape: Z = 64, three experts; ragged 83-row tail of the real HCPimage size
    run_case([379, 379, 379], 64, "poe", 83, seed=15)


def test_shape_limits_and_variants():
    """Kernel limits and less common variants: three hidden layers, the maximum hidden width (127) with the
    maximum latent (64), one hidden layer, linear (non_linear=False) stacks, an odd ROI count."""
    run_case([50, 61], 12, "mopoe", 100, seed=31, hidden=(64, 48, 32), c_dim=5)
    run_case([77], 64, "poe", 130, seed=32, hidden=(127,), c_dim=29)
    run_case([33, 45, 29], 7, "moe", 64, seed=33, hidden=(40, 24), c_dim=3, non_linear=False)
    run_case([129], 5, "gpoe", 257 - 1, seed=34, hidden=(16, 127, 8), c_dim=3)


def test_properties_full_size():
    """Size-independent properties at full size: (i) rows beyond n_rows never contribute; (ii) a zero
    learning rate leaves the parameters bit-identical; (iii) row order inside a batch does not change the
    loss; (iv) two identical jobs in one launch stay bit-identical over 8 fused steps."""
    g = torch.Generator().manual_seed(21)
    dims, Z, B = [379, 379, 379], 10, 200
    spec = nm.ModelSpec(dims, [110, 110], Z, 29)
    P = nm.ParamLayout(spec).init_reference_rule(3)
    xs = [torch.randn(B, d, generator=g) for d in dims]
    c = onehot(g, B)
    eps = torch.randn(B, Z, generator=g)

    def loss_of(xs_, c_, eps_, lr=1e-4, steps=0):
        job = nm.Job(spec, [nm.Table(x, c_, DEV) for x in xs_], combine="gpoe", state=P, lr=lr)
        job.set_eps(eps_)
        js = nm.JobSet([job])
        if steps:
            js.train(steps)
        else:
            js.grads(0, export=False)
        torch.cuda.synchronize()
        return job
    base = loss_of(xs, c, eps)
    # (iii) permutation of the rows (inputs, covariates and draws together)
    perm = torch.randperm(B, generator=g)
    pj = loss_of([x[perm] for x in xs], c[perm], eps[perm])
    assert abs(float(base.loss_log[0, 0]) - float(pj.loss_log[0, 0])) <= 2e-6 * abs(float(base.loss_log[0, 0]))
    assert float((base.grads - pj.grads).abs().max()) <= 2e-3 * float(base.grads.abs().max())
    # (ii) lr = 0
    z = loss_of(xs, c, eps, lr=0.0, steps=3)
    assert torch.equal(z.params.cpu(), nm.ParamLayout(spec).flatten(P))
    # (i) + (iv): two jobs, one launch, ragged table (600 rows = 256 + 256 + 88)
    x6 = [torch.randn(600, d, generator=g) for d in dims]
    c6 = onehot(g, 600)
    jobs = [nm.Job(spec, [nm.Table(x, c6, DEV) for x in x6], combine="gpoe", state=P, seed=5) for _ in range(2)]
    nm.JobSet(jobs).train(8)
    torch.cuda.synchronize()
    assert torch.equal(jobs[0].params.cpu(), jobs[1].params.cpu())
    assert torch.isfinite(jobs[0].loss_log[:8]).all()
