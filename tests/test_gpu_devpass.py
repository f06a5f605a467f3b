"""The compact deviation-pass kernel (nm_devpass: 128-row tiles, two workgroups per CU, dead 16-row tiles skipped) against
the general forward-only kernel (nm_forward) -- which tests/test_gpu_parity.py and tests/test_gpu_fullsize.py hold to the
oracle and to the reference's golden numbers: row by row the two run the same arithmetic with the same draws, so the
exports out_sqerr and out_loc must agree BIT FOR BIT at every shape the compact kernel admits (out_rowdev to fp32 summation
order: both kernels add a row's four column-group sums with LDS atomics); and directly against the oracle at one shape.  Reference: multimodal_kfold_train_cvae_supervised_regression.py:163-192."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import multi_modal_normative_modeling_amd as nm
from oracle import cvae_ref as R

DEV = "cuda:0"


def _job(N, D, hidden, Z, c_dim, seed, eps=None, non_linear=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, D, generator=g) * 1.1
    c = torch.zeros(N, c_dim)
    c[torch.arange(N), torch.randint(0, c_dim, (N,), generator=g)] = 1
    tab = nm.Table(x, c, DEV)
    spec = nm.ModelSpec([D], list(hidden), Z, c_dim, non_linear)
    job = nm.Job(spec, [tab], combine="poe", seed=seed, init_seed=seed, n_tiles_ws=tab.n_tiles)
    if eps is not None:
        job.set_eps(eps)
    job.enable_exports(loc=True, sqerr=True, rowdev=True, latent=False)
    return job, x, c


@pytest.mark.parametrize("N,D,hidden,Z,c_dim,inject", [
    (1064, 379, (110, 110), 10, 29, False),      # the benchmark's pass: 4 full 256-row tiles + a ragged 40-row one
    (1064, 379, (110, 110), 10, 29, True),       # ... with an injected draw
    (300, 1137, (110, 110), 10, 29, False),      # early fusion, 18 output chunks
    (129, 61, (112,), 32, 5, True),              # one hidden layer at the width limit, latent at the limit (Zs = 32)
    (517, 90, (64, 48, 32), 12, 3, False),       # three hidden layers, Z a multiple of 4
    (40, 116, (110, 110), 10, 2, True),          # fewer rows than one 128-row tile; raw float covariates' width
    (256, 379, (110, 110), 10, 29, False),       # exactly one 256-row tile
])
def test_devpass_equals_general_forward_bit_for_bit(N, D, hidden, Z, c_dim, inject):
    nt = (N + 255) // 256
    eps = torch.randn(nt, 256, Z, generator=torch.Generator().manual_seed(7)) if inject else None
    res = []
    for compact in (False, True):
        job, _, _ = _job(N, D, hidden, Z, c_dim, seed=11, eps=eps)
        js = nm.JobSet([job])
        assert js.devpass_ok()
        js.forward(loss=not compact)
        torch.cuda.synchronize()
        res.append((job.out_sqerr[0].cpu().clone(), job.out_rowdev[0].cpu().clone(), job.out_loc[0].cpu().clone()))
    for a, b, what in zip(res[0], res[1], ("out_sqerr", "out_rowdev", "out_loc")):
        if what == "out_rowdev":      # (both kernels add the row sums of their four column groups with LDS atomics: fp32 order)
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-9), (what, float((a - b).abs().max()))
        else:
            assert torch.equal(a, b), (what, float((a - b).abs().max()))
    assert float(res[1][0][:N].abs().max()) > 0
    assert float(res[1][0][N:].abs().max()) == 0.0 if res[1][0].shape[0] > N else True      # rows past the table: zeros


def test_devpass_many_models_share_a_table_and_match_the_oracle():
    """24 models over one 600-row table in one launch (the sweep's form), against the oracle on the same draws: the
    squared residuals within the distance between the fp32 and the bf16-operand oracle (the bound of the general kernel's
    own test), the per-subject means consistent with the matrix."""
    N, D, Z, cd = 600, 379, 10, 29
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, D, generator=g)
    c = torch.zeros(N, cd)
    c[torch.arange(N), torch.randint(0, cd, (N,), generator=g)] = 1
    tab = nm.Table(x, c, DEV)
    nt = tab.n_tiles
    eps = torch.randn(nt, 256, Z, generator=g)
    jobs = []
    for j in range(24):
        job = nm.Job(nm.ModelSpec([D], [110, 110], Z, cd), [tab], combine="poe", seed=j, init_seed=100 + j)
        job.set_eps(eps)
        job.enable_exports(loc=False, sqerr=True, rowdev=True, latent=False)
        jobs.append(job)
    js = nm.JobSet(jobs)
    js.forward(loss=False)
    torch.cuda.synchronize()
    rs = R.Spec([D], [110, 110], Z, cd)
    e = eps.reshape(-1, Z)[:N]
    for j in (0, 7, 23):
        P = jobs[j].state_dict()
        out = {}
        for mode in ("fp32", "bf16"):
            R.set_operand_rounding(mode)
            try:
                out[mode] = (x - R.forward_multimodal(P, rs, [x], [c.long()], "poe", e)["locs"][0].detach()) ** 2
            finally:
                R.set_operand_rounding("fp32")
        got = jobs[j].out_sqerr[0][:N].cpu()
        noise = float((out["bf16"] - out["fp32"]).abs().max())
        assert float((got - out["bf16"]).abs().max()) <= 1.5 * noise + 1e-5, j
        assert float((got - out["fp32"]).abs().max()) <= 3.0 * noise + 1e-5, j
        np.testing.assert_allclose(jobs[j].out_rowdev[0][:N].cpu().numpy(), got.sum(1).numpy() / D, rtol=2e-5, atol=1e-7)


def test_devpass_refuses_what_it_cannot_run():
    """Several experts, a wide first layer, latent exports: JobSet.forward falls back to the general kernel (same exports)."""
    job, _, _ = _job(200, 50, (120, 64), 8, 3, seed=5)                 # first hidden width 120 > 112
    assert not nm.JobSet([job]).devpass_ok()
    nm.JobSet([job]).forward(loss=False)                                 # runs on nm_forward
    torch.cuda.synchronize()
    assert float(job.out_sqerr[0][:200].abs().max()) > 0
    job2, _, _ = _job(200, 50, (64, 64), 8, 3, seed=5)
    job2.enable_exports(loc=True, sqerr=True, rowdev=True, latent=True)
    assert not nm.JobSet([job2]).devpass_ok()
