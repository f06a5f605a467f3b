"""Row-split launch (nm_launch_rowsplit: k = 2 / 4 workgroups share one (model, modality), each owning 256 / k rows of the
batch; fp32 weight-gradient partials summed in slice order, Adam sweep split k ways) against

  * the whole-batch launch of the same kernels: the same bf16 operands and per-row arithmetic, so gradients agree to fp32
    summation order (not bit for bit: a different association of the same sums);
  * the oracle / the reference's golden numbers at the bounds of tests/test_gpu_parity.py;
  * itself: n steps in one launch == n launches of one step, bit for bit (a stale shadow image or partial across a
    hand-off would break it), and run to run bit for bit (fixed summation order).

Reference: the train step of cVAE.py:1166-1196 / multimodal_kfold_train_cvae_supervised.py:177-199."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import _lib
from oracle import cvae_ref as R
from tests.golden_util import Golden
from tests.hip_harness import DEV, make_job, swap_batch, oracle_step0

CASES = ["mm1_small", "mm1_h1", "mm3_poe", "mm3_gpoe", "mm3_moe", "mm3_mopoe", "mm4_uca_gpoe", "mm2_z64", "cfgA_T1w",
         "cfgA_T1w_tail83"]


def rel_l2(a, r):
    return float((a - r).norm()) / (float(r.norm()) + 1e-30)


def oracle_in_mode(g, mode):
    R.set_operand_rounding(mode)
    try:
        return oracle_step0(g)
    finally:
        R.set_operand_rounding("fp32")


@pytest.mark.parametrize("k", [2, 4])
@pytest.mark.parametrize("name", CASES)
def test_rowsplit_gradients_and_loss(name, k):
    """One forward + backward, k row slices per modality: every summed gradient against the whole-batch launch (fp32
    summation order only), against the bf16-operand oracle and the reference's fp32 gradient; the loss row against the
    reference's own numbers (1e-4 on the reconstruction loss).  The small goldens (19 rows) leave slices empty, the
    83-row tail leaves one slice ragged and two empty."""
    g = Golden(name)
    whole = make_job(g, 0)
    whole.enable_exports()
    nm.JobSet([whole]).grads(0, export=True)
    job = make_job(g, 0)
    job.enable_exports()
    js = nm.JobSet([job])
    js.grads(0, rowsplit=k)                                   # (NM_F_EXPORT on: every slice stores its rows of the exports)
    js.check_split_errors(block=True)
    torch.cuda.synchronize()
    # exports: per row the same arithmetic as the whole-batch launch -- reconstructions, squared errors, per-subject
    # deviations, joint posterior and the draw (the eager classes' backward runs row-split and reads them)
    B = g.B
    for m in range(g.M):
        assert torch.equal(job.out_loc[m][:B], whole.out_loc[m][:B]), m
        assert torch.equal(job.out_sqerr[m][:B], whole.out_sqerr[m][:B]), m
        assert torch.allclose(job.out_rowdev[m][:B], whole.out_rowdev[m][:B], rtol=2e-6, atol=1e-9), m
    assert torch.equal(job.out_mu[:B], whole.out_mu[:B]) and torch.equal(job.out_logvar[:B], whole.out_logvar[:B])
    assert torch.equal(job.out_z[:B], whole.out_z[:B])
    _, loss16, grads16 = oracle_in_mode(g, "bf16")
    _, _, grads32 = oracle_in_mode(g, "fp32")
    row, wrow = job.loss_log[0].cpu(), whole.loss_log[0].cpu()
    ref = g.z["loss0"]
    assert abs(float(row[2]) - ref[2]) <= 1e-4 * abs(ref[2]), ("ll", float(row[2]), ref[2])          # north star
    assert abs(float(row[0]) - ref[0]) <= 1e-4 * abs(ref[0]), ("total", float(row[0]), ref[0])
    assert abs(float(row[1]) - ref[1]) <= 5e-3 * abs(ref[1]) + 1e-5, ("kl", float(row[1]), ref[1])
    assert abs(float(row[2]) - float(loss16["ll"])) <= 2e-6 * abs(ref[2])
    for i in range(3 + g.M):                                  # total, kl, ll, ll_m: the whole-batch launch's, re-associated
        assert abs(float(row[i]) - float(wrow[i])) <= 2e-6 * abs(float(wrow[i])) + 1e-7, (i, float(row[i]), float(wrow[i]))
    got, wgot = job.grads_dict(), whole.grads_dict()
    for key in got:
        a, w = got[key].flatten(), wgot[key].flatten()
        r32, r16 = grads32[key].flatten(), grads16[key].flatten()
        if float(r32.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, key
            continue
        # same operands, same per-row arithmetic: only the order of the fp32 row sums differs
        assert float((a - w).abs().max()) <= 2e-5 * float(w.abs().max()) + 1e-9, (key, float((a - w).abs().max()), float(w.abs().max()))
        assert rel_l2(a, r16) < 3e-2, (key, "bf16-oracle", rel_l2(a, r16))
        assert rel_l2(a, r32) < 0.15, (key, "fp32", rel_l2(a, r32))
        if a.numel() >= 8:
            assert float(torch.nn.functional.cosine_similarity(a, r32, dim=0)) > 0.99, key


@pytest.mark.parametrize("k", [2, 4])
@pytest.mark.parametrize("name", ["mm3_gpoe", "mm4_uca_gpoe", "cfgA_T1w_tail83", "mm2_z64"])
def test_rowsplit_adam_trajectory(name, k):
    """Fused train steps (the partial sums, the sweep's Adam, the bf16 shadow images it rewrites) against the reference's
    own trajectory (golden) and against the whole-batch launch on the same batches."""
    g = Golden(name)
    job, whole = make_job(g, 0), make_job(g, 0)
    js, ws = nm.JobSet([job]), nm.JobSet([whole])
    lr = 1e-4
    for s in range(g.n_steps):
        if s > 0:
            swap_batch(job, g, s)
            swap_batch(whole, g, s)
        js.train(1, rowsplit=k)
        ws.train(1, rowsplit=1, split=False)
        js.check_split_errors(block=True)
        torch.cuda.synchronize()
        row = job.loss_log[0].cpu()
        ref = g.z[f"loss{s}"]
        assert abs(float(row[2]) - ref[2]) <= 1e-4 * abs(ref[2]), (s, float(row[2]), ref[2])
        d = (job.params - whole.params).abs()
        # Adam's first steps move a weight by ~lr sign(g): a gradient within summation-order noise of zero may flip
        assert float(d.max()) <= 2.0 * lr * (s + 1) + 1e-6
        assert int((d > 0.05 * lr).sum()) <= 0.005 * (s + 1) * d.numel() + 2, (s, int((d > 0.05 * lr).sum()), d.numel())
        wref = g.weights(f"w{s + 1}")
        if wref:
            sd = job.state_dict()
            n_tot = n_bad = 0
            for key, v in wref.items():
                dd = (sd[key] - v).abs()
                assert float(dd.max()) <= 2.0 * lr * (s + 1) + 1e-6, (key, s, float(dd.max()))
                n_tot += dd.numel()
                n_bad += int((dd > 0.25 * lr).sum())
            assert n_bad <= (0.03 + 0.01 * s) * n_tot + 2, (n_bad, n_tot)
            mref, vref = g.adam(f"a{s + 1}")
            m_hip, v_hip = job.adam_dicts()
            for key in mref:
                if float(mref[key].abs().max()) == 0.0 or mref[key].numel() < 8:
                    continue
                assert rel_l2(m_hip[key], mref[key]) < 0.15, key
                assert rel_l2(v_hip[key], vref[key]) < 0.30, key


def _table_job(g, n_rows, seed, n_eps=7):
    """A model with golden `g`'s shapes and weights on a seeded n_rows table (several batches, ragged tail)."""
    gen = torch.Generator().manual_seed(seed)
    xs = [torch.randn(n_rows, d, generator=gen) for d in g.dims]
    c = torch.zeros(n_rows, g.c_dim)
    c[torch.arange(n_rows), torch.randint(0, g.c_dim, (n_rows,), generator=gen)] = 1
    eps = torch.randn(n_eps, 256, g.Z, generator=gen)
    spec = nm.ModelSpec(g.dims, g.hidden, g.Z, g.c_dim)
    job = nm.Job(spec, [nm.Table(x, c, DEV) for x in xs], combine=g.combine, state=g.weights("w0"))
    job.set_eps(eps)
    return job


@pytest.mark.parametrize("k", [2, 4])
@pytest.mark.parametrize("name", ["mm1_small", "mm3_gpoe", "mm4_uca_gpoe"])
def test_rowsplit_one_launch_equals_stepwise_and_is_reproducible(name, k):
    """7 steps over a 600-row table (batches 256 / 256 / 88, wrapping) in ONE launch == 7 launches of one step == the same
    launch run again, bit for bit: parameters, moments, loss log.  Inside one launch the slices' partials, the shadow
    images the sweep rewrites and the expert statistics cross workgroups through the in-kernel hand-offs; between
    launches through the kernel boundary -- any stale read inside the launch shows as a difference."""
    g = Golden(name)
    res = []
    for mode in ("fused", "stepwise", "fused", "no helpers", "3 helpers"):
        job = _table_job(g, 600, seed=5)
        js = nm.JobSet([job])
        if mode == "stepwise":
            for _ in range(7):
                js.train(1, rowsplit=k)
        else:       # (default: every idle CU of the group's XCD helps with the Adam sweep -- same arithmetic per parameter)
            js.train(7, rowsplit=k, helpers={"no helpers": 0, "3 helpers": 3}.get(mode))
        js.check_split_errors(block=True)
        torch.cuda.synchronize()
        res.append((job.params.cpu().clone(), job.adam_m.cpu().clone(), job.adam_v.cpu().clone(), job.loss_log[:7].cpu().clone()))
    for other, what in ((res[1], "stepwise"), (res[2], "second run"), (res[3], "no helpers"), (res[4], "3 helpers")):
        for a, b, t in zip(res[0], other, ("params", "adam_m", "adam_v", "loss_log")):
            assert torch.equal(a, b), (what, t, float((a - b).abs().max()))
    # and the trajectory is the whole-batch launch's up to summation order
    whole = _table_job(g, 600, seed=5)
    nm.JobSet([whole]).train(7, rowsplit=1, split=False)
    torch.cuda.synchronize()
    d = (res[0][0] - whole.params.cpu()).abs()
    assert float(d.max()) <= 2.0 * 1e-4 * 7 + 1e-6
    assert int((d > 0.05 * 1e-4).sum()) <= 0.02 * d.numel() + 2
    assert torch.allclose(res[0][3][:, :3], whole.loss_log[:7, :3].cpu(), rtol=2e-4, atol=1e-5)


def test_rowsplit_many_models_pick_and_refusal():
    """Twenty 3-modality models (the reference's grid: 5 folds x 4 procedures) as 4 slices each = 240 workgroups: every
    model gets what it gets alone; JobSet picks k from the set size; a set that could not be resident is refused."""
    g = Golden("mm3_gpoe")
    alone = make_job(g, 0)
    alone.seed = 3
    alone.set_eps(None)
    nm.JobSet([alone]).train(3, rowsplit=4)
    jobs = [make_job(g, 0) for _ in range(20)]
    for i, j in enumerate(jobs):
        j.seed = i
        j.set_eps(None)                                   # in-kernel draw, keyed by (seed, step, row, z)
    js = nm.JobSet(jobs)
    assert js.rowsplit_k() == 4                           # 60 groups -> 64 x 4 = 256 workgroups
    js.train(3)
    js.check_split_errors(block=True)
    torch.cuda.synchronize()
    assert torch.equal(jobs[3].params.cpu(), alone.params.cpu())
    assert not torch.equal(jobs[3].params.cpu(), jobs[4].params.cpu())
    five = nm.JobSet([make_job(g, 0) for _ in range(5)])
    assert five.rowsplit_k() == 4 and five.rowsplit_helpers(4) == 12          # 15 groups -> 16 x (4 + 12) = 256 workgroups
    # five models with the idle CUs lent to their sweeps == the same five without helpers == a model alone, bit for bit
    for i, j in enumerate(five.jobs):
        j.seed = i
        j.set_eps(None)
    five.train(3)
    five.check_split_errors(block=True)
    bare = [make_job(g, 0) for _ in range(5)]
    for i, j in enumerate(bare):
        j.seed = i
        j.set_eps(None)
    nm.JobSet(bare).train(3, rowsplit=4, helpers=0)
    torch.cuda.synchronize()
    for a, b in zip(five.jobs, bare):
        assert torch.equal(a.params.cpu(), b.params.cpu()) and torch.equal(a.adam_v.cpu(), b.adam_v.cpu())
    assert torch.equal(five.jobs[3].params.cpu(), alone.params.cpu())
    assert nm.JobSet([make_job(g, 0) for _ in range(40)]).rowsplit_k() == 2     # 120 groups x 2 = 240
    big = nm.JobSet([make_job(g, 0) for _ in range(96)])
    assert big.rowsplit_k() == 1
    for j in big.jobs:
        j._ensure_rowsplit(2)
    ptr = big._upload(2)
    st = torch.cuda.current_stream().cuda_stream
    assert _lib.load().nm_launch_rowsplit(ptr, 96, 3, 2, 0, 0, 1, _lib.NM_F_BACKWARD | _lib.NM_F_ADAM, 0, st) == -16


def test_rowsplit_full_size_se_model_vs_oracle():
    """The metric's shape -- 3 x 379 ROI, batch 256, gPoE -- through 4 slices per modality: gradients and loss against
    the oracle (fp32: the 1e-4 bound on the reconstruction loss; bf16 operands: every gradient)."""
    gen = torch.Generator().manual_seed(12)
    dims, Z, c_dim, B = [379, 379, 379], 10, 29, 256
    spec = nm.ModelSpec(dims, [110, 110], Z, c_dim, True)
    P = nm.ParamLayout(spec).init_reference_rule(12)
    xs = [torch.randn(B, d, generator=gen) * 1.2 for d in dims]
    c = torch.zeros(B, c_dim)
    c[torch.arange(B), torch.randint(0, c_dim - 2, (B,), generator=gen)] = 1
    c[torch.arange(B), c_dim - 2 + torch.randint(0, 2, (B,), generator=gen)] = 1
    eps = torch.randn(B, Z, generator=gen)
    rs = R.Spec(dims, [110, 110], Z, c_dim, True)
    res = {}
    for mode in ("fp32", "bf16"):
        R.set_operand_rounding(mode)
        try:
            leaves = {k_: v.clone().requires_grad_(True) for k_, v in P.items()}
            fwd = R.forward_multimodal(leaves, rs, xs, [c.long()] * 3, "gpoe", eps)
            loss = R.loss_multimodal(rs, xs, fwd)
            loss["total"].sum().backward()
            res[mode] = (loss, {k_: v.grad for k_, v in leaves.items()})
        finally:
            R.set_operand_rounding("fp32")
    for k in (2, 4):
        job = nm.Job(spec, [nm.Table(x, c, DEV) for x in xs], combine="gpoe", state=P)
        job.set_eps(eps)
        js = nm.JobSet([job])
        js.grads(0, rowsplit=k)
        js.check_split_errors(block=True)
        torch.cuda.synchronize()
        row = job.loss_log[0].cpu()
        ll32 = float(res["fp32"][0]["ll"])
        assert abs(float(row[2]) - ll32) <= 1e-4 * abs(ll32), (k, float(row[2]), ll32)
        got = job.grads_dict()
        for key, r16 in res["bf16"][1].items():
            assert float((got[key] - r16).norm()) <= 4e-2 * float(r16.norm()) + 1e-9, (k, key)


def test_rowsplit_handoff_timeout_is_reported():
    """A workgroup that never arrives (NM_F_FAULT_INJECT): the others give up at their bounded spin, the job's error word
    is set and the host raises instead of handing out parameters."""
    g = Golden("mm3_gpoe")
    job = make_job(g, 0)
    js = nm.JobSet([job])
    before = job.params.cpu().clone()
    js._launch_rowsplit(2, 0, 1, _lib.NM_F_BACKWARD | _lib.NM_F_ADAM | _lib.NM_F_FAULT_INJECT)
    with pytest.raises(_lib.NmError):
        js.check_split_errors(block=True)
    torch.cuda.synchronize()
    assert torch.equal(job.params.cpu(), before)          # nothing computed from missing partials reached the parameters
