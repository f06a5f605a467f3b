"""Seeded random shapes against the oracle: every case draws the number of modalities, ROI counts, hidden
stack, latent width, covariate width, batch size, combiner and activation within the kernel's limits and runs
the full forward + ELBO + backward comparison of tests/test_gpu_fullsize.py::run_case (fp32 oracle at the
north-star bound on the reconstruction loss, bf16-operand oracle for latents / reconstructions / gradients).
Guards the shape-dependent paths: ragged batches, partial tiles, 1..3 hidden layers, 1..4 experts, chunked
output layers with short last chunks, every fusion rule."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_fullsize import run_case


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.integers(1, 5))
    L = int(rng.integers(1, 4))
    dims = [int(rng.integers(3, 421)) for _ in range(M)]
    hidden = [int(rng.integers(8, 128)) for _ in range(L)]
    c_dim = int(rng.integers(3, 30))
    Z = int(rng.integers(1, min(64, 127 - c_dim) + 1))
    B = int(rng.choice([1, 7, 19, 64, 83, 200, 255, 256]))
    combine = str(rng.choice(["poe", "gpoe", "moe", "mopoe"]))
    non_linear = bool(rng.integers(0, 2))
    return dims, Z, combine, B, hidden, c_dim, non_linear


@pytest.mark.parametrize("seed", range(24))
def test_random_shape_matches_oracle(seed):
    dims, Z, combine, B, hidden, c_dim, non_linear = _draw(seed)
    # the 1e-4 bound of the north star is stated for batch 256; the bf16 operand noise of the reconstruction loss
    # averages out over rows, so tiny batches get the bound scaled by sqrt(256 / B) (the comparison with the
    # bf16-operand oracle inside run_case stays at 5e-6 for every batch size)
    run_case(dims, Z, combine, B, seed=seed, hidden=tuple(hidden), c_dim=c_dim, non_linear=non_linear,
             ll32_tol=1e-4 * max(1.0, (256.0 / B) ** 0.5))


def _draw_wide(seed):
    """Shapes beyond the fused kernel's tile: up to 5 layers of up to 400 columns, latent up to 128."""
    rng = np.random.default_rng(5000 + seed)
    M = int(rng.integers(1, 4))
    L = int(rng.integers(1, 6))
    dims = [int(rng.integers(3, 300)) for _ in range(M)]
    hidden = [int(rng.integers(8, 400)) for _ in range(L)]
    c_dim = int(rng.integers(3, 30))
    Z = int(rng.integers(1, 129))
    if all(h <= 127 for h in hidden) and Z <= 64 and Z + c_dim <= 127:
        hidden[int(rng.integers(0, L))] = int(rng.integers(128, 400))       # make sure it IS a general-shape model
    B = int(rng.choice([1, 19, 83, 200, 256]))
    combine = str(rng.choice(["poe", "gpoe", "moe", "mopoe"]))
    return dims, Z, combine, B, hidden, c_dim, bool(rng.integers(0, 2))


@pytest.mark.parametrize("seed", range(10))
def test_random_wide_shape_matches_oracle(seed):
    """The same comparison on the general-shape path (nm_launch_wide): block boundaries at 128 columns in every position
    (partial last blocks, latent > 64, latent + c_dim > 127, deeper stacks)."""
    dims, Z, combine, B, hidden, c_dim, non_linear = _draw_wide(seed)
    job = run_case(dims, Z, combine, B, seed=seed, hidden=tuple(hidden), c_dim=c_dim, non_linear=non_linear,
                   ll32_tol=1e-4 * max(1.0, (256.0 / B) ** 0.5))
    assert job.spec.wide


@pytest.mark.parametrize("seed", range(100, 110))
def test_random_shape_fused_adam_steps(seed):
    """Three fused train steps (forward + ELBO + backward + Adam inside the kernel, one launch, the batch index
    walking over a ragged table) against the oracle's Adam trajectory with the kernel's operand rounding: almost
    every parameter within 5 % of one learning-rate step, none further than the steps taken."""
    import multi_modal_normative_modeling_amd as nm
    from oracle import cvae_ref as R
    from tests.test_gpu_fullsize import onehot
    dims, Z, combine, _, hidden, c_dim, non_linear = _draw(seed)
    rng = np.random.default_rng(seed)
    N = int(rng.choice([300, 512, 531, 700]))               # 2-3 batches per epoch, last one ragged for 300 / 531 / 700
    g = torch.Generator().manual_seed(seed)
    spec = nm.ModelSpec(list(dims), list(hidden), Z, c_dim, non_linear)
    P = nm.ParamLayout(spec).init_reference_rule(seed)
    xs = [torch.randn(N, d, generator=g) for d in dims]
    c = onehot(g, N, c_dim)
    n_steps, lr = 3, 1e-4
    eps = torch.randn(n_steps, 256, Z, generator=g)
    job = nm.Job(spec, [nm.Table(x, c, "cuda:0") for x in xs], combine=combine, state=P, lr=lr)
    job.set_eps(eps)
    nm.JobSet([job]).train(n_steps)
    torch.cuda.synchronize()
    rs = R.Spec(list(dims), list(hidden), Z, c_dim, non_linear)
    P16 = {k: v.clone() for k, v in P.items()}
    opt = R.Adam(P16, R.param_names(rs), lr=lr)
    nb = (N + 255) // 256
    R.set_operand_rounding("bf16")
    try:
        for s in range(n_steps):
            lo, hi = (s % nb) * 256, min(N, (s % nb + 1) * 256)
            R.train_step(P16, opt, rs, [x[lo:hi] for x in xs], [c[lo:hi].long()] * len(dims), combine, eps[s, : hi - lo])
    finally:
        R.set_operand_rounding("fp32")
    sd = job.state_dict()
    n_tot = sum(v.numel() for v in P16.values())
    n_off = sum(int(((sd[k] - P16[k]).abs() > 0.05 * lr).sum()) for k in P16)
    worst = max(float((sd[k] - P16[k]).abs().max()) for k in P16)
    assert worst <= 2.0 * lr * n_steps + 1e-6, worst
    assert n_off <= 0.02 * n_steps * n_tot + 2, (n_off, n_tot)


@pytest.mark.parametrize("seed", range(200, 206))
def test_random_shape_head_models(seed):
    """Regression and end-to-end models at random shapes (residual widths that straddle chunk and modality
    boundaries, 1-3 classifier blocks of random width up to 512, 2-4 classes, ragged batches): losses and every gradient
    against the oracle with the kernels' operand rounding."""
    import multi_modal_normative_modeling_amd as nm
    from oracle import cvae_ref as R
    rng = np.random.default_rng(seed)
    M = int(rng.integers(1, 4))
    dims = [int(rng.integers(5, 300)) for _ in range(M)]
    hidden = [int(rng.integers(8, 100)) for _ in range(int(rng.integers(1, 3)))]
    Z = int(rng.integers(2, 33))
    B = int(rng.choice([5, 64, 130, 256]))
    g = torch.Generator().manual_seed(seed)
    xes = [torch.randn(B, d, generator=g) for d in dims]
    eps = torch.randn(B, Z, generator=g)

    def check(got, P, skip=()):
        for k, v in P.items():
            if v.grad is None or k.endswith(skip) or float(v.grad.norm()) == 0.0:
                continue
            a, r = got[k].flatten().float(), v.grad.flatten()
            if a.numel() < 8:
                assert float((a - r).norm()) <= 0.2 * float(r.norm()) + 1e-6, k
                continue
            cos = float(torch.nn.functional.cosine_similarity(a, r, dim=0))
            assert cos > 0.98 and float((a - r).norm() / r.norm()) < 0.2, (k, cos)

    # ---- regression ----
    torch.manual_seed(seed)
    c2 = torch.rand(B, 2, generator=g) * 3
    fi = torch.randn(B, 1, generator=g) + 1.0
    reg = nm.cVAE_multimodal_regression(dims, hidden, Z, 2, modalities=M, non_linear=True)
    reg.to("cuda:0")
    sd0 = {k: v.clone() for k, v in reg.state_dict().items()}
    reg._eps_override = eps
    out = reg.forward_multimodal([x.to("cuda:0") for x in xes], [c2.to("cuda:0")] * M, "gpoe")
    lo = reg.loss_function_multimodal(xes, out, fi.to("cuda:0"), lambda_reg=0.5)
    reg.optimizer1.zero_grad()
    lo["total"].backward()
    got = {n: p.grad.detach().cpu() for n, p in reg.named_parameters() if p.grad is not None}
    spec = R.Spec(dims, hidden, Z, 2, True, kind="regression")
    P = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    R.set_operand_rounding("bf16")
    try:
        fw = R.forward_regression(P, spec, xes, [c2] * M, "gpoe", eps)
        ol = R.loss_regression(spec, xes, fw, fi, lambda_reg=0.5)
        ol["total"].backward()
    finally:
        R.set_operand_rounding("fp32")
    assert abs(float(lo["regression"]) - float(ol["regression"])) <= 1e-2 * float(ol["regression"]) + 1e-6
    assert abs(float(lo["total"]) - float(ol["total"])) <= 2e-4 * abs(float(ol["total"]))
    check(got, P)

    # ---- end-to-end ----
    n_layers = int(rng.integers(1, 4))
    layers = [int(rng.integers(4, 129)) for _ in range(n_layers)]
    if seed % 2:                                   # odd seeds: one block wider than 128 (the head in 128-column tiles, <= 512)
        layers[seed % n_layers] += 128 * (1 + seed % 3)
    ncls = int(rng.integers(2, 5))
    c7 = torch.rand(B, 7, generator=g)
    labels = torch.randint(0, ncls, (B,), generator=g)
    torch.manual_seed(seed + 1)
    e2e = nm.cVAE_multimodal_endtoend(dims, hidden, Z, 7, modalities=M, non_linear=True, classifier_layers=layers,
                                      dropout_rate=0.0, num_classes=ncls)
    e2e.to("cuda:0")
    e2e.train()
    sd0 = {k: v.clone() for k, v in e2e.state_dict().items()}
    e2e._eps_override = eps
    fwd = e2e.forward([x.to("cuda:0") for x in xes], [c7.to("cuda:0")] * M)
    # the hinge is defined for two classes (labels 0 / 1 weight the two branches): keep it out for ncls > 2
    wc = 0.6 if ncls == 2 else 0.0
    le = e2e.loss_function(xes, fwd, labels.to("cuda:0"), margin=0.4, weightcontrastive=wc)
    e2e.optimizer.zero_grad()
    le["total_loss"].backward()
    got = {n: p.grad.detach().cpu() for n, p in e2e.named_parameters() if p.grad is not None}
    spec = R.Spec(dims, hidden, Z, 7, True, kind="endtoend", classifier_layers=layers)
    P = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd0.items()}
    R.set_operand_rounding("bf16")
    try:
        of = R.forward_endtoend(P, spec, xes, [c7] * M, eps, training=True)
        oe = R.loss_endtoend(spec, xes, of, labels, margin=0.4, weightcontrastive=wc)
        oe["total_loss"].backward()
    finally:
        R.set_operand_rounding("fp32")
    assert abs(float(le["classification_loss"]) - float(oe["classification_loss"])) <= 2e-2 * float(oe["classification_loss"]) + 1e-5
    assert abs(float(le["total_loss"]) - float(oe["total_loss"])) <= 5e-3 * abs(float(oe["total_loss"]))
    check(got, P, skip=tuple(f"classifier.{4 * i}.bias" for i in range(n_layers)))


def test_distinct_covariates_per_modality():
    """The API allows a different covariate matrix per modality (cs[i], cVAE.py:1174): then every decoder builds
    its own z | c | 1 input (no reuse of the first decoder's copy); same covariates -> shared copy.  Both paths
    against the oracle."""
    import multi_modal_normative_modeling_amd as nm
    from oracle import cvae_ref as R
    from tests.test_gpu_fullsize import onehot
    dims, hidden, Z, c_dim, B = [40, 55, 33], [32, 24], 6, 5, 100
    g = torch.Generator().manual_seed(77)
    spec = nm.ModelSpec(dims, hidden, Z, c_dim, True)
    P = nm.ParamLayout(spec).init_reference_rule(7)
    xs = [torch.randn(B, d, generator=g) for d in dims]
    eps = torch.randn(B, Z, generator=g)
    rs = R.Spec(dims, hidden, Z, c_dim, True)
    for shared in (False, True):
        cs = [onehot(g, B, c_dim) for _ in dims]
        if shared:
            cs = [cs[0]] * 3
        job = nm.Job(spec, [nm.Table(x, c, "cuda:0") for x, c in zip(xs, cs)], combine="gpoe", state=P)
        assert job.struct().shared_cov == (1 if shared else 0)
        job.set_eps(eps)
        job.enable_exports(sqerr=False, rowdev=False)
        nm.JobSet([job]).grads(0)
        torch.cuda.synchronize()
        R.set_operand_rounding("bf16")
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            fwd = R.forward_multimodal(leaves, rs, xs, [c.long() for c in cs], "gpoe", eps)
            loss = R.loss_multimodal(rs, xs, fwd)
            loss["total"].sum().backward()
        finally:
            R.set_operand_rounding("fp32")
        assert abs(float(job.loss_log[0, 2]) - float(loss["ll"])) <= 5e-6 * abs(float(loss["ll"]))
        for m in range(3):
            l16 = fwd["locs"][m].detach()
            assert float((job.out_loc[m][:B].cpu() - l16).abs().max()) <= 3e-3 * float(l16.abs().max()), (shared, m)
        got = job.grads_dict()
        for k, v in leaves.items():
            if v.grad is not None and float(v.grad.norm()) > 0:
                assert float((got[k].flatten() - v.grad.flatten()).norm()) <= 4e-2 * float(v.grad.norm()) + 1e-9, (shared, k)
