"""Pin the CPU oracle (oracle/cvae_ref.py) against golden vectors produced by the reference
itself (oracle/gen_golden.py imports /root/reference/cVAE.py in the build container).

fp32 on both sides, same torch build -> tolerances are a few ulps of the summed quantities.
"""
import numpy as np
import pytest
import torch

from oracle import cvae_ref as R
from tests.golden_util import GOLDEN as GOLDEN_DIR, Golden

MM_CASES = ["mm3_poe", "mm3_gpoe", "mm3_moe", "mm3_mopoe", "mm1_small", "mm4_uca_gpoe", "mm1_h1", "mm2_z64",
            "cfgA_T1w", "cfgA_T1w_tail83"]


def spec_of(g: Golden, kind="multimodal"):
    return R.Spec(input_dims=g.dims, hidden=g.hidden, latent=g.Z, c_dim=g.c_dim, non_linear=True, kind=kind)


@pytest.mark.parametrize("name", MM_CASES)
def test_state_dict_names_and_shapes(name):
    g = Golden(name)
    spec = spec_of(g)
    w = g.weights("w0")
    assert list(w.keys()) == R.param_names(spec)                       # reference state_dict() order
    shapes = R.param_shapes(spec)
    for k, v in w.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


@pytest.mark.parametrize("name", MM_CASES)
def test_forward_loss_grads(name):
    g = Golden(name)
    spec = spec_of(g)
    P = g.weights("w0")
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xes = g.xs(0)
    c = g.t("c")[0]
    eps = g.t("eps")[0]
    fwd = R.forward_multimodal(leaves, spec, xes, [c.long()] * g.M, g.combine, eps)
    loss = R.loss_multimodal(spec, xes, fwd)
    torch.testing.assert_close(fwd["mu"], g.t("mu"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(fwd["logvar"], g.t("logvar"), rtol=1e-5, atol=1e-6)
    for m in range(g.M):
        torch.testing.assert_close(fwd["locs"][m], g.t(f"loc{m}"), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(fwd["scales"][m], g.t(f"scale{m}")[:1], rtol=1e-6, atol=0)  # Normal broadcasts scale to [B, D]
    ref_loss = g.z["loss0"]
    assert abs(float(loss["total"]) - ref_loss[0]) <= 2e-6 * abs(ref_loss[0])
    assert abs(float(loss["kl"]) - ref_loss[1]) <= 2e-6 * abs(ref_loss[1]) + 1e-7
    assert abs(float(loss["ll"]) - ref_loss[2]) <= 2e-6 * abs(ref_loss[2])
    loss["total"].sum().backward()
    gg = g.grads("g0")
    for k, ref in gg.items():
        got = leaves[k].grad
        assert got is not None, k
        scale = float(ref.abs().max()) + 1e-12
        assert float((got - ref).abs().max()) <= 2e-5 * scale + 1e-7, k
    # tensors the reference left without grad (e.g. alpha when combine != gpoe) must be grad-free here too
    for k in leaves:
        if k not in gg:
            assert leaves[k].grad is None or float(leaves[k].grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("name", MM_CASES)
def test_adam_trajectory(name):
    g = Golden(name)
    spec = spec_of(g)
    P = g.weights("w0")
    # the reference optimizer owns every registered tensor (cVAE.py:1111-1116)
    opt = R.Adam(P, R.optimizer_param_names(spec), lr=1e-4)
    for s in range(g.n_steps):
        xes = g.xs(s)
        c = g.t("c")[s].long()
        loss, grads, _ = R.train_step(P, opt, spec, xes, [c] * g.M, g.combine, g.t("eps")[s])
        ref_loss = g.z[f"loss{s}"]
        assert abs(float(loss["total"]) - ref_loss[0]) <= 1e-5 * abs(ref_loss[0]), s
        tag = f"w{s + 1}"
        wref = g.weights(tag)
        if wref:
            for k, v in wref.items():
                # Adam moves each weight by <= lr per step; compare in units of lr
                assert float((P[k] - v).abs().max()) <= 2e-6, (k, s)
            mref, vref = g.adam(f"a{s + 1}")
            for k in mref:
                sc = float(mref[k].abs().max()) + 1e-12
                assert float((opt.m[k] - mref[k]).abs().max()) <= 1e-4 * sc + 1e-9, k
                sc = float(vref[k].abs().max()) + 1e-20
                assert float((opt.v[k] - vref[k]).abs().max()) <= 1e-4 * sc + 1e-12, k


def test_single_class_forward_loss():
    g = Golden("single_small")
    spec = spec_of(g, kind="single")
    P = g.weights("w0")
    assert list(P.keys()) == R.param_names(spec)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    x = g.t("x0")
    c = g.t("c").long()
    fwd = R.forward_multimodal(leaves, spec, [x], [c], "poe", g.t("eps"))
    loss = R.loss_multimodal(spec, [x], fwd)
    torch.testing.assert_close(fwd["mu"], g.t("mu"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(fwd["locs"][0], g.t("loc0"), rtol=1e-5, atol=1e-6)
    ref_loss = g.z["loss0"]
    assert abs(float(loss["total"]) - ref_loss[0]) <= 2e-6 * abs(ref_loss[0])
    loss["total"].sum().backward()
    for k, ref in g.grads("g0").items():
        sc = float(ref.abs().max()) + 1e-12
        assert float((leaves[k].grad - ref).abs().max()) <= 2e-5 * sc + 1e-7, k
    # pred_recon of class cVAE decodes mu (no draw), cVAE.py:547-553
    P1 = g.weights("w1")                     # the generator called pred_* after one optimizer step
    with torch.no_grad():
        mu, lv = R.encoder_fwd(P1, spec, 0, x, c)
        loc, _ = R.decoder_fwd(P1, spec, 0, mu, c)
    np.testing.assert_allclose(loc.numpy(), g.z["pred_recon"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mu.numpy(), g.z["pred_latent"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lv.exp().numpy(), g.z["pred_latent_var"], rtol=1e-5, atol=1e-6)


def test_deviation_passes():
    g = Golden("dev_small")
    spec = spec_of(g)
    P = g.weights("w0")
    xs = g.xs()
    c_raw = g.t("c_raw")
    for m in range(g.M):
        dev, loc = R.deviation_unimodal(P, spec, m, xs[m], c_raw, g.t("eps_uni")[m])
        np.testing.assert_allclose(loc.numpy(), g.z[f"uni_loc{m}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dev.numpy(), g.z[f"uni_dev{m}"], rtol=2e-5, atol=1e-6)
    preds = R.pred_recon(P, spec, xs, g.t("c_onehot").long(), g.combine, g.t("eps_joint"))
    for m in range(g.M):
        np.testing.assert_allclose(preds[m].numpy(), g.z[f"joint_pred{m}"], rtol=1e-5, atol=1e-6)
        d = R.reconstruction_deviation(xs[m], preds[m])
        np.testing.assert_allclose(d.numpy(), g.z[f"joint_dev{m}"], rtol=2e-5, atol=1e-7)


def test_committed_csv_known_answers():
    """G7: the reference's own committed CSV quintuple pins err_roi = (x - x_hat)^2 and
    err = mean_d err_roi (multimodal_kfold_test_cvae_supervised.py:141-149)."""
    z = np.load(GOLDEN_DIR / "csv_layouts.npz", allow_pickle=False)
    x = torch.from_numpy(z["adni_x"])
    xhat = torch.from_numpy(z["adni_xhat"])
    err_roi = R.reconstruction_deviation_roi(x, xhat).numpy()
    err = R.reconstruction_deviation(x, xhat).numpy()
    assert np.abs(err_roi - z["adni_err_roi"]).max() < 1e-6
    assert np.abs(err - z["adni_err"]).max() < 1e-7
    cols = list(z["adni_cols"])
    assert cols[:4] == ["participant_id", "DIA", "AGE", "PTGENDER"]
    ficols = list(z["adni_fi_cols"])
    assert ficols[4:] == [str(i) for i in range(1, len(cols) - 4 + 1)]
    assert list(z["adni_err_cols"]) == ["participant_id", "DIA", "AGE", "PTGENDER", "Reconstruction error"]


@pytest.mark.parametrize("name", ["dmvae3", "dmvae3_shared", "wdmvae3_shared", "mmvaeplus3_shared"])
def test_dm_family_oracle_matches_reference(name):
    """DMVAE / WeightedDMVAE / mmVAEPlus restated (oracle.dm_*) against the reference classes' own numbers: forward,
    losses, every gradient, and the 3-step Adam trajectory (cVAE.py:1491-1747, 1895-2002)."""
    g = Golden(name)
    spec = R.DmSpec(g.dims, g.hidden, g.Z, g.c_dim, str(g.z["cls"]))
    P = g.weights("w0")
    assert list(P.keys()) == R.dm_param_names(spec)
    opt = R.Adam(P, R.dm_param_names(spec))
    for s in range(g.n_steps):
        loss, grads, fwd = R.dm_train_step(P, opt, spec, g.xs(s), g.t("eps")[s])
        ref = g.t(f"loss{s}")
        for i, k in enumerate(("total", "kl", "ll")):
            assert abs(float(loss[k]) - float(ref[i])) <= 2e-6 * abs(float(ref[i])) + 1e-7, (s, k)
        if s == 0:
            if g.t("mu").numel():                       # (c_dim >= latent: no shared latent at all)
                assert float((fwd["mu_c"].detach() - g.t("mu")).abs().max()) <= 1e-6
            for m in range(g.M):
                assert float((fwd["x_recons"][m].detach() - g.t(f"loc{m}")).abs().max()) <= 1e-6
            for k, gr in g.grads("g0").items():
                assert float((grads[k] - gr).abs().max()) <= 2e-5 * float(gr.abs().max()) + 1e-9, k
    for k, w in g.weights(f"w{g.n_steps}").items():
        assert float((P[k] - w).abs().max()) <= 2e-6, k


@pytest.mark.parametrize("name", ["mvtcae3_poe", "mvtcae3_gpoe", "mvtcae3_mopoe"])
def test_mvtcae_oracle_matches_reference(name):
    """mvtCAE restated (oracle.mvt_*) against the reference class: the variance-as-logvar product of experts, the clamp,
    the total-correlation term, losses, gradients and the 3-step trajectory (cVAE.py:1754-1893)."""
    g = Golden(name)
    spec = R.Spec(g.dims, g.hidden, g.Z, g.c_dim)
    P = g.weights("w0")
    opt = R.Adam(P, R.param_names(spec))
    for s in range(g.n_steps):
        c = g.t("c")[s].long()
        loss, grads, fwd = R.mvt_train_step(P, opt, spec, g.xs(s), [c] * g.M, g.combine, g.t("eps")[s])
        ref = g.t(f"loss{s}")
        for i, k in enumerate(("total", "kl", "ll", "tc")):
            assert abs(float(loss[k]) - float(ref[i])) <= 5e-6 * abs(float(ref[i])) + 1e-7, (s, k, float(loss[k]), float(ref[i]))
        if s == 0:
            assert float((fwd["mu"].detach() - g.t("mu")).abs().max()) <= 1e-6
            assert float((fwd["logvar"].detach() - g.t("logvar")).abs().max()) <= 1e-5
            for k, gr in g.grads("g0").items():
                assert float((grads[k] - gr).abs().max()) <= 2e-5 * float(gr.abs().max()) + 1e-9, k
    for k, w in g.weights(f"w{g.n_steps}").items():
        assert float((P[k] - w).abs().max()) <= 2e-6, k
