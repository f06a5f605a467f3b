"""Oracle restatements of cVAE_multimodal_regression / cVAE_multimodal_endtoend against golden vectors
generated from the reference classes (oracle/gen_golden.py)."""
import numpy as np
import torch

from oracle import cvae_ref as R
from tests.golden_util import Golden


def _close(a, b, rel=2e-5):
    return abs(float(a) - float(b)) <= rel * abs(float(b)) + 1e-7


def test_regression_forward_loss_grads():
    g = Golden("reg3_gpoe")
    spec = R.Spec(g.dims, g.hidden, g.Z, g.c_dim, kind="regression")
    P = g.weights("w0")
    assert list(P.keys()) == R.param_names(spec)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xes, c = g.xs(0), g.t("c")[0]
    fwd = R.forward_regression(leaves, spec, xes, [c] * g.M, g.combine, g.t("eps")[0])
    loss = R.loss_regression(spec, xes, fwd, g.t("fi")[0])
    ref = g.z["loss0"]
    assert _close(loss["total"], ref[0]) and _close(loss["kl"], ref[1]) and _close(loss["ll"], ref[2]) and _close(loss["regression"], ref[3])
    torch.testing.assert_close(fwd["fi_pred"], g.t("fi_pred"), rtol=1e-5, atol=1e-6)
    loss["total"].sum().backward()
    for k, r in g.grads("g0").items():
        sc = float(r.abs().max()) + 1e-12
        assert float((leaves[k].grad - r).abs().max()) <= 3e-5 * sc + 1e-7, k


def test_endtoend_forward_loss_grads():
    g = Golden("e2e3")
    layers = [int(v) for v in g.z["layers"]]
    spec = R.Spec(g.dims, g.hidden, g.Z, g.c_dim, kind="endtoend", classifier_layers=layers)
    W = g.weights("w0")
    names = R.param_names(spec)
    assert [k for k in W if "running" not in k and "num_batches" not in k] == names
    leaves = {k: W[k].clone().requires_grad_(True) for k in names}
    xes, c = g.xs(0), g.t("c")[0]
    labels = g.t("labels")[0]
    margin, wc = (float(v) for v in g.z["margin_wc"])
    fwd = R.forward_endtoend(leaves, spec, xes, [c] * g.M, g.t("eps")[0], training=True)
    loss = R.loss_endtoend(spec, xes, fwd, labels, margin, wc)
    keys = ["total_loss", "recon_loss_health", "recon_loss_disease", "kl_loss", "classification_loss", "contrastive_loss"]
    for k, r in zip(keys, g.z["loss0"]):
        assert _close(loss[k], r), k
    torch.testing.assert_close(fwd["logits"], g.t("logits"), rtol=1e-4, atol=1e-5)
    loss["total_loss"].backward()
    for k, r in g.grads("g0").items():
        sc = float(r.abs().max()) + 1e-12
        assert float((leaves[k].grad - r).abs().max()) <= 5e-5 * sc + 1e-7, k


def test_mmjsd_is_poe_without_bypass():
    """mmJSD (cVAE.py:1354-1448): the oracle's cVAE_multimodal forward / loss with combine = 'poe' reproduces the
    reference class's losses, latent and gradients (its JSD term is identically zero)."""
    g = Golden("mmjsd3")
    spec = R.Spec(g.dims, g.hidden, g.Z, g.c_dim)
    P = g.weights("w0")
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    c = g.t("c")[0].long()
    fwd = R.forward_multimodal(leaves, spec, g.xs(0), [c] * g.M, "poe", g.t("eps")[0])
    loss = R.loss_multimodal(spec, g.xs(0), fwd)
    loss["total"].sum().backward()
    ref = g.z["loss0"]
    assert _close(loss["total"], ref[0]) and _close(loss["kl"], ref[1]) and _close(loss["ll"], ref[2])
    assert float((fwd["mu"] - g.t("mu")).abs().max()) < 1e-5
    gref = g.grads("g0")
    for k, v in leaves.items():
        r = gref.get(k)
        if k.startswith("alpha_m_list"):
            assert r is None or float(r.abs().max()) == 0.0
            continue
        assert float((v.grad - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-8, k
