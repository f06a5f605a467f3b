#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (build container only).

Imports ``/root/reference/cVAE.py`` (read-only mount, never copied), drives its classes on
seeded synthetic inputs with an explicit reparameterisation draw, and writes
``tests/golden/*.npz``.  The fixtures hold data only: inputs, weights keyed by the
reference's ``state_dict`` names, and expected outputs.  The reference itself never travels
to the GPU box; tests compare oracle <-> golden here and HIP <-> oracle/golden there.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
from contextlib import contextmanager
from pathlib import Path

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference"
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _import_reference():
    sys.path.insert(0, REF)
    import cVAE as ref  # noqa
    return ref


@contextmanager
def fixed_eps(eps_list):
    """Make every ``torch.randn_like`` call inside the reference return the next preset draw."""
    it = iter(eps_list)
    orig = torch.randn_like

    def fake(t, *a, **k):
        e = next(it)
        assert e.shape == t.shape, (e.shape, t.shape)
        return e.clone()

    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig


def onehot_cov(g, B, c_dim):
    """27 age bins + 2 sex bins when c_dim == 29, else split c_dim-2 / 2 the same way."""
    na = c_dim - 2
    a = torch.randint(0, na, (B,), generator=g)
    s = torch.randint(0, 2, (B,), generator=g)
    c = torch.zeros(B, c_dim)
    c[torch.arange(B), a] = 1
    c[torch.arange(B), na + s] = 1
    return c


def sd_np(model, prefix="w:"):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def grads_np(model, prefix="g:"):
    out = {}
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[prefix + k] = p.grad.detach().cpu().numpy().copy()
    return out


def adam_np(opt, model, prefix):
    out = {}
    name_of = {id(p): k for k, p in model.named_parameters()}
    for p, st in opt.state.items():
        k = name_of[id(p)]
        out[f"{prefix}m:{k}"] = st["exp_avg"].detach().numpy().copy()
        out[f"{prefix}v:{k}"] = st["exp_avg_sq"].detach().numpy().copy()
    return out


def case_multimodal(ref, name, dims, c_dim, hidden, Z, B, combine, n_steps, seed, int_cov=True, store_steps=(1,),
                    cls="cVAE_multimodal"):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    M = len(dims)
    model = getattr(ref, cls)(input_dim_list=list(dims), hidden_dim=list(hidden), latent_dim=Z, c_dim=c_dim,
                                learning_rate=1e-4, modalities=M, non_linear=True)
    out = {"meta": np.array([M, c_dim, Z, B, n_steps]), "dims": np.array(dims), "hidden": np.array(hidden),
           "combine": np.array(combine)}
    out.update(sd_np(model, "w0:"))
    xs = [torch.randn(n_steps, B, d, generator=g) * 1.3 + 0.1 for d in dims]
    c = torch.stack([onehot_cov(g, B, c_dim) for _ in range(n_steps)])
    eps = torch.randn(n_steps, B, Z, generator=g)
    for m in range(M):
        out[f"x{m}"] = xs[m].numpy()
    out["c"] = c.numpy()
    out["eps"] = eps.numpy()
    for s in range(n_steps):
        xes = [xs[m][s] for m in range(M)]
        cc = c[s].long() if int_cov else c[s]        # MyDataset_labels casts one-hot to int64 (utils_vae.py:24)
        cs = [cc for _ in range(M)]
        with fixed_eps([eps[s]]):
            fwd = model.forward_multimodal(xes, cs, combine)
        loss = model.loss_function_multimodal(xes, fwd)
        model.optimizer1.zero_grad()
        loss["total"].backward()
        if s == 0:
            out["mu"] = fwd["mu_multimodal"].detach().numpy().copy()
            out["logvar"] = fwd["logvar_multimodal"].detach().numpy().copy()
            for m in range(M):
                out[f"loc{m}"] = fwd["x_recons"][m].loc.detach().numpy().copy()
                out[f"scale{m}"] = fwd["x_recons"][m].scale.detach().numpy().copy()
            out.update(grads_np(model, "g0:"))
        out[f"loss{s}"] = np.array([float(loss["total"]), float(loss["kl"]), float(loss["ll"])], dtype=np.float64)
        model.optimizer1.step()
        if (s + 1) in store_steps:
            out.update(sd_np(model, f"w{s + 1}:"))
            out.update(adam_np(model.optimizer1, model, f"a{s + 1}:"))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, {k: float(out[k][0]) for k in out if k.startswith("loss")})


def case_dm(ref, name, cls, dims, c_dim, hidden, Z, B, n_steps, seed, store_steps=(1,)):
    """DMVAE / WeightedDMVAE / mmVAEPlus (cVAE.py:1491-1747, 1895-2002): covariate-free ReLU encoders / sigmoid decoders,
    the first min(c_dim, Z) latent columns private, the rest fused by ProductOfExperts2 and sampled; inputs in [0, 1]."""
    import contextlib, io
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    M = len(dims)
    model = getattr(ref, cls)(input_dim_list=list(dims), hidden_dim=list(hidden), latent_dim=Z, c_dim=c_dim,
                              learning_rate=1e-4, modalities=M, non_linear=True)
    S = min(c_dim, Z)
    Zc = Z - S
    out = {"meta": np.array([M, c_dim, Z, B, n_steps]), "dims": np.array(dims), "hidden": np.array(hidden),
           "combine": np.array("poe"), "cls": np.array(cls)}
    out.update(sd_np(model, "w0:"))
    xs = [torch.rand(n_steps, B, d, generator=g) for d in dims]
    eps = torch.randn(n_steps, B, Z, generator=g)          # the first Zc columns are the draw of the shared latent
    for m in range(M):
        out[f"x{m}"] = xs[m].numpy()
    out["eps"] = eps.numpy()
    out["c"] = np.zeros((n_steps, B, 0), dtype=np.float32)
    for s in range(n_steps):
        xes = [xs[m][s] for m in range(M)]
        with fixed_eps([eps[s][:, :Zc]]):
            fwd = model.forward_multimodal(xes, None, "poe")
        with contextlib.redirect_stdout(io.StringIO()):     # WeightedDMVAE prints per step (cVAE.py:1702)
            loss = model.loss_function_multimodal(xes, fwd)
        model.optimizer1.zero_grad()
        loss["total"].backward()
        if s == 0:
            out["mu"] = fwd["mu_c"].detach().numpy().copy()
            out["logvar"] = fwd["logvar_c"].detach().numpy().copy()
            for m in range(M):
                out[f"loc{m}"] = fwd["x_recons"][m].detach().numpy().copy()
            out.update(grads_np(model, "g0:"))
        out[f"loss{s}"] = np.array([float(loss["total"]), float(loss["kl"]), float(loss["ll"])], dtype=np.float64)
        model.optimizer1.step()
        if (s + 1) in store_steps:
            out.update(sd_np(model, f"w{s + 1}:"))
            out.update(adam_np(model.optimizer1, model, f"a{s + 1}:"))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, {k: [float(v) for v in out[k]] for k in out if k.startswith("loss")}, list(sd_np(model, "").keys())[:3])


def case_mvt(ref, name, dims, c_dim, hidden, Z, B, combine, n_steps, seed, store_steps=(1,)):
    """mvtCAE (cVAE.py:1754-1893): case_multimodal with the tc term logged as a fourth loss entry."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    M = len(dims)
    model = ref.mvtCAE(input_dim_list=list(dims), hidden_dim=list(hidden), latent_dim=Z, c_dim=c_dim, learning_rate=1e-4,
                       modalities=M, non_linear=True)
    out = {"meta": np.array([M, c_dim, Z, B, n_steps]), "dims": np.array(dims), "hidden": np.array(hidden),
           "combine": np.array(combine)}
    out.update(sd_np(model, "w0:"))
    xs = [torch.randn(n_steps, B, d, generator=g) * 1.3 + 0.1 for d in dims]
    c = torch.stack([onehot_cov(g, B, c_dim) for _ in range(n_steps)])
    eps = torch.randn(n_steps, B, Z, generator=g)
    for m in range(M):
        out[f"x{m}"] = xs[m].numpy()
    out["c"], out["eps"] = c.numpy(), eps.numpy()
    for s in range(n_steps):
        xes = [xs[m][s] for m in range(M)]
        cs = [c[s].long() for _ in range(M)]
        with fixed_eps([eps[s]]):
            fwd = model.forward_multimodal(xes, cs, combine)
        loss = model.loss_function_multimodal(xes, fwd)
        model.optimizer1.zero_grad()
        loss["total"].backward()
        if s == 0:
            out["mu"] = fwd["mu_multimodal"].detach().numpy().copy()
            out["logvar"] = fwd["logvar_multimodal"].detach().numpy().copy()
            for m in range(M):
                out[f"loc{m}"] = fwd["x_recons"][m].loc.detach().numpy().copy()
            out.update(grads_np(model, "g0:"))
        out[f"loss{s}"] = np.array([float(loss["total"]), float(loss["kl"]), float(loss["ll"]), float(loss["tc"])], dtype=np.float64)
        model.optimizer1.step()
        if (s + 1) in store_steps:
            out.update(sd_np(model, f"w{s + 1}:"))
            out.update(adam_np(model.optimizer1, model, f"a{s + 1}:"))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, {k: [float(v) for v in out[k]] for k in out if k.startswith("loss")})


def case_single(ref, name, D, c_dim, hidden, Z, B, seed):
    """class cVAE (cVAE.py:391-562): forward + loss_function + pred_recon/pred_latent."""
    import pandas as pd
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = ref.cVAE(input_dim=D, hidden_dim=list(hidden), latent_dim=Z, c_dim=c_dim, learning_rate=1e-4,
                     non_linear=True)
    x = torch.randn(B, D, generator=g)
    c = onehot_cov(g, B, c_dim)
    eps = torch.randn(B, Z, generator=g)
    out = {"meta": np.array([1, c_dim, Z, B, 1]), "dims": np.array([D]), "hidden": np.array(hidden)}
    out.update({k: v for k, v in sd_np(model, "w0:").items() if "discriminator" not in k})
    with fixed_eps([eps]):
        fwd = model.forward(x, c.long())
    loss = model.loss_function(x, fwd)
    model.optimizer1.zero_grad()
    loss["total"].backward()
    out.update({k: v for k, v in grads_np(model, "g0:").items() if "discriminator" not in k})
    out.update({"x0": x.numpy(), "c": c.numpy(), "eps": eps.numpy(),
                "mu": fwd["mu"].detach().numpy(), "logvar": fwd["logvar"].detach().numpy(),
                "loc0": fwd["x_recon"].loc.detach().numpy(), "scale0": fwd["x_recon"].scale.detach().numpy(),
                "loss0": np.array([float(loss["total"]), float(loss["kl"]), float(loss["ll"])])})
    model.optimizer1.step()
    out.update({k: v for k, v in sd_np(model, "w1:").items() if "discriminator" not in k})
    # pred_recon uses mu (no draw) for the single-modality class, cVAE.py:547-553
    xdf = pd.DataFrame(x.numpy())
    out["pred_recon"] = model.pred_recon(xdf, c.long().numpy(), torch.device("cpu"))
    lat, latvar = model.pred_latent(xdf, c.long().numpy(), torch.device("cpu"))
    out["pred_latent"] = lat
    out["pred_latent_var"] = latvar
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name)


def case_deviation(ref, name, dims, c_dim, hidden, Z, N, combine, seed):
    """(i) regression-script unimodal (x - x_hat)^2 with raw float covariates
    (multimodal_kfold_train_cvae_supervised_regression.py:183-188);
    (ii) pred_recon + reconstruction_deviation_multimodal (cVAE.py:1198-1211)."""
    import pandas as pd
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    M = len(dims)
    model = ref.cVAE_multimodal(input_dim_list=list(dims), hidden_dim=list(hidden), latent_dim=Z, c_dim=c_dim,
                                learning_rate=1e-4, modalities=M, non_linear=True)
    out = {"meta": np.array([M, c_dim, Z, N, 0]), "dims": np.array(dims), "hidden": np.array(hidden),
           "combine": np.array(combine)}
    out.update(sd_np(model, "w0:"))
    xs = [torch.randn(N, d, generator=g) for d in dims]
    c_onehot = onehot_cov(g, N, c_dim)
    c_raw = torch.rand(N, c_dim, generator=g) * 3.0           # raw-float covariates, as in the regression script
    eps_uni = torch.randn(M, N, Z, generator=g)
    eps_joint = torch.randn(N, Z, generator=g)
    for m in range(M):
        out[f"x{m}"] = xs[m].numpy()
        with torch.no_grad(), fixed_eps([eps_uni[m]]):
            mu, logvar = model.encode(xs[m], c_raw, m)
            z = model.reparameterise(mu, logvar)
            loc = model.decode(z, c_raw, m).loc
            out[f"uni_dev{m}"] = ((xs[m] - loc) ** 2).numpy()
            out[f"uni_loc{m}"] = loc.numpy()
    dfs = [pd.DataFrame(x.numpy()) for x in xs]
    with fixed_eps([eps_joint]):
        preds = model.pred_recon(dfs, c_onehot.numpy(), torch.device("cpu"), combine)
    devs = model.reconstruction_deviation_multimodal([d.values for d in dfs], preds)
    for m in range(M):
        out[f"joint_pred{m}"] = preds[m]
        out[f"joint_dev{m}"] = np.asarray(devs[m])
    out.update({"c_onehot": c_onehot.numpy(), "c_raw": c_raw.numpy(), "eps_uni": eps_uni.numpy(),
                "eps_joint": eps_joint.numpy()})
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name)


def case_regression(ref, name, dims, c_dim, hidden, Z, B, combine, n_steps, seed):
    """cVAE_multimodal_regression (cVAE.py:2211-2346): trunk + regressor on concatenated residuals."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    M = len(dims)
    model = ref.cVAE_multimodal_regression(input_dim_list=list(dims), hidden_dim=list(hidden), latent_dim=Z, c_dim=c_dim,
                                           learning_rate=1e-4, modalities=M, non_linear=True)
    out = {"meta": np.array([M, c_dim, Z, B, n_steps]), "dims": np.array(dims), "hidden": np.array(hidden),
           "combine": np.array(combine)}
    out.update(sd_np(model, "w0:"))
    xs = [torch.randn(n_steps, B, d, generator=g) for d in dims]
    c = torch.rand(n_steps, B, c_dim, generator=g) * 2.0            # raw float covariates (AGE, PTGENDER), :107
    fi = torch.randn(n_steps, B, 1, generator=g) * 0.5 + 1.0
    eps = torch.randn(n_steps, B, Z, generator=g)
    for m in range(M):
        out[f"x{m}"] = xs[m].numpy()
    out.update({"c": c.numpy(), "eps": eps.numpy(), "fi": fi.numpy()})
    for s_ in range(n_steps):
        xes = [xs[m][s_] for m in range(M)]
        with fixed_eps([eps[s_]]):
            fwd = model.forward_multimodal(xes, [c[s_]] * M, combine)
        loss = model.loss_function_multimodal(xes, fwd, fi[s_], lambda_reg=1.0)
        model.optimizer1.zero_grad()
        loss["total"].backward()
        if s_ == 0:
            out["fi_pred"] = fwd["fi_pred"].detach().numpy().copy()
            out["mu"] = fwd["mu_multimodal"].detach().numpy().copy()
            for m in range(M):
                out[f"loc{m}"] = fwd["x_recons"][m].loc.detach().numpy().copy()
            out.update(grads_np(model, "g0:"))
        out[f"loss{s_}"] = np.array([float(loss["total"]), float(loss["kl"]), float(loss["ll"]), float(loss["regression"])])
        model.optimizer1.step()
    out.update(sd_np(model, f"w{n_steps}:"))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, out["loss0"])


def case_endtoend(ref, name, dims, c_dim, hidden, Z, B, layers, n_steps, seed, margin=1.0, wc=1.0):
    """cVAE_multimodal_endtoend (cVAE.py:2021-2207), classifier in train() mode with dropout_rate = 0
    (BatchNorm batch statistics; Dropout(0.5) masks are RNG-bound and excluded from parity)."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    M = len(dims)
    model = ref.cVAE_multimodal_endtoend(input_dim_list=list(dims), hidden_dim=list(hidden), latent_dim=Z, c_dim=c_dim,
                                         modalities=M, non_linear=True, classifier_layers=list(layers), dropout_rate=0.0,
                                         num_classes=2)
    model.train()
    out = {"meta": np.array([M, c_dim, Z, B, n_steps]), "dims": np.array(dims), "hidden": np.array(hidden),
           "layers": np.array(layers), "margin_wc": np.array([margin, wc])}
    out.update(sd_np(model, "w0:"))
    xs = [torch.randn(n_steps, B, d, generator=g) for d in dims]
    c = torch.stack([onehot_cov(g, B, c_dim) for _ in range(n_steps)])
    labels = torch.randint(0, 2, (n_steps, B), generator=g)
    eps = torch.randn(n_steps, B, Z, generator=g)
    for m in range(M):
        out[f"x{m}"] = xs[m].numpy()
    out.update({"c": c.numpy(), "eps": eps.numpy(), "labels": labels.numpy()})
    keys = ["total_loss", "recon_loss_health", "recon_loss_disease", "kl_loss", "classification_loss", "contrastive_loss"]
    for s_ in range(n_steps):
        xes = [xs[m][s_] for m in range(M)]
        with fixed_eps([eps[s_]]):
            fwd = model.forward(xes, [c[s_]] * M)
        loss = model.loss_function(xes, fwd, labels[s_], margin, wc)
        model.optimizer.zero_grad()
        loss["total_loss"].backward()
        if s_ == 0:
            out["logits"] = fwd["logits"].detach().numpy().copy()
            out["mu"] = fwd["mu"].detach().numpy().copy()
            out["logvar"] = fwd["logvar"].detach().numpy().copy()
            for m in range(M):
                out[f"loc_h{m}"] = fwd["x_recons_health"][m].loc.detach().numpy().copy()
                out[f"loc_d{m}"] = fwd["x_recons_disease"][m].loc.detach().numpy().copy()
            out.update(grads_np(model, "g0:"))
        out[f"loss{s_}"] = np.array([float(loss[k]) for k in keys])
        model.optimizer.step()
    out.update(sd_np(model, f"w{n_steps}:"))
    model.eval()
    out["predict"] = model.predict([xs[m][0] for m in range(M)], [c[0]] * M).numpy()
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, out["loss0"])


def case_fusion_ops(ref, name, M, B, Z, seed):
    """The public expert-fusion methods of the hot-path classes on fixed inputs: cVAE_multimodal.combine_latent
    (cVAE.py:1144-1164) for the four combiners and the M = 1 bypass, product_of_experts / mixture_of_experts /
    mixture_of_product_of_experts (:1118-1126), the same on cVAE_multimodal_regression (:2265-2307), mvtCAE's
    (ProductOfExperts2 on what it is given, the 1e-6 clamp, total_correlation :1782-1869) and mmJSD.combine_latent (:1399-1402)."""
    g = torch.Generator().manual_seed(seed)
    mus = torch.randn(M, B, Z, generator=g)
    variances = torch.exp(0.7 * torch.randn(M, B, Z, generator=g))
    alpha = torch.randn(M, generator=g)
    out = {"meta": np.array([M, 0, Z, B, 0]), "mus": mus.numpy(), "variances": variances.numpy(), "alpha": alpha.numpy()}
    dims = [5] * M
    for cname in ("cVAE_multimodal", "cVAE_multimodal_regression", "mvtCAE"):
        model = getattr(ref, cname)(dims, [8, 8], Z, 2, modalities=M)
        with torch.no_grad():
            for m in range(M):
                model.alpha_m_list[m].copy_(alpha[m:m + 1])
            for comb in ("poe", "gpoe", "moe", "mopoe"):
                mu, var = model.combine_latent(mus, variances, comb)
                out[f"{cname}.combine_latent.{comb}.mu"], out[f"{cname}.combine_latent.{comb}.var"] = mu.numpy(), var.numpy()
            for meth in ("product_of_experts", "mixture_of_experts", "mixture_of_product_of_experts"):
                mu, var = getattr(model, meth)(mus, variances)
                out[f"{cname}.{meth}.mu"], out[f"{cname}.{meth}.var"] = mu.numpy(), var.numpy()
            if cname != "mvtCAE":
                mu, var = model.combine_latent(mus[:1], variances[:1], "gpoe")          # single-expert bypass
                out[f"{cname}.combine_latent.single.mu"], out[f"{cname}.combine_latent.single.var"] = mu.numpy(), var.numpy()
            else:
                small = 1e-8 * variances                                                 # the clamp at 1e-6
                mu, var = model.combine_latent(mus, small, "moe")
                out["mvtCAE.combine_latent.clamped.mu"], out["mvtCAE.combine_latent.clamped.var"] = mu.numpy(), var.numpy()
                tc = model.total_correlation(mus, mus.mean(0))
                out["mvtCAE.total_correlation"] = np.asarray(float(tc), dtype=np.float32)
    jsd = ref.mmJSD(dims, [8, 8], Z, 2, modalities=M)
    with torch.no_grad():
        mu, var = jsd.combine_latent(mus, torch.log(variances))
    out["mmJSD.combine_latent.mu"], out["mmJSD.combine_latent.var"] = mu.numpy(), var.numpy()
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"[golden] {name}: {len(out)} arrays")


def case_csv_headers(name):
    """G7/G8: layout facts of the committed CSV artefacts (first line + IID column only, plus a
    known-answer slice of one (x, x_hat, err_roi, err) quintuple)."""
    import pandas as pd
    out = {}
    p = Path(REF) / "regression_outputs" / "deviation_fold_0_T1w_sMRI_roiwise.csv"
    df = pd.read_csv(p)
    out["roiwise_header"] = np.array(list(df.columns))
    out["roiwise_iid"] = df["IID"].to_numpy()
    with open(p) as f:
        f.readline()
        out["roiwise_row0_text"] = np.array(f.readline().strip())
    out["roiwise_row0_vals"] = df.iloc[0, 1:].to_numpy(dtype=np.float32)
    base = Path(REF) / "deviation" / "supervised_cvae" / "ADNI" / "UCA-gPoE" / "av45"
    nrm = pd.read_csv(base / "normalized_av45.csv")
    rec = pd.read_csv(base / "reconstruction_av45.csv")
    err_roi = pd.read_csv(base / "reconstruction_error_roi_av45.csv")
    err = pd.read_csv(base / "reconstruction_error_av45.csv")
    fi = pd.read_csv(base / "deviation_as_feature_importance_av45.csv")
    n = 64
    meta_cols = ["participant_id", "DIA", "AGE", "PTGENDER"]
    roi_cols = [c for c in nrm.columns if c not in meta_cols]
    out["adni_cols"] = np.array(list(nrm.columns))
    out["adni_fi_cols"] = np.array(list(fi.columns))
    out["adni_err_cols"] = np.array(list(err.columns))
    out["adni_x"] = nrm[roi_cols].to_numpy()[:n]
    out["adni_xhat"] = rec[roi_cols].to_numpy()[:n]
    out["adni_err_roi"] = err_roi[roi_cols].to_numpy()[:n]
    out["adni_err"] = err["Reconstruction error"].to_numpy()[:n]
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    ref = _import_reference()
    only = set(sys.argv[1:])          # optional: names of the cases to (re)generate
    if only:
        import builtins
        keep = lambda fn: (lambda r, name, *a, **k: fn(r, name, *a, **k) if name in only else None)
        g = globals()
        for fname in ("case_multimodal", "case_single", "case_deviation", "case_regression", "case_endtoend", "case_dm", "case_mvt",
                      "case_fusion_ops"):
            g[fname] = keep(g[fname])
        orig_csv = g["case_csv_headers"]
        g["case_csv_headers"] = lambda name: orig_csv(name) if name in only else None
    # small shapes: every combiner, ragged rows, float vs int covariates
    for comb in ("poe", "gpoe", "moe", "mopoe"):
        case_multimodal(ref, f"mm3_{comb}", (23, 17, 29), 7, (24, 16), 6, 19, comb, 3, seed=100, store_steps=(1, 3))
    case_multimodal(ref, "mm1_small", (37,), 7, (24, 16), 6, 19, "gpoe", 5, seed=101, store_steps=(1, 5))
    case_multimodal(ref, "mm4_uca_gpoe", (12, 11, 13, 36), 7, (24, 16), 6, 32, "gpoe", 2, seed=102, store_steps=(2,))
    case_multimodal(ref, "mm1_h1", (21,), 4, (20,), 5, 16, "poe", 2, seed=103, store_steps=(2,))
    case_multimodal(ref, "mm2_z64", (40, 33), 29, (48, 40), 64, 48, "poe", 2, seed=104, store_steps=(2,))
    # BASELINE config A: D=379, c=29, H=[110,110], Z=10, B=256 (one step; full-size grads)
    case_multimodal(ref, "cfgA_T1w", (379,), 29, (110, 110), 10, 256, "gpoe", 2, seed=7, store_steps=(2,))
    # ragged tail of the real HCPimage size (83 rows)
    case_multimodal(ref, "cfgA_T1w_tail83", (379,), 29, (110, 110), 10, 83, "gpoe", 1, seed=8, store_steps=())
    case_single(ref, "single_small", 37, 7, (24, 16), 6, 19, seed=105)
    case_deviation(ref, "dev_small", (23, 17, 29), 5, (24, 16), 6, 40, "gpoe", seed=106)
    case_csv_headers("csv_layouts")
    case_fusion_ops(ref, "fusion_ops", 3, 19, 6, seed=120)
    case_regression(ref, "reg3_gpoe", (23, 17, 29), 2, (24, 16), 6, 32, "gpoe", 3, seed=107)
    case_endtoend(ref, "e2e3", (23, 17, 29), 7, (24, 16), 8, 32, (16, 8), 3, seed=108)
    # baseline zoo: mmJSD (cVAE.py:1354-1448) = product of experts without the single-expert
    # bypass; its JSD term compares the joint posterior with itself and is identically zero
    case_multimodal(ref, "mmjsd3", (23, 17, 29), 7, (24, 16), 6, 19, "gpoe", 3, seed=109, store_steps=(3,), cls="mmJSD")
    # DMVAE family: (a) the configuration the scripts run (c_dim = 29 >= latent: every latent column is private, the
    # models are deterministic autoencoders), (b) c_dim < latent: 3 private + 7 shared columns
    case_dm(ref, "dmvae3", "DMVAE", (23, 17, 29), 29, (24, 16), 10, 19, 3, seed=110, store_steps=(3,))
    case_dm(ref, "dmvae3_shared", "DMVAE", (23, 17, 29), 3, (24, 16), 10, 32, 3, seed=111, store_steps=(3,))
    case_dm(ref, "wdmvae3_shared", "WeightedDMVAE", (23, 17, 29), 3, (24, 16), 10, 32, 3, seed=112, store_steps=(3,))
    case_dm(ref, "mmvaeplus3_shared", "mmVAEPlus", (23, 17, 29), 4, (24, 16), 12, 19, 3, seed=113, store_steps=(3,))
    # mvtCAE: its 'poe' (ProductOfExperts2 on variances) and a standard combiner, both with the variance clamp and the tc term
    for comb in ("poe", "gpoe", "mopoe"):
        case_mvt(ref, f"mvtcae3_{comb}", (23, 17, 29), 7, (24, 16), 6, 19, comb, 3, seed=114, store_steps=(3,))


if __name__ == "__main__":
    main()
