"""CPU oracle for the cVAE hot path (TEST INFRASTRUCTURE, NOT THE PRODUCT).

This file is a plain fp32 PyTorch-CPU restatement of the reference's conditional-VAE
train step and deviation pass.  It exists so that the HIP path can be *checked*:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package never imports anything from
``oracle/`` and fails loudly when its HIP library is missing.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself: ``oracle/gen_golden.py`` imports ``/root/reference/cVAE.py`` in the build
container, drives its classes on seeded inputs with an explicit ``eps`` and
commits the results as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against them.

Every function cites the reference file:line it restates (paths relative to
``/root/reference``).  Parameters are kept in a flat ``dict`` keyed by the
reference's own ``state_dict`` names so weights interchange with it.

All randomness is explicit: ``eps`` (the reparameterisation draw of
``cVAE.py:418-421``) is always an argument.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

LEAKY_SLOPE = 0.01          # F.leaky_relu default, cVAE.py:167,203
LOG_SQRT_2PI = math.log(math.sqrt(2.0 * math.pi))


# ----------------------------------------------------------------------------------------
# model description
# ----------------------------------------------------------------------------------------
@dataclass
class Spec:
    """Shape description of one cVAE_multimodal-family model.

    Mirrors the constructor arguments of cVAE.py:1087-1116 (``input_dim_list``,
    ``hidden_dim``, ``latent_dim``, ``c_dim``, ``modalities``, ``non_linear``).
    """
    input_dims: Sequence[int]
    hidden: Sequence[int]
    latent: int
    c_dim: int
    non_linear: bool = True
    kind: str = "multimodal"      # "single" (cVAE), "multimodal", "regression", "endtoend"
    classifier_layers: Sequence[int] = field(default_factory=list)   # endtoend only
    num_classes: int = 2

    @property
    def M(self) -> int:
        return len(self.input_dims)

    def enc_sizes(self, m: int) -> List[int]:
        # cVAE.py:153  layer_sizes_encoder = [input_dim + c_dim] + hidden_dims(+latent)
        return [self.input_dims[m] + self.c_dim] + list(self.hidden) + [self.latent]

    def dec_sizes(self, m: int) -> List[int]:
        # cVAE.py:183-188  hidden_dims reversed, first entry + c_dim, then input_dim
        hd = (list(self.hidden) + [self.latent])[::-1]
        sizes = hd + [self.input_dims[m]]
        sizes[0] = hd[0] + self.c_dim
        return sizes


def _enc_prefix(spec: Spec, m: int) -> str:
    return "encoder." if spec.kind == "single" else f"encoder_list.{m}."


def _dec_prefix(spec: Spec, m: int, bank: str = "") -> str:
    if spec.kind == "single":
        return "decoder."
    if spec.kind == "endtoend":
        return f"decoder_list_{bank}.{m}."
    return f"decoder_list.{m}."


def param_names(spec: Spec) -> List[str]:
    """Names in the order the reference registers them (= ``state_dict()`` order).

    cVAE_multimodal: alpha_m_list, encoder_list, decoder_list   (cVAE.py:1106-1108)
    cVAE_multimodal_regression: encoder_list, decoder_list, alpha_m_list, regressor (cVAE.py:2230-2253)
    cVAE (single): encoder, decoder (cVAE.py:406-407; the unused discriminator is not part of the path)
    cVAE_multimodal_endtoend: encoder_list, decoder_list_health, decoder_list_disease, classifier (cVAE.py:2044-2054)
    """
    names: List[str] = []

    def enc(m):
        p = _enc_prefix(spec, m)
        n_hidden = len(spec.hidden)
        out = []
        for i in range(n_hidden):
            out += [f"{p}encoder_layers.{i}.weight", f"{p}encoder_layers.{i}.bias"]
        out += [f"{p}enc_mean_layer.weight", f"{p}enc_mean_layer.bias",
                f"{p}enc_logvar_layer.weight", f"{p}enc_logvar_layer.bias"]
        return out

    def dec(m, bank=""):
        p = _dec_prefix(spec, m, bank)
        n_hidden = len(spec.hidden)
        out = [f"{p}logvar_out"]                    # registered first, cVAE.py:193-194 order of attributes
        for i in range(n_hidden):
            out += [f"{p}decoder_layers.{i}.weight", f"{p}decoder_layers.{i}.bias"]
        out += [f"{p}decoder_mean_layer.weight", f"{p}decoder_mean_layer.bias"]
        return out

    if spec.kind == "single":
        names += enc(0) + dec(0)
    elif spec.kind == "multimodal":
        names += [f"alpha_m_list.{m}" for m in range(spec.M)]
        for m in range(spec.M):
            names += enc(m)
        for m in range(spec.M):
            names += dec(m)
    elif spec.kind == "regression":
        for m in range(spec.M):
            names += enc(m)
        for m in range(spec.M):
            names += dec(m)
        names += [f"alpha_m_list.{m}" for m in range(spec.M)]
        names += ["regressor.0.weight", "regressor.0.bias", "regressor.2.weight",
                  "regressor.2.bias", "regressor.4.weight", "regressor.4.bias"]
    elif spec.kind == "endtoend":
        for m in range(spec.M):
            names += enc(m)
        for bank in ("health", "disease"):
            for m in range(spec.M):
                names += dec(m, bank)
        sizes = [spec.latent] + list(spec.classifier_layers)
        li = 0
        for i in range(len(sizes) - 1):
            names += [f"classifier.classifier.{li}.weight", f"classifier.classifier.{li}.bias",
                      f"classifier.classifier.{li + 1}.weight", f"classifier.classifier.{li + 1}.bias"]
            li += 4
        names += [f"classifier.classifier.{li}.weight", f"classifier.classifier.{li}.bias"]
    else:
        raise ValueError(spec.kind)
    return names


def param_shapes(spec: Spec) -> Dict[str, tuple]:
    shapes: Dict[str, tuple] = {}

    def enc(m):
        p = _enc_prefix(spec, m)
        s = spec.enc_sizes(m)
        for i in range(len(spec.hidden)):
            shapes[f"{p}encoder_layers.{i}.weight"] = (s[i + 1], s[i])
            shapes[f"{p}encoder_layers.{i}.bias"] = (s[i + 1],)
        for h in ("enc_mean_layer", "enc_logvar_layer"):
            shapes[f"{p}{h}.weight"] = (s[-1], s[-2])
            shapes[f"{p}{h}.bias"] = (s[-1],)

    def dec(m, bank=""):
        p = _dec_prefix(spec, m, bank)
        s = spec.dec_sizes(m)
        shapes[f"{p}logvar_out"] = (1, spec.input_dims[m])
        for i in range(len(spec.hidden)):
            shapes[f"{p}decoder_layers.{i}.weight"] = (s[i + 1], s[i])
            shapes[f"{p}decoder_layers.{i}.bias"] = (s[i + 1],)
        shapes[f"{p}decoder_mean_layer.weight"] = (s[-1], s[-2])
        shapes[f"{p}decoder_mean_layer.bias"] = (s[-1],)

    for m in range(spec.M):
        enc(m)
        if spec.kind == "endtoend":
            dec(m, "health")
            dec(m, "disease")
        else:
            dec(m)
    if spec.kind in ("multimodal", "regression"):
        for m in range(spec.M):
            shapes[f"alpha_m_list.{m}"] = (1,)
    if spec.kind == "regression":
        tot = sum(spec.input_dims)
        shapes.update({"regressor.0.weight": (128, tot), "regressor.0.bias": (128,),
                       "regressor.2.weight": (64, 128), "regressor.2.bias": (64,),
                       "regressor.4.weight": (1, 64), "regressor.4.bias": (1,)})
    if spec.kind == "endtoend":
        sizes = [spec.latent] + list(spec.classifier_layers)
        li = 0
        for i in range(len(sizes) - 1):
            shapes[f"classifier.classifier.{li}.weight"] = (sizes[i + 1], sizes[i])
            shapes[f"classifier.classifier.{li}.bias"] = (sizes[i + 1],)
            shapes[f"classifier.classifier.{li + 1}.weight"] = (sizes[i + 1],)   # BatchNorm1d gamma
            shapes[f"classifier.classifier.{li + 1}.bias"] = (sizes[i + 1],)
            li += 4
        shapes[f"classifier.classifier.{li}.weight"] = (spec.num_classes, sizes[-1])
        shapes[f"classifier.classifier.{li}.bias"] = (spec.num_classes,)
    return {n: shapes[n] for n in param_names(spec)}


def init_params(spec: Spec, seed: int = 42) -> Dict[str, torch.Tensor]:
    """Reference init rule: nn.Linear default U(+-1/sqrt(fan_in)) for weight and bias
    (cVAE.py:155-159,190-192), ``logvar_out`` = -3 (cVAE.py:179,193), ``alpha`` ~ N(0,1)
    (cVAE.py:1106), BatchNorm gamma=1 / beta=0.  The draw ORDER differs from the
    reference's constructor, so values are not the reference's for a given seed;
    parity tests always load explicit weights."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(spec).items():
        if name.endswith("logvar_out"):
            out[name] = torch.full(shape, -3.0)
        elif name.startswith("alpha_m_list"):
            out[name] = torch.randn(shape, generator=g)
        elif len(shape) == 2:
            bound = 1.0 / math.sqrt(shape[1])
            out[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            # bias: needs fan_in of the matching weight
            w = out.get(name[:-4] + "weight")
            if w is not None and w.dim() == 2:
                bound = 1.0 / math.sqrt(w.shape[1])
                out[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
            elif name.endswith("weight"):      # BatchNorm gamma
                out[name] = torch.ones(shape)
            else:                              # BatchNorm beta
                out[name] = torch.zeros(shape)
    return out


# ----------------------------------------------------------------------------------------
# operand-rounding mode
# ----------------------------------------------------------------------------------------
# The HIP path is specified (BASELINE.json north_star) to run every Linear as a bf16 MFMA GEMM
# with fp32 accumulation.  "fp32" below is the reference's own arithmetic; "bf16" restates the
# SAME algorithm with the GEMM operands (activations, weights, and in backward the incoming
# deltas) rounded to bfloat16 -- everything else (bias add, activation, NLL, KL, fusion, Adam)
# stays fp32.  LeakyReLU makes gradients discontinuous in the pre-activations, so an fp32-vs-bf16
# comparison of gradients sees isolated sign flips; HIP-vs-"bf16" is the tight elementwise check,
# HIP-vs-"fp32"/golden carries the 1e-4 reconstruction-loss bound.
_OPERANDS = "fp32"


def set_operand_rounding(mode: str):
    global _OPERANDS
    if mode not in ("fp32", "bf16"):
        raise ValueError(mode)
    _OPERANDS = mode


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class _LinearBf16(torch.autograd.Function):
    """y = bf16(a) @ bf16(W)^T + b with fp32 accumulation; backward rounds the incoming delta
    to bf16 once and uses it for dgrad, wgrad and the bias gradient (ones-column of the wgrad)."""

    @staticmethod
    def forward(ctx, a, W, b):
        ctx.save_for_backward(a, W)
        return _bf(a) @ _bf(W).T + b

    @staticmethod
    def backward(ctx, go):
        a, W = ctx.saved_tensors
        gb = _bf(go)
        return gb @ _bf(W), gb.T @ _bf(a), gb.sum(0)


def linear(a, W, b):
    if _OPERANDS == "bf16":
        return _LinearBf16.apply(a, W, b)
    return torch.nn.functional.linear(a, W, b)


# ----------------------------------------------------------------------------------------
# A1 / A2  encoder, decoder      (cVAE.py:140-172, 174-206)
# ----------------------------------------------------------------------------------------
def _act(h: torch.Tensor, non_linear: bool) -> torch.Tensor:
    return torch.nn.functional.leaky_relu(h, LEAKY_SLOPE) if non_linear else h


def encoder_fwd(P, spec: Spec, m: int, x: torch.Tensor, c: torch.Tensor):
    """cVAE.py:161-172.  ``c`` may be int64 one-hot (train) or float; ``torch.cat``
    promotes to float32 exactly as in the reference."""
    p = _enc_prefix(spec, m)
    h = torch.cat((x, c.to(x.dtype)), dim=1)
    for i in range(len(spec.hidden)):
        h = _act(linear(h, P[f"{p}encoder_layers.{i}.weight"], P[f"{p}encoder_layers.{i}.bias"]), spec.non_linear)
    mu = linear(h, P[f"{p}enc_mean_layer.weight"], P[f"{p}enc_mean_layer.bias"])
    logvar = linear(h, P[f"{p}enc_logvar_layer.weight"], P[f"{p}enc_logvar_layer.bias"])
    return mu, logvar


def decoder_fwd(P, spec: Spec, m: int, z: torch.Tensor, c: torch.Tensor, bank: str = ""):
    """cVAE.py:197-206.  Returns (loc, scale) of the Normal; scale = exp(logvar_out)**0.5."""
    p = _dec_prefix(spec, m, bank)
    h = torch.cat((z, c.reshape(-1, spec.c_dim).to(z.dtype)), dim=1)
    for i in range(len(spec.hidden)):
        h = _act(linear(h, P[f"{p}decoder_layers.{i}.weight"], P[f"{p}decoder_layers.{i}.bias"]), spec.non_linear)
    loc = linear(h, P[f"{p}decoder_mean_layer.weight"], P[f"{p}decoder_mean_layer.bias"])
    scale = P[f"{p}logvar_out"].exp().pow(0.5)
    return loc, scale


# ----------------------------------------------------------------------------------------
# A3-A5  reparameterise / KL / LL
# ----------------------------------------------------------------------------------------
def reparameterise(mu, logvar, eps):
    """cVAE.py:418-421 with the draw made explicit."""
    return mu + eps * torch.exp(0.5 * logvar)


def calc_kl(mu, logvar):
    """cVAE.py:429-430 / 1138-1139."""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1).mean(0)


def normal_log_prob(x, loc, scale):
    """torch.distributions.Normal.log_prob: -(x-loc)^2/(2 scale^2) - log(scale) - log(sqrt(2 pi))."""
    var = scale ** 2
    return -((x - loc) ** 2) / (2 * var) - scale.log() - LOG_SQRT_2PI


def compute_ll(x, loc, scale):
    """cVAE.py:14-15: log_prob(x).sum(1, keepdims=True).mean(0) -> shape [1]."""
    return normal_log_prob(x, loc, scale).sum(1, keepdim=True).mean(0)


# ----------------------------------------------------------------------------------------
# A7  expert fusion      (cVAE.py:986-998, 1000-1011, 1060-1083, 1144-1164)
# ----------------------------------------------------------------------------------------
def combine_latent(mus, variances, combine: str, alphas: Optional[Sequence[torch.Tensor]] = None,
                   single_bypass: bool = True):
    if single_bypass and mus.shape[0] == 1:           # cVAE.py:1146-1147
        return mus[0], variances[0]
    combine = combine.lower()
    if combine == "poe":
        T = 1.0 / variances
        return torch.sum(mus * T, dim=0) / torch.sum(T, dim=0), 1.0 / torch.sum(T, dim=0)
    if combine == "gpoe":
        M = mus.shape[0]
        a = torch.softmax(torch.stack([p for p in alphas]), dim=0).reshape(M, 1, 1)
        mu = torch.sum(mus * a / variances, dim=0) / torch.sum(a / variances, dim=0)
        return mu, 1 / torch.sum(a / variances, dim=0)
    if combine == "moe":
        w = 1.0 / mus.shape[0]
        return torch.sum(mus * w, dim=0), torch.sum(variances * w, dim=0)
    if combine == "mopoe":
        T = 1.0 / variances
        poe_mu = torch.sum(mus * T, dim=0) / torch.sum(T, dim=0)
        poe_var = 1.0 / torch.sum(T, dim=0)
        mus2 = torch.cat((mus, poe_mu.unsqueeze(0)), dim=0)
        var2 = torch.cat((variances, poe_var.unsqueeze(0)), dim=0)
        w = 1.0 / mus2.shape[0]
        return torch.sum(mus2 * w, dim=0), torch.sum(var2 * w, dim=0)
    raise ValueError("No such combination method")       # cVAE.py:1163


# ----------------------------------------------------------------------------------------
# A6 / A8 / A9  forward + loss
# ----------------------------------------------------------------------------------------
def forward_multimodal(P, spec: Spec, xes, cs, combine: str, eps):
    """cVAE.py:1166-1182 (also cVAE.forward :435-443 when spec.kind == 'single')."""
    if spec.kind == "single":
        mu, logvar = encoder_fwd(P, spec, 0, xes[0], cs[0])
        z = reparameterise(mu, logvar, eps)
        loc, scale = decoder_fwd(P, spec, 0, z, cs[0])
        return {"locs": [loc], "scales": [scale], "mu": mu, "logvar": logvar, "z": z,
                "mus": mu.unsqueeze(0), "logvars": logvar.unsqueeze(0)}
    enc = [encoder_fwd(P, spec, m, xes[m], cs[m]) for m in range(spec.M)]
    mus = torch.stack([e[0] for e in enc])
    logvars = torch.stack([e[1] for e in enc])
    variances = torch.exp(logvars)
    alphas = [P[f"alpha_m_list.{m}"] for m in range(spec.M)]
    mu, var = combine_latent(mus, variances, combine, alphas)
    logvar = torch.log(var)                               # exp -> log round trip, cVAE.py:1175-1178
    z = reparameterise(mu, logvar, eps)
    dec = [decoder_fwd(P, spec, m, z, cs[m]) for m in range(spec.M)]
    return {"locs": [d[0] for d in dec], "scales": [d[1] for d in dec], "mu": mu, "logvar": logvar,
            "z": z, "mus": mus, "logvars": logvars}


def loss_multimodal(spec: Spec, xes, fwd):
    """cVAE.py:1187-1196 (KL is added once PER MODALITY) / cVAE.loss_function :491-504."""
    total = 0.0
    kl_sum = 0.0
    ll_sum = 0.0
    lls = []
    for m in range(spec.M):
        kl = calc_kl(fwd["mu"], fwd["logvar"])
        ll = compute_ll(xes[m], fwd["locs"][m], fwd["scales"][m])
        total = total + (kl - ll)
        kl_sum = kl_sum + kl
        ll_sum = ll_sum + ll
        lls.append(ll)
    return {"total": total, "kl": kl_sum, "ll": ll_sum, "ll_m": lls}


# ----------------------------------------------------------------------------------------
# A10  Adam (torch.optim.Adam defaults, cVAE.py:1111-1116): lr=1e-4, betas=(0.9,0.999), eps=1e-8
# ----------------------------------------------------------------------------------------
class Adam:
    """Hand-written restatement of torch.optim.Adam (no amsgrad, no weight decay)."""

    def __init__(self, P: Dict[str, torch.Tensor], names: Sequence[str], lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.names = list(names)
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.t = 0
        self.m = {n: torch.zeros_like(P[n]) for n in self.names}
        self.v = {n: torch.zeros_like(P[n]) for n in self.names}

    @torch.no_grad()
    def step(self, P, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for n in self.names:
            g = grads[n]
            if g is None:
                continue
            self.m[n].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[n].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(self.eps)
            P[n].addcdiv_(self.m[n], denom, value=-self.lr / bc1)


def optimizer_param_names(spec: Spec) -> List[str]:
    """Which tensors the reference's optimizer owns (cVAE.py:1111-1116, 404, 2254-2260, 2057-2063)."""
    return param_names(spec)


def train_step(P, opt: Adam, spec: Spec, xes, cs, combine: str, eps):
    """One iteration of the hot loop multimodal_kfold_train_cvae_supervised.py:193-199."""
    leaves = {n: P[n].detach().clone().requires_grad_(True) for n in opt.names}
    fwd = forward_multimodal(leaves, spec, xes, cs, combine, eps)
    loss = loss_multimodal(spec, xes, fwd)
    tot = loss["total"]
    grads_list = torch.autograd.grad(tot.sum(), [leaves[n] for n in opt.names], allow_unused=True)
    grads = {n: g for n, g in zip(opt.names, grads_list)}
    opt.step(P, grads)
    return ({k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss.items() if k != "ll_m"}
            | {"ll_m": [l.detach() for l in loss["ll_m"]]}, grads, fwd)


# ----------------------------------------------------------------------------------------
# A11  deviation passes
# ----------------------------------------------------------------------------------------
@torch.no_grad()
def deviation_unimodal(P, spec: Spec, m: int, x, c, eps):
    """multimodal_kfold_train_cvae_supervised_regression.py:183-188: unimodal posterior,
    SAMPLED z, per-ROI squared residual."""
    mu, logvar = encoder_fwd(P, spec, m, x, c)
    z = reparameterise(mu, logvar, eps)
    loc, _ = decoder_fwd(P, spec, m, z, c)
    return (x - loc) ** 2, loc


@torch.no_grad()
def pred_recon(P, spec: Spec, xes, c, combine: str, eps):
    """cVAE.py:1198-1208 (joint latent, sampled z; covariates cast to long there)."""
    cs = [c for _ in range(spec.M)]
    fwd = forward_multimodal(P, spec, xes, cs, combine, eps)
    return [l for l in fwd["locs"]]


def reconstruction_deviation(x, x_pred):
    """utils_vae.py:147-148 / cVAE.py:1210-1211: sum_d (x - x_pred)^2 / D per subject."""
    return ((x - x_pred) ** 2).sum(1) / x.shape[1]


def reconstruction_deviation_roi(x, x_pred):
    """utils_vae.py:151-152."""
    return (x - x_pred) ** 2


# ----------------------------------------------------------------------------------------
# A14  cVAE_multimodal_regression      (cVAE.py:2211-2346)
# ----------------------------------------------------------------------------------------
def forward_regression(P, spec: Spec, xes, cs, combine: str, eps):
    """cVAE.py:2309-2328: trunk as cVAE_multimodal, then fi_pred = regressor(cat_m(x_m - loc_m)),
    regressor = Linear(sum D, 128) - ReLU - Linear(128, 64) - ReLU - Linear(64, 1)."""
    trunk = Spec(spec.input_dims, spec.hidden, spec.latent, spec.c_dim, spec.non_linear, kind="multimodal")
    fwd = forward_multimodal(P, trunk, xes, cs, combine, eps)
    diffs = torch.cat([xes[m] - fwd["locs"][m] for m in range(spec.M)], dim=1)
    h = torch.relu(linear(diffs, P["regressor.0.weight"], P["regressor.0.bias"]))
    h = torch.relu(linear(h, P["regressor.2.weight"], P["regressor.2.bias"]))
    fwd["fi_pred"] = linear(h, P["regressor.4.weight"], P["regressor.4.bias"])
    return fwd


def loss_regression(spec: Spec, xes, fwd, true_fi, lambda_reg: float = 1.0):
    """cVAE.py:2330-2346: sum_m (KL - LL_m) + lambda * MSE(fi_pred, FI)."""
    trunk = Spec(spec.input_dims, spec.hidden, spec.latent, spec.c_dim, spec.non_linear, kind="multimodal")
    loss = loss_multimodal(trunk, xes, fwd)
    reg = torch.nn.functional.mse_loss(fwd["fi_pred"].squeeze(), true_fi.squeeze())
    loss["regression"] = reg
    loss["total"] = loss["total"] + lambda_reg * reg
    return loss


# ----------------------------------------------------------------------------------------
# A13  cVAE_multimodal_endtoend        (cVAE.py:2004-2207)
# ----------------------------------------------------------------------------------------
def classifier_fwd(P, spec: Spec, z, training: bool, bn_stats: Optional[dict] = None, drop_masks=None, drop_p: float = 0.0):
    """Classifier (cVAE.py:2004-2018): (Linear - BatchNorm1d - ReLU - Dropout)* - Linear.  Dropout (nn.Dropout in
    train mode, :2012): `drop_masks` = one [B, width] 0/1 keep mask per block (torch draws it from the global
    generator; parity runs inject the mask the kernel used), kept activations scaled by 1 / (1 - drop_p); None: the
    identity (rate 0, or eval).  In training mode BatchNorm uses the batch statistics; in eval mode the running
    statistics passed in `bn_stats`."""
    h = z
    li = 0
    for _ in range(len(spec.classifier_layers)):
        p = f"classifier.classifier.{li}"
        h = linear(h, P[p + ".weight"], P[p + ".bias"])
        q = f"classifier.classifier.{li + 1}"
        if training:
            h = torch.nn.functional.batch_norm(h, None, None, P[q + ".weight"], P[q + ".bias"], True, 0.1, 1e-5)
        else:
            h = torch.nn.functional.batch_norm(h, bn_stats[q + ".running_mean"], bn_stats[q + ".running_var"],
                                               P[q + ".weight"], P[q + ".bias"], False, 0.1, 1e-5)
        h = torch.relu(h)
        if training and drop_masks is not None and drop_p > 0.0:
            h = h * drop_masks[li // 4].to(h.dtype) / (1.0 - drop_p)
        li += 4
    p = f"classifier.classifier.{li}"
    return linear(h, P[p + ".weight"], P[p + ".bias"])


def forward_endtoend(P, spec: Spec, xes, cs, eps, training: bool = True, bn_stats=None, drop_masks=None, drop_p: float = 0.0):
    """cVAE.py:2106-2123: shared encoders -> PoE WITHOUT the single-expert bypass (cVAE.py:2083-2090)
    -> z -> health and disease decoder banks + classifier(z)."""
    enc = [encoder_fwd(P, spec, m, xes[m], cs[m]) for m in range(spec.M)]
    mus = torch.stack([e[0] for e in enc])
    logvars = torch.stack([e[1] for e in enc])
    T = 1 / torch.exp(logvars)
    mu = torch.sum(mus * T, dim=0) / torch.sum(T, dim=0)
    logvar = torch.log(1 / torch.sum(T, dim=0))
    z = reparameterise(mu, logvar, eps)
    locs_h, locs_d, scales_h, scales_d = [], [], [], []
    for m in range(spec.M):
        lh, sh = decoder_fwd(P, spec, m, z, cs[m], "health")
        ld, sd = decoder_fwd(P, spec, m, z, cs[m], "disease")
        locs_h.append(lh); locs_d.append(ld); scales_h.append(sh); scales_d.append(sd)
    return {"locs_h": locs_h, "locs_d": locs_d, "scales_h": scales_h, "scales_d": scales_d, "mu": mu, "logvar": logvar,
            "z": z, "logits": classifier_fwd(P, spec, z, training, bn_stats, drop_masks, drop_p)}


def loss_endtoend(spec: Spec, xes, fwd, labels, margin=1.0, weightcontrastive=0.1, weight_kl=0.1, weight_rec=0.1):
    """cVAE.py:2140-2200."""
    rh = rd = 0.0
    dh, dd = [], []
    for m in range(spec.M):
        rh = rh + (-normal_log_prob(xes[m], fwd["locs_h"][m], fwd["scales_h"][m]).sum(dim=1).mean())
        rd = rd + (-normal_log_prob(xes[m], fwd["locs_d"][m], fwd["scales_d"][m]).sum(dim=1).mean())
        dh.append(((xes[m] - fwd["locs_h"][m]) ** 2).mean(dim=1))           # compute_deviation, cVAE.py:2134-2138
        dd.append(((xes[m] - fwd["locs_d"][m]) ** 2).mean(dim=1))
    dev_h = torch.stack(dh).mean(dim=0)
    dev_d = torch.stack(dd).mean(dim=0)
    contrastive = torch.mean((1 - labels) * torch.relu(margin + dev_h - dev_d) + labels * torch.relu(margin + dev_d - dev_h))
    kl = -0.5 * torch.sum(1 + fwd["logvar"] - fwd["mu"].pow(2) - fwd["logvar"].exp(), dim=1).mean()
    ce = torch.nn.functional.cross_entropy(fwd["logits"], labels)
    total = weight_rec * (rh + rd) + weight_kl * kl + ce + weightcontrastive * contrastive
    return {"total_loss": total, "recon_loss_health": rh, "recon_loss_disease": rd, "kl_loss": kl,
            "classification_loss": ce, "contrastive_loss": contrastive}


# ----------------------------------------------------------------------------------------
# N4  DMVAE / WeightedDMVAE / mmVAEPlus (cVAE.py:1453-1747, 1895-2002)
# ----------------------------------------------------------------------------------------
@dataclass
class DmSpec:
    """Constructor arguments of the DMVAE family.  VariationalEncoder (cVAE.py:1453-1466): x -> relu(fc1) -> relu(fc2)
    -> fc_mu / fc_logvar (no covariates); VariationalDecoder (:1468-1479): z -> relu(fc1) -> relu(fc2) -> sigmoid(fc_out).
    s_dim = c_dim (:1506): the first s_dim columns of mu are the modality's private latent, the rest is shared."""
    input_dims: Sequence[int]
    hidden: Sequence[int]          # exactly two widths
    latent: int
    c_dim: int
    cls: str = "DMVAE"             # DMVAE | WeightedDMVAE | mmVAEPlus

    @property
    def M(self) -> int:
        return len(self.input_dims)

    @property
    def n_private(self) -> int:
        return min(self.c_dim, self.latent)    # mu[:, :s_dim] of a [B, latent] tensor

    @property
    def beta(self) -> float:
        return 0.05 if self.cls == "mmVAEPlus" else 1.0        # cVAE.py:1911 / :1507


def dm_param_names(spec: DmSpec) -> List[str]:
    """state_dict order: a module's own parameters (`weights`) come before its children's."""
    names = ["weights"] if spec.cls == "WeightedDMVAE" else []
    for m in range(spec.M):
        for l in ("fc1", "fc2", "fc_mu", "fc_logvar"):
            names += [f"encoder_list.{m}.{l}.weight", f"encoder_list.{m}.{l}.bias"]
    for m in range(spec.M):
        for l in ("fc1", "fc2", "fc_out"):
            names += [f"decoder_list.{m}.{l}.weight", f"decoder_list.{m}.{l}.bias"]
    return names


def dm_forward(P, spec: DmSpec, xes, eps):
    """forward_multimodal of the DMVAE family (cVAE.py:1536-1558): eps [B, >= Z - s] is the draw of the shared latent."""
    S, Zc = spec.n_private, spec.latent - spec.n_private
    mus, lvs = [], []
    for m in range(spec.M):
        p = f"encoder_list.{m}."
        h = torch.relu(linear(xes[m], P[p + "fc1.weight"], P[p + "fc1.bias"]))
        h = torch.relu(linear(h, P[p + "fc2.weight"], P[p + "fc2.bias"]))
        mus.append(linear(h, P[p + "fc_mu.weight"], P[p + "fc_mu.bias"]))
        lvs.append(linear(h, P[p + "fc_logvar.weight"], P[p + "fc_logvar.bias"]))
    mu_c = torch.stack([m_[:, S:] for m_ in mus])
    lv_c = torch.stack([l_[:, S:] for l_ in lvs])
    var_inv = 1.0 / torch.exp(lv_c)                                           # ProductOfExperts2, cVAE.py:1481-1489
    mu_j = torch.sum(mu_c * var_inv, dim=0) / torch.sum(var_inv, dim=0)
    lv_j = torch.log(1.0 / torch.sum(var_inv, dim=0))
    z = mu_j + eps[:, :Zc] * torch.exp(0.5 * lv_j)
    recons = []
    for m in range(spec.M):
        p = f"decoder_list.{m}."
        zc = torch.cat((z, mus[m][:, :S]), dim=1)
        h = torch.relu(linear(zc, P[p + "fc1.weight"], P[p + "fc1.bias"]))
        h = torch.relu(linear(h, P[p + "fc2.weight"], P[p + "fc2.bias"]))
        recons.append(torch.sigmoid(linear(h, P[p + "fc_out.weight"], P[p + "fc_out.bias"])))
    return {"x_recons": recons, "mu_c": mu_j, "logvar_c": lv_j}


def dm_loss(P, spec: DmSpec, xes, fwd):
    """loss_function_multimodal (cVAE.py:1563-1575, :1693-1710, :1967-1979): KL of the shared posterior once per modality,
    ll_i = -0.5 sum (x_i - x_hat_i)^2 mean over rows; total = beta * kl - ll (weights[i] on both terms for WeightedDMVAE)."""
    klb = -0.5 * torch.sum(1 + fwd["logvar_c"] - fwd["mu_c"].pow(2) - torch.exp(fwd["logvar_c"]), dim=1).mean(0)
    kl, ll = 0, 0
    for i in range(spec.M):
        w = P["weights"][i] if spec.cls == "WeightedDMVAE" else 1.0
        kl = kl + klb * w
        ll = ll + (-0.5 * torch.sum((xes[i] - fwd["x_recons"][i]) ** 2, dim=1).mean(0)) * w
    return {"total": kl * spec.beta - ll, "kl": kl, "ll": ll}


def dm_train_step(P, opt: Adam, spec: DmSpec, xes, eps):
    leaves = {n: P[n].detach().clone().requires_grad_(True) for n in opt.names}
    fwd = dm_forward(leaves, spec, xes, eps)
    loss = dm_loss(leaves, spec, xes, fwd)
    gl = torch.autograd.grad(loss["total"], [leaves[n] for n in opt.names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(opt.names, gl)}
    opt.step(P, grads)
    return {k: v.detach() for k, v in loss.items()}, grads, fwd


# ----------------------------------------------------------------------------------------
# N4  mvtCAE (cVAE.py:1754-1893)
# ----------------------------------------------------------------------------------------
MVT_BETA = 1e-4          # cVAE.py:1771
MVT_LL_W = 1e-5          # cVAE.py:1877 (the log-likelihood enters the total with a PLUS sign)
MVT_VAR_FLOOR = 1e-6     # cVAE.py:1823


def mvt_combine(mus, variances, combine: str, alphas):
    """mvtCAE.combine_latent (cVAE.py:1808-1825): no single-expert bypass; 'poe' hands the VARIANCES to ProductOfExperts2
    as if they were log variances and takes the returned log variance as the joint variance (:1782-1783, 1481-1489); the
    other combiners as in cVAE_multimodal; the joint variance is clamped at 1e-6."""
    combine = combine.lower()
    if combine == "poe":
        var_inv = 1.0 / torch.exp(variances)
        mu = torch.sum(mus * var_inv, dim=0) / torch.sum(var_inv, dim=0)
        var = torch.log(1.0 / torch.sum(var_inv, dim=0))
    else:
        mu, var = combine_latent(mus, variances, combine, alphas, single_bypass=False)
    return mu, torch.clamp(var, min=MVT_VAR_FLOOR)


def mvt_total_correlation(mus, mu_joint, latent: int):
    """mvtCAE.total_correlation (cVAE.py:1862-1869) as written: per latent column, logsumexp over the batch of the joint
    mean minus its own (scalar) mean, minus the mean over experts of logsumexp over the batch of the expert means."""
    tc = 0
    for i in range(latent):
        a = mu_joint[:, i].logsumexp(dim=0) - mu_joint[:, i].logsumexp(dim=0).mean()
        b = torch.stack([mus[j][:, i].logsumexp(dim=0) for j in range(len(mus))]).mean(dim=0)
        tc = tc + a - b
    return tc


def mvt_forward(P, spec: Spec, xes, cs, combine: str, eps):
    enc = [encoder_fwd(P, spec, m, xes[m], cs[m]) for m in range(spec.M)]
    mus = torch.stack([e[0] for e in enc])
    variances = torch.exp(torch.stack([e[1] for e in enc]))
    alphas = [P[f"alpha_m_list.{m}"] for m in range(spec.M)]
    mu, var = mvt_combine(mus, variances, combine, alphas)
    logvar = torch.log(var)
    z = reparameterise(mu, logvar, eps)
    dec = [decoder_fwd(P, spec, m, z, cs[m]) for m in range(spec.M)]
    return {"locs": [d[0] for d in dec], "scales": [d[1] for d in dec], "mu": mu, "logvar": logvar, "z": z, "mus": mus}


def mvt_loss(spec: Spec, xes, fwd):
    """cVAE.py:1871-1882: sum over the modalities of kl + 1e-5 * ll_i + beta * tc."""
    out = {"total": 0, "kl": 0, "ll": 0, "tc": 0}
    for i in range(spec.M):
        kl = calc_kl(fwd["mu"], fwd["logvar"])
        ll = compute_ll(xes[i], fwd["locs"][i], fwd["scales"][i])
        tc = mvt_total_correlation(fwd["mus"], fwd["mu"], spec.latent)
        out["total"] = out["total"] + kl + MVT_LL_W * ll + MVT_BETA * tc
        out["kl"] = out["kl"] + kl
        out["ll"] = out["ll"] + ll
        out["tc"] = out["tc"] + tc
    return out


def mvt_train_step(P, opt: Adam, spec: Spec, xes, cs, combine: str, eps):
    leaves = {n: P[n].detach().clone().requires_grad_(True) for n in opt.names}
    fwd = mvt_forward(leaves, spec, xes, cs, combine, eps)
    loss = mvt_loss(spec, xes, fwd)
    gl = torch.autograd.grad(loss["total"].sum(), [leaves[n] for n in opt.names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(opt.names, gl)}
    opt.step(P, grads)
    return {k: v.detach() for k, v in loss.items()}, grads, fwd
