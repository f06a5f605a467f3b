"""The reference's own class surface (cVAE.py) on top of the HIP path.

``cVAE`` (cVAE.py:391-562) and ``cVAE_multimodal`` (cVAE.py:1087-1211) keep their constructor
signatures, method names, returned dict keys, attribute names (``optimizer1``, ``encoder_list``,
``decoder_list``, ``alpha_m_list``) and ``state_dict`` keys, so the reference's train / test
loops run unchanged:

    fwd_rtn = model.forward_multimodal(xes, cs, combine)        # multimodal_kfold_train_...:193
    loss = model.loss_function_multimodal(xes, fwd_rtn)         # :194
    model.optimizer1.zero_grad(); loss['total'].backward(); model.optimizer1.step()   # :197-199

Underneath, ``forward_multimodal`` is ONE launch of the fused step kernel (forward + ELBO +
backward, gradients to a flat buffer), ``.backward()`` publishes those gradients and
``optimizer1.step()`` is the flat Adam kernel.  The sweep (``sweep.py``) does not use this eager
surface: it keeps whole training runs inside the persistent kernel.

All arithmetic runs in libnmhip.so; there is no CPU fallback (NmError without a GPU).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import Job, JobSet, Table, adam_step, require_gpu
from .layout import ModelSpec, ParamLayout


class NormalLike:
    """What callers use of ``torch.distributions.Normal`` (cVAE.py:206): loc / scale / mean / log_prob."""

    def __init__(self, loc: torch.Tensor, scale: torch.Tensor):
        self.loc, self.scale = loc, scale

    @property
    def mean(self):
        return self.loc

    def log_prob(self, x):
        var = self.scale ** 2
        return -((x - self.loc) ** 2) / (2 * var) - self.scale.log() - 0.9189385332046727


class _Holder(nn.Module):
    """Parameter holder reproducing one reference sub-module's parameter names."""

    def __init__(self, **params):
        super().__init__()
        for k, v in params.items():
            self.register_parameter(k, v)


class _LossFn(torch.autograd.Function):
    """Gives the loss scalars a ``.backward()`` that publishes the gradients the kernel produced."""

    @staticmethod
    def forward(ctx, anchor, model, which, value):
        ctx.model, ctx.which = model, which
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.model._publish_grads(ctx.which, g)
        return None, None, None, None


class _Adam:
    """``optimizer1`` of the reference (torch.optim.Adam, lr = learning_rate, cVAE.py:1111-1116)."""

    def __init__(self, model, lr):
        self.model = model
        self.param_groups = [{"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8}]
        self.t = 0
        self.lr = lr            # the reference's loop assigns this attribute; like there, it is inert

    def zero_grad(self, set_to_none: bool = True):
        self.model._grads_ready = False

    def step(self):
        m = self.model
        if not m._grads_ready:
            raise RuntimeError("optimizer1.step() called without a preceding loss.backward()")
        g = self.param_groups[0]
        self.t += 1
        adam_step(m._job.params, m._pending, m._job.adam_m, m._job.adam_v, self.t, lr=g["lr"], betas=g["betas"],
                  eps=g["eps"])
        m.layout.kernel_to_nat(m._job.params, m._flat.data)     # the module's parameter views follow
        # (the step count lives here: the job's own optimizer clock only drives the fused launches, and leaving it
        #  alone keeps the uploaded descriptor valid from call to call)
        m._job.params_changed()


class _Base(nn.Module):
    def _setup(self, spec: ModelSpec, learning_rate: float, kl_weight: float):
        self.spec = spec
        self.layout = ParamLayout(spec)
        self._device = None
        self._lr = learning_rate
        self._kl_weight = kl_weight
        # "natural" flat buffer (tensors row-major, back to back): the module tree below holds VIEWS into it; the
        # kernels' own tiled copy (Job.params) is refreshed from it before every launch
        self._flat = nn.Parameter(self.layout.nat_flatten(self.layout.init_reference_rule(
            int(torch.initial_seed() % (2 ** 31)))), requires_grad=False)
        self._job: Optional[Job] = None
        self._grads_ready = False
        self._pending = None
        self._last = None
        self._anchor = torch.zeros(1, requires_grad=True)
        self._build_tree()

    # -- module tree with the reference's parameter names (views into the flat buffer) --------------
    def _views(self):
        """name -> view into the natural flat buffer; built once per buffer (to() replaces the buffer and rebuilds)."""
        # (keyed on the storage, not on the Parameter object: nn.Module._apply -- .cuda(), .float(), .to(dtype) -- and
        #  `_flat.data = ...` swap the storage under the same Parameter)
        vc = self.__dict__.get("_view_cache")
        key = (self._flat.data_ptr(), self._flat.device, self._flat.dtype)
        if vc is None or vc[0] != key:
            vc = (key, self.layout.nat_views(self._flat.data))
            self.__dict__["_view_cache"] = vc
            self.__dict__["_nat_grads"] = None
            self.__dict__["_grad_views"] = None
        return vc[1]

    def _build_tree(self):
        self.__dict__["_view_cache"] = None
        self.__dict__["_nat_grads"] = None
        self.__dict__["_grad_views"] = None
        v = {k: nn.Parameter(t, requires_grad=False) for k, t in self._views().items()}
        s, L = self.spec, len(self.spec.hidden)
        if s.is_dm:                                 # encoder_list.{m}.fc1 ... / decoder_list.{m}.fc_out (cVAE.py:1453-1479)
            def stack(prefix, layers):
                mods = []
                for m in range(s.M):
                    mod = nn.Module()
                    for l in layers:
                        mod.add_module(l, _Holder(weight=v[f"{prefix}.{m}.{l}.weight"], bias=v[f"{prefix}.{m}.{l}.bias"]))
                    mods.append(mod)
                return nn.ModuleList(mods)
            if "weights" in v:
                self.register_parameter("weights", v["weights"])
            self.encoder_list = stack("encoder_list", ("fc1", "fc2", "fc_mu", "fc_logvar"))
            self.decoder_list = stack("decoder_list", ("fc1", "fc2", "fc_out"))
            return

        def enc(m):
            p = s.enc_prefix(m)
            layers = nn.ModuleList([_Holder(weight=v[f"{p}encoder_layers.{i}.weight"], bias=v[f"{p}encoder_layers.{i}.bias"])
                                    for i in range(L)])
            e = nn.Module()
            e.encoder_layers = layers
            e.enc_mean_layer = _Holder(weight=v[f"{p}enc_mean_layer.weight"], bias=v[f"{p}enc_mean_layer.bias"])
            e.enc_logvar_layer = _Holder(weight=v[f"{p}enc_logvar_layer.weight"], bias=v[f"{p}enc_logvar_layer.bias"])
            return e

        def dec(m):
            p = s.dec_prefix(m)
            d = nn.Module()
            d.register_parameter("logvar_out", v[f"{p}logvar_out"])
            d.decoder_layers = nn.ModuleList([_Holder(weight=v[f"{p}decoder_layers.{i}.weight"],
                                                      bias=v[f"{p}decoder_layers.{i}.bias"]) for i in range(L)])
            d.decoder_mean_layer = _Holder(weight=v[f"{p}decoder_mean_layer.weight"], bias=v[f"{p}decoder_mean_layer.bias"])
            return d

        if s.kind == "single":
            self.encoder, self.decoder = enc(0), dec(0)
        else:
            self.alpha_m_list = nn.ParameterList([v[f"alpha_m_list.{m}"] for m in range(s.M)])
            self.encoder_list = nn.ModuleList([enc(m) for m in range(s.M)])
            self.decoder_list = nn.ModuleList([dec(m) for m in range(s.M)])
        if s.kind == "regression":                # nn.Sequential(Linear, ReLU, Linear, ReLU, Linear), cVAE.py:2249-2253
            self.regressor = nn.Module()
            for i in (0, 2, 4):
                self.regressor.add_module(str(i), _Holder(weight=v[f"regressor.{i}.weight"], bias=v[f"regressor.{i}.bias"]))

    def state_dict(self, *a, **k):
        return {n: t.detach().cpu().clone() for n, t in self._views().items()}

    def load_state_dict(self, state, strict: bool = True):
        self._flat.data.copy_(self.layout.nat_flatten(state, device=self._flat.device))
        return self

    def to(self, device):
        dev = require_gpu(device)
        if self._device != dev:
            self._flat.data = self._flat.data.to(dev)
            self.__dict__["_view_cache"] = None
            self.__dict__["_nat_grads"] = None
            self.__dict__["_grad_views"] = None
            self._build_tree()
            self._device = dev
            self._job = None
        return self

    # -- one launch: forward + ELBO + backward --------------------------------------------------------
    def _ensure_device(self, like: torch.Tensor):
        if self._device is None:
            self.to(like.device if like.is_cuda else "cuda:0")

    def _draw(self, B: int) -> torch.Tensor:
        """The reparameterisation draw (torch.randn_like(mu), cVAE.py:1132).  Tests inject a fixed draw
        through ``_eps_override``."""
        ov = getattr(self, "_eps_override", None)
        if ov is not None:
            return torch.as_tensor(ov, dtype=torch.float32).to(self._dev())
        return torch.randn(B, self.spec.latent, device=self._dev())

    def _dev(self):
        if self._device is None:
            self.to("cuda:0")          # raises NmError when no MI355X is visible
        return self._device

    def _run(self, xes: Sequence[torch.Tensor], cs: Sequence[torch.Tensor], combine: str, flags: int, eps=None,
             kl_w=None, ll_w=1.0):
        self._ensure_device(xes[0])
        # One Job, one JobSet, one set of table buffers from call to call: a batch of the same shape is packed into the
        # existing tables (Table.repack) and the draw into the existing buffer, so the descriptor on the device stays
        # valid and a call costs its kernels, not a re-upload (2.16 ms per step, host-bound).
        j = self._job
        reuse = (j is not None and j.combine == combine.lower() and len(j.tables) == len(xes)
                 and all(t.repack(x, c) for t, x, c in zip(j.tables, xes, cs)))
        if not reuse:
            tables = [Table(x, c, self._device) for x, c in zip(xes, cs)]
            if j is None or j.combine != combine.lower() or j.tables[0].rows_alloc != tables[0].rows_alloc:
                j = self._job = Job(self.spec, tables, combine=combine, state=_Base.state_dict(self), lr=self._lr,
                                    kl_weight=self._kl_weight, loss_cap=1,
                                    single_bypass=getattr(self, "_single_bypass", self.spec.kind != "endtoend"))
                j.enable_exports()
                self._js = JobSet([j])
            j.tables = tables
            j.touch()
        self.layout.nat_to_kernel(self._flat.data, j.params)     # the module's parameters (views, maybe edited) -> kernel layout
        j.params_changed()
        if int(all(t.c_key == j.tables[0].c_key for t in j.tables)) != getattr(j, "_shared_cov", -1):
            j.touch()                                              # (the decoders share z | c | 1 only while the covariates are shared)
        kl_new = self._kl_weight if kl_w is None else kl_w
        ll_new = getattr(self, "_ll_weight", 1.0) if ll_w == 1.0 else ll_w
        if (j.kl_weight, j.ll_weight) != (kl_new, ll_new):
            j.kl_weight, j.ll_weight = kl_new, ll_new
            j.touch()
        B, Z = int(xes[0].shape[0]), self.spec.latent
        nt = j.tables[0].n_tiles
        if nt > 1 and (flags & (_lib.NM_F_BACKWARD | _lib.NM_F_GRADS)):
            raise ValueError(f"a train step takes one batch of at most {_lib.NM_BATCH} rows, got {B} "
                             f"(forward-only calls -- pred_recon, pred_latent, encode, decode -- take any number)")
        if eps is None:
            eps = torch.randn(B, Z, device=self._device)            # torch.randn_like(mu), cVAE.py:1132
        # forward-only calls over more than one 256-row tile: tile t runs as step t and reads draw block t
        if j.eps is None or tuple(j.eps.shape) != (nt, _lib.NM_BATCH, Z):
            j.set_eps(torch.zeros(nt, _lib.NM_BATCH, Z, dtype=torch.float32, device=self._device))
        j.eps.view(nt * _lib.NM_BATCH, Z)[:B].copy_(torch.as_tensor(eps, dtype=torch.float32).reshape(B, Z))
        if j.step != 0:
            j.step = 0
            j.touch()
        if (flags & _lib.NM_F_BACKWARD) and nt == 1 and self.spec.kind == "multimodal" and self._js.split_parts() > 1:
            # (the row-split launch would cut this kernel from ~210 to ~105 us, measured -- and leave the step where it is: the
            #  loop through this class costs ~0.45 ms of HOST time per step, tools/facade_host_time.py)
            # One workgroup per modality (bit-identical, ~2.3x faster).  Its workgroups wait for each other; if a hand-off times
            # out (another stream or process holding CUs) they leave the launch BEFORE the loss row is written, and what sits in
            # loss_log / grads is the previous call's.  Reading the error words here would stall the stream every step, so the
            # loss row is poisoned first: a launch that did not finish leaves NaN, which every loss this call returns -- and,
            # through _publish_grads, every gradient -- then carries; the NmError itself is raised by the next call's upload.
            j.loss_log[0].fill_(float("nan"))
            self._js._launch_split(0, 1, flags | getattr(self, "_fault_inject", 0))
        else:
            self._js._launch(0, 1, nt, flags)
        return j, B

    def _publish_grads(self, which: str, g: torch.Tensor):
        xes, cs, combine, eps = self._last
        if which == "total":
            pend = self._job.grads
        else:       # d kl / d theta or d ll / d theta on their own: one more launch with the other term off
            kl_w, ll_w = (self._kl_weight, 0.0) if which == "kl" else (0.0, -1.0)
            j, _ = self._run(xes, cs, combine, _lib.NM_F_BACKWARD | _lib.NM_F_GRADS, eps=eps, kl_w=kl_w, ll_w=ll_w)
            pend = j.grads
        # (the upstream gradient stays on the device: reading it as a Python float would stall the stream every step;
        #  times 1 or NaN: NaN if the launch that produced `pend` left its loss row poisoned, see _run)
        ok = self._job.loss_log[0, 0] * 0.0 + 1.0
        self._pending = pend * (g.to(pend.device).reshape(()) * ok)
        self._grads_ready = True
        self._assign_grads()

    def _assign_grads(self):
        """p.grad of every parameter = a view of ONE buffer in the module's (natural) layout, refreshed in place from the
        kernel-layout gradients: two index kernels per backward instead of one un-tiling per tensor."""
        if getattr(self, "_nat_grads", None) is None or self._nat_grads.device != self._pending.device:
            self._nat_grads = torch.zeros(self.layout.nat_total, dtype=torch.float32, device=self._pending.device)
            gv = self.layout.nat_views(self._nat_grads)
            self._grad_views = [(p, gv[name]) for name, p in self._named_views() if name in gv]
        self.layout.kernel_to_nat(self._pending, self._nat_grads)
        for p, v in self._grad_views:
            p.grad = v

    def _named_views(self):
        out = []
        for name, p in self.named_parameters():
            if name != "_flat":
                out.append((name, p))
        return out

    def zero_grad(self, set_to_none: bool = True):
        """nn.Module.zero_grad walks named_parameters() -- 0.15 ms for this module tree, a third of a step of the reference's
        loop through this class.  The only gradients this module ever holds are the views _assign_grads set: drop those."""
        gv = self.__dict__.get("_grad_views")
        if gv is None or not set_to_none:
            return super().zero_grad(set_to_none)
        for p, _ in gv:
            p.grad = None

    def reparameterise(self, mu, logvar):
        return mu + torch.randn_like(mu) * torch.exp(0.5 * logvar)          # cVAE.py:418-421

    def calc_kl(self, mu, logvar):
        return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1).mean(0)

    def calc_ll(self, x, x_recon):
        return x_recon.log_prob(x).sum(1, keepdims=True).mean(0)

    def sample_from_normal(self, normal):
        return normal.loc

    def _scale(self, m: int) -> torch.Tensor:
        return self._views()[f"{self.spec.dec_prefix(m)}logvar_out"].exp().pow(0.5)

    def _loss_dict(self, B):
        row = self._job.loss_log[0]
        mk = lambda which, val: _LossFn.apply(self._anchor, self, which, val)
        return {"total": mk("total", row[_lib_loss("TOTAL")]), "kl": mk("kl", row[_lib_loss("KL")]),
                "ll": mk("ll", row[_lib_loss("LL")].reshape(1))}


def _lib_loss(which):
    return {"TOTAL": 0, "KL": 1, "LL": 2}[which]


class _ExpertOps:
    """The reference's public expert-fusion methods (cVAE.py:1118-1126, 1144-1164; :2265-2307 on the regression class)
    as forward-only launches of nm_combine_latent: tensors [M, B, Z] in, (mu, variance) [B, Z] on the device out."""
    _fuse_bypass = True          # combine_latent returns a single expert as it is (cVAE.py:1146-1147)
    _fuse_floor = 0.0
    _bad_combine = "No such combination method"

    def _fuse(self, mus, variances, combine: str, in_log: bool = False, out_log: bool = False, bypass: bool = False,
              floor: float = 0.0):
        dev = self._dev()
        mu = torch.as_tensor(mus, dtype=torch.float32).to(dev).contiguous()
        var = torch.as_tensor(variances, dtype=torch.float32).to(dev).contiguous()
        if mu.shape != var.shape or mu.dim() < 2:
            raise ValueError(f"mus / variances must be [M, ...] tensors of equal shape, got {tuple(mu.shape)} / {tuple(var.shape)}")
        M, n = int(mu.shape[0]), int(mu[0].numel())
        # The launch builds no autograd graph.  The reference's methods are plain differentiable torch (cVAE.py:986-1083,
        # 1144-1164), so inputs that carry a graph take the same arithmetic as a torch expression: rare -- the
        # train step differentiates inside the kernel -- but then silently detaching would be wrong.
        if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in (mus, variances)):
            var_t = torch.exp(var) if in_log else var
            if M == 1 and bypass:
                return mu[0], var_t[0]
            if combine == "moe":
                omu, ovar = mu.mean(0), var_t.mean(0)
            else:
                w = 1.0 / var_t
                if combine == "gpoe":
                    if M != self.modalities:
                        raise ValueError(f"gpoe needs one expert per modality ({self.modalities}), got {M}")
                    al = torch.softmax(torch.cat([self._views()[f"alpha_m_list.{m}"].reshape(1) for m in range(M)]).to(dev), 0)
                    w = al.reshape((M,) + (1,) * (mu.dim() - 1)) * w
                S = w.sum(0)
                omu, ovar = (mu * w).sum(0) / S, 1.0 / S
                if combine == "mopoe":
                    omu, ovar = (mu.sum(0) + omu) / (M + 1), (var_t.sum(0) + ovar) / (M + 1)
            if out_log:
                ovar = torch.log(ovar)
            if floor > 0:
                ovar = torch.clamp(ovar, min=floor)
            return omu, ovar
        alpha = None
        if combine == "gpoe":                       # softmax(alpha_m_list) inside the kernel (cVAE.py:1155)
            alpha = torch.cat([self._views()[f"alpha_m_list.{m}"].reshape(1) for m in range(self.modalities)]).to(dev).contiguous()
            if M != self.modalities:
                raise ValueError(f"gpoe needs one expert per modality ({self.modalities}), got {M}")
        out_mu, out_var = torch.empty_like(mu[0]), torch.empty_like(mu[0])
        lib = _lib.load()
        _lib.check(lib.nm_combine_latent(mu.data_ptr(), var.data_ptr(), M, n, _lib.NM_COMBINE[combine],
                                         alpha.data_ptr() if alpha is not None else None, int(bypass), int(in_log), int(out_log),
                                         float(floor), out_mu.data_ptr(), out_var.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream), "nm_combine_latent")
        return out_mu, out_var

    def product_of_experts(self, mus, variances):
        return self._fuse(mus, variances, "poe")

    def mixture_of_experts(self, mus, variances):
        return self._fuse(mus, variances, "moe")

    def mixture_of_product_of_experts(self, mus, variances):
        return self._fuse(mus, variances, "mopoe")

    def combine_latent(self, mus, variances, combine):
        combine = combine.lower()
        if self._fuse_bypass and int(mus.shape[0]) == 1:
            combine = "poe"                          # (the bypass returns before the reference looks at `combine`)
        if combine not in ("poe", "gpoe", "moe", "mopoe"):
            raise ValueError(self._bad_combine.format(combine=combine))
        return self._fuse(mus, variances, combine, bypass=self._fuse_bypass, floor=self._fuse_floor)


class cVAE_multimodal(_ExpertOps, _Base):
    """cVAE.py:1087-1211."""

    def __init__(self, input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate=0.0001, modalities=3,
                 non_linear=False):
        super().__init__()
        if modalities != len(input_dim_list):
            raise ValueError("modalities must equal len(input_dim_list)")
        self.input_dim_list = list(input_dim_list)
        self.hidden_dim = list(hidden_dim) + [latent_dim]
        self.latent_dim, self.c_dim, self.modalities, self.learning_rate = latent_dim, c_dim, modalities, learning_rate
        self._setup(ModelSpec(list(input_dim_list), list(hidden_dim), latent_dim, c_dim, non_linear, "multimodal"),
                    learning_rate, kl_weight=float(modalities))
        self.optimizer1 = _Adam(self, learning_rate)

    def forward_multimodal(self, xes, cs, combine):
        if combine.lower() not in _lib.NM_COMBINE:
            raise ValueError("No such combination method")
        self.zero_grad()
        eps = self._draw(int(xes[0].shape[0]))
        j, B = self._run(xes, cs, combine, _lib.NM_F_BACKWARD | _lib.NM_F_GRADS | _lib.NM_F_EXPORT, eps=eps)
        self._last = (list(xes), list(cs), combine, eps)
        x_recons = [NormalLike(j.out_loc[m][:B].clone(), self._scale(m)) for m in range(self.modalities)]
        return {"x_recons": x_recons, "mu_multimodal": j.out_mu[:B].clone(), "logvar_multimodal": j.out_logvar[:B].clone()}

    def loss_function_multimodal(self, xes, fwd_rtn):
        return self._loss_dict(int(xes[0].shape[0]))

    def encode(self, x, c, m):
        """Encoder m alone: (mu, logvar) -- forward-only launch of a one-modality view."""
        one = self._unimodal(m)
        j, B = one._run([x], [c], "poe", _lib.NM_F_EXPORT, eps=torch.zeros(int(x.shape[0]), self.spec.latent,
                                                                          device=self._dev()))
        return j.out_mu[:B].clone(), j.out_logvar[:B].clone()

    def decode(self, z, c, m):
        """Decoder m alone on a given z (cVAE.py:1135)."""
        one = self._unimodal(m)
        x0 = torch.zeros(int(z.shape[0]), self.input_dim_list[m], device=self._dev())
        j, B = one._run([x0], [c], "poe", _lib.NM_F_EXPORT | _lib.NM_F_ZGIVEN, eps=z)
        return NormalLike(j.out_loc[0][:B].clone(), self._scale(m))

    def _unimodal(self, m: int) -> "cVAE_multimodal":
        self._dev()
        cache = self.__dict__.setdefault("_uni", {})
        if m not in cache:
            u = cVAE_multimodal([self.input_dim_list[m]], self.spec.hidden, self.latent_dim, self.c_dim,
                                self.learning_rate, 1, self.spec.non_linear)
            u.to(self._device)
            cache[m] = u
        u = cache[m]
        sd = self.state_dict()
        u.load_state_dict({k: sd[k.replace("_list.0.", f"_list.{m}.")] for k in u.layout.names})
        return u

    def pred_recon(self, xes, c, DEVICE, combine):
        """cVAE.py:1198-1208: DataFrames in, numpy reconstructions (joint latent, sampled z) out."""
        xs = [torch.tensor(np.asarray(x.values if hasattr(x, "values") else x), dtype=torch.float32) for x in xes]
        ct = torch.tensor(np.asarray(c), dtype=torch.long)
        self._dev()
        j, B = self._run(xs, [ct] * self.modalities, combine, _lib.NM_F_EXPORT)
        return [j.out_loc[m][:B].cpu().numpy() for m in range(self.modalities)]

    def reconstruction_deviation_multimodal(self, xes, x_preds):
        return [np.sum((xes[m] - x_preds[m]) ** 2, axis=1) / xes[m].shape[1] for m in range(self.modalities)]


class DMVAE(_Base):
    """cVAE.py:1491-1598 (baseline zoo) on the step kernel: covariate-free ReLU encoders, the first
    min(c_dim, latent) latent columns private to their modality (handed to its decoder as they are), the rest fused by
    ProductOfExperts2 and sampled; sigmoid decoders with ll = -0.5 sum (x - x_hat)^2; total = beta * kl - ll with the KL
    counted once per modality.  `x_recons` are plain tensors here, as in the reference."""
    _kind, _beta = "dmvae", 1.0

    def __init__(self, input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate=0.0001, modalities=3,
                 non_linear=False):
        super().__init__()
        if modalities != len(input_dim_list):
            raise ValueError("modalities must equal len(input_dim_list)")
        self.input_dim_list, self.hidden_dim = list(input_dim_list), list(hidden_dim)
        self.latent_dim, self.c_dim, self.s_dim, self.beta = latent_dim, c_dim, c_dim, self._beta
        self.modalities, self.learning_rate, self.non_linear = modalities, learning_rate, non_linear
        self._setup(ModelSpec(list(input_dim_list), list(hidden_dim), latent_dim, c_dim, True, self._kind), learning_rate,
                    kl_weight=float(modalities) * self._beta)
        self._single_bypass = False
        self.optimizer1 = _Adam(self, learning_rate)

    def _tables_c(self, n):
        return torch.zeros(n, 0)                    # the networks take no covariates (encode(x, c, m) ignores c)

    def forward_multimodal(self, xes, cs, combine):
        self.zero_grad()
        B = int(xes[0].shape[0])
        eps = self._draw(B)
        cz = [self._tables_c(B)] * self.modalities
        j, B = self._run(xes, cz, "poe", _lib.NM_F_BACKWARD | _lib.NM_F_GRADS | _lib.NM_F_EXPORT, eps=eps)
        self._last = (list(xes), cz, "poe", eps)
        zc = self.latent_dim - self.spec.n_private
        return {"x_recons": [j.out_loc[m][:B].clone() for m in range(self.modalities)],
                "mu_c": j.out_mu[:B, :zc].clone(), "logvar_c": j.out_logvar[:B, :zc].clone()}

    def loss_function_multimodal(self, xes, fwd_rtn):
        row = self._job.loss_log[0]
        total = _LossFn.apply(self._anchor, self, "total", row[0])
        # losses['kl'] is the sum over the modalities BEFORE beta (cVAE.py:1569-1574); the log holds beta * that
        return {"total": total, "kl": (row[1] / self.beta).clone(), "ll": row[2].clone()}

    def pred_recon(self, xes, cs, device, combine):
        xs = [torch.tensor(np.asarray(x.values if hasattr(x, "values") else x), dtype=torch.float32) for x in xes]
        self._dev()
        j, B = self._run(xs, [self._tables_c(xs[0].shape[0])] * self.modalities, "poe", _lib.NM_F_EXPORT)
        return [j.out_loc[m][:B].cpu().numpy() for m in range(self.modalities)]

    def reparameterize(self, mu, logvar):
        return self.reparameterise(mu, logvar)

    def reconstruction_deviation_multimodal(self, xes, x_preds):
        out = []
        for m in range(self.modalities):
            x = np.asarray(xes[m].values if hasattr(xes[m], "values") else xes[m])
            out.append(np.sum((x - np.asarray(x_preds[m])) ** 2, axis=1) / x.shape[1])
        return out


class mmVAEPlus(DMVAE):
    """cVAE.py:1895-2002: DMVAE with beta = 0.05."""
    _kind, _beta = "mmvaeplus", 0.05


class WeightedDMVAE(DMVAE):
    """cVAE.py:1620-1747: DMVAE whose kl_i and ll_i are multiplied by learnable `weights[i]` (initialised
    |N(0, 1)|).  The reference prints weight / KL / LL per modality from inside loss_function_multimodal (:1702);
    that print is not reproduced."""
    _kind = "weighted_dmvae"

    def loss_function_multimodal(self, xes, fwd_rtn):
        row = self._job.loss_log[0]
        total = _LossFn.apply(self._anchor, self, "total", row[0])
        return {"total": total, "kl": row[1].clone(), "ll": row[2].clone()}


class mvtCAE(cVAE_multimodal):
    """cVAE.py:1754-1893 (baseline zoo) on the step kernel: cVAE_multimodal's encoders / decoders /
    alpha_m_list; no single-expert bypass; `combine='poe'` is ProductOfExperts2 fed with the VARIANCES where it expects
    log variances (:1782-1783 -- kept as written: NM_COMBINE_POE2V); the joint variance is clamped at 1e-6; the loss is
    sum_i [kl + 1e-5 ll_i + 1e-4 tc] (the log-likelihood with a plus sign, :1877), tc = the total-correlation term of
    :1862-1869 whose joint half is identically zero.  `qz_xs` = the experts' stacked means, as the reference returns them
    (read back from the kernel's workspace); the loss itself is formed inside the kernel."""

    def __init__(self, input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate=0.0001, modalities=3, non_linear=False):
        super().__init__(input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate, modalities, non_linear)
        self.beta = 0.0001
        self._setup(ModelSpec(list(input_dim_list), list(hidden_dim), latent_dim, c_dim, non_linear, "mvtcae"), learning_rate,
                    kl_weight=float(modalities))
        self._single_bypass = False
        self._ll_weight = -1e-5
        self.optimizer1 = _Adam(self, learning_rate)

    _fuse_bypass = False         # cVAE.py:1800-1825: no single-expert bypass, joint variance clamped at 1e-6
    _fuse_floor = 1e-6

    @staticmethod
    def _kernel_combine(combine):
        if combine.lower() not in ("poe", "gpoe", "moe", "mopoe"):
            raise ValueError("No such combination method")
        return "poe2v" if combine.lower() == "poe" else combine

    def product_of_experts(self, mus, variances):
        """ProductOfExperts2 applied to what it is given (cVAE.py:1782-1783, 1481-1489): the second argument is read as
        LOG variances and a log variance comes back -- the reference's own call passes variances; kept as written."""
        return self._fuse(mus, variances, "poe", in_log=True, out_log=True)

    def combine_latent(self, mus, variances, combine):
        combine = combine.lower()
        if combine not in ("poe", "gpoe", "moe", "mopoe"):
            raise ValueError("No such combination method")
        if combine == "poe":
            return self._fuse(mus, variances, "poe", in_log=True, out_log=True, floor=self._fuse_floor)
        return self._fuse(mus, variances, combine, floor=self._fuse_floor)

    def total_correlation(self, qz_xs, qz_x):
        """cVAE.py:1859-1866 on the device (nm_total_correlation).  qz_xs: [M, B, Z] tensor or list of [B, Z]; `qz_x`
        enters the reference's expression only as a scalar minus its own mean, i.e. not at all."""
        dev = self._dev()
        q = (torch.stack([torch.as_tensor(t, dtype=torch.float32) for t in qz_xs]) if isinstance(qz_xs, (list, tuple))
             else torch.as_tensor(qz_xs, dtype=torch.float32)).to(dev).contiguous()
        out = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().nm_total_correlation(q.data_ptr(), int(q.shape[0]), int(q.shape[1]), int(q.shape[2]),
                                                    out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "nm_total_correlation")
        return out[0]

    def forward_multimodal(self, xes, cs, combine):
        out = super().forward_multimodal(xes, cs, self._kernel_combine(combine))
        # 'qz_xs': the stacked per-expert means [M, B, Z] (cVAE.py:1845) -- the fused kernels leave them in the job's workspace
        j = self._job
        off = int(_lib.load().nm_workspace_offset(C.byref(j.struct()), 0))
        if off >= 0:
            M, Z, Zs, B = self.spec.M, self.spec.latent, (self.spec.latent + 15) // 16 * 16, int(out["mu_multimodal"].shape[0])
            raw = j._ws[off: off + M * _lib.NM_BATCH * Zs * 4].view(torch.float32).view(M, _lib.NM_BATCH, Zs)
            out["qz_xs"] = raw[:, :B, :Z].clone()
        else:
            out["qz_xs"] = None
        out["qz_x"] = out["mu_multimodal"]
        return out

    def loss_function_multimodal(self, xes, fwd_rtn):
        row = self._job.loss_log[0]
        total = _LossFn.apply(self._anchor, self, "total", row[0])
        return {"total": total, "kl": row[1].clone(), "ll": row[2].clone().reshape(1),
                "tc": (row[_lib.NM_LOSS_TC] * self.modalities).clone()}

    def pred_recon(self, xes, c, DEVICE, combine):
        return super().pred_recon(xes, c, DEVICE, self._kernel_combine(combine))


class mmJSD(cVAE_multimodal):
    """cVAE.py:1354-1448 (baseline zoo).  Same encoders / decoders / `alpha_m_list` and the same
    ELBO sum as cVAE_multimodal; the latent is always the plain product of experts (`combine_latent` ignores the
    `combine` argument and has no single-expert bypass), and the JSD term of its loss compares the joint posterior
    with itself (`multimodal_jsd([mu_multimodal] * M, ...)`, :1426), i.e. it is identically zero with zero
    gradient.  So the step kernel runs it unchanged: PoE, bypass off; `alpha_m_list` receives no gradient."""

    def __init__(self, input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate=0.0001, modalities=3, non_linear=False):
        super().__init__(input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate, modalities, non_linear)
        self._single_bypass = False

    def combine_latent(self, mus, logvars):
        """cVAE.py:1399-1402: plain product of experts on LOG variances, the joint variance comes back; no bypass."""
        return self._fuse(mus, logvars, "poe", in_log=True)

    def forward_multimodal(self, xes, cs, combine):
        return super().forward_multimodal(xes, cs, "poe")

    def pred_recon(self, xes, c, DEVICE, combine):
        return super().pred_recon(xes, c, DEVICE, "poe")

    def reparameterize(self, mu, logvar):
        return self.reparameterise(mu, logvar)


class cVAE(_Base):
    """cVAE.py:391-562 (encoder + decoder; the unused discriminator is not part of the hot path)."""

    def __init__(self, input_dim, hidden_dim, latent_dim, c_dim, learning_rate=0.0001, modalities=4, non_linear=False):
        super().__init__()
        self.input_dim, self.latent_dim, self.c_dim, self.learning_rate = input_dim, latent_dim, c_dim, learning_rate
        self.hidden_dim = list(hidden_dim) + [latent_dim]
        self.modalities = modalities
        self._setup(ModelSpec([input_dim], list(hidden_dim), latent_dim, c_dim, non_linear, "single"), learning_rate, 1.0)
        self.optimizer1 = _Adam(self, learning_rate)

    def forward(self, x, c):
        self.zero_grad()
        eps = self._draw(int(x.shape[0]))
        j, B = self._run([x], [c], "poe", _lib.NM_F_BACKWARD | _lib.NM_F_GRADS | _lib.NM_F_EXPORT, eps=eps)
        self._last = ([x], [c], "poe", eps)
        return {"x_recon": NormalLike(j.out_loc[0][:B].clone(), self._scale(0)), "mu": j.out_mu[:B].clone(),
                "logvar": j.out_logvar[:B].clone()}

    def loss_function(self, x, fwd_rtn):
        return self._loss_dict(int(x.shape[0]))

    def encode(self, x, c):
        j, B = self._run([x], [c], "poe", _lib.NM_F_EXPORT)
        return j.out_mu[:B].clone(), j.out_logvar[:B].clone()

    def decode(self, z, c):
        x0 = torch.zeros(int(z.shape[0]), self.input_dim, device=self._dev())
        j, B = self._run([x0], [c], "poe", _lib.NM_F_EXPORT | _lib.NM_F_ZGIVEN, eps=z)
        return NormalLike(j.out_loc[0][:B].clone(), self._scale(0))

    def pred_latent(self, x, c, DEVICE):
        """cVAE.py:539-545."""
        xt = torch.as_tensor(np.asarray(x.to_numpy() if hasattr(x, "to_numpy") else x), dtype=torch.float32)
        mu, logvar = self.encode(xt, torch.as_tensor(np.asarray(c), dtype=torch.long))
        return mu.cpu().numpy(), logvar.exp().cpu().numpy()

    def pred_recon(self, x, c, DEVICE):
        """cVAE.py:547-553: decodes mu (no draw)."""
        xt = torch.as_tensor(np.asarray(x.to_numpy() if hasattr(x, "to_numpy") else x), dtype=torch.float32)
        ct = torch.as_tensor(np.asarray(c), dtype=torch.long)
        mu, _ = self.encode(xt, ct)
        return self.decode(mu, ct).loc.cpu().numpy()


# =========================================================================================================
# Models with a head on top of the trunk: cVAE_multimodal_regression, cVAE_multimodal_endtoend
# =========================================================================================================
# The trunk (encoders, fusion, decoders, ELBO, their backward and Adam) is the step kernel; the heads are their
# own HIP kernels (nm_head_regression, nm_head_classifier).  They depend on the trunk only through x_hat, z and
# the per-subject deviations, so they hand back extra loss gradients (nm_modality_t.dloc_extra / dloc_rowcoef,
# nm_job_t.dz_extra) that the trunk's backward launch adds to the ELBO's.


class _HeadBase(_Base):
    """Models with a head on top of the trunk (regressor / classifier).  The head is its own HIP kernel that
    runs between two trunk launches: the first exports x_hat / z / per-subject deviations, the head turns them
    into its loss and into extra gradients on x_hat and z (nm_modality_t.dloc_extra / dloc_rowcoef,
    nm_job_t.dz_extra), the second runs the trunk's forward + backward with those added."""

    def _named_views(self):
        return [(n, p) for n, p in self.named_parameters() if n in self.layout.offsets]


class _RegTotal(torch.autograd.Function):
    """total = ELBO part + lambda * MSE; backward runs the head kernel (d MSE / d x_hat, regressor gradients)
    and then the trunk launch that adds that extra gradient to the ELBO's."""

    @staticmethod
    def forward(ctx, anchor, model, value):
        ctx.model = model
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.model._backward_native(float(g))
        return None, None, None


class cVAE_multimodal_regression(_ExpertOps, _HeadBase):
    """cVAE.py:2211-2346: cVAE_multimodal + a regressor on the concatenated residuals.  Trunk and regressor
    both run in HIP (nm_launch + nm_head_regression); the regressor's tensors live in the same flat
    parameter buffer under the reference's names regressor.{0,2,4}.{weight,bias}."""

    def __init__(self, input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate=0.0001, modalities=3,
                 non_linear=False):
        super().__init__()
        self.input_dim_list, self.latent_dim, self.c_dim = list(input_dim_list), latent_dim, c_dim
        self.hidden_dim = list(hidden_dim) + [latent_dim]
        self.modalities, self.learning_rate, self.non_linear = modalities, learning_rate, non_linear
        self._setup(ModelSpec(list(input_dim_list), list(hidden_dim), latent_dim, c_dim, non_linear, "regression"),
                    learning_rate, kl_weight=float(modalities))
        self.mse_loss = nn.MSELoss()
        self.optimizer1 = _Adam(self, learning_rate)

    _bad_combine = "Invalid combine strategy: {combine}"         # cVAE.py:2307

    def forward_multimodal(self, xes, cs, combine):
        if combine.lower() not in _lib.NM_COMBINE:
            raise ValueError("No such combination method")
        eps = self._draw(int(xes[0].shape[0]))
        j, B = self._run(xes, cs, combine, _lib.NM_F_EXPORT, eps=eps)
        j.fi_target = None
        j.touch()
        JobSet([j]).head_regression(backward=False)                                     # cVAE.py:2318-2321
        self._last = (list(xes), list(cs), combine, eps)
        return {"x_recons": [NormalLike(j.out_loc[m][:B].clone(), self._scale(m)) for m in range(self.modalities)],
                "mu_multimodal": j.out_mu[:B].clone(), "logvar_multimodal": j.out_logvar[:B].clone(),
                "fi_pred": j.out_fi_pred[:B].clone().reshape(B, 1)}

    def loss_function_multimodal(self, xes, fwd_rtn, true_fi, lambda_reg=1.0):
        j = self._job
        row = j.loss_log[0]
        fi = torch.as_tensor(true_fi, dtype=torch.float32).to(self._device).reshape(-1)
        ra = j.tables[0].rows_alloc
        j.fi_target = torch.zeros(ra, device=self._device)
        j.fi_target[:fi.numel()] = fi
        self._lambda = float(lambda_reg)
        reg = self.mse_loss(fwd_rtn["fi_pred"].squeeze(), fi.squeeze())                 # cVAE.py:2340
        total = _RegTotal.apply(self._anchor, self, row[0] + lambda_reg * reg)
        return {"total": total, "kl": row[1].clone(), "ll": row[2].clone().reshape(1), "regression": reg}

    def _backward_native(self, g: float):
        j = self._job
        j.reg_lambda = self._lambda * g
        j.kl_weight, j.ll_weight = self._kl_weight * g, g
        j.step = 0
        j.touch()
        # one launch: trunk forward (same draws), head forward / backward, trunk backward with d MSE / d x_hat
        JobSet([j]).grads_head(0)
        j.kl_weight, j.ll_weight = self._kl_weight, 1.0
        j.touch()
        self._pending = j.grads
        self._grads_ready = True
        self._assign_grads()

    def encode(self, x, c, m):
        return cVAE_multimodal.encode(self, x, c, m)

    def decode(self, z, c, m):
        return cVAE_multimodal.decode(self, z, c, m)

    def _unimodal(self, m):
        return cVAE_multimodal._unimodal(self, m)


class _E2ETotal(torch.autograd.Function):
    """total_loss of the end-to-end model; backward runs the classifier head's backward (d CE / d z, hinge row
    coefficients, its own gradients) and then the trunk launch that adds them to the weighted ELBO gradient."""

    @staticmethod
    def forward(ctx, anchor, model, value):
        ctx.model = model
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.model._backward_native(float(g))
        return None, None, None


class cVAE_multimodal_endtoend(_HeadBase):
    """cVAE.py:2021-2207: shared encoders, PoE (no single-expert bypass), health + disease decoder banks,
    classifier on z, loss = w_rec (NLL_h + NLL_d) + w_kl KL + CE + w_c contrastive hinge.  Trunk (nm_launch) and
    classifier head (nm_head_classifier: Linear - BatchNorm1d - ReLU - Dropout blocks, cross entropy, hinge) both
    run in HIP; the classifier's tensors, BatchNorm running statistics included, live in the flat parameter
    buffer under the reference's names classifier.classifier.{i}.*."""

    def __init__(self, input_dim_list, hidden_dim, latent_dim, c_dim, learning_rate=0.0001, modalities=3,
                 non_linear=False, classifier_layers=[128, 64], dropout_rate=0.5, num_classes=2):
        super().__init__()
        self.input_dim_list, self.latent_dim, self.c_dim = list(input_dim_list), latent_dim, c_dim
        self.hidden_dim = list(hidden_dim) + [latent_dim]
        self.modalities, self.learning_rate, self.non_linear, self.num_classes = modalities, learning_rate, non_linear, num_classes
        self.dropout_rate = float(dropout_rate)
        self._nbt = [0] * len(classifier_layers)            # BatchNorm1d.num_batches_tracked
        self._setup(ModelSpec(list(input_dim_list), list(hidden_dim), latent_dim, c_dim, non_linear, "endtoend",
                              tuple(classifier_layers), num_classes), learning_rate, kl_weight=0.1)
        self._ll_weight = 0.1
        self._margin, self._wc = 1.0, 0.1
        self.optimizer = _Adam(self, learning_rate)

    def _build_tree(self):
        v = {k: nn.Parameter(t, requires_grad=False) for k, t in self._views().items()}
        s, L = self.spec, len(self.spec.hidden)
        if s.is_dm:                                 # encoder_list.{m}.fc1 ... / decoder_list.{m}.fc_out (cVAE.py:1453-1479)
            def stack(prefix, layers):
                mods = []
                for m in range(s.M):
                    mod = nn.Module()
                    for l in layers:
                        mod.add_module(l, _Holder(weight=v[f"{prefix}.{m}.{l}.weight"], bias=v[f"{prefix}.{m}.{l}.bias"]))
                    mods.append(mod)
                return nn.ModuleList(mods)
            if "weights" in v:
                self.register_parameter("weights", v["weights"])
            self.encoder_list = stack("encoder_list", ("fc1", "fc2", "fc_mu", "fc_logvar"))
            self.decoder_list = stack("decoder_list", ("fc1", "fc2", "fc_out"))
            return

        def enc(m):
            p = s.enc_prefix(m)
            e = nn.Module()
            e.encoder_layers = nn.ModuleList([_Holder(weight=v[f"{p}encoder_layers.{i}.weight"], bias=v[f"{p}encoder_layers.{i}.bias"])
                                              for i in range(L)])
            e.enc_mean_layer = _Holder(weight=v[f"{p}enc_mean_layer.weight"], bias=v[f"{p}enc_mean_layer.bias"])
            e.enc_logvar_layer = _Holder(weight=v[f"{p}enc_logvar_layer.weight"], bias=v[f"{p}enc_logvar_layer.bias"])
            return e

        def dec(m, bank):
            p = s.dec_prefix(m, bank)
            d = nn.Module()
            d.register_parameter("logvar_out", v[f"{p}logvar_out"])
            d.decoder_layers = nn.ModuleList([_Holder(weight=v[f"{p}decoder_layers.{i}.weight"],
                                                      bias=v[f"{p}decoder_layers.{i}.bias"]) for i in range(L)])
            d.decoder_mean_layer = _Holder(weight=v[f"{p}decoder_mean_layer.weight"], bias=v[f"{p}decoder_mean_layer.bias"])
            return d

        self.encoder_list = nn.ModuleList([enc(m) for m in range(s.M)])
        self.decoder_list_health = nn.ModuleList([dec(m, "health") for m in range(s.M)])
        self.decoder_list_disease = nn.ModuleList([dec(m, "disease") for m in range(s.M)])
        self.classifier = nn.Module()                          # Classifier.classifier = nn.Sequential(...), cVAE.py:2015
        seq = nn.Module()
        n = len(s.classifier_layers)
        for i in range(n):
            p = f"classifier.classifier.{4 * i}"
            seq.add_module(str(4 * i), _Holder(weight=v[f"{p}.weight"], bias=v[f"{p}.bias"]))
            q = f"classifier.classifier.{4 * i + 1}"
            seq.add_module(str(4 * i + 1), _Holder(weight=v[f"{q}.weight"], bias=v[f"{q}.bias"],
                                                   running_mean=v[f"{q}.running_mean"], running_var=v[f"{q}.running_var"]))
        p = f"classifier.classifier.{4 * n}"
        seq.add_module(str(4 * n), _Holder(weight=v[f"{p}.weight"], bias=v[f"{p}.bias"]))
        self.classifier.classifier = seq

    def state_dict(self, *a, **k):
        out = {}
        for name, t in _Base.state_dict(self).items():
            out[name] = t
            if name.endswith(".running_var"):                  # reference key order: ..., running_var, num_batches_tracked
                i = (int(name.split(".")[2]) - 1) // 4
                out[name[: -len("running_var")] + "num_batches_tracked"] = torch.tensor(self._nbt[i], dtype=torch.int64)
        return out

    def load_state_dict(self, state, strict: bool = True):
        for k, v in state.items():
            if k.endswith("num_batches_tracked"):
                self._nbt[(int(k.split(".")[2]) - 1) // 4] = int(v)
        return _Base.load_state_dict(self, {k: v for k, v in state.items() if not k.endswith("num_batches_tracked")})

    def _scale_bank(self, m, bank):
        return self._views()[f"{self.spec.dec_prefix(m, bank)}logvar_out"].exp().pow(0.5)

    def _head_setup(self, j, use_mu: bool, labels=None):
        j.cls_train, j.cls_use_mu = bool(self.training), use_mu
        j.cls_dropout = self.dropout_rate
        j.cls_margin, j.cls_w_ce, j.cls_w_contrast = self._margin, 1.0, self._wc
        if labels is None:
            j.labels = None
        else:
            j.set_labels(labels)
        j.touch()

    def forward(self, xes, cs):
        eps = self._draw(int(xes[0].shape[0]))
        j, B = self._run(xes, cs, "poe", _lib.NM_F_EXPORT, eps=eps)
        self._head_setup(j, use_mu=False)
        JobSet([j]).head_classifier(backward=False, bn_stats=self.training)                  # cVAE.py:2117
        if self.training:
            self._nbt = [n + 1 for n in self._nbt]
        M = self.modalities
        self._last = (list(xes), list(cs), "poe", eps)
        return {"x_recons_health": [NormalLike(j.out_loc[m][:B].clone(), self._scale_bank(m, "health")) for m in range(M)],
                "x_recons_disease": [NormalLike(j.out_loc[M + m][:B].clone(), self._scale_bank(m, "disease")) for m in range(M)],
                "mu": j.out_mu[:B].clone(), "logvar": j.out_logvar[:B].clone(),
                "logits": j.out_logits[:B, : self.num_classes].clone()}

    def _unimodal(self, m: int, bank: str) -> "cVAE_multimodal":
        """One-modality view (encoder m + decoder m of one bank) for the stand-alone encode / decode calls."""
        self._dev()
        cache = self.__dict__.setdefault("_uni", {})
        if (m, bank) not in cache:
            u = cVAE_multimodal([self.input_dim_list[m]], self.spec.hidden, self.latent_dim, self.c_dim,
                                self.learning_rate, 1, self.spec.non_linear)
            u.to(self._device)
            cache[(m, bank)] = u
        u = cache[(m, bank)]
        sd = _Base.state_dict(self)
        st = {}
        for k in u.layout.names:
            if k.startswith("encoder_list.0."):
                st[k] = sd[k.replace("encoder_list.0.", f"encoder_list.{m}.")]
            elif k.startswith("decoder_list.0."):
                st[k] = sd[k.replace("decoder_list.0.", f"decoder_list_{bank}.{m}.")]
            else:
                st[k] = torch.zeros(u.layout.shapes[k])          # alpha of the one-expert view: unused (bypass)
        u.load_state_dict(st)
        return u

    def encode(self, xes, cs):
        """cVAE.py:2064-2076: every modality's own (mu, logvar), stacked [M, B, Z]."""
        mus, lvs = [], []
        for m in range(self.modalities):
            mu, lv = self._unimodal(m, "health").encode(xes[m], cs[m], 0)
            mus.append(mu); lvs.append(lv)
        return torch.stack(mus), torch.stack(lvs)

    def combine_latent(self, mus, logvars):
        """cVAE.py:2083-2090 (product of experts on log variances, a log variance back; no single-expert bypass)."""
        return _ExpertOps._fuse(self, mus, logvars, "poe", in_log=True, out_log=True)

    def decode(self, z, cs, group):
        """cVAE.py:2092-2104: every modality's reconstruction from one decoder bank."""
        if group not in ("health", "disease"):
            raise ValueError("group must be 'health' or 'disease'")
        return [NormalLike(self._unimodal(m, group).decode(z, cs[m], 0).loc, self._scale_bank(m, group))
                for m in range(self.modalities)]

    def calc_kl(self, mu, logvar):
        return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1).mean()

    def calc_recon_loss(self, x, x_recon):
        return -x_recon.log_prob(x.to(x_recon.loc.device)).sum(dim=1).mean()              # cVAE.py:2128-2132

    def compute_deviation(self, x, x_recon):
        return ((x - x_recon.mean) ** 2).mean(dim=1)                          # cVAE.py:2134-2138

    def loss_function(self, xes, fwd_rtn, labels, margin=1.0, weightcontrastive=0.1, weight_kl=0.1, weight_rec=0.1):
        j, M = self._job, self.modalities
        self._margin, self._wc = float(margin), float(weightcontrastive)
        self._kl_weight, self._ll_weight = float(weight_kl), float(weight_rec)
        # cross entropy and hinge come from the head kernel (same dropout mask: the generator is keyed by the step)
        self._head_setup(j, use_mu=False, labels=labels)
        JobSet([j]).head_classifier(backward=False)
        row = j.loss_log[0]
        ll_m = row[3:3 + 2 * M]
        recon_h, recon_d = -ll_m[:M].sum(), -ll_m[M:].sum()
        kl = self.calc_kl(fwd_rtn["mu"], fwd_rtn["logvar"])
        ce, contrastive = row[_lib.NM_LOSS_CE].clone(), row[_lib.NM_LOSS_CONTRAST].clone()
        value = weight_rec * (recon_h + recon_d) + weight_kl * kl + ce + weightcontrastive * contrastive
        total = _E2ETotal.apply(self._anchor, self, value.detach())
        return {"total_loss": total, "recon_loss_health": recon_h.clone(), "recon_loss_disease": recon_d.clone(),
                "kl_loss": kl, "classification_loss": ce, "contrastive_loss": contrastive}

    def _backward_native(self, g: float):
        j = self._job
        j.prepare_classifier()
        j.dz_extra.zero_()
        for t in j.dloc_rowcoef:
            t.zero_()
        j.cls_w_ce, j.cls_w_contrast = g, self._wc * g
        j.kl_weight, j.ll_weight = self._kl_weight * g, self._ll_weight * g
        j.step = 0
        j.touch()
        js = JobSet([j])
        js.head_classifier(backward=True, grads=True)            # d CE / d z, hinge row coefficients, classifier grads
        js._launch(0, 1, 1, _lib.NM_F_BACKWARD | _lib.NM_F_GRADS)  # trunk: weighted ELBO + those extras
        j.dz_extra = None
        j.dloc_rowcoef = [None] * len(j.kmods)
        j.cls_w_ce, j.cls_w_contrast = 1.0, self._wc
        j.kl_weight, j.ll_weight = self._kl_weight, self._ll_weight
        j.touch()
        self._pending = j.grads
        self._grads_ready = True
        self._assign_grads()

    def predict(self, xes, cs):
        with torch.no_grad():
            j, B = self._run(xes, cs, "poe", _lib.NM_F_EXPORT)
            self._head_setup(j, use_mu=True)                     # classifier(mu_combined), cVAE.py:2202-2207
            JobSet([j]).head_classifier(backward=False, bn_stats=self.training)
            if self.training:
                self._nbt = [n + 1 for n in self._nbt]
            return j.out_logits[:B, : self.num_classes].clone()
