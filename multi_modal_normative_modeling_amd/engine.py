"""Host side of the HIP path: ROI tables resident in HBM, job descriptors, launches.

A *job* is one independent model of the sweep (a (fold, procedure) cell): its packed ROI
tables, flat fp32 parameters, Adam moments and workspace.  A *JobSet* is the device array
of descriptors one kernel launch runs -- one workgroup per job.
PyTorch is used for device memory and streams only; all arithmetic is in libnmhip.so.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import zlib
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .layout import ModelSpec, ParamLayout

BATCH = _lib.NM_BATCH


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device) -> int:
    """The current HIP stream of `device` as an integer handle.  (torch.cuda.current_stream() builds a Stream object --
    ~6 us a call, eight calls per step of the eager classes; the raw getter is what torch.cuda's own internals use.)"""
    if _RAW_STREAM is not None:
        d = torch.device(device)
        return int(_RAW_STREAM(d.index if d.index is not None else torch.cuda.current_device()))
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.NmError("no HIP device visible: the cVAE hot path runs on MI355X only (no CPU fallback)")
    return torch.device(device if device is not None else "cuda:0")


class Table:
    """One modality's ROI table in HBM.

    ``x`` [N, D] and covariates ``c`` [N, C] (any float/int dtype; the reference's
    ``torch.cat((x, c))`` promotes to float32, cVAE.py:163) become
      x_f32 [rows_alloc, x_pitch] fp32, zero rows beyond N, pitch = D rounded up to 4 (residual / NLL side)
      xb    [rows_alloc, Kx] bf16  x | c | 1 | 0             (MFMA operand side)
    with rows_alloc a multiple of 256 and Kx a multiple of 32.
    """

    def __init__(self, x, c, device=None):
        dev = require_gpu(device)
        lib = _lib.load()
        # content key of the covariates: tables with equal keys carry the same covariate block, which lets the
        # decoders of a model share one z | c | 1 input (nm_job_t.shared_cov).  Host data: shape, dtype and two
        # checksums of the bytes; device tensors: identity of the storage (no device-to-host copy, no sync)
        if torch.is_tensor(c) and c.is_cuda:
            self.c_key = ("dev", c.data_ptr(), tuple(c.shape), str(c.dtype), c._version)
            self._keep = (c,)            # the key is the storage's identity: the storage must outlive the table
        else:
            c_host = c.detach().contiguous().numpy() if torch.is_tensor(c) else np.ascontiguousarray(np.asarray(c))
            self.c_key = (tuple(c_host.shape), str(c_host.dtype), zlib.crc32(c_host.tobytes()), zlib.adler32(c_host.tobytes()))
        x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        c = torch.as_tensor(np.asarray(c) if not torch.is_tensor(c) else c)
        if x.dim() != 2 or c.dim() != 2 or x.shape[0] != c.shape[0]:
            raise ValueError(f"x must be [N, D] and c [N, C] with equal N, got {tuple(x.shape)} / {tuple(c.shape)}")
        self.N, self.D = int(x.shape[0]), int(x.shape[1])
        self.C = int(c.shape[1])
        self.rows_alloc = max(1, math.ceil(self.N / BATCH)) * BATCH
        self.Kx = (self.D + self.C + 1 + 31) // 32 * 32
        self.Cz = (self.C + 1 + 7) // 8 * 8
        xs = x.to(device=dev, dtype=torch.float32).contiguous()
        cs = c.to(device=dev, dtype=torch.float32).contiguous()
        self.x_pitch = (self.D + 3) // 4 * 4
        self.x_f32 = torch.empty(self.rows_alloc, self.x_pitch, dtype=torch.float32, device=dev)
        # xb: 64-column chunk images of 256-row tiles, [tiles][chunks][256][72] (nm_modality_t.xb)
        self.xb = torch.empty(int(lib.nm_xb_elems(self.rows_alloc, self.Kx)), dtype=torch.bfloat16, device=dev)
        self.cz = torch.empty(self.rows_alloc, self.Cz, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.nm_pack_table(xs.data_ptr(), cs.data_ptr() if self.C > 0 else None, self.N, self.rows_alloc,
                                     self.D, self.C, self.Kx, self.xb.data_ptr(), self.x_f32.data_ptr(),
                                     self.x_pitch, self.cz.data_ptr(), self.Cz, _stream_ptr(dev)), "nm_pack_table")
        self.device = dev

    def repack(self, x, c) -> bool:
        """Overwrite this table's contents with a new batch of the same shape (the eager facade calls the model once per
        batch: the buffers, and with them the job descriptor, stay as they are).  False if the shapes differ."""
        if not (torch.is_tensor(x) and torch.is_tensor(c)) or x.dim() != 2 or c.dim() != 2:
            return False
        if int(x.shape[0]) != self.N or int(x.shape[1]) != self.D or int(c.shape[1]) != self.C or int(c.shape[0]) != self.N:
            return False
        lib = _lib.load()
        dev = self.device
        self.c_key = (("dev", c.data_ptr(), tuple(c.shape), str(c.dtype), c._version) if c.is_cuda else
                      ("host", id(c), tuple(c.shape), str(c.dtype), c._version))
        xs = x.to(device=dev, dtype=torch.float32).contiguous()
        cs = c.to(device=dev, dtype=torch.float32).contiguous()
        self._keep = (c, xs, cs)
        _lib.check(lib.nm_pack_table(xs.data_ptr(), cs.data_ptr() if self.C > 0 else None, self.N, self.rows_alloc,
                                     self.D, self.C, self.Kx, self.xb.data_ptr(), self.x_f32.data_ptr(),
                                     self.x_pitch, self.cz.data_ptr(), self.Cz, _stream_ptr(dev)), "nm_pack_table")
        return True

    def packed_rows(self) -> torch.Tensor:
        """The packed operand rows x | c | 1 | 0 as a [rows_alloc, Kx] tensor (diagnostics / tests)."""
        nch = (self.Kx + 63) // 64
        t = self.xb.view(self.rows_alloc // BATCH, nch, BATCH, 72)[..., :64]
        return t.permute(0, 2, 1, 3).reshape(self.rows_alloc, nch * 64)[:, :self.Kx]

    @property
    def n_tiles(self) -> int:
        return self.rows_alloc // BATCH


class Job:
    """Parameters + optimizer state + tables of one model."""

    def __init__(self, spec: ModelSpec, tables: Sequence[Table], combine: str = "poe", state: Optional[Dict] = None,
                 lr: float = 1e-4, betas=(0.9, 0.999), adam_eps: float = 1e-8, kl_weight: Optional[float] = None,
                 ll_weight: float = 1.0, seed: int = 0, loss_cap: int = 1024, single_bypass: bool = True,
                 init_seed: int = 42, n_tiles_ws: int = 1):
        self.layout = ParamLayout(spec)
        self.spec = spec
        if len(tables) != spec.M:
            raise ValueError(f"need {spec.M} tables, got {len(tables)}")
        for m, t in enumerate(tables):
            if t.D != spec.input_dims[m] or t.C != spec.net_c_dim:
                raise ValueError(f"table {m}: D={t.D}, C={t.C} do not match the model ({spec.input_dims[m]}, {spec.net_c_dim})")
            if t.N != tables[0].N:
                raise ValueError("all modalities must hold the same subjects (rows)")
        combine_l = combine.lower()
        if combine_l not in _lib.NM_COMBINE:
            raise ValueError("No such combination method")            # cVAE.py:1163
        self.combine = combine_l
        self.tables = list(tables)
        dev = tables[0].device
        self.device = dev
        if state is None:
            state = self.layout.init_reference_rule(init_seed)
        self.params = self.layout.flatten(state, device=dev)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.grads = torch.zeros_like(self.params)
        self.lr, self.betas, self.adam_eps = float(lr), (float(betas[0]), float(betas[1])), float(adam_eps)
        # cVAE_multimodal adds KL once per modality (cVAE.py:1189-1195); class cVAE once (cVAE.py:497-500); the DMVAE
        # family: once per modality times beta (1.0, mmVAEPlus 0.05; cVAE.py:1507, 1911, 1570)
        dm_beta = {"dmvae": 1.0, "weighted_dmvae": 1.0, "mmvaeplus": 0.05}.get(spec.kind, 1.0)
        self.kl_weight = float(spec.M * dm_beta if kl_weight is None else kl_weight)
        if spec.is_dm:
            single_bypass = False                     # ProductOfExperts2 is always evaluated (cVAE.py:1547)
        # mvtCAE (cVAE.py:1754-1893): no single-expert bypass, 'poe' = ProductOfExperts2 fed with variances, joint variance
        # clamped at 1e-6, total = sum_i [kl + 1e-5 ll_i + beta tc] with beta = 1e-4 (the log-likelihood enters with a PLUS)
        self.var_floor, self.tc_weight = 0.0, 0.0
        if spec.kind == "mvtcae":
            single_bypass = False
            self.var_floor, self.tc_weight = 1e-6, spec.M * 1e-4
            if combine.lower() == "poe":
                self.combine = "poe2v"
            if ll_weight == 1.0:
                ll_weight = -1e-5
        self.kmods = spec.kernel_modalities()         # decoders the kernel runs: (table, has_encoder, prefix)
        self.dz_extra: Optional[torch.Tensor] = None  # d L_extra / d z          [rows_alloc, Z]
        self.dloc_extra: List[Optional[torch.Tensor]] = [None] * len(self.kmods)   # d L_extra / d x_hat [rows_alloc, x_pitch]
        self.ll_weight = float(ll_weight)
        # classifier head (kind == "endtoend" with classifier_layers)
        self.dloc_rowcoef: List[Optional[torch.Tensor]] = [None] * len(self.kmods)      # hinge row coefficients [rows_alloc]
        self.labels: Optional[torch.Tensor] = None    # int32 [rows_alloc]
        self.out_logits: Optional[torch.Tensor] = None
        self.cls_train, self.cls_use_mu = True, False
        self.cls_dropout, self.cls_margin, self.cls_w_ce, self.cls_w_contrast = 0.0, 1.0, 1.0, 0.1
        self.dephase_sleeps = 0          # set by JobSet (see nm_job_t.dephase)
        self.reg_lambda = 1.0                         # regression head (kind == "regression")
        self.fi_target: Optional[torch.Tensor] = None # [rows_alloc]
        self.out_fi_pred: Optional[torch.Tensor] = None
        self.reg_resid: Optional[torch.Tensor] = None # bf16 residual chunk images (nm_job_t.reg_resid)
        self.reg_dres: Optional[torch.Tensor] = None  # bf16 d MSE / d x_hat chunk images of the batch in flight
        self.single_bypass = bool(single_bypass)
        self.seed = int(seed)
        self.t = 0                       # optimizer steps taken
        self.step = 0                    # data steps taken (selects the batch)
        self.loss_cap = int(loss_cap)
        self.loss_log = torch.zeros(self.loss_cap, _lib.NM_LOSS_STRIDE, dtype=torch.float32, device=dev)
        self.eps: Optional[torch.Tensor] = None
        self.eps_cap = 1
        self.lr_table: Optional[torch.Tensor] = None   # fp64 [n]: learning rate of optimizer step t = lr_table[(t - 1) % n]
        self._ws = None
        self._ws_tiles = 0
        self._version = 0                # bumped whenever the descriptor would change
        self._wsh = None                 # bf16 shadow images of the weights (nm_job_t.wsh)
        self._gpart = None               # row-split launch: k slices of fp32 gradient partials (nm_job_t.gpart)
        self._gpart_k = 0
        self.shadow_dirty = True         # params were written by the host: nm_sync_shadow before the next launch
        self._ensure_workspace(n_tiles_ws)
        # optional exports
        self.out_mu = self.out_logvar = self.out_z = None
        nk = len(self.kmods)
        self.out_loc: List[Optional[torch.Tensor]] = [None] * nk
        self.out_sqerr: List[Optional[torch.Tensor]] = [None] * nk
        self.out_rowdev: List[Optional[torch.Tensor]] = [None] * nk

    # -- buffers -------------------------------------------------------------------------------
    def _ensure_workspace(self, n_tiles: int):
        if self._ws is not None and self._ws_tiles >= n_tiles:
            return
        lib = _lib.load()
        probe = _lib.NmJob()
        probe.M, probe.L, probe.Z = len(self.kmods), len(self.spec.hidden), self.spec.latent
        probe.C, probe.wide = self.spec.net_c_dim, int(self.spec.wide)
        for i, h in enumerate(self.spec.hidden):
            probe.H[i] = h
        probe.cls_layers, probe.cls_classes = len(self.spec.classifier_layers), (self.spec.num_classes if self.spec.classifier_layers else 0)
        for i, w in enumerate(self.spec.classifier_layers):      # (blocks wider than 128: the head's workspace is in tiles)
            probe.cls_width[i] = w
        probe.reg_head = 1 if self.spec.kind == "regression" else 0
        probe.M_enc = self.spec.M
        for k, (m, _, _) in enumerate(self.kmods):
            probe.mod[k].D = self.tables[m].D
            probe.mod[k].Kx = self.tables[m].Kx
        self.ws_bytes = int(lib.nm_workspace_bytes(C.byref(probe)))
        if self._wsh is None:
            if self.spec.wide and self.spec.kind != "regression":   # the general-shape path reads the fp32 master: no shadow images
                self._wsh = torch.zeros(256, dtype=torch.uint8, device=self.device)
            else:                                    # (general-shape regression model: the regressor's first-layer images only)
                nb = int(lib.nm_fill_shadow(C.byref(probe)))
                if nb < 0:
                    _lib.check(nb, "nm_fill_shadow")
                self._wsh = torch.zeros(nb, dtype=torch.uint8, device=self.device)
        self._ws = torch.zeros(self.ws_bytes * n_tiles, dtype=torch.uint8, device=self.device)
        self._ws_tiles = n_tiles
        self._version += 1

    @property
    def gpart_stride(self) -> int:
        return (int(self.params.numel()) + 255) // 256 * 256

    def _ensure_rowsplit(self, k: int):
        """Buffers of a row-split launch with k slices per (model, modality): k workspace tiles, k gradient-partial slices."""
        self._ensure_workspace(k)
        if self._gpart is None or self._gpart_k < k:
            self._gpart = torch.zeros(k * self.gpart_stride, dtype=torch.float32, device=self.device)
            self._gpart_k = k
            self._version += 1

    def devpass_ok(self) -> bool:
        """nm_devpass_ok for this job, plus: no latent exports asked for (the general forward kernel writes those)."""
        s = self.spec
        return (not s.wide and len(self.kmods) == 1 and s.M == 1 and self.single_bypass and s.n_private == 0 and not s.is_dm
                and self.tc_weight == 0.0 and s.kind != "weighted_dmvae" and s.hidden[0] <= 112 and (s.latent + 15) // 16 * 16 <= 32
                and self.out_mu is None and self.out_logvar is None and self.out_z is None)

    def rowsplit_ok(self) -> bool:
        """Can this model run row-split (nm_rowsplit_ok: plain cVAE / cVAE_multimodal-type models on the fused kernel)?"""
        s = self.spec
        return (not s.wide and s.kind in ("single", "multimodal") and len(self.kmods) == s.M and s.M <= _lib.NM_MAX_EXP
                and self.tc_weight == 0.0)

    def set_fi(self, fi):
        """Regression target per table row (FI, ..._regression.py:86-87); padded with zeros to rows_alloc."""
        ra = self.tables[0].rows_alloc
        f = torch.as_tensor(fi, dtype=torch.float32).reshape(-1)
        if f.numel() != self.tables[0].N:
            raise ValueError(f"fi has {f.numel()} values for {self.tables[0].N} table rows")
        self.fi_target = torch.zeros(ra, device=self.device)
        self.fi_target[: f.numel()] = f.to(self.device)
        self._version += 1

    def set_labels(self, labels):
        """Class label per table row (classifier head); padded with zeros to rows_alloc."""
        ra = self.tables[0].rows_alloc
        l = torch.as_tensor(labels).reshape(-1).to(torch.int32)
        if l.numel() != self.tables[0].N:
            raise ValueError(f"labels has {l.numel()} values for {self.tables[0].N} table rows")
        self.labels = torch.zeros(ra, dtype=torch.int32, device=self.device)
        self.labels[: l.numel()] = l.to(self.device)
        self._version += 1

    def prepare_classifier(self):
        """Buffers the end-to-end train loop needs: exports (latent, per-subject deviations, logits) and the
        slots the head's backward fills (d CE / d z, hinge row coefficients)."""
        if self.out_z is None or any(o is None for o in self.out_rowdev) or self.out_logits is None:
            self.enable_exports(loc=False, sqerr=False, rowdev=True, latent=True)
        ra = self.tables[0].rows_alloc
        if self.dz_extra is None:
            self.dz_extra = torch.zeros(ra, self.spec.latent, device=self.device)
            self._version += 1
        if any(d is None for d in self.dloc_rowcoef):
            self.dloc_rowcoef = [torch.zeros(ra, device=self.device) for _ in self.kmods]
            self._version += 1

    def prepare_regression(self):
        """Buffers the regression head needs: the residual chunk images the trunk exports for it (one set per 256-row
        tile of the table), the d MSE / d x_hat images it hands back (one set: the batch in flight), the predictions."""
        nq = sum((self.tables[m].D + 63) // 64 for m in range(self.spec.M))
        img = 256 * 72 * 2
        if self.reg_resid is None or self.reg_resid.numel() < self.tables[0].n_tiles * nq * img:
            self.reg_resid = torch.zeros(self.tables[0].n_tiles * nq * img, dtype=torch.uint8, device=self.device)
            self.reg_dres = torch.zeros(nq * img, dtype=torch.uint8, device=self.device)
            self._version += 1
        if self.out_fi_pred is None:
            self.out_fi_pred = torch.zeros(self.tables[0].rows_alloc, device=self.device)
            self._version += 1

    def touch(self):
        """Call after changing tables / step / t / hyper-parameters by hand: forces a descriptor re-upload."""
        self._version += 1

    def set_lr_table(self, lrs):
        """Per-step learning rates (optimizer step t, 1-based, runs at lrs[(t - 1) % len]): the schedules that really
        reach the optimizer in the reference (`param_group['lr'] = clr`, multimodal_kfold_cvae_nmmlp.py:376-381;
        prep.cyclic_lr builds that one).  None: the constant lr."""
        self.lr_table = None if lrs is None else torch.as_tensor(np.asarray(lrs, dtype=np.float64)).to(self.device).contiguous()
        self._version += 1

    def set_eps(self, eps: Optional[torch.Tensor]):
        """Explicit reparameterisation draws [n_steps, 256, Z] (parity mode); None = in-kernel generator."""
        self._version += 1
        if eps is None:
            self.eps, self.eps_cap = None, 1
            return
        e = torch.as_tensor(eps, dtype=torch.float32)
        if e.dim() == 2:
            e = e.unsqueeze(0)
        if e.shape[1] != BATCH or e.shape[2] != self.spec.latent:
            pad = torch.zeros(e.shape[0], BATCH, self.spec.latent, dtype=torch.float32)
            pad[:, :e.shape[1]] = e
            e = pad
        self.eps = e.to(self.device).contiguous()
        self.eps_cap = int(self.eps.shape[0])

    def enable_exports(self, loc=True, sqerr=True, rowdev=True, latent=True):
        ra = self.tables[0].rows_alloc
        Z = self.spec.latent
        self._version += 1
        if latent:
            self.out_mu = torch.zeros(ra, Z, device=self.device)
            self.out_logvar = torch.zeros(ra, Z, device=self.device)
            self.out_z = torch.zeros(ra, Z, device=self.device)
        if self.spec.kind == "regression":
            self.out_fi_pred = torch.zeros(ra, device=self.device)
        if self.spec.kind == "endtoend" and self.spec.classifier_layers:
            self.out_logits = torch.zeros(ra, _lib.NM_MAX_CLASSES, device=self.device)
        for j, (m, _, _) in enumerate(self.kmods):
            t = self.tables[m]
            # storage rows share the fp32 table's pitch (16-byte stores in the kernel); the views are [rows, D]
            self.out_loc[j] = torch.zeros(ra, t.x_pitch, device=self.device)[:, :t.D] if loc else None
            self.out_sqerr[j] = torch.zeros(ra, t.x_pitch, device=self.device)[:, :t.D] if sqerr else None
            self.out_rowdev[j] = torch.zeros(ra, device=self.device) if rowdev else None

    # -- descriptor ----------------------------------------------------------------------------
    def struct(self) -> _lib.NmJob:
        s, j = self.spec, _lib.NmJob()
        j.M, j.M_enc, j.C, j.L, j.Z = len(self.kmods), s.M, s.net_c_dim, len(s.hidden), s.latent
        # DMVAE family: ReLU, sigmoid / squared-error output, private latent columns, learnable loss weights
        j.act_slope = 0.0 if s.is_dm else 0.01
        j.out_kind = 1 if s.is_dm else 0
        j.n_private = s.n_private
        j.w_off = self.layout.offsets["weights"] if s.kind == "weighted_dmvae" else -1
        j.var_floor, j.tc_weight = self.var_floor, self.tc_weight
        for i, h in enumerate(s.hidden):
            j.H[i] = h
        j.combine = _lib.NM_COMBINE[self.combine]
        j.single_bypass = 1 if self.single_bypass else 0
        j.n_rows = self.tables[0].N
        j.non_linear = 1 if (s.non_linear or s.is_dm) else 0        # (torch.relu unconditionally, cVAE.py:1462-1463)
        j.dephase = int(self.dephase_sleeps)
        k0 = self.tables[0].c_key
        j.shared_cov = 1 if (k0 is not None and all(t.c_key == k0 for t in self.tables)) else 0
        self._shared_cov = int(j.shared_cov)
        j.wide = int(s.wide)
        j.loss_cap, j.eps_cap = self.loss_cap, self.eps_cap
        j.lr, j.beta1, j.beta2, j.adam_eps = self.lr, self.betas[0], self.betas[1], self.adam_eps
        j.adam_off = self.t - self.step
        j.lr_table = self.lr_table.data_ptr() if self.lr_table is not None else None
        j.lr_cap = int(self.lr_table.numel()) if self.lr_table is not None else 0
        j.kl_weight, j.ll_weight = self.kl_weight, self.ll_weight
        j.params, j.adam_m, j.adam_v = self.params.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr()
        j.grads = self.grads.data_ptr()
        j.eps = self.eps.data_ptr() if self.eps is not None else None
        j.seed = self.seed
        j.loss_log = self.loss_log.data_ptr()
        j.workspace = self._ws.data_ptr()
        j.workspace_stride = self.ws_bytes
        j.n_params = int(self.params.numel())
        j.gpart = self._gpart.data_ptr() if self._gpart is not None else None
        j.gpart_stride = self.gpart_stride
        j.wsh = self._wsh.data_ptr()
        j.out_mu = self.out_mu.data_ptr() if self.out_mu is not None else None
        j.out_logvar = self.out_logvar.data_ptr() if self.out_logvar is not None else None
        j.out_z = self.out_z.data_ptr() if self.out_z is not None else None
        j.dz_extra = self.dz_extra.data_ptr() if self.dz_extra is not None else None
        self.layout.fill_head(j)
        if j.reg_head:
            self.prepare_regression()
            j.reg_resid, j.reg_dres = self.reg_resid.data_ptr(), self.reg_dres.data_ptr()
        j.reg_lambda = self.reg_lambda
        j.fi_target = self.fi_target.data_ptr() if self.fi_target is not None else None
        j.out_fi_pred = self.out_fi_pred.data_ptr() if self.out_fi_pred is not None else None
        if j.cls_classes and (self.out_z is None or self.out_mu is None):
            j.cls_layers, j.cls_classes = 0, 0        # the head reads the exported latent
        j.cls_train, j.cls_use_mu = int(self.cls_train), int(self.cls_use_mu)
        j.cls_dropout, j.cls_margin = self.cls_dropout, self.cls_margin
        j.cls_w_ce, j.cls_w_contrast = self.cls_w_ce, self.cls_w_contrast
        j.labels = self.labels.data_ptr() if self.labels is not None else None
        j.out_logits = self.out_logits.data_ptr() if self.out_logits is not None else None
        j.dz_out = self.dz_extra.data_ptr() if self.dz_extra is not None else None
        for k, (m, _, _) in enumerate(self.kmods):
            t = self.tables[m]
            md = j.mod[k]
            md.D, md.Kx, md.x_pitch, md.Cz = t.D, t.Kx, t.x_pitch, t.Cz
            md.x_f32, md.xb, md.cz = t.x_f32.data_ptr(), t.xb.data_ptr(), t.cz.data_ptr()
            self.layout.fill_modality(md, k)
            md.out_loc = self.out_loc[k].data_ptr() if self.out_loc[k] is not None else None
            md.out_sqerr = self.out_sqerr[k].data_ptr() if self.out_sqerr[k] is not None else None
            md.out_rowdev = self.out_rowdev[k].data_ptr() if self.out_rowdev[k] is not None else None
            md.dloc_extra = self.dloc_extra[k].data_ptr() if self.dloc_extra[k] is not None else None
            md.dloc_rowcoef = self.dloc_rowcoef[k].data_ptr() if self.dloc_rowcoef[k] is not None else None
            j.rowcoef_out[k] = md.dloc_rowcoef
        if not s.wide or s.kind == "regression":
            nb = int(_lib.load().nm_fill_shadow(C.byref(j)))       # shadow-image offsets of every modality
            if nb != self._wsh.numel():
                raise _lib.NmError(f"shadow size changed: {nb} vs {self._wsh.numel()} bytes")
        _lib.check(_lib.load().nm_validate_job(C.byref(j)), "nm_validate_job")
        return j

    # -- state_dict interchange (reference key names) --------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().cpu().clone() for k, v in self.layout.unflatten(self.params).items()}

    def load_state_dict(self, state: Dict[str, torch.Tensor]):
        self.params.copy_(self.layout.flatten(state, device=self.device))
        self.shadow_dirty = True

    def params_changed(self):
        """Call after writing ``params`` from the host side (an external optimizer, a hand edit): the bf16 shadow
        images the kernels read are rebuilt before the next launch."""
        self.shadow_dirty = True

    def grads_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().cpu().clone() for k, v in self.layout.unflatten(self.grads).items()}

    def adam_dicts(self):
        m = {k: v.detach().cpu().clone() for k, v in self.layout.unflatten(self.adam_m).items()}
        v = {k: v.detach().cpu().clone() for k, v in self.layout.unflatten(self.adam_v).items()}
        return m, v

    @property
    def batches_per_epoch(self) -> int:
        return math.ceil(self.tables[0].N / BATCH)


class JobSet:
    """Device array of job descriptors = the unit one launch runs (one workgroup per job)."""

    def __init__(self, jobs: Sequence[Job]):
        if not jobs:
            raise ValueError("empty job set")
        self.jobs = list(jobs)
        self.wide = bool(jobs[0].spec.wide)
        if any(bool(j.spec.wide) != self.wide for j in jobs):
            raise ValueError("a job set holds either fused-kernel shapes or general-shape (wide) models, not both")
        self.device = jobs[0].device
        self.lib = _lib.load()
        self._dev = None
        self._sig = None
        self._split_pending = False      # a split launch has run since the hand-off error words were last read

    def check_split_errors(self, block: bool = True):
        """Raise NmError if a hand-off of a split launch (one workgroup per modality) timed out: the job's workgroups left
        that launch, its parameters are not to be trusted.  The error words are fetched asynchronously (a small kernel + a
        copy into pinned memory behind the launch); block=True -- everything that reads results: assert_finite, losses,
        sweep.save_model -- waits for the words of every split launch so far; block=False -- before the next launch of
        this set -- only looks at words that have already arrived, so a loop of launches is never stalled by the check."""
        infl = getattr(self, "_err_inflight", None)
        if infl is not None:
            ev, host = infl
            if block:
                ev.synchronize()
            if ev.query():
                self._err_inflight = None
                bad = host.nonzero().flatten().tolist()
                if bad:
                    raise _lib.NmError(f"split launch: the hand-off between the workgroups of job(s) {bad[:8]} timed out "
                                       f"(the parts of a model must all be resident at once: another stream or process "
                                       f"occupying CUs breaks that); their parameters are not valid -- re-run with NMHIP_SPLIT=0")
        if self._split_pending and self._dev is not None and getattr(self, "_err_inflight", None) is None:
            n = len(self.jobs)
            if getattr(self, "_err_dev", None) is None:
                self._err_dev = torch.zeros(n, dtype=torch.int32, device=self.device)
                self._err_host = torch.zeros(n, dtype=torch.int32).pin_memory()
            _lib.check(self.lib.nm_split_errors(self._dev.data_ptr(), n, self._err_dev.data_ptr(), 1,
                                                _stream_ptr(self.device)), "nm_split_errors")
            self._err_host.copy_(self._err_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._err_inflight = (ev, self._err_host)
            self._split_pending = False
            if block:
                self.check_split_errors(True)

    def _set_dephase(self):
        """Start offsets of the jobs of a long launch (nm_job_t.dephase): workgroup b lands on XCD b mod 8 (observed
        placement, speed only), so the 32 jobs that share an XCD -- its L2 and its link to memory -- are spread
        evenly over one step, and the XCDs are staggered against each other by a fraction of that spacing.
        NMHIP_DEPHASE = 0 switches it off, "xcd" is the round-1 scheme (one offset per XCD)."""
        mode = os.environ.get("NMHIP_DEPHASE", "cu")
        scale = float(os.environ.get("NMHIP_DEPHASE_SCALE", "1"))
        n = len(self.jobs)
        key = (mode, scale, n)
        vals = getattr(self, "_dephase_vals", None)
        # (host time of a launch matters: it precedes the launch -- so the offsets are computed once per set; a job that
        #  another set has re-timed in between is noticed by comparing the values this set assigned)
        if getattr(self, "_dephase_key", None) == key and vals is not None and all(j.dephase_sleeps == v for j, v in zip(self.jobs, vals)):
            return
        self._dephase_key = key
        for b, j in enumerate(self.jobs):
            # ~2.05 ns per parameter and step with the chip full (measured: 0.73 ms for 355 k parameters)
            step_us = j.layout.n_params * 2.05e-3
            if mode == "0" or n < 16:
                frac = 0.0
            elif mode == "xcd":
                frac = (b & 7) / 8
            else:
                per_xcd = max(1, (n + 7) // 8)
                frac = ((b >> 3) + (b & 7) / 8) / per_xcd
            s = int(round(step_us * frac * scale))
            if s != j.dephase_sleeps:
                j.dephase_sleeps = s
                j._version += 1
        self._dephase_vals = [j.dephase_sleeps for j in self.jobs]

    def _upload(self, n_tiles: int = 1):
        """Descriptor array on the device; rebuilt only when a job's descriptor changed (the
        optimizer step count rides on adam_off = t - step, constant while both advance)."""
        self.check_split_errors(block=False)
        self._set_dephase()
        for j in self.jobs:
            j._ensure_workspace(n_tiles)
        sig = tuple((j._version, j.t - j.step) for j in self.jobs)
        if self._dev is None or sig != self._sig:
            arr = (_lib.NmJob * len(self.jobs))(*[j.struct() for j in self.jobs])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._dev = host.to(self.device)
            self._sig = sig
        if self.wide:                                # (no shadow images but a regression head's first layer)
            for j in self.jobs:
                j.shadow_dirty = j.shadow_dirty and j.spec.kind == "regression"
        if any(j.shadow_dirty for j in self.jobs):
            _lib.check(self.lib.nm_sync_shadow(self._dev.data_ptr(), len(self.jobs), _stream_ptr(self.device)), "nm_sync_shadow")
            for j in self.jobs:
                j.shadow_dirty = False
        return self._dev.data_ptr()

    def _launch(self, step0, steps_per_tile, n_tiles, flags, scalar_tr=False):
        ptr = self._upload(n_tiles)
        fn = self.lib.nm_launch_wide if self.wide else (self.lib.nm_launch_scalar_tr if scalar_tr else self.lib.nm_launch)
        _lib.check(fn(ptr, len(self.jobs), int(step0), int(steps_per_tile), int(n_tiles), int(flags),
                      _stream_ptr(self.device)), "nm_launch_wide" if self.wide else "nm_launch")

    def _launch_split(self, step0, n_steps, flags):
        """nm_launch_split: every model as one workgroup per modality (small sets; see split_parts)."""
        ptr = self._upload(1)
        _lib.check(self.lib.nm_launch_split(ptr, len(self.jobs), len(self.jobs[0].kmods), int(step0), int(n_steps), int(flags),
                                            _stream_ptr(self.device)), "nm_launch_split")
        self._split_pending = True

    def split_parts(self) -> int:
        """Workgroups per model for a training launch: the M modalities of a model as separate workgroups when the
        set is small enough for all of them to be resident at once (nm_launch_split), else 1.  NMHIP_SPLIT=0 / 1
        forces one / insists on several."""
        M = len(self.jobs[0].kmods)
        mode = os.environ.get("NMHIP_SPLIT", "auto")
        if mode == "0" or M < 2 or self.wide or any(len(j.kmods) != M for j in self.jobs):
            return 1
        if not hasattr(self, "_cus"):
            self._cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        cus = self._cus
        fits = (len(self.jobs) + 7) // 8 * 8 * M <= cus
        return M if fits else 1

    def rowsplit_k(self) -> int:
        """Row slices per (model, modality) for a training launch (nm_launch_rowsplit): the largest k in {4, 2} for which
        all ceil(jobs * M / 8) * 8 * k workgroups are resident at once, 1 if the set is too large for that or a model
        needs the whole batch in one workgroup.  NMHIP_ROWSPLIT = 0 switches it off, 2 / 4 cap k."""
        mode = os.environ.get("NMHIP_ROWSPLIT", "auto")
        M = len(self.jobs[0].kmods)
        if mode == "0" or self.wide or any(len(j.kmods) != M or not j.rowsplit_ok() for j in self.jobs):
            return 1
        if not hasattr(self, "_cus"):
            self._cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        groups = (len(self.jobs) * M + 7) // 8 * 8
        kmax = int(mode) if mode in ("2", "4") else 4
        for k in (4, 2):
            if k <= kmax and groups * k <= self._cus:
                return k
        return 1

    def rowsplit_helpers(self, k: int) -> int:
        """Helper workgroups per (model, modality) of a row-split launch: the CUs the k slices leave idle join the Adam
        sweep (nmhip.h: nm_launch_rowsplit).  A group stays on one XCD (32 CUs); measured (one and five 3 x 379 models,
        k = 4): 6 helpers give all of the gain, beyond 12 the extra arrivals at the hand-off cost what the shorter sweep
        saves -- so at most 12.  NMHIP_RS_HELPERS pins it."""
        env = os.environ.get("NMHIP_RS_HELPERS", "auto")
        if not hasattr(self, "_cus"):
            self._cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        groups = (len(self.jobs) * len(self.jobs[0].kmods) + 7) // 8 * 8
        room = max(0, min(self._cus // groups, 32) - k)
        return min(int(env), room) if env != "auto" else min(room, 12)

    def _launch_rowsplit(self, k: int, step0: int, n_steps: int, flags: int, helpers: Optional[int] = None):
        for j in self.jobs:
            j._ensure_rowsplit(k)
        ptr = self._upload(k)
        h = self.rowsplit_helpers(k) if helpers is None else int(helpers)
        # start offsets over ~one step's time once the launch fills a good part of the chip (measured: 0.37 ns per
        # parameter and step for one model at k = 4); NMHIP_RS_SPREAD scales it, 0 switches it off
        wgs = len(self.jobs) * len(self.jobs[0].kmods) * k
        scale = float(os.environ.get("NMHIP_RS_SPREAD", "1"))
        spread = int(self.jobs[0].layout.n_params * 0.37e-3 * (4 / k) * scale) if wgs >= 96 else 0
        _lib.check(self.lib.nm_launch_rowsplit(ptr, len(self.jobs), len(self.jobs[0].kmods), int(k), h, int(step0), int(n_steps),
                                               int(flags), spread, _stream_ptr(self.device)), "nm_launch_rowsplit")
        self._split_pending = True

    def train(self, n_steps: int, scalar_tr: bool = False, profile: bool = False, split: Optional[bool] = None,
              rowsplit: Optional[int] = None, helpers: Optional[int] = None):
        """n_steps fused train steps per job in ONE launch (forward + ELBO + backward + Adam).  Small sets put several
        workgroups behind a model: k row slices per (model, modality) (rowsplit=None: rowsplit_k(); results agree with the
        one-workgroup launch to fp32 summation order), else one workgroup per modality (split=None: automatically;
        bit-identical to the one-workgroup launch)."""
        step0 = self.jobs[0].step
        if any(j.step != step0 for j in self.jobs):
            raise ValueError("jobs of one set must be at the same step")
        flags = _lib.NM_F_BACKWARD | _lib.NM_F_ADAM | (_lib.NM_F_PROFILE if profile else 0)
        k = (self.rowsplit_k() if split is None else 1) if rowsplit is None else int(rowsplit)
        if k > 1 and not scalar_tr:
            self._launch_rowsplit(k, step0, n_steps, flags, helpers)
            for j in self.jobs:
                j.step += n_steps
                j.t += n_steps
            return
        parts = self.split_parts() if split is None else (len(self.jobs[0].kmods) if split else 1)
        if parts > 1 and not scalar_tr:
            ptr = self._upload(1)
            _lib.check(self.lib.nm_launch_split(ptr, len(self.jobs), parts, int(step0), int(n_steps), int(flags),
                                                _stream_ptr(self.device)), "nm_launch_split")
            self._split_pending = True
        else:
            self._launch(step0, n_steps, 1, flags, scalar_tr)
        for j in self.jobs:
            j.step += n_steps
            j.t += n_steps

    def grads(self, step: Optional[int] = None, export: bool = True, scalar_tr: bool = False, split: bool = False,
              rowsplit: int = 1, helpers: Optional[int] = None):
        """forward + loss + backward for one step; gradients land in job.grads (no update)."""
        s = self.jobs[0].step if step is None else step
        flags = _lib.NM_F_BACKWARD | _lib.NM_F_GRADS | (_lib.NM_F_EXPORT if export else 0)
        if rowsplit > 1:
            self._launch_rowsplit(rowsplit, s, 1, flags, helpers)
        elif split:
            ptr = self._upload(1)
            _lib.check(self.lib.nm_launch_split(ptr, len(self.jobs), len(self.jobs[0].kmods), int(s), 1, int(flags),
                                                _stream_ptr(self.device)), "nm_launch_split")
            self._split_pending = True
        else:
            self._launch(s, 1, 1, flags, scalar_tr)

    def devpass_ok(self) -> bool:
        """Can the set's deviation pass run on the compact kernel (nm_devpass: 128-row tiles, two workgroups per CU)?
        The conditions of nm_devpass_ok, read off the jobs (building 256 descriptors per launch to ask the library would
        cost more than the pass; tests/test_cabi_cpu.py holds the two to each other)."""
        if self.wide or os.environ.get("NMHIP_DEVPASS", "1") == "0":
            return False
        return all(j.devpass_ok() for j in self.jobs)

    def forward(self, tile0: int = 0, n_tiles: Optional[int] = None, loss: bool = True):
        """forward-only over row tiles (one workgroup per (job, 256-row tile)); fills the exports.  loss=False: only the
        per-ROI / per-subject deviations and the reconstruction are wanted (the deviation pass of
        ..._regression.py:163-192) -- one-expert sets then run on the compact kernel, two workgroups per CU."""
        nt = self.jobs[0].tables[0].n_tiles if n_tiles is None else n_tiles
        if not loss and self.devpass_ok():
            ptr = self._upload(1)
            _lib.check(self.lib.nm_devpass(ptr, len(self.jobs), int(tile0) * 2, int(nt) * 2, int(getattr(self, "_dv_flags", 0)),
                                           _stream_ptr(self.device)), "nm_devpass")
            return
        self._launch(tile0, 1, nt, _lib.NM_F_EXPORT)

    def head_regression(self, backward: bool, grads: bool = True, adam: bool = False, step: int = 0, tile0: int = 0,
                        n_tiles: int = 1):
        """nm_head_regression on the residual images a preceding forward() / NM_F_EXPORT launch left in job.reg_resid:
        fills out_fi_pred and loss_log[..][NM_LOSS_REG]; with backward also job.reg_dres (d MSE / d x_hat, bf16 chunk
        images) and the regressor's gradients / Adam update (cVAE.py:2309-2346).  Training runs through
        train_regression (one launch for trunk and head)."""
        ptr = self._upload(max(n_tiles, 1))
        flags = (_lib.NM_F_BACKWARD if backward else 0) | (_lib.NM_F_GRADS if grads and backward else 0) | \
                (_lib.NM_F_ADAM if adam and backward else 0)
        _lib.check(self.lib.nm_head_regression(ptr, len(self.jobs), int(step), int(tile0), int(n_tiles), int(flags),
                                               _stream_ptr(self.device)), "nm_head_regression")

    def head_classifier(self, backward: bool, grads: bool = True, adam: bool = False, bn_stats: bool = False,
                        step: int = 0, tile0: int = 0, n_tiles: int = 1):
        """nm_head_classifier on the latent / deviations a preceding NM_F_EXPORT launch exported: fills out_logits
        and loss_log[..][NM_LOSS_CE / NM_LOSS_CONTRAST]; with backward also dz_extra, the hinge row coefficients
        and the classifier's gradients / Adam update (cVAE.py:2004-2018, 2140-2200)."""
        ptr = self._upload(max(n_tiles, 1))
        flags = (_lib.NM_F_BACKWARD if backward else 0) | (_lib.NM_F_GRADS if grads and backward else 0) | \
                (_lib.NM_F_ADAM if adam and backward else 0) | (_lib.NM_F_BNSTATS if bn_stats else 0)
        _lib.check(self.lib.nm_head_classifier(ptr, len(self.jobs), int(step), int(tile0), int(n_tiles), int(flags),
                                               _stream_ptr(self.device)), "nm_head_classifier")

    def _train_head(self, step0: int, n_steps: int, flags: int = 0):
        """nm_train_steps_head: all n_steps in one persistent launch, the trunk's forward evaluated once per step."""
        ptr = self._upload(1)
        _lib.check(self.lib.nm_train_steps_head(ptr, len(self.jobs), int(step0), int(n_steps), int(flags),
                                                _stream_ptr(self.device)), "nm_train_steps_head")

    def train_endtoend(self, n_steps: int, fused: bool = True):
        """n_steps train steps of cVAE_multimodal_endtoend jobs on the device, no host sync (the loop of
        multimodal_kfold_cvae_nmpmcont.py:257-303): per step (i) forward with latent and per-subject deviations
        exported, (ii) the classifier head: forward (train-mode BatchNorm / Dropout), cross entropy, contrastive
        hinge, backward, its Adam update, d CE / d z and the hinge row coefficients, (iii) the trunk's backward + Adam
        with those extra gradients.  fused (default): one persistent launch for all steps (nm_train_steps_head);
        fused=False: the three-launches-per-step form it replaced (trunk forward twice), kept as a cross-check -- and the
        form a trunk on the general-shape path or a classifier with blocks wider than 128 (-Layers "256 128 64") runs in:
        the persistent head kernel holds the one-tile classifier only."""
        step0 = self.jobs[0].step
        for j in self.jobs:
            if j.spec.kind != "endtoend" or not j.spec.classifier_layers or j.labels is None:
                raise ValueError("train_endtoend needs end-to-end jobs with a classifier and labels set")
            if j.step != step0:
                raise ValueError("jobs of one set must be at the same step")
            j.cls_train, j.cls_use_mu = True, False
            j.prepare_classifier()
        nb = self.jobs[0].batches_per_epoch
        if any(j.batches_per_epoch != nb for j in self.jobs):
            raise ValueError("jobs of one set must have the same number of batches")
        tiled_head = any(w > 128 for j in self.jobs for w in j.spec.classifier_layers)
        if fused and not self.wide and not tiled_head:
            self._train_head(step0, n_steps, _lib.NM_F_BNSTATS)
        else:
            for s in range(step0, step0 + n_steps):
                self._launch(s, 1, 1, _lib.NM_F_EXPORT)
                self.head_classifier(backward=True, grads=False, adam=True, bn_stats=True, step=s, tile0=s % nb)
                self._launch(s, 1, 1, _lib.NM_F_BACKWARD | _lib.NM_F_ADAM)
        for j in self.jobs:
            j.step += n_steps
            j.t += n_steps

    def train_regression(self, n_steps: int):
        """n_steps train steps of cVAE_multimodal_regression jobs in one persistent launch, no host sync (the loop of
        multimodal_kfold_train_cvae_supervised_regression.py:112-125): per step (i) the trunk's forward, leaving the
        residuals as bf16 chunk images, (ii) the regressor: forward, MSE, backward, its Adam update, d MSE / d x_hat,
        (iii) the trunk's backward + Adam with that extra gradient (nm_train_steps_head)."""
        step0 = self.jobs[0].step
        for j in self.jobs:
            if j.spec.kind != "regression" or j.fi_target is None:
                raise ValueError("train_regression needs regression jobs with fi_target set")
            if j.step != step0:
                raise ValueError("jobs of one set must be at the same step")
            j.prepare_regression()
        nb = self.jobs[0].batches_per_epoch
        if any(j.batches_per_epoch != nb for j in self.jobs):
            raise ValueError("jobs of one set must have the same number of batches")
        if self.wide:
            # a trunk on the general-shape path: three launches per step (residual images out, the regressor with its
            # update and d MSE / d x_hat, the trunk's backward + Adam with that extra gradient)
            for s in range(step0, step0 + n_steps):
                self._launch(s, 1, 1, _lib.NM_F_EXPORT)
                self.head_regression(backward=True, grads=False, adam=True, step=s, tile0=s % nb)
                self._launch(s, 1, 1, _lib.NM_F_BACKWARD | _lib.NM_F_ADAM)
        else:
            self._train_head(step0, n_steps)
        for j in self.jobs:
            j.step += n_steps
            j.t += n_steps

    def grads_head(self, step: int = 0, bn_stats: bool = False):
        """Gradients of one head-model step's total loss into job.grads, no update (the eager facade's backward)."""
        if self.wide:                      # (regression model on a general-shape trunk: the three-launch form)
            nb = self.jobs[0].batches_per_epoch
            self._launch(step, 1, 1, _lib.NM_F_EXPORT)
            self.head_regression(backward=True, grads=True, adam=False, step=step, tile0=step % nb)
            self._launch(step, 1, 1, _lib.NM_F_BACKWARD | _lib.NM_F_GRADS)
            return
        self._train_head(step, 1, _lib.NM_F_GRADS | (_lib.NM_F_BNSTATS if bn_stats else 0))

    def losses(self) -> torch.Tensor:
        """[n_jobs, loss_cap, 8] on the host."""
        self.check_split_errors()
        return torch.stack([j.loss_log for j in self.jobs]).cpu()

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def assert_finite(self):
        """Failure detection, once per epoch / run rather than per step (the reference prints NaNs from inside
        its hot loop, cVAE.py:1169-1172): one device reduction over every job's loss ring; raises NmError naming
        the first job whose log holds a non-finite value.  Also the point where a timed-out hand-off of a split
        launch surfaces at the latest (check_split_errors)."""
        self.check_split_errors()
        logs = torch.stack([j.loss_log for j in self.jobs])
        ok = torch.isfinite(logs).flatten(1).all(dim=1)
        if not bool(ok.all()):
            bad = int((~ok).nonzero()[0])
            raise _lib.NmError(f"non-finite loss in job {bad} of {len(self.jobs)} (step {self.jobs[bad].step})")


def adam_step(params: torch.Tensor, grads: torch.Tensor, m: torch.Tensor, v: torch.Tensor, t: int, lr=1e-4,
              betas=(0.9, 0.999), eps=1e-8):
    """Flat Adam on device buffers (nm_adam_step); t is the 1-based step count."""
    lib = _lib.load()
    _lib.check(lib.nm_adam_step(params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), params.numel(),
                                lr, betas[0], betas[1], eps, t, _stream_ptr(params.device)), "nm_adam_step")
