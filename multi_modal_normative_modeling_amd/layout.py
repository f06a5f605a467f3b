"""Model shape description and the flat fp32 parameter layout the HIP kernels address.

Tensors keep the reference's ``state_dict`` names, shapes and order (cVAE.py:140-206,
1087-1116), so ``load_state_dict``/``state_dict`` interchange weights with the reference.
Inside the kernel's flat buffer a weight matrix [N][K] is stored as 16 x 16 fp32 tiles,
``[ceil(N/16)][ceil(K/16)][16][16]`` zero padded (1 KiB per tile, tile-aligned): the Adam sweep of
the weight-gradient phase streams parameters and moments tile by tile, one lane-linear 16-byte
access per lane.  Vectors (biases, logvar_out, alpha, BatchNorm tensors) are stored plain, 16-byte
aligned.  ``flatten`` / ``unflatten`` convert between that buffer and the reference's tensors; the
eager facade keeps a second, plain ("natural") buffer for parameter views (``nat_*``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch

from . import _lib

DM_KINDS = ("dmvae", "weighted_dmvae", "mmvaeplus")
ALIGN = 4   # floats
HEAD_CHUNK = 64   # regression head: every modality's residual columns are padded to whole chunks (XCH in nmhip.hip)
TILE = 16   # weight matrices are stored as TILE x TILE tiles (wt_off() in nmhip.hip)
REGRESSOR_WIDTHS = (128, 64, 1)   # cVAE.py:2249-2253


@dataclass
class ModelSpec:
    """Constructor arguments of the reference model classes (cVAE.py:391-399, 1087-1095)."""
    input_dims: Sequence[int]
    hidden: Sequence[int]
    latent: int
    c_dim: int
    non_linear: bool = True
    # "single" = class cVAE, "multimodal" = cVAE_multimodal, "regression" = cVAE_multimodal_regression (trunk),
    # "endtoend" = cVAE_multimodal_endtoend (trunk: shared encoders + health / disease decoder banks),
    # "mvtcae" = mvtCAE (cVAE.py:1754-1893): cVAE_multimodal's modules with its own fusion quirk, variance clamp and loss;
    # "dmvae" / "weighted_dmvae" / "mmvaeplus" = the DMVAE family (cVAE.py:1491-1747, 1895-2002): covariate-free ReLU
    # encoders, sigmoid decoders with a squared-error likelihood, the first min(c_dim, latent) latent columns private
    kind: str = "multimodal"
    # end-to-end model only: hidden widths of the Classifier (cVAE.py:2004-2018); () = trunk without a classifier
    classifier_layers: Sequence[int] = ()
    num_classes: int = 2

    @property
    def M(self) -> int:
        return len(self.input_dims)

    @property
    def is_dm(self) -> bool:
        return self.kind in DM_KINDS

    @property
    def net_c_dim(self) -> int:
        """Covariate width the networks see: the DMVAE family ignores c (VariationalEncoder / Decoder take x / z only)."""
        return 0 if self.is_dm else self.c_dim

    @property
    def n_private(self) -> int:
        """DMVAE family: s_dim = c_dim (cVAE.py:1506), mu[:, :s_dim] of a [B, latent] tensor."""
        return min(self.c_dim, self.latent) if self.is_dm else 0

    @property
    def wide(self) -> bool:
        """A shape beyond the fused step kernel's [256][128] tile (hidden width > 127, latent > 64, latent + c_dim > 127):
        it runs through the general-shape path (nm_launch_wide, csrc/nm_wide.inc)."""
        return (any(h > _lib.NM_MAX_WIDTH for h in self.hidden) or self.latent > _lib.NM_MAX_LATENT
                or self.latent + self.net_c_dim > _lib.NM_MAX_WIDTH)

    def validate(self):
        n_dec = self.M * (2 if self.kind == "endtoend" else 1)
        if not (1 <= self.M <= _lib.NM_MAX_EXP) or n_dec > _lib.NM_MAX_MOD:
            raise ValueError(f"modalities must be 1..{_lib.NM_MAX_EXP} (and at most {_lib.NM_MAX_MOD} decoders), got {self.M}")
        if not (1 <= len(self.hidden) <= _lib.NM_MAX_HID):
            raise ValueError(f"hidden layers must be 1..{_lib.NM_MAX_HID}, got {len(self.hidden)}")
        if any(h < 1 or h > _lib.NM_WIDE_MAX_WIDTH for h in self.hidden):
            raise ValueError(f"hidden widths must be 1..{_lib.NM_WIDE_MAX_WIDTH}, got {list(self.hidden)}")
        if not (1 <= self.latent <= _lib.NM_WIDE_MAX_LATENT):
            raise ValueError(f"latent_dim must be 1..{_lib.NM_WIDE_MAX_LATENT}, got {self.latent}")
        if self.wide and self.kind == "mvtcae" and self.M * self.latent > 256:
            # (its total-correlation term keeps experts x latent log-sum-exps in 256 floats of LDS: nm_validate_job -19)
            raise ValueError(f"mvtCAE beyond the fused kernel's tile needs modalities x latent_dim <= 256, got {self.M} x {self.latent}")
        if self.is_dm and len(self.hidden) != 2:
            raise ValueError("the DMVAE family has exactly two hidden layers (hidden_dims[0], hidden_dims[1])")
        if len(self.classifier_layers) > _lib.NM_MAX_CLS or any(w < 1 or w > _lib.NM_MAX_CLS_WIDTH for w in self.classifier_layers):
            raise ValueError(f"classifier_layers: at most {_lib.NM_MAX_CLS} widths in 1..{_lib.NM_MAX_CLS_WIDTH}, "
                             f"got {list(self.classifier_layers)}")
        if self.classifier_layers and not (2 <= self.num_classes <= _lib.NM_MAX_CLASSES):
            raise ValueError(f"num_classes must be 2..{_lib.NM_MAX_CLASSES}")

    # encoder / decoder layer sizes exactly as the reference computes them
    def enc_sizes(self, m: int) -> List[int]:
        return [self.input_dims[m] + self.net_c_dim] + list(self.hidden) + [self.latent]      # cVAE.py:153

    def dec_sizes(self, m: int) -> List[int]:
        hd = (list(self.hidden) + [self.latent])[::-1]                                    # cVAE.py:183
        sizes = hd + [self.input_dims[m]]
        sizes[0] = hd[0] + self.net_c_dim                                                 # cVAE.py:188
        return sizes

    def enc_prefix(self, m: int) -> str:
        return "encoder." if self.kind == "single" else f"encoder_list.{m}."

    def dec_prefix(self, m: int, bank: str = "") -> str:
        if self.kind == "single":
            return "decoder."
        if self.kind == "endtoend":
            return f"decoder_list_{bank}.{m}."
        return f"decoder_list.{m}."

    def kernel_modalities(self):
        """(table index, has_encoder, decoder prefix) of every decoder the kernel runs.  The end-to-end
        model's disease bank follows the health bank as decoder-only modalities on the same tables."""
        if self.kind == "endtoend":
            return ([(m, True, self.dec_prefix(m, "health")) for m in range(self.M)] +
                    [(m, False, self.dec_prefix(m, "disease")) for m in range(self.M)])
        return [(m, True, self.dec_prefix(m)) for m in range(self.M)]


def tensor_table(spec: ModelSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in the reference's registration order."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    L = len(spec.hidden)

    def enc(m):
        p, s = spec.enc_prefix(m), spec.enc_sizes(m)
        for i in range(L):
            out.append((f"{p}encoder_layers.{i}.weight", (s[i + 1], s[i])))
            out.append((f"{p}encoder_layers.{i}.bias", (s[i + 1],)))
        for h in ("enc_mean_layer", "enc_logvar_layer"):
            out.append((f"{p}{h}.weight", (s[-1], s[-2])))
            out.append((f"{p}{h}.bias", (s[-1],)))

    def dec(m, bank=""):
        p, s = spec.dec_prefix(m, bank), spec.dec_sizes(m)
        out.append((f"{p}logvar_out", (1, spec.input_dims[m])))
        for i in range(L):
            out.append((f"{p}decoder_layers.{i}.weight", (s[i + 1], s[i])))
            out.append((f"{p}decoder_layers.{i}.bias", (s[i + 1],)))
        out.append((f"{p}decoder_mean_layer.weight", (s[-1], s[-2])))
        out.append((f"{p}decoder_mean_layer.bias", (s[-1],)))

    if spec.is_dm:                           # cVAE.py:1453-1479; a module's own parameter (`weights`) precedes its children's
        if spec.kind == "weighted_dmvae":
            out.append(("weights", (spec.M,)))
        for m in range(spec.M):
            s = spec.enc_sizes(m)
            for l, (n_out, n_in) in (("fc1", (s[1], s[0])), ("fc2", (s[2], s[1])), ("fc_mu", (s[3], s[2])), ("fc_logvar", (s[3], s[2]))):
                out.append((f"encoder_list.{m}.{l}.weight", (n_out, n_in)))
                out.append((f"encoder_list.{m}.{l}.bias", (n_out,)))
        for m in range(spec.M):
            s = spec.dec_sizes(m)
            for l, (n_out, n_in) in (("fc1", (s[1], s[0])), ("fc2", (s[2], s[1])), ("fc_out", (s[3], s[2]))):
                out.append((f"decoder_list.{m}.{l}.weight", (n_out, n_in)))
                out.append((f"decoder_list.{m}.{l}.bias", (n_out,)))
    elif spec.kind == "single":
        enc(0)
        dec(0)
    elif spec.kind in ("multimodal", "mvtcae"):      # mvtCAE registers the same modules in the same order (cVAE.py:1775-1777)
        for m in range(spec.M):
            out.append((f"alpha_m_list.{m}", (1,)))
        for m in range(spec.M):
            enc(m)
        for m in range(spec.M):
            dec(m)
    elif spec.kind == "regression":          # cVAE.py:2230-2253: encoder_list, decoder_list, alpha_m_list, regressor
        for m in range(spec.M):
            enc(m)
        for m in range(spec.M):
            dec(m)
        for m in range(spec.M):
            out.append((f"alpha_m_list.{m}", (1,)))
        sizes = [sum(spec.input_dims)] + list(REGRESSOR_WIDTHS)
        for i in range(len(REGRESSOR_WIDTHS)):             # nn.Sequential indices 0, 2, 4 (ReLU at 1, 3)
            out.append((f"regressor.{2 * i}.weight", (sizes[i + 1], sizes[i])))
            out.append((f"regressor.{2 * i}.bias", (sizes[i + 1],)))
    elif spec.kind == "endtoend":            # cVAE.py:2044-2054: encoder_list, decoder_list_health, decoder_list_disease (+ classifier)
        for m in range(spec.M):
            enc(m)
        for bank in ("health", "disease"):
            for m in range(spec.M):
                dec(m, bank)
        if spec.classifier_layers:               # nn.Sequential: Linear 4i, BatchNorm1d 4i+1, ReLU, Dropout; Linear 4n
            sizes = [spec.latent] + list(spec.classifier_layers)
            for i in range(len(sizes) - 1):
                out.append((f"classifier.classifier.{4 * i}.weight", (sizes[i + 1], sizes[i])))
                out.append((f"classifier.classifier.{4 * i}.bias", (sizes[i + 1],)))
                for t in ("weight", "bias", "running_mean", "running_var"):
                    out.append((f"classifier.classifier.{4 * i + 1}.{t}", (sizes[i + 1],)))
            n = 4 * (len(sizes) - 1)
            out.append((f"classifier.classifier.{n}.weight", (spec.num_classes, sizes[-1])))
            out.append((f"classifier.classifier.{n}.bias", (spec.num_classes,)))
    else:
        raise ValueError(f"unknown model kind {spec.kind!r}")
    return out


class ParamLayout:
    def __init__(self, spec: ModelSpec):
        spec.validate()
        self.spec = spec
        self.names: List[str] = []
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        self.offsets: Dict[str, int] = {}          # kernel (tiled) buffer
        self.tiles: Dict[str, Tuple[int, int]] = {}   # weight matrices: (row tiles, column tiles)
        self.nat_offsets: Dict[str, int] = {}      # natural buffer: tensors row-major, back to back
        # kernel-side column of every reference column, for matrices the kernel keeps with gaps: regressor.0.weight has
        # each modality's columns padded to whole 64-column chunks (nm_job_t.reg_w, include/nmhip.h)
        self.colmap: Dict[str, torch.Tensor] = {}
        self._kcols: Dict[str, int] = {}
        if spec.kind == "regression":
            cols, base = [], 0
            for d in spec.input_dims:
                cols.append(torch.arange(d) + base)
                base += (d + HEAD_CHUNK - 1) // HEAD_CHUNK * HEAD_CHUNK
            self.colmap["regressor.0.weight"] = torch.cat(cols)
            self._kcols = {"regressor.0.weight": base}
        off = nat = 0
        for name, shape in tensor_table(spec):
            self.names.append(name)
            self.shapes[name] = shape
            if name.endswith(".weight") and len(shape) == 2:
                off = (off + TILE * TILE - 1) // (TILE * TILE) * (TILE * TILE)
                kcols = self._kcols[name] if name in self.colmap else shape[1]
                nt, kt = (shape[0] + TILE - 1) // TILE, (kcols + TILE - 1) // TILE
                self.tiles[name] = (nt, kt)
                n = nt * kt * TILE * TILE
            else:
                n = math.prod(shape)
            self.offsets[name] = off
            off += (n + ALIGN - 1) // ALIGN * ALIGN
            self.nat_offsets[name] = nat
            nat += (math.prod(shape) + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.nat_total = nat
        self.n_params = sum(math.prod(s) for s in self.shapes.values())
        self._perm = None

    def numel(self, name: str) -> int:
        return math.prod(self.shapes[name])

    # -- kernel (tiled) buffer ---------------------------------------------------------------------
    def _tile_view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        """[NT, 16, KT, 16] view of a weight matrix inside the kernel buffer (element [a, r, b, c] = W[16a + r][16b + c])."""
        nt, kt = self.tiles[name]
        o = self.offsets[name]
        return flat[o:o + nt * kt * TILE * TILE].view(nt, kt, TILE, TILE).permute(0, 2, 1, 3)

    def get(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        """One tensor out of the kernel buffer (weights: a copy; vectors: a view)."""
        if name in self.tiles:
            n, k = self.shapes[name]
            nt, kt = self.tiles[name]
            full = self._tile_view(flat, name).reshape(nt * TILE, kt * TILE)
            if name in self.colmap:
                return full[:n][:, self.colmap[name].to(flat.device)]
            return full[:n, :k]
        o = self.offsets[name]
        return flat[o:o + self.numel(name)].view(self.shapes[name])

    def put(self, flat: torch.Tensor, name: str, t: torch.Tensor):
        t = t.to(dtype=torch.float32, device=flat.device)
        if name in self.tiles:
            n, k = self.shapes[name]
            nt, kt = self.tiles[name]
            pad = torch.zeros(nt * TILE, kt * TILE, dtype=torch.float32, device=flat.device)
            if name in self.colmap:
                pad[:n, self.colmap[name].to(flat.device)] = t
            else:
                pad[:n, :k] = t
            self._tile_view(flat, name).copy_(pad.view(nt, TILE, kt, TILE))
        else:
            self.get(flat, name).copy_(t)

    def unflatten(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {n: self.get(flat, n) for n in self.names}

    def flatten(self, state: Dict[str, torch.Tensor], device=None) -> torch.Tensor:
        flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        for n in self.names:
            if n not in state:
                raise KeyError(f"state_dict is missing {n!r}")
            t = state[n]
            if tuple(t.shape) != tuple(self.shapes[n]):
                raise ValueError(f"{n}: expected shape {self.shapes[n]}, got {tuple(t.shape)}")
            self.put(flat, n, t)
        return flat

    # -- natural buffer (eager facade: parameter VIEWS with the reference's names) -------------------
    def nat_views(self, nat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {n: nat[self.nat_offsets[n]:self.nat_offsets[n] + self.numel(n)].view(self.shapes[n]) for n in self.names}

    def nat_flatten(self, state: Dict[str, torch.Tensor], device=None) -> torch.Tensor:
        nat = torch.zeros(self.nat_total, dtype=torch.float32, device=device)
        for n, v in self.nat_views(nat).items():
            if tuple(state[n].shape) != tuple(self.shapes[n]):
                raise ValueError(f"{n}: expected shape {self.shapes[n]}, got {tuple(state[n].shape)}")
            v.copy_(state[n].to(torch.float32))
        return nat

    def perm(self, device=None) -> torch.Tensor:
        """perm[i] = position in the kernel buffer of element i of the natural buffer (-1 for alignment gaps)."""
        if self._perm is None:
            if self.total >= 1 << 24:
                raise ValueError("model too large for the float index trick")
            idx = torch.arange(self.total, dtype=torch.float32)       # exact below 2^24
            nat = torch.full((self.nat_total,), -1.0)
            for n, v in self.nat_views(nat).items():
                v.copy_(self.get(idx, n))
            self._perm = nat.to(torch.int64)
        return self._perm if device is None else self._perm.to(device)

    def _perm_pair(self, device):
        """(positions in the natural buffer that hold an element, their positions in the kernel buffer), on `device`,
        built once per device: both conversions below are then two index kernels, no mask, no host round trip."""
        key = str(device)
        cache = self.__dict__.setdefault("_perm_dev", {})
        if key not in cache:
            p = self.perm()
            ok = (p >= 0).nonzero().flatten()
            cache[key] = (ok.to(device), p[ok].to(device))
        return cache[key]

    def nat_to_kernel(self, nat: torch.Tensor, out: torch.Tensor):
        """natural buffer -> kernel buffer (16 x 16 tiles, zero padding)."""
        i_nat, i_ker = self._perm_pair(nat.device)
        out.zero_()
        out.index_copy_(0, i_ker, nat.index_select(0, i_nat))
        return out

    def kernel_to_nat(self, flat: torch.Tensor, out: torch.Tensor):
        i_nat, i_ker = self._perm_pair(flat.device)
        out.index_copy_(0, i_nat, flat.index_select(0, i_ker))
        return out

    def init_reference_rule(self, seed: int = 42) -> Dict[str, torch.Tensor]:
        """nn.Linear default init U(+-1/sqrt(fan_in)) for weight and bias, logvar_out = -3,
        alpha ~ N(0,1)  (cVAE.py:155-159, 179, 190-194, 1106)."""
        g = torch.Generator().manual_seed(seed)
        out: Dict[str, torch.Tensor] = {}
        for n in self.names:
            shape = self.shapes[n]
            if n == "weights":                         # torch.abs(torch.randn(modalities)), cVAE.py:1650
                out[n] = torch.randn(shape, generator=g).abs()
            elif n.endswith("logvar_out"):
                out[n] = torch.full(shape, -3.0)
            elif len(shape) == 1 and n.endswith((".weight", ".running_var")):      # BatchNorm1d defaults
                out[n] = torch.ones(shape)
            elif n.endswith(".running_mean") or (n.endswith(".bias") and n[:-4] + "running_mean" in self.shapes):
                out[n] = torch.zeros(shape)
            elif n.startswith("alpha_m_list"):
                out[n] = torch.randn(shape, generator=g)
            elif n.endswith(".weight"):
                bound = 1.0 / math.sqrt(shape[1])
                out[n] = (torch.rand(shape, generator=g) * 2 - 1) * bound
            else:
                fan_in = self.shapes[n[:-4] + "weight"][1]
                bound = 1.0 / math.sqrt(fan_in)
                out[n] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        return out

    def fill_modality(self, md: "_lib.NmModality", j: int):
        """Tensor offsets of kernel modality j (see ModelSpec.kernel_modalities)."""
        spec, o = self.spec, self.offsets
        m, has_enc, dp = spec.kernel_modalities()[j]
        ep = spec.enc_prefix(m)
        if spec.is_dm:
            md.enc_w[0], md.enc_b[0] = o[f"{ep}fc1.weight"], o[f"{ep}fc1.bias"]
            md.enc_w[1], md.enc_b[1] = o[f"{ep}fc2.weight"], o[f"{ep}fc2.bias"]
            md.mu_w, md.mu_b = o[f"{ep}fc_mu.weight"], o[f"{ep}fc_mu.bias"]
            md.lv_w, md.lv_b = o[f"{ep}fc_logvar.weight"], o[f"{ep}fc_logvar.bias"]
            md.dec_w[0], md.dec_b[0] = o[f"{dp}fc1.weight"], o[f"{dp}fc1.bias"]
            md.dec_w[1], md.dec_b[1] = o[f"{dp}fc2.weight"], o[f"{dp}fc2.bias"]
            md.out_w, md.out_b = o[f"{dp}fc_out.weight"], o[f"{dp}fc_out.bias"]
            md.logvar_out, md.alpha = -1, -1
            return
        for i in range(len(spec.hidden)):
            md.enc_w[i] = o[f"{ep}encoder_layers.{i}.weight"] if has_enc else 0
            md.enc_b[i] = o[f"{ep}encoder_layers.{i}.bias"] if has_enc else 0
            md.dec_w[i] = o[f"{dp}decoder_layers.{i}.weight"]
            md.dec_b[i] = o[f"{dp}decoder_layers.{i}.bias"]
        md.mu_w = o[f"{ep}enc_mean_layer.weight"] if has_enc else 0
        md.mu_b = o[f"{ep}enc_mean_layer.bias"] if has_enc else 0
        md.lv_w = o[f"{ep}enc_logvar_layer.weight"] if has_enc else 0
        md.lv_b = o[f"{ep}enc_logvar_layer.bias"] if has_enc else 0
        md.logvar_out = o[f"{dp}logvar_out"]
        md.out_w, md.out_b = o[f"{dp}decoder_mean_layer.weight"], o[f"{dp}decoder_mean_layer.bias"]
        md.alpha = o.get(f"alpha_m_list.{m}", -1) if has_enc else -1

    def fill_head(self, job: "_lib.NmJob"):
        """Offsets of the head tensors: regressor (kind == "regression"), classifier (kind == "endtoend")."""
        s = self.spec
        job.cls_layers, job.cls_classes = 0, 0
        if s.kind == "endtoend" and s.classifier_layers:
            job.cls_layers, job.cls_classes = len(s.classifier_layers), s.num_classes
            for i, w in enumerate(s.classifier_layers):
                job.cls_width[i] = w
                job.cls_w[i] = self.offsets[f"classifier.classifier.{4 * i}.weight"]
                job.cls_b[i] = self.offsets[f"classifier.classifier.{4 * i}.bias"]
                job.cls_bn_w[i] = self.offsets[f"classifier.classifier.{4 * i + 1}.weight"]
                job.cls_bn_b[i] = self.offsets[f"classifier.classifier.{4 * i + 1}.bias"]
                job.cls_bn_mean[i] = self.offsets[f"classifier.classifier.{4 * i + 1}.running_mean"]
                job.cls_bn_var[i] = self.offsets[f"classifier.classifier.{4 * i + 1}.running_var"]
            n = len(s.classifier_layers)
            job.cls_w[n] = self.offsets[f"classifier.classifier.{4 * n}.weight"]
            job.cls_b[n] = self.offsets[f"classifier.classifier.{4 * n}.bias"]
        if self.spec.kind != "regression":
            job.reg_head = 0
            return
        job.reg_head = 1
        for i in range(3):
            job.reg_w[i] = self.offsets[f"regressor.{2 * i}.weight"]
            job.reg_b[i] = self.offsets[f"regressor.{2 * i}.bias"]
