"""Input preparation on the device: a fold's ROI tables straight from the raw cohort in HBM.

``DeviceCohort`` uploads the cohort once (fp64, as the reference's DataFrames hold it); ``fold_tables`` then builds the
``Table`` objects of one (fold, procedure) -- RobustScaler fit on the fold's rows, rank-quantile one-hot covariates,
early-fusion column concat, bf16 chunk images + fp32 copy + covariate block -- with three kernels of libnmhip.so
(csrc/nm_prep.hip) and no pass over the data on the host.  Bit-identical to ``prep.fold_train_tables`` followed by
``Table(x, c)`` (tests/test_gpu_prep.py); the fold's row indices (KFold, bootstrap ids) still come from ``prep``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, prep
from .engine import BATCH, Table, _stream_ptr, require_gpu


class DeviceCohort:
    def __init__(self, cohort: prep.SyntheticCohort, device=None):
        self.device = require_gpu(device)
        self.cohort = cohort
        self.x: Dict[str, torch.Tensor] = {m: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float64)).to(self.device)
                                           for m, v in cohort.x.items()}
        self.age = torch.as_tensor(np.asarray(cohort.age, dtype=np.float64)).to(self.device)
        self.gender = torch.as_tensor(np.asarray(cohort.gender, dtype=np.float64)).to(self.device)
        self._edges: Dict[Tuple[int, int], torch.Tensor] = {}
        self._fold_cache: Dict[tuple, object] = {}

    def _sources(self, modality: str):
        """(device array of source pointers, device array of widths, n_src, D) -- several sources = early fusion."""
        names = self.cohort.modalities if (modality not in self.x and prep.is_fusion(modality)) else [modality]
        tens = [self.x[n] for n in names]
        ptrs = torch.tensor([t.data_ptr() for t in tens], dtype=torch.int64, device=self.device)
        widths = torch.tensor([t.shape[1] for t in tens], dtype=torch.int32, device=self.device)
        return ptrs, widths, len(tens), int(sum(t.shape[1] for t in tens)), tens

    def edges(self, n: int, q: int) -> torch.Tensor:
        if (n, q) not in self._edges:
            self._edges[(n, q)] = torch.as_tensor(prep.qcut_edges(n, q)).to(self.device)
        return self._edges[(n, q)]

    def scaler_fit(self, modality: str, rows: torch.Tensor):
        """RobustScaler().fit on the given rows -> (center, scale) fp64 device tensors [D]."""
        lib = _lib.load()
        ptrs, widths, n_src, D, keep = self._sources(modality)
        center = torch.empty(D, dtype=torch.float64, device=self.device)
        scale = torch.empty(D, dtype=torch.float64, device=self.device)
        _lib.check(lib.nm_prep_scaler_fit(ptrs.data_ptr(), widths.data_ptr(), n_src, D, rows.data_ptr(), int(rows.numel()),
                                          center.data_ptr(), scale.data_ptr(), _stream_ptr(self.device)), "nm_prep_scaler_fit")
        return center, scale

    def one_hot(self, rows: torch.Tensor, age_bins: int = 27, gender_bins: int = 2) -> torch.Tensor:
        """one_hot_covariates of the given rows -> fp32 [n, 29] on the device."""
        lib = _lib.load()
        n = int(rows.numel())
        c = torch.empty(n, age_bins + gender_bins, dtype=torch.float32, device=self.device)
        _lib.check(lib.nm_prep_onehot(self.age.data_ptr(), self.gender.data_ptr(), rows.data_ptr(), n,
                                      self.edges(n, age_bins).data_ptr(), age_bins, self.edges(n, gender_bins).data_ptr(),
                                      gender_bins, c.data_ptr(), _stream_ptr(self.device)), "nm_prep_onehot")
        return c

    def table(self, modality: str, rows: torch.Tensor, center: torch.Tensor, scale: torch.Tensor, c: torch.Tensor) -> Table:
        """The packed Table of `rows` of one modality, scaled by (center, scale), with covariates c [n, C]."""
        lib = _lib.load()
        ptrs, widths, n_src, D, keep = self._sources(modality)
        n, Cc = int(rows.numel()), int(c.shape[1])
        t = Table.__new__(Table)
        t.device = self.device
        t.N, t.D, t.C = n, D, Cc
        t.rows_alloc = max(1, math.ceil(n / BATCH)) * BATCH
        t.Kx = (D + Cc + 1 + 31) // 32 * 32
        t.Cz = (Cc + 1 + 7) // 8 * 8
        t.x_pitch = (D + 3) // 4 * 4
        t.x_f32 = torch.empty(t.rows_alloc, t.x_pitch, dtype=torch.float32, device=self.device)
        t.xb = torch.empty(int(lib.nm_xb_elems(t.rows_alloc, t.Kx)), dtype=torch.bfloat16, device=self.device)
        t.cz = torch.empty(t.rows_alloc, t.Cz, dtype=torch.bfloat16, device=self.device)
        t.c_key = ("dev", c.data_ptr(), tuple(c.shape), str(c.dtype), c._version)
        _lib.check(lib.nm_pack_table_raw(ptrs.data_ptr(), widths.data_ptr(), n_src, rows.data_ptr(), n, center.data_ptr(),
                                         scale.data_ptr(), c.data_ptr(), t.rows_alloc, D, Cc, t.Kx, t.xb.data_ptr(),
                                         t.x_f32.data_ptr(), t.x_pitch, t.cz.data_ptr(), t.Cz, _stream_ptr(self.device)),
                   "nm_pack_table_raw")
        t._keep = (c,)
        return t

    def fold_tables(self, modalities: Sequence[str], train_idx: np.ndarray) -> List[Table]:
        """prep.fold_train_tables + Table(...) on the device: scaler fit on the fold's rows, covariates binned on the
        same rows (multimodal_kfold_train_cvae_supervised.py:101-126)."""
        rows = torch.as_tensor(np.asarray(train_idx, dtype=np.int32)).to(self.device)
        c = self.one_hot(rows)
        out = []
        for m in modalities:
            center, scale = self.scaler_fit(m, rows)
            out.append(self.table(m, rows, center, scale, c))
        return out

    def fold_tables_cached(self, key, modalities: Sequence[str], train_idx: np.ndarray, with_covariates: bool = True) -> List[Table]:
        """fold_tables with the (fold key, modality) tables and the fold's covariates built once: the models of a fold
        share them (same HBM buffers: one copy in L2 / Infinity Cache for all of them, nm_job_t.shared_cov)."""
        ck = (key, "rows")
        if ck not in self._fold_cache:
            rows = torch.as_tensor(np.asarray(train_idx, dtype=np.int32)).to(self.device)
            self._fold_cache[ck] = (rows, self.one_hot(rows))
        rows, c = self._fold_cache[ck]
        if not with_covariates:
            c = torch.zeros(int(rows.numel()), 0, dtype=torch.float32, device=self.device)
        out = []
        for m in modalities:
            tk = (key, m, with_covariates)
            if tk not in self._fold_cache:
                center, scale = self.scaler_fit(m, rows)
                self._fold_cache[tk] = self.table(m, rows, center, scale, c)
            out.append(self._fold_cache[tk])
        return out
