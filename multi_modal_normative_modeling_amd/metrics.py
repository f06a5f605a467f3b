"""Post-hoc metrics on the device: host side of csrc/nm_metrics.hip.

`posthoc_metrics` = compute_classification_performance(method='roc') of
multimodal_kfold_cvae_group_analysis_1x1.py:105-157 for many score sets at once (one workgroup per set);
`confusion_metrics` = evaluate() of multimodal_kfold_cvae_nmpmcont.py:29-70 from hard predictions.
Scores stay on the GPU (they are the per-subject mean deviations the forward pass exported); only the
[n_sets, 8] fp64 result table comes back.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from .engine import _stream_ptr, require_gpu

POSTHOC_COLUMNS = ("roc_auc", "threshold", "accuracy", "recall", "specificity", "significance_ratio", "n_pos", "n_neg")
CONFUSION_COLUMNS = ("accuracy", "auroc", "sensitivity", "specificity", "f1_score", "precision", "n_pos", "n_neg")


def _segments(parts: Sequence[torch.Tensor], device, dtype):
    sizes = [int(p.numel()) for p in parts]
    if not sizes:
        raise ValueError("no score sets")
    off = torch.zeros(len(sizes) + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.int64), 0).to(torch.int32)
    flat = torch.cat([p.reshape(-1).to(device=device, dtype=dtype) for p in parts]) if sum(sizes) else \
        torch.zeros(1, dtype=dtype, device=device)
    return flat.contiguous(), off.to(device), sizes


def posthoc_metrics(scores: Sequence[torch.Tensor], positive: Sequence[torch.Tensor],
                    thresholds: Optional[Sequence[float]] = None, device="cuda:0") -> torch.Tensor:
    """One row per set: roc_auc, threshold (Youden's J unless given), accuracy, recall, specificity,
    significance_ratio, n_pos, n_neg.  `positive[i] != 0` marks the class counted as 1 (disease for
    training_class == 'nm', group_analysis_1x1.py:118-121)."""
    dev = require_gpu(device)
    if len(scores) != len(positive):
        raise ValueError("scores and positive must have the same number of sets")
    s, off, sizes = _segments(scores, dev, torch.float32)
    l, _, sizes_l = _segments(positive, dev, torch.int32)
    if sizes != sizes_l:
        raise ValueError("every score set needs one label per score")
    if max(sizes) > _lib.NM_METRICS_MAX_N:
        raise ValueError(f"at most {_lib.NM_METRICS_MAX_N} scores per set, got {max(sizes)}")
    out = torch.empty(len(sizes), _lib.NM_METRICS_STRIDE, dtype=torch.float64, device=dev)
    thr = None
    if thresholds is not None:
        thr = torch.tensor([float(t) for t in thresholds], dtype=torch.float64, device=dev)
        if thr.numel() != len(sizes):
            raise ValueError("one threshold per set")
    _lib.check(_lib.load().nm_posthoc_metrics(s.data_ptr(), l.data_ptr(), off.data_ptr(), len(sizes), max(max(sizes), 1),
                                               thr.data_ptr() if thr is not None else None, out.data_ptr(),
                                               _stream_ptr(dev)), "nm_posthoc_metrics")
    return out


def confusion_metrics(pred: Sequence[torch.Tensor], labels: Sequence[torch.Tensor], device="cuda:0") -> torch.Tensor:
    """One row per set: accuracy, auroc, sensitivity, specificity, f1_score, precision, n_pos, n_neg."""
    dev = require_gpu(device)
    p, off, sizes = _segments(pred, dev, torch.int32)
    l, _, sizes_l = _segments(labels, dev, torch.int32)
    if sizes != sizes_l:
        raise ValueError("every prediction needs one label")
    out = torch.empty(len(sizes), _lib.NM_METRICS_STRIDE, dtype=torch.float64, device=dev)
    _lib.check(_lib.load().nm_confusion_metrics(p.data_ptr(), l.data_ptr(), off.data_ptr(), len(sizes), out.data_ptr(),
                                                 _stream_ptr(dev)), "nm_confusion_metrics")
    return out
