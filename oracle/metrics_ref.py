"""TEST INFRASTRUCTURE (CPU oracle) -- post-hoc metrics of the sweep, restated in numpy.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
computes these numbers with csrc/nm_metrics.hip.

Follows
  * compute_classification_performance(method='roc'), multimodal_kfold_cvae_group_analysis_1x1.py:105-157:
    roc_curve -> auc -> Youden-J threshold -> confusion counts -> accuracy / recall / specificity /
    significance ratio;
  * evaluate(), multimodal_kfold_cvae_nmpmcont.py:29-70 (accuracy, auroc, recall, specificity, f1 on hard
    predictions).
The ROC arithmetic lives in a third-party dependency: scikit-learn (environment.yml pins 1.6.1; this image has
1.7.2, same algorithm since 1.3: the curve starts at threshold +inf).  `roc_points` restates
sklearn.metrics._ranking._binary_clf_curve + roc_curve(drop_intermediate=True); tests/test_metrics_cpu.py
pins it against the installed scikit-learn on random, tied, constant and anti-correlated score sets.
"""
from __future__ import annotations

import numpy as np


def roc_points(labels, scores):
    """(fpr, tpr, thresholds, fps, tps) as sklearn.metrics.roc_curve returns them (drop_intermediate=True)."""
    y = (np.asarray(labels) != 0).astype(np.float64)
    s = np.asarray(scores)
    order = np.argsort(s, kind="mergesort")[::-1]
    s, y = s[order], y[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    thr = s[idx]
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    thr = np.r_[np.inf, thr]
    fpr = fps / fps[-1] if fps[-1] > 0 else np.repeat(np.nan, fps.shape)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.repeat(np.nan, tps.shape)
    return fpr, tpr, thr, fps, tps


def posthoc_metrics(scores, labels, optimal_threshold=None):
    """[roc_auc, threshold, accuracy, recall, specificity, significance_ratio, n_pos, n_neg]
    (group_analysis_1x1.py:125-155).  labels != 0 = the positive class of that script's `labels` list."""
    scores = np.asarray(scores)
    lab = (np.asarray(labels) != 0).astype(int)
    P, N = int(lab.sum()), int((1 - lab).sum())
    if P == 0 or N == 0 or scores.size == 0:
        return np.array([np.nan] * 6 + [P, N], dtype=np.float64)
    fpr, tpr, thr, _, _ = roc_points(lab, scores)
    roc_auc = float(np.trapezoid(tpr, fpr))                       # sklearn.metrics.auc
    if optimal_threshold is None:
        optimal_threshold = thr[int(np.argmax(tpr - fpr))]        # Youden's J, :131-134
    pred = (scores >= optimal_threshold).astype(int)              # :147
    acc = float((pred == lab).mean())
    TP = int(np.sum((pred == 1) & (lab == 1)))
    FN = int(np.sum((pred == 0) & (lab == 1)))
    TN = int(np.sum((pred == 0) & (lab == 0)))
    FP = int(np.sum((pred == 1) & (lab == 0)))
    with np.errstate(divide="ignore", invalid="ignore"):
        sig = np.float64(roc_auc) / np.float64(1 - roc_auc)
    return np.array([roc_auc, float(optimal_threshold), acc, TP / (TP + FN), TN / (TN + FP), sig, P, N],
                    dtype=np.float64)


def confusion_metrics(pred, labels):
    """[accuracy, auroc, sensitivity, specificity, f1, precision, n_pos, n_neg] of evaluate()
    (nmpmcont.py:52-69) for binary hard predictions."""
    p = (np.asarray(pred) != 0)
    l = (np.asarray(labels) != 0)
    TP, FP = int(np.sum(p & l)), int(np.sum(p & ~l))
    TN, FN = int(np.sum(~p & ~l)), int(np.sum(~p & l))
    n = p.size
    acc = (TP + TN) / n if n else np.nan
    sens = TP / (TP + FN) if TP + FN else 0.0                     # recall_score: 0 (with a warning) when undefined
    with np.errstate(divide="ignore", invalid="ignore"):
        spec = float(np.float64(TN) / np.float64(TN + FP))        # tn / (tn + fp), numpy semantics
    auroc = 0.5 * (TP / (TP + FN) + TN / (TN + FP)) if (TP + FN) and (TN + FP) else np.nan   # ValueError -> nan, :56-59
    f1 = 2 * TP / (2 * TP + FP + FN) if 2 * TP + FP + FN else 0.0
    prec = TP / (TP + FP) if TP + FP else 0.0
    return np.array([acc, auroc, sens, spec, f1, prec, TP + FN, TN + FP], dtype=np.float64)
