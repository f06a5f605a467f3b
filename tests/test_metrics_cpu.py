"""The metrics oracle (oracle/metrics_ref.py) pinned against the installed scikit-learn, which is the library
the reference calls for these numbers (multimodal_kfold_cvae_group_analysis_1x1.py:125-126,
multimodal_kfold_cvae_nmpmcont.py:50-62)."""
import warnings

import numpy as np
import pytest
from sklearn import metrics as skm

from oracle import metrics_ref as MR


def _cases():
    rng = np.random.default_rng(7)
    out = []
    for n, kind in [(213, "normal"), (213, "ties"), (64, "anti"), (17, "coarse"), (1000, "normal"), (5, "ties"),
                    (300, "constant"), (2, "normal"), (257, "mixed")]:
        lab = (rng.random(n) < 0.3).astype(np.int32)
        if lab.sum() == 0:
            lab[0] = 1
        if lab.sum() == n:
            lab[0] = 0
        if kind == "normal":
            s = rng.normal(size=n) + 0.8 * lab
        elif kind == "ties":
            s = np.round(rng.normal(size=n) + 0.8 * lab, 1)
        elif kind == "anti":
            s = rng.normal(size=n) - 1.5 * lab
        elif kind == "coarse":
            s = rng.integers(0, 4, size=n).astype(float)
        elif kind == "constant":
            s = np.full(n, 0.25)
        else:
            s = np.where(rng.random(n) < 0.5, np.round(rng.normal(size=n), 0), rng.normal(size=n)) + 0.5 * lab
        out.append((s.astype(np.float32), lab))
    return out


@pytest.mark.parametrize("case", range(9))
def test_roc_points_and_metrics_match_sklearn(case):
    s, lab = _cases()[case]
    fpr, tpr, thr = skm.roc_curve(lab, s)
    f2, t2, th2, _, _ = MR.roc_points(lab, s)
    assert np.array_equal(fpr, f2) and np.array_equal(tpr, t2) and np.array_equal(thr, th2)
    m = MR.posthoc_metrics(s, lab)
    assert m[0] == skm.auc(fpr, tpr)
    assert abs(m[0] - skm.roc_auc_score(lab, s)) < 1e-12
    best = thr[np.argmax(tpr - fpr)]
    assert m[1] == best
    pred = (s >= best).astype(int)
    assert m[2] == skm.accuracy_score(lab, pred)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert m[3] == skm.recall_score(lab, pred)
        tn, fp, fn, tp = skm.confusion_matrix(lab, pred, labels=[0, 1]).ravel()
    assert m[4] == tn / (tn + fp)
    assert m[6] == lab.sum() and m[7] == len(lab) - lab.sum()


def test_given_threshold_and_single_class():
    s, lab = _cases()[0]
    m = MR.posthoc_metrics(s, lab, optimal_threshold=0.3)
    pred = (s >= 0.3).astype(int)
    assert m[1] == 0.3 and m[2] == (pred == lab).mean()
    one = MR.posthoc_metrics(s, np.zeros_like(lab))
    assert np.isnan(one[:6]).all() and one[6] == 0 and one[7] == len(lab)


@pytest.mark.parametrize("seed", range(4))
def test_confusion_metrics_match_sklearn(seed):
    rng = np.random.default_rng(seed)
    n = 150
    lab = (rng.random(n) < 0.4).astype(int)
    pred = np.where(rng.random(n) < 0.75, lab, 1 - lab)
    if seed == 3:
        pred = np.zeros(n, dtype=int)          # never predicts the positive class
    m = MR.confusion_metrics(pred, lab)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert m[0] == skm.accuracy_score(lab, pred)
        assert abs(m[1] - skm.roc_auc_score(lab, pred)) < 1e-15
        assert m[2] == skm.recall_score(lab, pred)
        assert m[5] == skm.precision_score(lab, pred)
        assert abs(m[4] - skm.f1_score(lab, pred)) < 1e-15
    tn, fp, fn, tp = skm.confusion_matrix(lab, pred).ravel()
    assert m[3] == tn / (tn + fp)
    only = MR.confusion_metrics(pred, np.ones(n, dtype=int))
    assert np.isnan(only[1])
