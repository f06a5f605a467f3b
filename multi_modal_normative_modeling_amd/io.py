"""On-disk formats of the deviation pass (same pandas calls as the reference, so the text is
byte-compatible with its tooling).

``deviation_fold_{fold}_{dataset_name}_roiwise.csv``: header ``IID,ROI_0..ROI_{D-1}``, one row per
subject in table order, float32 values  (multimodal_kfold_train_cvae_supervised_regression.py:190-192).
The five CSV kinds of the test script (multimodal_kfold_test_cvae_supervised.py:116-154).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Sequence

import numpy as np
import pandas as pd

META_COLS = ["participant_id", "DIA", "AGE", "PTGENDER"]


def roiwise_filename(fold: int, dataset_name: str) -> str:
    return f"deviation_fold_{fold}_{dataset_name}_roiwise.csv"


def write_roiwise_csv(out_dir, fold: int, dataset_name: str, iids: Sequence[int], deviation_roi: np.ndarray) -> Path:
    dev = np.asarray(deviation_roi, dtype=np.float32)
    df_out = pd.DataFrame(dev, columns=[f"ROI_{i}" for i in range(dev.shape[1])])
    df_out.insert(0, "IID", list(iids))
    path = Path(out_dir) / roiwise_filename(fold, dataset_name)
    path.parent.mkdir(parents=True, exist_ok=True)
    df_out.to_csv(path, index=False)
    return path


def write_test_csvs(out_dir, dataset_name: str, covariates: pd.DataFrame, roi_columns: Sequence[str], x: np.ndarray,
                    x_hat: np.ndarray) -> Dict[str, Path]:
    """normalized_ / reconstruction_ / reconstruction_error_ / reconstruction_error_roi_ /
    deviation_as_feature_importance_{name}.csv with the reference's column layouts."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    cov = covariates[META_COLS].copy()
    paths = {}
    cov = cov.reset_index(drop=True)

    def with_rois(values):                      # same frame as `df[columns_name] = values` (…test….py:123-124), built in one go
        return pd.concat([cov, pd.DataFrame(np.asarray(values), columns=list(roi_columns))], axis=1)

    normalized = with_rois(x)
    paths["normalized"] = out_dir / f"normalized_{dataset_name}.csv"
    normalized.to_csv(paths["normalized"], index=False)
    recon = with_rois(x_hat)
    paths["reconstruction"] = out_dir / f"reconstruction_{dataset_name}.csv"
    recon.to_csv(paths["reconstruction"], index=False)
    err = cov.copy()
    err["Reconstruction error"] = np.sum((x - x_hat) ** 2, axis=1) / x.shape[1]
    paths["reconstruction_error"] = out_dir / f"reconstruction_error_{dataset_name}.csv"
    err.to_csv(paths["reconstruction_error"], index=False)
    err_roi = with_rois((x - x_hat) ** 2)
    paths["reconstruction_error_roi"] = out_dir / f"reconstruction_error_roi_{dataset_name}.csv"
    err_roi.to_csv(paths["reconstruction_error_roi"], index=False)
    fi = err_roi.rename(columns=dict(zip(roi_columns, map(str, range(1, len(roi_columns) + 1)))))
    paths["deviation_as_feature_importance"] = out_dir / f"deviation_as_feature_importance_{dataset_name}.csv"
    fi.to_csv(paths["deviation_as_feature_importance"], index=False)
    return paths
