"""On-disk formats of the deviation pass (same pandas calls as the reference, so the text is
byte-compatible with its tooling).

``deviation_fold_{fold}_{dataset_name}_roiwise.csv``: header ``IID,ROI_0..ROI_{D-1}``, one row per
subject in table order, float32 values  (multimodal_kfold_train_cvae_supervised_regression.py:190-192).
The five CSV kinds of the test script (multimodal_kfold_test_cvae_supervised.py:116-154).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional, Sequence

import numpy as np
import pandas as pd

from . import prep

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


# ---- the reference's input layout -----------------------------------------------------------
# data/<resource>/y.csv: IID, participant_id, DIA, AGE, PTGENDER (+ FI for HCPimage); data/<resource>/<modality>.csv:
# IID + the ROI columns.  multimodal_kfold_train_cvae_supervised.py:49-50, 84-91 and utils.py:110-168 read them per
# fold and modality through pd.merge on IID; here they are read once into a prep.Cohort.
_NON_ROI = ("IID", "participant_id", "DIA", "AGE", "PTGENDER", "FI", "Session_ID", "Run_ID")


def read_cohort(resource_dir, resource: str = "HCPimage", modalities: Optional[Sequence[str]] = None) -> prep.Cohort:
    """y.csv (rows with missing values dropped, utils.py:124) inner-merged on IID with every modality table of the
    resource (get_datasets_name order).  Rows follow the first modality's file (pd.merge keeps the left table's order,
    utils.py:117); the other modalities are aligned to it by IID -- the reference feeds the modalities' DataLoaders side by
    side and so assumes the files agree.  ROI columns = every column of the modality file that is not an id / covariate,
    in file order.  DIA is mapped to the cohort's convention 1 = healthy control through get_hc_label (utils.py:760-774).
    A missing FI column (only HCPimage carries it) reads as zeros."""
    resource_dir = Path(resource_dir)
    if resource not in prep.DATASET_MODALITIES:
        raise ValueError("Unknown dataset: {}".format(resource))
    names = list(modalities) if modalities is not None else [m for m in prep.DATASET_MODALITIES[resource]
                                                             if (resource_dir / f"{m}.csv").exists()]
    if not names:
        raise FileNotFoundError(f"no modality tables of {resource} under {resource_dir}")
    y = pd.read_csv(resource_dir / "y.csv").dropna()
    for col in ("IID", "DIA", "AGE", "PTGENDER"):
        if col not in y.columns:
            raise ValueError(f"{resource_dir / 'y.csv'} has no column {col!r}")
    tables = {m: pd.read_csv(resource_dir / f"{m}.csv") for m in names}
    keep = set(y["IID"])
    for t in tables.values():
        keep &= set(t["IID"])
    first = tables[names[0]]
    order = [i for i in first["IID"].tolist() if i in keep]
    if len(set(order)) != len(order):
        raise ValueError(f"{names[0]}.csv repeats IIDs")
    yi = y.drop_duplicates("IID").set_index("IID").loc[order]
    x = {}
    for m, t in tables.items():
        roi = [c for c in t.columns if c not in _NON_ROI]
        x[m] = t.drop_duplicates("IID").set_index("IID").loc[order, roi].to_numpy(dtype=np.float64)
    hc = prep.HC_LABEL[resource]
    fi = yi["FI"].to_numpy(dtype=np.float64) if "FI" in yi.columns else np.zeros(len(order))
    return prep.Cohort(iid=np.asarray(order), age=yi["AGE"].to_numpy(dtype=np.float64), gender=yi["PTGENDER"].to_numpy(dtype=np.float64),
                       dia=(yi["DIA"].to_numpy() == hc).astype(np.int64), fi=fi, x=x, resource=resource)


def write_cohort(cohort: prep.Cohort, resource_dir, roi_names: Optional[dict] = None) -> None:
    """The inverse of read_cohort: a cohort (e.g. the synthetic one) in the reference's ./data/<resource>/ layout, plus
    the early-fusion table the way early_fusion_modalities.py:23-35 builds it (columns `<roi>_<modality>`, modality-major)."""
    resource_dir = Path(resource_dir)
    resource_dir.mkdir(parents=True, exist_ok=True)
    hc = prep.HC_LABEL[cohort.resource]
    dia = np.where(np.asarray(cohort.dia) == 1, hc, 0)
    pd.DataFrame({"IID": cohort.iid, "participant_id": cohort.iid, "DIA": dia, "AGE": cohort.age, "PTGENDER": cohort.gender,
                  "FI": cohort.fi}).to_csv(resource_dir / "y.csv", index=False)
    fused = []
    for m, xm in cohort.x.items():
        cols = list(roi_names[m]) if roi_names and m in roi_names else [f"ROI_{i}" for i in range(xm.shape[1])]
        df = pd.DataFrame(xm, columns=cols)
        df.insert(0, "IID", cohort.iid)
        df.to_csv(resource_dir / f"{m}.csv", index=False)
        fused.append(pd.DataFrame(xm, columns=[f"{c}_{m}" for c in cols]))
    fdf = pd.concat(fused, axis=1)
    fdf.insert(0, "IID", cohort.iid)
    fdf.to_csv(resource_dir / f"{cohort.fusion_name}.csv", index=False)
