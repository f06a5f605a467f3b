"""The k-fold x procedure sweep sharded over the GPUs of one node.

Every (fold, procedure[, grid point]) cell is an independent model (the reference runs them as
sequential loop iterations, multimodal_kfold_train_cvae_supervised.py:68,82), so the cells are
dealt round-robin by descending cost to one process per GPU; each process trains its cells
concurrently inside the persistent step kernel and writes its own ROI-wise CSVs.  The only
collective is one all_gather of a small fp32 metric table at the end (RCCL on GPUs, gloo in the
CPU tests) -- there is no data-path exchange.
"""
from __future__ import annotations

import argparse
import time
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import io, metrics, prep, workload
from .engine import Job, JobSet, Table
from .layout import ModelSpec

# one row per cell in the table the final all_gather carries
METRIC_COLUMNS = ("job_id", "fold", "proc_id", "final_total_loss", "steps_per_s", "roc_auc", "threshold", "accuracy",
                  "sensitivity", "specificity", "mean_dev_hc", "mean_dev_dx")
N_METRICS = len(METRIC_COLUMNS)


# -Model values of the train script (multimodal_kfold_train_cvae_supervised.py:149-157) -> (ModelSpec.kind, single-expert
# bypass, combine forced to 'poe')
MODEL_KINDS = {"cVAE_multimodal": ("multimodal", True, False), "mmJSD": ("multimodal", False, True),
               "DMVAE": ("dmvae", False, True), "WeightedDMVAE": ("weighted_dmvae", False, True),
               "mvtCAE": ("mvtcae", False, False), "mmVAEPlus": ("mmvaeplus", False, True)}


@dataclass(frozen=True)
class Cell:
    job_id: int
    fold: int
    proc_id: int
    procedure: str
    replica: int = 0
    resource: str = "HCPimage"

    @property
    def cost(self) -> float:
        # (relative cost only: every base modality counted at the HCPimage width, the early-fusion table as their sum)
        mods, _ = workload.procedure_modalities(self.procedure, self.resource)
        n_base = len(prep.DATASET_MODALITIES.get(self.resource, prep.HCP_MODALITIES))
        dims = [379 * n_base if prep.is_fusion(m) else 379 for m in mods]
        return workload.step_work(dims)["bytes"]


def plan_cells(procedures: Sequence[str], n_folds: int, replicas: int = 1, resource: str = "HCPimage") -> List[Cell]:
    cells, jid = [], 0
    for r in range(replicas):
        for p_id, proc in enumerate(procedures):
            for k in range(n_folds):
                cells.append(Cell(jid, k, p_id, proc, r, resource))
                jid += 1
    return cells


def assign(cells: Sequence[Cell], rank: int, world: int) -> List[Cell]:
    """Static round-robin by descending cost; ties keep job_id order."""
    order = sorted(cells, key=lambda c: (-c.cost, c.job_id))
    return order[rank::world]


def gather_metrics(local: torch.Tensor, max_rows: int, device=None) -> torch.Tensor:
    """all_gather of the per-rank metric table [max_rows, N_METRICS] (unused rows = NaN)."""
    import torch.distributed as dist
    pad = torch.full((max_rows, N_METRICS), float("nan"), dtype=torch.float32, device=device or local.device)
    pad[: local.shape[0]] = local.to(pad.device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        one = pad[~torch.isnan(pad[:, 0])].cpu()
        return one[torch.argsort(one[:, 0])]
    bufs = [torch.empty_like(pad) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, pad)
    allm = torch.cat(bufs).cpu()
    allm = allm[~torch.isnan(allm[:, 0])]
    return allm[torch.argsort(allm[:, 0])]


def run_cells(cohort: prep.SyntheticCohort, cells: Sequence[Cell], n_folds: int, epochs: int, device, out_dir=None,
              lr: float = 1e-4, steps_per_launch: int = 64, oversample_percentage: Optional[float] = None,
              hidden: Sequence[int] = workload.HIDDEN, latent: int = workload.LATENT,
              per_procedure_dirs: bool = False, model: str = "cVAE_multimodal", models_dir=None) -> torch.Tensor:
    """Train the given cells concurrently, run the ROI-wise deviation pass, return the metric rows.
    `oversample_percentage` switches the training rows to the train script's own recipe (utils.generate_kfold_ids:
    KFold over healthy + other, bootstrap resample with replacement, merged back in table order); None = the plain
    KFold split of the regression script."""
    if not cells:
        return torch.empty(0, N_METRICS)
    # per_procedure_dirs: several procedures in one sweep that share modalities write the same (fold, modality) file
    # names -- the CSVs then go to <out_dir>/<procedure>/ (the sweep entry point below does that)
    folds = prep.kfold_indices(len(cohort.iid), n_folds, 42)
    if oversample_percentage is not None:
        hc = cohort.dia == 1
        ids = prep.generate_kfold_ids(cohort.iid[hc], cohort.iid[~hc], oversample_percentage, n_folds)
        folds = [(prep.rows_of_ids(cohort.iid, tr), prep.rows_of_ids(cohort.iid, te)) for tr, te in ids]
    if model not in MODEL_KINDS:
        raise ValueError(f"Model '{model}' is not recognized. Available models are: {', '.join(MODEL_KINDS)}")   # :170-171
    kind, bypass, force_poe = MODEL_KINDS[model]
    jobs: List[Job] = []
    # the fold's tables are built on the device from the raw cohort (scaler fit, covariate bins, early-fusion concat,
    # packing: prep_device.py), once per (fold, modality); the cells of a fold share them
    from .prep_device import DeviceCohort
    dc = DeviceCohort(cohort, device)
    for c in cells:
        mods, combine = workload.procedure_modalities(c.procedure, cohort.resource)
        # (the DMVAE family's networks take no covariates: its tables are packed without the covariate block)
        tables = dc.fold_tables_cached(c.fold, mods, folds[c.fold][0], with_covariates=kind not in ("dmvae", "weighted_dmvae", "mmvaeplus"))
        spec = ModelSpec([t.D for t in tables], list(hidden), int(latent), workload.C_DIM, True, kind)
        jobs.append(Job(spec, tables, combine="poe" if force_poe else combine, lr=lr, seed=1000 * c.fold + c.job_id,
                        init_seed=42 + c.job_id, loss_cap=max(8, epochs * 8), single_bypass=bypass))
    # cells of different shapes take different time per step: group by shape so a launch is balanced
    groups: Dict[tuple, List[int]] = {}
    for i, j in enumerate(jobs):
        groups.setdefault(tuple(j.spec.input_dims), []).append(i)
    t0 = time.perf_counter()
    total_steps = 0
    for idxs in groups.values():
        js = JobSet([jobs[i] for i in idxs])
        n = epochs * jobs[idxs[0]].batches_per_epoch
        done = 0
        while done < n:
            k = min(steps_per_launch, n - done)
            js.train(k)
            done += k
        total_steps += n * len(idxs)
        js.assert_finite()
    torch.cuda.synchronize(device)
    if models_dir is not None:
        # what the train script leaves for the test script (cVAE_model.pkl per fold, ...train...py:211-212), as the
        # state_dict under the reference's key names + the constructor arguments
        # ... plus the fold's own train / test subjects (the reference's train_ids_{fold}.csv / test_ids_{fold}.csv,
        # utils.py:88-93): the `test` subcommand scores exactly the rows this model did not train on, whichever
        # fold recipe (-O / -TrainingClass) produced them
        for c, j in zip(cells, jobs):
            save_model(Path(models_dir) / c.procedure / f"{c.fold:03d}", j, model,
                       train_ids=cohort.iid[folds[c.fold][0]], test_ids=cohort.iid[folds[c.fold][1]])
    sps = total_steps / max(time.perf_counter() - t0, 1e-9)
    # deviation pass: ONE forward-only launch for every (cell, modality) of this rank (unimodal views of the trained
    # models on the all-subject tables, one workgroup per (view, 256-row tile)); the per-subject score (mean over
    # modalities of the ROI-mean deviation) stays on the GPU and feeds the metrics kernel
    # (group_analysis_1x1.py:105-157); the ROI-wise matrices go to the host only when CSVs are asked for
    scores, finals = [], []
    dx = torch.as_tensor(cohort.dia == 0)
    views = [(i, m, name) for i, c in enumerate(cells) for m, name in enumerate(workload.procedure_modalities(c.procedure, cohort.resource)[0])]
    devs = deviation_roiwise_many([(jobs[i], m, name) for i, m, name in views], cohort, device, want_matrix=out_dir is not None)
    per_cell: Dict[int, list] = {}
    for (i, m, name), (dev, rowdev) in zip(views, devs):
        per_cell.setdefault(i, []).append(rowdev)
        if out_dir is not None:
            c = cells[i]
            io.write_roiwise_csv(Path(out_dir) / c.procedure if per_procedure_dirs else out_dir, c.fold,
                                 name if c.replica == 0 else f"{name}_r{c.replica}", cohort.iid, dev)
    for i, (c, j) in enumerate(zip(cells, jobs)):
        last = (j.step - 1) % j.loss_cap
        finals.append(float(j.loss_log[last, 0]))
        scores.append(torch.stack(per_cell[i]).mean(dim=0))
    pm = metrics.posthoc_metrics(scores, [dx] * len(cells), device=device).cpu()
    rows = []
    for i, c in enumerate(cells):
        sc = scores[i].cpu()
        rows.append([c.job_id, c.fold, c.proc_id, finals[i], sps, float(pm[i, 0]), float(pm[i, 1]), float(pm[i, 2]),
                     float(pm[i, 3]), float(pm[i, 4]), float(sc[~dx].mean()), float(sc[dx].mean())])
    return torch.tensor(rows, dtype=torch.float32)


def deviation_roiwise_many(views, cohort: prep.SyntheticCohort, device, want_matrix: bool = True,
                           covariates: Optional[np.ndarray] = None):
    """deviation_roiwise for a list of (trained job, modality index, modality name) in ONE launch: the all-subject
    table of a modality (scaler re-fit on all subjects, ..._regression.py:177-186) is built once and shared by every
    view of that modality; each view is the unimodal model encoder m / decoder m of its job.  Returns
    [(ROI-wise matrix on the host or None, per-subject ROI-mean deviation on the device)] in the order of `views`."""
    if not views:
        return []
    from .layout import ParamLayout
    c = prep.one_hot_covariates(cohort.age, cohort.gender) if covariates is None else covariates
    tables: Dict[tuple, Table] = {}
    ones = []
    for job, m, name in views:
        with_c = job.spec.net_c_dim > 0
        if (name, with_c) not in tables:
            src = prep.source_table(cohort, name)
            center, scale = prep.robust_scaler_fit(src.astype(np.float32))
            xs_ = prep.robust_scaler_transform(src.astype(np.float32), center, scale).astype(np.float32)
            tables[(name, with_c)] = Table(xs_, c if with_c else np.zeros((len(xs_), 0), dtype=np.float32), device)
        table = tables[(name, with_c)]
        spec1 = ModelSpec([job.spec.input_dims[m]], list(job.spec.hidden), job.spec.latent, job.spec.c_dim, job.spec.non_linear,
                          job.spec.kind if job.spec.kind in ("mvtcae", "dmvae", "weighted_dmvae", "mmvaeplus") else "multimodal")
        sd = job.state_dict()
        st = {k: (sd[k][m:m + 1] if k == "weights" else sd[k.replace("_list.0.", f"_list.{m}.")]) for k in ParamLayout(spec1).names}
        one = Job(spec1, [table], combine="poe", state=st, seed=job.seed + 7919 * (m + 1), n_tiles_ws=table.n_tiles,
                  single_bypass=job.single_bypass)
        one.enable_exports(loc=False, sqerr=want_matrix, rowdev=True, latent=False)
        ones.append(one)
    # one launch per distinct table height (all-subject tables of one cohort: a single launch)
    by_tiles: Dict[int, List[int]] = {}
    for i, o in enumerate(ones):
        by_tiles.setdefault(o.tables[0].n_tiles, []).append(i)
    for idxs in by_tiles.values():
        JobSet([ones[i] for i in idxs]).forward(loss=False)      # (one-expert views: the compact deviation-pass kernel)
    torch.cuda.synchronize(device)
    out = []
    for one in ones:
        N = one.tables[0].N
        out.append((one.out_sqerr[0][:N].cpu().numpy() if want_matrix else None, one.out_rowdev[0][:N].clone()))
    return out


def deviation_roiwise(job: Job, m: int, cohort: prep.SyntheticCohort, name: str, device, want_matrix: bool = True,
                      covariates: Optional[np.ndarray] = None):
    """ROI-wise deviation of ALL subjects through modality m's own encoder/decoder with a sampled z
    and a scaler re-fit on all subjects -- exactly the pass of
    multimodal_kfold_train_cvae_supervised_regression.py:163-192.  Returns (ROI-wise matrix on the host or
    None, per-subject ROI-mean deviation on the device, IIDs)."""
    src = prep.source_table(cohort, name)
    center, scale = prep.robust_scaler_fit(src.astype(np.float32))
    x = prep.robust_scaler_transform(src.astype(np.float32), center, scale).astype(np.float32)
    c = prep.one_hot_covariates(cohort.age, cohort.gender) if covariates is None else covariates
    spec1 = ModelSpec([job.spec.input_dims[m]], list(job.spec.hidden), job.spec.latent, job.spec.c_dim)
    sd = job.state_dict()
    from .layout import ParamLayout
    st = {k: sd[k.replace("_list.0.", f"_list.{m}.")] for k in ParamLayout(spec1).names}
    table = Table(x, c, device)
    one = Job(spec1, [table], combine="poe", state=st, seed=job.seed + 7919 * (m + 1), n_tiles_ws=table.n_tiles)
    one.enable_exports(loc=False, sqerr=want_matrix, rowdev=True, latent=False)
    JobSet([one]).forward(loss=False)
    torch.cuda.synchronize(device)
    dev = one.out_sqerr[0][: table.N].cpu().numpy() if want_matrix else None
    return dev, one.out_rowdev[0][: table.N].clone(), cohort.iid


def evaluate_regression(y_true: np.ndarray, y_pred: np.ndarray) -> Dict[str, float]:
    """RMSE / MAE / R2 / MAPE exactly as evaluate_regression of
    multimodal_kfold_train_cvae_supervised_regression.py:30-35 forms them."""
    y_true = np.asarray(y_true, dtype=np.float64).reshape(-1)
    y_pred = np.asarray(y_pred, dtype=np.float64).reshape(-1)
    err = y_true - y_pred
    ss_res, ss_tot = float((err ** 2).sum()), float(((y_true - y_true.mean()) ** 2).sum())
    return {"RMSE": float(np.sqrt((err ** 2).mean())), "MAE": float(np.abs(err).mean()),
            "R2": 1.0 - ss_res / ss_tot if ss_tot > 0 else float("nan"),
            "MAPE": float(np.mean(np.abs(err / (y_true + 1e-6))) * 100)}


def run_regression_folds(cohort: prep.SyntheticCohort, folds_to_run: Sequence[int], n_folds: int, epochs: int, device,
                         out_dir=None, modalities: Sequence[str] = prep.HCP_MODALITIES, combine: str = "gpoe",
                         lr: float = 1e-4, lambda_reg: float = 1.0, hidden: Sequence[int] = workload.HIDDEN,
                         latent: int = workload.LATENT):
    """The whole of multimodal_kfold_train_cvae_supervised_regression.py:52-192 for the given folds, all folds
    training concurrently: cVAE_multimodal_regression on RobustScaler-ed ROI tables with the two raw covariates
    (AGE, PTGENDER) and the FI target; FI prediction on the held-out fold (fold_{k}_pred.npy / _true.npy and
    RMSE / MAE / R2 / MAPE); ROI-wise deviation of every subject per modality
    (deviation_fold_{k}_{name}_roiwise.csv).  Deviation from the script: batches are taken in table order and
    aligned across modalities (its three independently shuffled DataLoaders pair different subjects, :94)."""
    folds = prep.kfold_indices(len(cohort.iid), n_folds, 42)
    cov_all = np.stack([cohort.age, cohort.gender], axis=1).astype(np.float32)
    jobs, scalers = [], []
    for k in folds_to_run:
        tr = folds[k][0]
        xs, sc = [], []
        for m in modalities:
            center, scale = prep.robust_scaler_fit(prep.source_table(cohort, m)[tr])
            xs.append(prep.robust_scaler_transform(prep.source_table(cohort, m)[tr], center, scale).astype(np.float32))
            sc.append((center, scale))
        scalers.append(sc)
        tables = [Table(x, cov_all[tr], device) for x in xs]
        spec = ModelSpec([t.D for t in tables], list(hidden), int(latent), 2, True, "regression")
        j = Job(spec, tables, combine=combine, lr=lr, seed=1000 * k, init_seed=42 + k, loss_cap=8)
        j.reg_lambda = float(lambda_reg)
        j.set_fi(cohort.fi[tr].astype(np.float32))
        jobs.append(j)
    js = JobSet(jobs)
    n = epochs * jobs[0].batches_per_epoch
    t0 = time.perf_counter()
    js.train_regression(n)
    torch.cuda.synchronize(device)
    js.assert_finite()
    sps = n * len(jobs) / max(time.perf_counter() - t0, 1e-9)
    results = []
    for k, j, sc in zip(folds_to_run, jobs, scalers):
        te = folds[k][1]
        # held-out FI prediction (:127-149): joint posterior, sampled z
        xs = [prep.robust_scaler_transform(prep.source_table(cohort, m)[te], *sc[i]).astype(np.float32) for i, m in enumerate(modalities)]
        tables = [Table(x, cov_all[te], device) for x in xs]
        ev = Job(j.spec, tables, combine=combine, state=j.state_dict(), seed=j.seed + 17, n_tiles_ws=tables[0].n_tiles)
        ev.enable_exports(loc=True, sqerr=False, rowdev=False, latent=False)
        es = JobSet([ev])
        es.forward()
        es.head_regression(backward=False, tile0=0, n_tiles=tables[0].n_tiles)
        torch.cuda.synchronize(device)
        pred = ev.out_fi_pred[: len(te)].cpu().numpy().reshape(-1, 1)
        true = cohort.fi[te].astype(np.float32).reshape(-1, 1)
        scores = evaluate_regression(true, pred)
        if out_dir is not None:
            out = Path(out_dir)
            out.mkdir(parents=True, exist_ok=True)
            np.save(out / f"fold_{k}_pred.npy", pred)
            np.save(out / f"fold_{k}_true.npy", true)
        # ROI-wise deviation of every subject, one modality at a time (:163-192)
        for i, m in enumerate(modalities):
            dev, _, iids = deviation_roiwise(j, i, cohort, m, device, want_matrix=out_dir is not None, covariates=cov_all)
            if out_dir is not None:
                io.write_roiwise_csv(out_dir, k, m, iids, dev)
        last = (j.step - 1) % j.loss_cap
        results.append({"fold": k, "steps_per_s": sps, "final_total": float(j.loss_log[last, 0]),
                        "final_mse": float(j.loss_log[last, 12]), **scores})
    return results


def run_endtoend_folds(cohort: prep.SyntheticCohort, folds_to_run: Sequence[int], n_folds: int, epochs: int, device,
                       modalities: Sequence[str] = prep.HCP_MODALITIES, latent: int = 64,
                       classifier_layers: Sequence[int] = (128, 64, 32), dropout_rate: float = 0.5, margin: float = 1.0,
                       weightcontrastive: float = 0.1, lr: float = 1e-4, hc_label: int = 1,
                       hidden: Sequence[int] = workload.HIDDEN):
    """multimodal_kfold_cvae_nmpmcont.py:180-320 for the given folds, all folds training concurrently:
    cVAE_multimodal_endtoend (shared encoders, PoE, health / disease decoder banks, classifier on z) on
    RobustScaler-ed tables with the 29 one-hot covariates and labels healthy = 0 / disease = 1 (:118), batches in
    table order (shuffle=False, :213); then evaluate() (:29-70) on the held-out fold -- eval mode, classifier on
    the joint mean, argmax -- with the confusion metrics computed on the device.  The cyclic learning-rate
    arithmetic of :266-270 is inert in the reference (it assigns an attribute the optimizer never reads), so
    the optimizer runs at its constructor lr here as there."""
    folds = prep.kfold_indices(len(cohort.iid), n_folds, 42)
    labels_all = (cohort.dia != hc_label).astype(np.int32)
    jobs, scalers = [], []
    for k in folds_to_run:
        tr = folds[k][0]
        xs, sc = [], []
        for m in modalities:
            center, scale = prep.robust_scaler_fit(prep.source_table(cohort, m)[tr])
            xs.append(prep.robust_scaler_transform(prep.source_table(cohort, m)[tr], center, scale).astype(np.float32))
            sc.append((center, scale))
        scalers.append(sc)
        cov = prep.one_hot_covariates(cohort.age[tr], cohort.gender[tr])
        tables = [Table(x, cov, device) for x in xs]
        # (hidden widths beyond the fused kernel's tile: the trunk runs on the general-shape path, three launches per step)
        spec = ModelSpec([t.D for t in tables], list(hidden), latent, workload.C_DIM, True, "endtoend",
                         tuple(classifier_layers), 2)
        j = Job(spec, tables, combine="poe", lr=lr, kl_weight=0.1, ll_weight=0.1, seed=1000 * k, init_seed=42 + k,
                loss_cap=8, single_bypass=False)
        j.cls_dropout, j.cls_margin, j.cls_w_contrast = float(dropout_rate), float(margin), float(weightcontrastive)
        j.set_labels(labels_all[tr])
        jobs.append(j)
    js = JobSet(jobs)
    n = epochs * jobs[0].batches_per_epoch
    t0 = time.perf_counter()
    js.train_endtoend(n)
    torch.cuda.synchronize(device)
    js.assert_finite()
    sps = n * len(jobs) / max(time.perf_counter() - t0, 1e-9)
    preds, labs, finals = [], [], []
    for k, j, sc in zip(folds_to_run, jobs, scalers):
        te = folds[k][1]
        xs = [prep.robust_scaler_transform(prep.source_table(cohort, m)[te], *sc[i]).astype(np.float32) for i, m in enumerate(modalities)]
        cov = prep.one_hot_covariates(cohort.age[te], cohort.gender[te])           # re-binned on the test rows (:203-209)
        tables = [Table(x, cov, device) for x in xs]
        ev = Job(j.spec, tables, combine="poe", state=j.state_dict(), seed=j.seed + 17, single_bypass=False,
                 n_tiles_ws=tables[0].n_tiles)
        ev.cls_train, ev.cls_use_mu = False, True                                   # model.eval(); predict(): classifier(mu)
        ev.enable_exports(loc=False, sqerr=False, rowdev=False, latent=True)
        es = JobSet([ev])
        es.forward()
        es.head_classifier(backward=False, tile0=0, n_tiles=tables[0].n_tiles)
        preds.append(torch.argmax(ev.out_logits[: len(te), :2], dim=1).to(torch.int32))
        labs.append(torch.as_tensor(labels_all[te]))
        last = (j.step - 1) % j.loss_cap
        finals.append([float(v) for v in j.loss_log[last, [0, 13, 14]]])
    cm = metrics.confusion_metrics(preds, labs, device=device).cpu()
    out = []
    for i, k in enumerate(folds_to_run):
        row = {"fold": k, "steps_per_s": sps, "final_trunk_total": finals[i][0], "final_ce": finals[i][1],
               "final_contrastive": finals[i][2]}
        row.update({name: float(cm[i, c]) for c, name in enumerate(metrics.CONFUSION_COLUMNS)})
        out.append(row)
    return out


def save_model(fold_dir, job: Job, model: str = "cVAE_multimodal", train_ids=None, test_ids=None) -> Path:
    """<fold_dir>/cVAE_model_state.pt: {'state_dict': reference-keyed tensors, 'input_dim_list', 'hidden_dim', 'latent_dim',
    'c_dim', 'model', 'combine'} -- loadable by the reference class (`load_state_dict`) and by load_model.  With the
    fold's subjects given, also train_ids.csv / test_ids.csv (one IID column, the files of utils.py:88-93)."""
    fold_dir = Path(fold_dir)
    fold_dir.mkdir(parents=True, exist_ok=True)
    if train_ids is not None and test_ids is not None:
        import pandas as pd
        pd.DataFrame({"IID": np.asarray(train_ids)}).to_csv(fold_dir / "train_ids.csv", index=False)
        pd.DataFrame({"IID": np.asarray(test_ids)}).to_csv(fold_dir / "test_ids.csv", index=False)
    sp = job.spec
    path = fold_dir / "cVAE_model_state.pt"
    torch.save({"state_dict": {k: v.cpu() for k, v in job.state_dict().items()}, "input_dim_list": list(sp.input_dims),
                "hidden_dim": list(sp.hidden), "latent_dim": int(sp.latent), "c_dim": int(sp.c_dim), "model": model,
                "combine": job.combine}, path)
    return path


def load_model(fold_dir, tables: Sequence[Table], device, seed: int = 0) -> Job:
    """The Job of a model saved by save_model, on the given tables (their widths must match the saved input_dim_list)."""
    ck = torch.load(Path(fold_dir) / "cVAE_model_state.pt", map_location="cpu", weights_only=True)
    kind, bypass, _ = MODEL_KINDS[ck["model"]]
    if [t.D for t in tables] != list(ck["input_dim_list"]):
        raise ValueError(f"tables of widths {[t.D for t in tables]} for a model trained on {ck['input_dim_list']}")
    spec = ModelSpec(list(ck["input_dim_list"]), list(ck["hidden_dim"]), int(ck["latent_dim"]), int(ck["c_dim"]), True, kind)
    return Job(spec, list(tables), combine=ck["combine"], state=ck["state_dict"], seed=seed, single_bypass=bypass,
               n_tiles_ws=tables[0].n_tiles)


def test_fold(job: Job, cohort: prep.SyntheticCohort, train_rows: np.ndarray, test_rows: np.ndarray, modalities: Sequence[str],
              combine: str, device, out_dir=None, roi_columns: Optional[Dict[str, Sequence[str]]] = None):
    """One fold of multimodal_kfold_test_cvae_supervised.py:64-153 for a trained model: per modality a RobustScaler
    fit on the fold's train rows and applied to the test rows (:86-92), covariates re-binned on the TEST rows
    (:94-99), `pred_recon` (joint latent, sampled z; one workgroup per 256-row tile) and
    `reconstruction_deviation_multimodal`, then the five CSV kinds per modality (:121-153).  Returns
    {modality: per-subject reconstruction error} (what the group analysis averages and scores)."""
    xs = []
    for m in modalities:
        src = prep.source_table(cohort, m)
        center, scale = prep.robust_scaler_fit(src[train_rows])
        xs.append(prep.robust_scaler_transform(src[test_rows], center, scale).astype(np.float32))
    # (the DMVAE family's networks take no covariates: net_c_dim = 0 and a table without the covariate block)
    cov = (prep.one_hot_covariates(cohort.age[test_rows], cohort.gender[test_rows]) if job.spec.net_c_dim > 0
           else np.zeros((len(test_rows), 0), dtype=np.float32))
    tables = [Table(x, cov, device) for x in xs]
    ev = Job(job.spec, tables, combine=combine, state=job.state_dict(), seed=job.seed + 31, n_tiles_ws=tables[0].n_tiles,
             single_bypass=job.single_bypass)
    ev.enable_exports(loc=True, sqerr=False, rowdev=True, latent=False)
    JobSet([ev]).forward()
    torch.cuda.synchronize(device)
    n = len(test_rows)
    errors = {}
    for i, m in enumerate(modalities):
        x_hat = ev.out_loc[i][:n].cpu().numpy()
        errors[m] = ev.out_rowdev[i][:n].cpu().numpy()
        if out_dir is not None:
            import pandas as pd
            meta = pd.DataFrame({"participant_id": cohort.iid[test_rows], "DIA": cohort.dia[test_rows],
                                 "AGE": cohort.age[test_rows], "PTGENDER": cohort.gender[test_rows]})
            cols = list(roi_columns[m]) if roi_columns and m in roi_columns else [f"{m}_{k}" for k in range(xs[i].shape[1])]
            io.write_test_csvs(Path(out_dir) / m, m, meta, cols, xs[i], x_hat)
    return errors


# ---------------------------------------------------------------------------------------------------------------
# Entry point: the reference's train script as ONE sharded sweep,
#     python -m torch.distributed.run --nproc-per-node N -m multi_modal_normative_modeling_amd.sweep -R HCPimage \
#         -P SM-T1w_sMRI SM-T2w_sMRI SM-fMRI UCA-gPoE -E 50 -K 5
# (flag names of multimodal_kfold_train_cvae_supervised.py:216-299; the bash drivers there loop over -P, here the
# procedures of one invocation form the grid).  plan_cells -> assign(rank, world) -> run_cells -> gather_metrics:
# every rank trains its cells inside the persistent kernel and writes their ROI-wise CSVs; the one collective is the
# all_gather of the metric table, which rank 0 writes as sweep_metrics.csv.
# ---------------------------------------------------------------------------------------------------------------
def build_parser():
    import argparse
    ap = argparse.ArgumentParser(prog="python -m multi_modal_normative_modeling_amd.sweep", description=__doc__)
    ap.add_argument("-R", "--dataset_resourse", dest="dataset_resourse", type=str, default="HCPimage",
                    help="dataset name (labels the output directory; the cohort is synthetic: the reference's data/ is not distributed)")
    ap.add_argument("-H", "--hz_para_list", dest="hz_para_list", nargs="+", type=int, default=[110, 110, 10],
                    help="hidden widths followed by the latent width")
    ap.add_argument("-C", "--combine", dest="combine", type=str, default=None, help="overrides the combine part of every -P")
    ap.add_argument("-P", "--procedure", dest="procedure", nargs="+", type=str, default=["UCA-gPoE"],
                    help="one or more of SM-<modality> | SE-<combine> | UCA-<combine>")
    ap.add_argument("-E", "--epochs", dest="epochs", type=int, default=200)
    ap.add_argument("-K", "--n_splits", dest="n_splits", type=int, default=10)
    ap.add_argument("-O", "--oversample_percentage", dest="oversample_percentage", type=float, default=1.0,
                    help="the reference's fold recipe (utils.generate_kfold_ids, utils.py:73-93, run by the train script at :65 whatever "
                         "the value): KFold over healthy + other, then a bootstrap resample of the train split WITH replacement, "
                         "size = oversample_percentage x the split (also at the default 1.0).  --plain-kfold trains on the KFold split "
                         "itself, every row once.  Either way the fold's train / test IIDs are saved with the model (--save-models) "
                         "and the `test` subcommand scores exactly the held-out rows")
    ap.add_argument("--plain-kfold", dest="plain_kfold", action="store_true",
                    help="train on the plain KFold(shuffle, random_state=42) split of the regression script (every train row once) "
                         "instead of the train script's bootstrap-resampled ids")
    ap.add_argument("-Model", "--model", dest="model", type=str, default="cVAE_multimodal")
    ap.add_argument("-SingleModality", "--single_modality", dest="single_modality", type=str, default=None)
    ap.add_argument("-Baselearningrate", "--base_learning_rate", dest="base_learning_rate", type=float, default=1e-4)
    ap.add_argument("-Maxlearningrate", "--max_learning_rate", dest="max_learning_rate", type=float, default=0.005,
                    help="accepted for compatibility: the train script's cyclic schedule never reaches the optimizer (it assigns an "
                         "attribute Adam does not read, :180-186), so training runs at the base rate there and here")
    ap.add_argument("-TrainingClass", "--training_class", dest="training_class", type=str, default="nm")
    ap.add_argument("--data-dir", dest="data_dir", type=str, default=None,
                    help="root of the reference's data layout: <data-dir>/<dataset_resourse>/y.csv and one <modality>.csv per "
                         "modality (IID + ROI columns); default: the synthetic cohort of SURVEY.md 8(d)")
    ap.add_argument("--subjects", type=int, default=1280, help="synthetic cohort size")
    ap.add_argument("--replicas", type=int, default=1, help="independent seeds per (fold, procedure) cell")
    ap.add_argument("--out-dir", type=str, default=None, help="deviation_fold_*_roiwise.csv + sweep_metrics.csv go here")
    ap.add_argument("--no-csv", action="store_true", help="skip the ROI-wise CSVs (metrics only)")
    ap.add_argument("--save-models", action="store_true",
                    help="write <out-dir>/<resource>/<procedure>/<fold:03d>/cVAE_model_state.pt (what the `test` subcommand loads)")
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend when WORLD_SIZE > 1 (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal of the multi-rank path on a one-GPU box: every rank uses cuda:0 (with --backend gloo)")
    return ap


def main(argv=None, _run_cells=None) -> torch.Tensor:
    """The sharded sweep; returns the gathered metric table on rank 0 (an empty tensor elsewhere).  `_run_cells`
    replaces run_cells in the CPU rehearsal tests (the real one needs a GPU)."""
    import os
    args = build_parser().parse_args(argv)
    if args.model not in MODEL_KINDS:                                         # multimodal_kfold_train_cvae_supervised.py:170-171
        raise ValueError(f"Model '{args.model}' is not recognized. Available models are: {', '.join(MODEL_KINDS)}")
    procedures = list(args.procedure)
    if args.single_modality:
        procedures = [f"SM-{args.single_modality}"]
    if args.combine:
        procedures = [p if p.startswith("SM-") else f"{p.split('-')[0]}-{args.combine}" for p in procedures]
    if args.dataset_resourse not in prep.DATASET_MODALITIES:
        raise ValueError("Unknown dataset: {}".format(args.dataset_resourse))   # utils.py:749
    for p in procedures:
        workload.procedure_modalities(p, args.dataset_resourse)               # raises on an unknown procedure
    hidden, latent = list(args.hz_para_list[:-1]), int(args.hz_para_list[-1])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    use_gpu = torch.cuda.is_available() and _run_cells is None
    if args.share_device:
        local_rank = 0
        # (processes sharing ONE GPU: launches whose workgroups wait for each other -- one workgroup per modality, row slices --
        #  could each hold half the CUs and time out on each other; the rehearsal runs every model as one workgroup)
        os.environ["NMHIP_ROWSPLIT"] = "0"
        os.environ["NMHIP_SPLIT"] = "0"
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    started = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and use_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        started = True
    if args.data_dir is not None:
        # the reference's ./data/<resource>/ layout (y.csv + one CSV per modality)
        from . import io as nm_io
        cohort = nm_io.read_cohort(Path(args.data_dir) / args.dataset_resourse, args.dataset_resourse)
    else:
        cohort = prep.synthetic_cohort(n=args.subjects, d=379, modalities=prep.DATASET_MODALITIES[args.dataset_resourse],
                                       resource=args.dataset_resourse)
    for p in procedures:
        missing = [m for m in workload.procedure_modalities(p, args.dataset_resourse)[0] if not prep.is_fusion(m) and m not in cohort.x]
        if missing:
            raise ValueError(f"procedure {p}: the cohort has no table for {missing} (has {cohort.modalities})")
    cells = plan_cells(procedures, args.n_splits, args.replicas, args.dataset_resourse)
    mine = assign(cells, rank, world)
    out_dir = None
    if args.out_dir is not None:
        out_dir = Path(args.out_dir) / args.dataset_resourse
        out_dir.mkdir(parents=True, exist_ok=True)
    runner = _run_cells or run_cells
    # the reference's own recipe by default (generate_kfold_ids at every -O, multimodal_kfold_train_cvae_supervised.py:65)
    oversample = None if args.plain_kfold else args.oversample_percentage
    # run_cells writes a cell's CSVs into <out>/<procedure>/ when procedures share modalities (per_procedure_dirs)
    kw = {} if _run_cells is not None else {"per_procedure_dirs": True, "model": args.model,
                                            "models_dir": out_dir if (args.save_models and out_dir is not None) else None}
    t_rank = time.perf_counter()
    local = runner(cohort, mine, args.n_splits, args.epochs, device, out_dir=None if args.no_csv else out_dir,
                   lr=args.base_learning_rate, oversample_percentage=oversample, hidden=hidden, latent=latent, **kw)
    # the sweep is a STRONG-scaling job (a fixed grid of cells dealt over the ranks): what a rank got and how long it
    # took is what makes an N-GPU run of a small grid readable (20 cells over 8 GPUs = 3/3/3/3/2/2/2/2)
    print(f"[sweep rank {rank}/{world}] cells {len(mine)} of {len(cells)} (cost share "
          f"{sum(c.cost for c in mine) / max(sum(c.cost for c in cells), 1e-9):.3f})  wall {time.perf_counter() - t_rank:.2f} s", flush=True)
    max_rows = (len(cells) + world - 1) // world
    table = gather_metrics(local.to(device) if (world > 1 and dist.get_backend() == "nccl") else local, max_rows)
    if rank == 0:
        if table.shape[0] != len(cells):
            raise RuntimeError(f"metric gather returned {table.shape[0]} rows for {len(cells)} cells")
        if out_dir is not None:
            import pandas as pd
            df = pd.DataFrame(table.numpy(), columns=list(METRIC_COLUMNS))
            df.insert(1, "procedure", [cells[int(j)].procedure for j in table[:, 0].tolist()])
            df.to_csv(out_dir / "sweep_metrics.csv", index=False)
        col = {n: i for i, n in enumerate(METRIC_COLUMNS)}
        for p_id, proc in enumerate(procedures):
            rows = table[table[:, col["proc_id"]] == p_id]
            print(f"[sweep] {proc:28s} cells {rows.shape[0]:3d}  AUC {float(rows[:, col['roc_auc']].mean()):.4f} "
                  f"+- {float(rows[:, col['roc_auc']].std(unbiased=False)):.4f}  final loss {float(rows[:, col['final_total_loss']].mean()):.2f}",
                  flush=True)
    if started:
        dist.destroy_process_group()
    return table if rank == 0 else torch.empty(0, N_METRICS)


def _cohort_from_args(args) -> prep.Cohort:
    if args.dataset_resourse not in prep.DATASET_MODALITIES:
        raise ValueError("Unknown dataset: {}".format(args.dataset_resourse))   # utils.py:749
    if args.data_dir is not None:
        from . import io as nm_io
        return nm_io.read_cohort(Path(args.data_dir) / args.dataset_resourse, args.dataset_resourse)
    return prep.synthetic_cohort(n=args.subjects, d=379, modalities=prep.DATASET_MODALITIES[args.dataset_resourse],
                                 resource=args.dataset_resourse)


def _driver_common(ap: argparse.ArgumentParser):
    ap.add_argument("--data-dir", dest="data_dir", type=str, default=None, help="root of the reference's data layout (see the train entry)")
    ap.add_argument("--subjects", type=int, default=1280, help="synthetic cohort size when no --data-dir is given")
    ap.add_argument("--out-dir", type=str, default=None)
    ap.add_argument("--folds", nargs="+", type=int, default=None, help="folds to run (default: all; under torch.distributed.run: this rank's share)")


def _my_folds(args) -> List[int]:
    import os
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    folds = list(range(args.n_splits)) if args.folds is None else list(args.folds)
    return folds[rank::world]


def main_regression(argv=None, _runner=None):
    """Command line of multimodal_kfold_train_cvae_supervised_regression.py:196-206 (same flag names): FI regression model
    on the folds of a cohort, every fold training concurrently in one launch; under torch.distributed.run the folds are
    dealt round-robin to the ranks (no collective: each rank writes its folds' files).  Returns the per-fold results."""
    import os
    ap = argparse.ArgumentParser(prog="python -m multi_modal_normative_modeling_amd.sweep regression", description=main_regression.__doc__)
    ap.add_argument("-R", "--dataset_resourse", dest="dataset_resourse", type=str, default="HCPimage",
                    help="(the reference defaults to ADNI, whose y.csv carries no FI column; HCPimage is the resource with FI)")
    ap.add_argument("-H", "--hz_para_list", dest="hz_para_list", nargs="+", type=int, default=[110, 110, 10])
    ap.add_argument("-C", "--combine", dest="combine", type=str, default="gpoe")
    ap.add_argument("-P", "--procedure", dest="procedure", type=str, default="UCA-gPoE")
    ap.add_argument("-E", "--epochs", dest="epochs", type=int, default=500)
    ap.add_argument("-K", "--n_splits", dest="n_splits", type=int, default=5)
    ap.add_argument("--batch_size", type=int, default=128,
                    help="accepted for compatibility: a step takes 256 rows (the tile the kernel is built around)")
    ap.add_argument("-BaseLR", "--base_learning_rate", dest="base_learning_rate", type=float, default=1e-4)
    _driver_common(ap)
    args = ap.parse_args(argv)
    if len(args.hz_para_list) < 2:
        raise ValueError("-H takes the hidden widths followed by the latent size (the script's hz_para_list)")
    cohort = _cohort_from_args(args)
    mods = prep.datasets_name(args.dataset_resourse, args.procedure)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    out_dir = None if args.out_dir is None else Path(args.out_dir) / args.dataset_resourse / "regression_outputs"
    runner = _runner or run_regression_folds
    res = runner(cohort, _my_folds(args), args.n_splits, args.epochs, device, out_dir=out_dir, modalities=mods,
                 combine=args.combine.lower(), lr=args.base_learning_rate, hidden=list(args.hz_para_list[:-1]),
                 latent=int(args.hz_para_list[-1]))
    for r in res:
        print("[regression] " + "  ".join(f"{k} {v:.5g}" if isinstance(v, float) else f"{k} {v}" for k, v in r.items()), flush=True)
    return res


def main_endtoend(argv=None, _runner=None):
    """Command line of multimodal_kfold_cvae_nmpmcont.py:344-445 (same flag names): the end-to-end model (shared encoders,
    PoE, health / disease decoder banks, classifier with cross entropy + contrastive hinge) on the folds of a cohort,
    every fold training concurrently in one launch, evaluate() on the held-out fold.  Flags the script parses but never
    uses (-C, -O, -Model, -Maxlearningrate, -Learningrateclassifier, -Weightkl, -Weightrec: its loss call passes margin and
    weightcontrastive only, :298) are accepted and ignored the same way."""
    import os
    ap = argparse.ArgumentParser(prog="python -m multi_modal_normative_modeling_amd.sweep endtoend", description=main_endtoend.__doc__)
    ap.add_argument("-R", "--dataset_resourse", dest="dataset_resourse", type=str, default="HCPimage")
    ap.add_argument("-H", "--hz_para_list", dest="hz_para_list", nargs="+", type=int, default=[110, 110, 64])
    ap.add_argument("-C", "--combine", dest="combine", type=str, default="poe")
    ap.add_argument("-P", "--procedure", dest="procedure", type=str, default="SE-PoE")
    ap.add_argument("-E", "--epochs", dest="epochs", type=int, default=50)
    ap.add_argument("-K", "--n_splits", dest="n_splits", type=int, default=5)
    ap.add_argument("-O", "--oversample_percentage", dest="oversample_percentage", type=float, default=1)
    ap.add_argument("-Model", "--model", dest="model", type=str, default="cVAE_multimodal")
    ap.add_argument("-SingleModality", "--single_modality", dest="single_modality", type=str, default=None)
    ap.add_argument("-Baselearningrate", "--base_learning_rate", dest="base_learning_rate", type=float, default=1e-4)
    ap.add_argument("-Maxlearningrate", "--max_learning_rate", dest="max_learning_rate", type=float, default=0.005)
    ap.add_argument("-Learningrateclassifier", "--learning_rate_classifier", dest="learning_rate_classifier", type=float, default=0.001)
    ap.add_argument("-Margin", "--margin", dest="margin", type=float, default=1)
    ap.add_argument("-Weightcontrastive", "--weightcontrastive", dest="weightcontrastive", type=float, default=1)
    ap.add_argument("-Weightkl", "--weight_kl", dest="weight_kl", type=float, default=1)
    ap.add_argument("-Weightrec", "--weight_rec", dest="weight_rec", type=float, default=1)
    ap.add_argument("-Dropout", "--dropout", dest="dropout", type=float, default=0.5)
    ap.add_argument("-Layers", "--layers", dest="layers", nargs="+", type=int, default=[128, 64, 32])
    _driver_common(ap)
    args = ap.parse_args(argv)
    cohort = _cohort_from_args(args)
    proc = f"SM-{args.single_modality}" if args.single_modality else args.procedure
    mods = [m for m in prep.datasets_name(args.dataset_resourse, proc)]
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    runner = _runner or run_endtoend_folds
    res = runner(cohort, _my_folds(args), args.n_splits, args.epochs, device, modalities=mods, latent=int(args.hz_para_list[-1]),
                 classifier_layers=tuple(args.layers), dropout_rate=args.dropout, margin=args.margin,
                 weightcontrastive=args.weightcontrastive, lr=args.base_learning_rate, hc_label=1,
                 hidden=list(args.hz_para_list[:-1]))
    for r in res:
        print("[endtoend] " + "  ".join(f"{k} {v:.5g}" if isinstance(v, float) else f"{k} {v}" for k, v in r.items()), flush=True)
    if args.out_dir is not None and res:
        import pandas as pd
        out = Path(args.out_dir) / args.dataset_resourse
        out.mkdir(parents=True, exist_ok=True)
        pd.DataFrame(res).to_csv(out / f"endtoend_metrics_rank{int(os.environ.get('RANK', '0'))}.csv", index=False)
    return res


def main_test(argv=None):
    """Command line of multimodal_kfold_test_cvae_supervised.py:180-187 (-R -H -C -P -K): for every fold of a procedure load
    the model the train entry saved (--save-models), scale the fold's test rows with the scaler of its train rows, re-bin
    the covariates on the test rows, reconstruct from the joint latent and write the five CSV kinds per modality under
    <models-dir>/<resource>/<procedure>/<fold:03d>/<modality>/, then the all-folds tables under
    <out-dir>/<resource>/<procedure>/<modality>/ (the script's deviation_dir, :147-175).  Returns {modality: [N] errors}."""
    import os
    import pandas as pd
    ap = argparse.ArgumentParser(prog="python -m multi_modal_normative_modeling_amd.sweep test", description=main_test.__doc__)
    ap.add_argument("-R", "--dataset_resourse", dest="dataset_resourse", type=str, default="HCPimage")
    ap.add_argument("-H", "--hz_para_list", dest="hz_para_list", nargs="+", type=int, default=[110, 110, 10])
    ap.add_argument("-C", "--combine", dest="combine", type=str, default=None)
    ap.add_argument("-P", "--procedure", dest="procedure", type=str, default="SE-gPoE")
    ap.add_argument("-K", "--n_splits", dest="n_splits", type=int, default=10)
    ap.add_argument("--models-dir", type=str, required=True, help="the --out-dir of the train entry (run with --save-models)")
    _driver_common(ap)
    args = ap.parse_args(argv)
    cohort = _cohort_from_args(args)
    mods, combine = workload.procedure_modalities(args.procedure, args.dataset_resourse)
    combine = (args.combine or combine).lower()
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    folds = prep.kfold_indices(len(cohort.iid), args.n_splits, 42)
    root = Path(args.models_dir) / args.dataset_resourse / args.procedure
    out_root = Path(args.out_dir or args.models_dir) / args.dataset_resourse / args.procedure
    from .prep_device import DeviceCohort
    dc = DeviceCohort(cohort, device)
    errors: Dict[str, list] = {m: [] for m in mods}
    my = _my_folds(args)
    for k in my:
        tr, te = folds[k]
        fold_dir = root / f"{k:03d}"
        # (the model is rebuilt on any tables of the right widths; test_fold puts it on the fold's test tables)
        meta = torch.load(fold_dir / "cVAE_model_state.pt", map_location="cpu", weights_only=True)
        no_cov = MODEL_KINDS[meta["model"]][0] in ("dmvae", "weighted_dmvae", "mmvaeplus")
        # the fold's subjects as the train entry recorded them (its -O / -TrainingClass recipe may differ from the
        # plain KFold); models saved without them: the plain KFold split
        if (fold_dir / "train_ids.csv").exists() and (fold_dir / "test_ids.csv").exists():
            tr = prep.rows_of_ids(cohort.iid, pd.read_csv(fold_dir / "train_ids.csv")["IID"].to_numpy())
            te = prep.rows_of_ids(cohort.iid, pd.read_csv(fold_dir / "test_ids.csv")["IID"].to_numpy())
        job = load_model(fold_dir, dc.fold_tables_cached(k, mods, tr, with_covariates=not no_cov), device, seed=1000 * k)
        # models whose class fixes the fusion (mmJSD, the DMVAE family: MODEL_KINDS[...][2]) reconstruct with the saved
        # one, whatever the procedure name says (mmJSD.pred_recon ignores its combine argument, cVAE.py:1405-1420)
        fold_combine = str(meta["combine"]).lower() if MODEL_KINDS[meta["model"]][2] else combine
        err = test_fold(job, cohort, tr, te, mods, fold_combine, device, out_dir=fold_dir)
        for m in mods:
            errors[m].append(err[m])
    for m in mods:                                   # all folds of this rank, one table per CSV kind (:147-175)
        (out_root / m).mkdir(parents=True, exist_ok=True)
        for kind in ("normalized", "reconstruction", "reconstruction_error", "reconstruction_error_roi", "deviation_as_feature_importance"):
            parts = [pd.read_csv(root / f"{k:03d}" / m / f"{kind}_{m}.csv") for k in my]
            if parts:
                pd.concat(parts, ignore_index=True).to_csv(out_root / m / f"{kind}_{m}.csv", index=False)
    out = {m: np.concatenate(v) if v else np.empty(0) for m, v in errors.items()}
    for m, v in out.items():
        print(f"[test] {args.procedure} {m}: {len(v)} subjects, mean reconstruction error {float(v.mean()) if len(v) else float('nan'):.5f}", flush=True)
    return out


def main_analysis(argv=None):
    """multimodal_kfold_cvae_group_analysis_1x1.py:160-235 on the files the `test` subcommand wrote: per fold the subjects'
    reconstruction errors averaged over the procedure's modalities (:205-209), healthy vs disease ROC-AUC, Youden-J
    threshold, accuracy / sensitivity / specificity (compute_classification_performance, :105-157, on the device:
    nm_posthoc_metrics) and the significance ratio auc / (1 - auc) (:231); prints the per-fold rows and mean +- std,
    writes <models-dir>/<resource>/<procedure>/group_analysis.csv.  Returns the [folds, 8] metric table."""
    import pandas as pd
    ap = argparse.ArgumentParser(prog="python -m multi_modal_normative_modeling_amd.sweep analysis", description=main_analysis.__doc__)
    ap.add_argument("-R", "--dataset_resourse", dest="dataset_resourse", type=str, default="HCPimage")
    ap.add_argument("-H", "--hz_para_list", dest="hz_para_list", nargs="+", type=int, default=[110, 110, 10])
    ap.add_argument("-C", "--combine", dest="combine", type=str, default=None)
    ap.add_argument("-P", "--procedure", dest="procedure", type=str, default="SE-gPoE")
    ap.add_argument("-E", "--epochs", dest="epochs", type=int, default=None)
    ap.add_argument("-K", "--n_splits", dest="n_splits", type=int, default=10)
    ap.add_argument("--models-dir", type=str, required=True, help="where the `test` subcommand wrote its per-fold CSVs")
    args = ap.parse_args(argv)
    mods, _ = workload.procedure_modalities(args.procedure, args.dataset_resourse)
    root = Path(args.models_dir) / args.dataset_resourse / args.procedure
    hc = prep.HC_LABEL.get(args.dataset_resourse, 1)
    scores, positive, folds = [], [], []
    for k in range(args.n_splits):
        files = [root / f"{k:03d}" / m / f"reconstruction_error_{m}.csv" for m in mods]
        if not all(f.exists() for f in files):
            continue
        dfs = [pd.read_csv(f) for f in files]
        err = sum(d["Reconstruction error"].to_numpy(dtype=np.float64) for d in dfs) / len(dfs)
        dia = dfs[0]["DIA"].to_numpy()
        # (files written from a prep.Cohort carry DIA in the cohort's convention 1 = healthy; raw tables the resource's label)
        healthy = (dia == 1) if set(np.unique(dia)) <= {0, 1} else (dia == hc)
        scores.append(torch.as_tensor(err, dtype=torch.float32))
        positive.append(torch.as_tensor(~healthy, dtype=torch.int32))
        folds.append(k)
    if not folds:
        raise FileNotFoundError(f"no reconstruction_error_*.csv of {mods} under {root}/<fold>/ -- run the `test` subcommand first")
    table = metrics.posthoc_metrics(scores, positive).cpu()
    df = pd.DataFrame(table.numpy(), columns=list(metrics.POSTHOC_COLUMNS))
    df.insert(0, "fold", folds)
    df.to_csv(root / "group_analysis.csv", index=False)
    for _, r in df.iterrows():
        print(f"[analysis] fold {int(r['fold'])}: AUC {r['roc_auc']:.4f}  accuracy {r['accuracy']:.4f}  sensitivity {r['recall']:.4f}  "
              f"specificity {r['specificity']:.4f}  significance ratio {r['significance_ratio']:.3f}", flush=True)
    print(f"[analysis] {args.procedure}: AUC {df['roc_auc'].mean():.4f} +- {df['roc_auc'].std(ddof=0):.4f}  accuracy {df['accuracy'].mean():.4f}  "
          f"sensitivity {df['recall'].mean():.4f}  specificity {df['specificity'].mean():.4f}", flush=True)
    return table


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "regression":
        main_regression(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "endtoend":
        main_endtoend(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "test":
        main_test(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "analysis":
        main_analysis(sys.argv[2:])
    else:
        main()
