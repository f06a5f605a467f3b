"""Host logic of the sweep (CPU): cell planning / sharding, the rank-based AUC, the CSV layouts,
and the ONE collective of the path -- the final metric all_gather -- with gloo, world_size 2."""
import os
import tempfile

import numpy as np
import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multi_modal_normative_modeling_amd import io, prep, sweep
from tests.golden_util import GOLDEN


def test_cells_cover_the_grid_once_and_balance():
    procs = ["SM-T1w_sMRI", "SM-T2w_sMRI", "SM-fMRI", "SM-" + prep.EARLY_FUSION]
    cells = sweep.plan_cells(procs, 5)
    assert len(cells) == 20
    for world in (1, 2, 4, 8):
        shards = [sweep.assign(cells, r, world) for r in range(world)]
        ids = sorted(c.job_id for s in shards for c in s)
        assert ids == list(range(20))                                   # every cell exactly once
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1
    # costliest cells (early fusion, 1137 ROI) are dealt first
    assert sweep.assign(cells, 0, 8)[0].procedure.endswith(prep.EARLY_FUSION)


def test_roiwise_csv_layout_matches_reference_file():
    z = np.load(GOLDEN / "csv_layouts.npz", allow_pickle=False)
    header = [str(h) for h in z["roiwise_header"]]
    iids = z["roiwise_iid"]
    vals = z["roiwise_row0_vals"]
    dev = np.tile(vals, (len(iids), 1)).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        p = io.write_roiwise_csv(d, 0, "T1w_sMRI", iids, dev)
        assert p.name == "deviation_fold_0_T1w_sMRI_roiwise.csv"       # ..._regression.py:192
        with open(p) as f:
            assert f.readline().strip().split(",") == header           # IID, ROI_0 .. ROI_{D-1}
            assert f.readline().strip() == str(z["roiwise_row0_text"])  # same float32 text as the reference's file
        back = pd.read_csv(p)
        assert (back["IID"].to_numpy() == iids).all()                   # row order = table order, bit-exact ids


def test_test_script_csv_kinds():
    cov = pd.DataFrame({"participant_id": [1, 2], "DIA": [1, 0], "AGE": [30, 31], "PTGENDER": [0, 1]})
    x = np.array([[1.0, 2.0, 3.0], [0.0, 1.0, 0.5]])
    xh = np.array([[0.5, 2.0, 2.0], [0.0, 0.0, 0.5]])
    with tempfile.TemporaryDirectory() as d:
        paths = io.write_test_csvs(d, "av45", cov, ["a", "b", "c"], x, xh)
        err = pd.read_csv(paths["reconstruction_error"])
        assert list(err.columns) == io.META_COLS + ["Reconstruction error"]
        np.testing.assert_allclose(err["Reconstruction error"], ((x - xh) ** 2).sum(1) / 3)
        fi = pd.read_csv(paths["deviation_as_feature_importance"])
        assert list(fi.columns) == io.META_COLS + ["1", "2", "3"]
        roi = pd.read_csv(paths["reconstruction_error_roi"])
        np.testing.assert_allclose(roi[["a", "b", "c"]].to_numpy(), (x - xh) ** 2)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cells = sweep.plan_cells(["SM-T1w_sMRI", "SM-fMRI", "SE-gPoE"], 3)
    mine = sweep.assign(cells, rank, world)
    # stand-in for the GPU work: each rank fills the metric rows of ITS cells
    local = torch.tensor([[c.job_id, c.fold, c.proc_id, 100.0 + c.job_id, 10.0 * (rank + 1)] + [0.5] * (sweep.N_METRICS - 5)
                          for c in mine], dtype=torch.float32)
    allm = sweep.gather_metrics(local, max_rows=math_ceil(len(cells), world))
    if rank == 0:
        torch.save(allm, out)
    dist.destroy_process_group()


def math_ceil(a, b):
    return (a + b - 1) // b


def test_metric_gather_gloo_world2():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "m.pt")
        port = 29500 + (os.getpid() % 500)
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        allm = torch.load(out, weights_only=True)
    assert allm.shape == (9, sweep.N_METRICS)
    assert allm[:, 0].tolist() == list(range(9))                        # every cell once, ordered by job id
    assert torch.allclose(allm[:, 3], 100.0 + allm[:, 0])
    assert set(allm[:, 4].tolist()) == {10.0, 20.0}                     # rows really came from both ranks


def _stub_run_cells(cohort, cells, n_folds, epochs, device, out_dir=None, lr=1e-4, oversample_percentage=None, hidden=None,
                    latent=None):
    """Stand-in for the GPU work of sweep.run_cells: one metric row per cell, recognisable values."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    rows = [[c.job_id, c.fold, c.proc_id, 100.0 + c.job_id, 10.0 * (rank + 1), 0.5 + 0.01 * c.fold] + [0.25] * (sweep.N_METRICS - 6)
            for c in cells]
    assert list(hidden) == [64, 32] and latent == 7 and epochs == 3 and abs(lr - 2e-4) < 1e-12
    return torch.tensor(rows, dtype=torch.float32).reshape(-1, sweep.N_METRICS)


def _cli_worker(rank, world, port, out_dir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank)})
    table = sweep.main(["-R", "HCPimage", "-P", "SM-T1w_sMRI", "SM-fMRI", "UCA-gPoE", "-E", "3", "-K", "5", "-H", "64", "32", "7",
                        "-Baselearningrate", "2e-4", "--subjects", "64", "--out-dir", out_dir, "--backend", "gloo"],
                       _run_cells=_stub_run_cells)
    assert (table.shape[0] == 15) == (rank == 0)


def test_sweep_cli_entry_gloo_world2():
    """The sharded entry point (reference flag names) under two ranks: plan -> assign -> run (stubbed) -> all_gather ->
    sweep_metrics.csv on rank 0 with every cell exactly once."""
    import pandas as pd
    with tempfile.TemporaryDirectory() as d:
        port = 29600 + (os.getpid() % 300)
        mp.spawn(_cli_worker, args=(2, port, d), nprocs=2, join=True)
        df = pd.read_csv(os.path.join(d, "HCPimage", "sweep_metrics.csv"))
    assert list(df["job_id"]) == list(range(15))
    assert set(df["procedure"]) == {"SM-T1w_sMRI", "SM-fMRI", "UCA-gPoE"}
    assert set(df["steps_per_s"]) == {10.0, 20.0}                     # rows came from both ranks
    # descending-cost round-robin: the five UCA cells (the expensive ones) are split 3 / 2 over the ranks
    uca = df[df["procedure"] == "UCA-gPoE"]
    assert sorted(uca["steps_per_s"].value_counts().tolist()) == [2, 3]


def test_sweep_cli_rejects_unknown_model_and_procedure():
    with pytest.raises(ValueError):
        sweep.main(["-Model", "nope", "--subjects", "32"], _run_cells=_stub_run_cells)
    with pytest.raises(ValueError):
        sweep.main(["-P", "XX-gPoE", "--subjects", "32"], _run_cells=_stub_run_cells)


def test_evaluate_regression_matches_sklearn():
    """sweep.evaluate_regression vs the sklearn calls of ..._regression.py:30-35."""
    from sklearn.metrics import mean_squared_error, mean_absolute_error, r2_score
    rng = np.random.default_rng(1)
    t = rng.normal(100, 15, size=(64, 1)).astype(np.float32)
    p = (t + rng.normal(0, 5, size=(64, 1))).astype(np.float32)
    got = sweep.evaluate_regression(t, p)
    assert abs(got["RMSE"] - np.sqrt(mean_squared_error(t, p))) < 1e-5
    assert abs(got["MAE"] - mean_absolute_error(t, p)) < 1e-5
    assert abs(got["R2"] - r2_score(t, p)) < 1e-6
    assert abs(got["MAPE"] - np.mean(np.abs((t - p) / (t + 1e-6))) * 100) < 1e-4
