#!/usr/bin/env python3
"""Small sweeps (the metric's literal shape: 5 folds; the reference's grid: 5 folds x 4 procedures = 20 models; one model):
sweep steps/s as one workgroup per model, one per modality (nm_launch_split) and k row slices per modality
(nm_launch_rowsplit).  `--trace`: per-wave interval cycles of workgroup 0 of the row-split kernel."""
import argparse, json, sys, time
from pathlib import Path
import ctypes as C

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import _lib, prep, workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procedure", default="SE-gPoE")
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--sets", default="1,5,20")
    ap.add_argument("--modes", default="wg,split,rs2,rs4")
    ap.add_argument("--trace", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cohort = prep.synthetic_cohort(n=1280, d=379)
    out = {}
    # (clocks: the first mode of a run measured ~5 % slow -- half a second of the same work first)
    wj = nm.JobSet(workload.build_sweep_jobs(cohort, a.procedure, 5, 5, dev, seed0=10_000))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        wj.train(128)
        torch.cuda.synchronize()
    del wj
    for n in [int(x) for x in a.sets.split(",")]:
        for mode in a.modes.split(","):
            sj = nm.JobSet(workload.build_sweep_jobs(cohort, a.procedure, 5, n, dev, seed0=10_000))
            import re
            mh = re.fullmatch(r"rs([24])h(\d+)", mode)
            kw = dict(rowsplit=int(mh.group(1)), helpers=int(mh.group(2))) if mh else \
                 {"wg": dict(split=False, rowsplit=1), "split": dict(split=True, rowsplit=1), "rs2": dict(rowsplit=2), "rs4": dict(rowsplit=4),
                  "rs2wt": dict(rowsplit=2, profile=True), "rs4wt": dict(rowsplit=4, profile=True)}[mode]
            try:
                sj.train(32, **kw)
                sj.check_split_errors(block=True)
            except _lib.NmError as e:
                out[f"{n}:{mode}"] = f"refused ({str(e)[:60]})"
                continue
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                sj.train(a.steps, **kw)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            sj.assert_finite()
            out[f"{n}:{mode}"] = {"steps_per_s": round(n * a.steps / best, 1), "us_per_sweep_step": round(best / a.steps * 1e6, 2)}
            out[f"{n}:{mode}"]["group_on_one_xcd"] = bool(float(sj.jobs[0].loss_log[0, 15]) == 0.0) if mode.startswith("rs") else None
            print(f"{n:3d} models  {mode:6s} {out[f'{n}:{mode}']}", flush=True)
            if a.trace and mode.startswith("rs"):
                lib = _lib.load()
                buf = (C.c_ulonglong * 512)()
                lib.nm_trace_read_rs(buf, 1)
                flags = _lib.NM_F_BACKWARD | _lib.NM_F_ADAM | _lib.NM_F_TRACE
                sj._launch_rowsplit(int(mode[2]), sj.jobs[0].step, 16, flags | (_lib.NM_F_PROFILE if mode.endswith("wt") else 0), kw.get("helpers"))
                torch.cuda.synchronize()
                lib.nm_trace_read_rs(buf, 1)
                tags = {0: "enc first layer", 1: "enc hidden", 2: "enc heads", 3: "handoff A + latent", 4: "z|c", 5: "dec hidden",
                        6: "out: wait+GEMM", 7: "out: epilogue", 8: "out: dlogvar+dgrad", 9: "out: wgrad partial", 10: "dec bwd dgrad",
                        11: "dec bwd wgrad", 12: "handoff B + fusion bwd", 13: "enc bwd prep+heads dgrad", 14: "enc bwd heads/hidden",
                        15: "enc bwd first layer", 16: "enc first layer: chunk 0 landed", 17: "step prologue", 18: "enc first layer: requests issued", 40: "(tail of run_step)", 41: "handoff C", 44: "sweep: table", 45: "sweep: tiles", 42: "sweep: vectors", 43: "handoff D"}
                tot = [sum(buf[w * 64 + t] for t in range(64)) / 16 for w in range(8)]
                print(f"   trace ({mode}, {n} models): cycles per step, wave 0 / mean of waves; total {tot[0]:.0f}")
                for t in range(64):
                    v = [buf[w * 64 + t] / 16 for w in range(8)]
                    if max(v) > 0:
                        print(f"     [{t:2d}] {tags.get(t, ''):28s} {v[0]:10.0f} {sum(v) / 8:10.0f}")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
