#!/usr/bin/env python3
"""GPU diagnostic: the random-shape comparison of tests/test_gpu_fuzz.py over many more seeds (not part of the suite)."""
import argparse, sys, traceback
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests.test_gpu_fuzz import _draw
from tests.test_gpu_fullsize import run_case

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=1000)
ap.add_argument("--count", type=int, default=150)
a = ap.parse_args()
bad = []
for seed in range(a.first, a.first + a.count):
    dims, Z, combine, B, hidden, c_dim, non_linear = _draw(seed)
    try:
        run_case(dims, Z, combine, B, seed=seed, hidden=tuple(hidden), c_dim=c_dim, non_linear=non_linear,
                 ll32_tol=1e-4 * max(1.0, (256.0 / B) ** 0.5))
    except Exception as e:                      # noqa: BLE001 -- diagnostic: collect and report every failing shape
        bad.append((seed, dims, Z, combine, B, hidden, c_dim, non_linear, repr(e)[:200]))
print(f"{a.count - len(bad)} / {a.count} shapes ok")
for b in bad:
    print("FAILED", b)
sys.exit(1 if bad else 0)
