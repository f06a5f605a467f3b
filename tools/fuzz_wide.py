#!/usr/bin/env python3
"""GPU diagnostic: further random general-shape models against the oracle (tests/test_gpu_fuzz.py's comparison, other seeds):
tools/fuzz_wide.py --first 10 --count 60"""
import argparse, sys, traceback
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests.test_gpu_fuzz import test_random_wide_shape_matches_oracle, _draw_wide
ap = argparse.ArgumentParser(); ap.add_argument("--first", type=int, default=10); ap.add_argument("--count", type=int, default=60)
a = ap.parse_args()
bad = 0
for seed in range(a.first, a.first + a.count):
    try:
        test_random_wide_shape_matches_oracle(seed)
    except Exception:
        bad += 1
        print("FAILED seed", seed, _draw_wide(seed)); traceback.print_exc()
    if (seed - a.first) % 10 == 9:
        print(f"seed {seed}: {bad} failures so far", flush=True)
print(f"{a.count} general-shape models, {bad} failures")
sys.exit(1 if bad else 0)
