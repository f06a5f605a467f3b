"""Helpers to read the golden fixtures written by oracle/gen_golden.py (data only)."""
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


class Golden:
    def __init__(self, name):
        self.z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
        meta = self.z["meta"]
        self.M, self.c_dim, self.Z, self.B, self.n_steps = (int(v) for v in meta)
        self.dims = [int(d) for d in self.z["dims"]]
        self.hidden = [int(h) for h in self.z["hidden"]]
        self.combine = str(self.z["combine"]) if "combine" in self.z.files else "poe"

    def weights(self, tag="w0"):
        pre = tag + ":"
        return {k[len(pre):]: torch.from_numpy(self.z[k].copy()) for k in self.z.files if k.startswith(pre)}

    def grads(self, tag="g0"):
        return self.weights(tag)

    def adam(self, tag):
        m = {k[len(tag) + 3:]: torch.from_numpy(self.z[k].copy()) for k in self.z.files if k.startswith(tag + ":m:")}
        v = {k[len(tag) + 3:]: torch.from_numpy(self.z[k].copy()) for k in self.z.files if k.startswith(tag + ":v:")}
        return m, v

    def t(self, key):
        return torch.from_numpy(np.asarray(self.z[key]).copy())

    def xs(self, step=None):
        out = []
        for m in range(self.M):
            x = self.t(f"x{m}")
            out.append(x if step is None else x[step])
        return out
