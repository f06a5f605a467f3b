#!/usr/bin/env python3
"""GPU diagnostic (not a test): per-tensor error of the HIP path against the fp32 oracle/golden and
against the oracle in bf16-operand mode.  Usage: python tests/diag_parity.py [case ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multi_modal_normative_modeling_amd as nm
from oracle import cvae_ref as R
from tests.golden_util import Golden
from tests.hip_harness import make_job, oracle_step0, rel_err

cases = sys.argv[1:] or ["mm1_small", "mm3_gpoe", "cfgA_T1w"]
for name in cases:
    g = Golden(name)
    job = make_job(g, 0)
    job.enable_exports()
    nm.JobSet([job]).grads(0)
    torch.cuda.synchronize()
    R.set_operand_rounding("fp32")
    f32 = oracle_step0(g)
    R.set_operand_rounding("bf16")
    b16 = oracle_step0(g)
    R.set_operand_rounding("fp32")
    B = g.B
    print(f"=== {name}: loss hip {job.loss_log[0, :3].cpu().tolist()} ref {g.z['loss0'].tolist()} "
          f"bf16-oracle {[float(b16[1][k]) for k in ('total', 'kl', 'll')]}")
    print(f"   mu: vs fp32 {rel_err(job.out_mu[:B].cpu(), f32[0]['mu'].detach()):.2e}  vs bf16 {rel_err(job.out_mu[:B].cpu(), b16[0]['mu'].detach()):.2e}")
    for m in range(g.M):
        loc = job.out_loc[m][:B].cpu()
        print(f"   loc{m}: vs fp32 {rel_err(loc, f32[0]['locs'][m].detach()):.2e}  vs bf16 {rel_err(loc, b16[0]['locs'][m].detach()):.2e}")
    got = job.grads_dict()
    for k in got:
        a, r32, r16 = got[k].flatten(), f32[2][k].flatten(), b16[2][k].flatten()
        cos = float(torch.nn.functional.cosine_similarity(a, r32, dim=0)) if a.numel() > 1 else float("nan")
        d = (a - r16).abs()
        sc = float(r16.abs().max()) + 1e-30
        print(f"   {k:48s} fp32: max {rel_err(a, r32):.2e} cos {cos:.5f} | bf16: max {float(d.max())/sc:.2e} "
              f"frac>1e-3 {float((d > 1e-3 * sc).float().mean()):.4f}")

# fused multi-step vs stepwise
g = Golden("mm1_small")
xs = torch.cat([g.xs(s)[0] for s in range(g.n_steps)])
x = torch.cat([xs] * 7)[:600]
c = torch.cat([g.t("c")[s] for s in range(g.n_steps)] * 7)[:600]
eps = torch.randn(7, 256, g.Z, generator=torch.Generator().manual_seed(5))
res = []
for mode in ("fused", "stepwise", "fused"):
    spec = nm.ModelSpec(g.dims, g.hidden, g.Z, g.c_dim)
    job = nm.Job(spec, [nm.Table(x, c, "cuda:0")], combine=g.combine, state=g.weights("w0"))
    job.set_eps(eps)
    js = nm.JobSet([job])
    if mode == "fused":
        js.train(7)
    else:
        for _ in range(7):
            js.train(1)
    torch.cuda.synchronize()
    res.append((job.params.cpu().clone(), job.loss_log[:7].cpu().clone()))
print("fused vs stepwise: max param diff", float((res[0][0] - res[1][0]).abs().max()), "loss diff",
      float((res[0][1] - res[1][1]).abs().max()))
print("fused vs fused   : max param diff", float((res[0][0] - res[2][0]).abs().max()), "loss diff",
      float((res[0][1] - res[2][1]).abs().max()))
print("losses fused", res[0][1][:, 0].tolist())
print("losses step ", res[1][1][:, 0].tolist())
