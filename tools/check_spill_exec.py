#!/usr/bin/env python3
"""Build-time guard against a code-generation hazard seen with this toolchain (ROCm 7.2 clang, gfx950): when a
lane-divergent branch sits in a region with register spills, the spill stores of values that are live in ALL
lanes can be placed at the head of the branch's merge block, before `s_or_b64 exec, exec, ...` restores the
execution mask -- lanes outside the branch then never store, and the later reload returns stale scratch.  (It
showed up as an NLL off by exactly 0.5*|logvar_out| per row for one ROI column of one test shape.)

The script compiles csrc/nmhip.hip and csrc/nm_rowsplit.hip to ISA and reports every basic block that touches scratch before its first
EXEC restore.  Exit status 1 if any is found.  Fix at the source: keep divergent branches out of high-pressure
regions (hoist descriptor loads out of per-lane selects, use selects / predicated stores instead of `if`)."""
import re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def scan(asm_text: str):
    bad, cur_label, cur = [], None, []

    def check(label, insts):
        for k, (ln, t) in enumerate(insts):
            if re.match(r"\s*s_or_b64 exec, exec,", t):
                sc = [(l, x.strip()) for l, x in insts[:k] if "scratch_" in x]
                if sc:
                    bad.append((label, ln, sc))
                return
            if re.match(r"\s*(s_and_saveexec|s_cbranch|s_branch|s_barrier)", t):
                return

    for i, l in enumerate(asm_text.split("\n")):
        if re.match(r"^\.LBB\d+_\d+:", l):
            if cur_label is not None:
                check(cur_label, cur)
            cur_label, cur = l.split(":")[0], []
        elif re.match(r"^[A-Za-z_][\w.$]*:", l):           # function label: new scope
            if cur_label is not None:
                check(cur_label, cur)
            cur_label, cur = l.split(":")[0], []
        elif l.strip() and not l.strip().startswith(";"):
            cur.append((i + 1, l))
    if cur_label is not None:
        check(cur_label, cur)
    return bad


def resources(asm_text: str):
    """{kernel name: {vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size}} from the
    code-object metadata at the end of the ISA listing."""
    out, cur = {}, None
    for l in asm_text.split("\n"):
        m = re.match(r"\s*\.name:\s+(\S+)", l)
        if m:
            cur = m.group(1)
        m2 = re.match(r"\s*\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", l)
        if m2 and cur and "kernel" in cur:
            out.setdefault(cur, {})[m2.group(1)] = int(m2.group(2))
    return out


# Ceilings per kernel family (substring of the mangled name).  The persistent kernels must not spill vector registers;
# no kernel may touch scratch at all (the classifier head once kept a pointer table in private memory: 88 bytes).  SGPR spills (v_writelane / v_readlane into a reserved
# VGPR, no memory traffic) are bounded at what the current source needs plus a margin, so that a change that makes the
# scalar pressure worse is seen at build time (round 2 built 458 / 512 without anyone looking).
LIMITS = {
    "nm_step_kernel": {"vgpr_spill_count": 0, "private_segment_fixed_size": 0, "sgpr_spill_count": 540},
    "nm_devpass_kernel": {"vgpr_spill_count": 0, "private_segment_fixed_size": 0, "sgpr_spill_count": 64, "vgpr_count": 128},
    "nm_rs_kernel": {"vgpr_spill_count": 0, "private_segment_fixed_size": 0, "sgpr_spill_count": 540},
    "nm_wide_step_kernel": {"vgpr_spill_count": 0, "private_segment_fixed_size": 0, "sgpr_spill_count": 400},
    "nm_head_step_kernel": {"vgpr_spill_count": 0, "private_segment_fixed_size": 0, "sgpr_spill_count": 580},
    "nm_reghead_kernel": {"vgpr_spill_count": 0, "private_segment_fixed_size": 0, "sgpr_spill_count": 16},
    "nm_clshead_kernel": {"vgpr_spill_count": 0, "private_segment_fixed_size": 0, "sgpr_spill_count": 140},   # (+ the tiled head)
}


def compile_isa() -> str:
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for name in ("nmhip", "nm_rowsplit", "nm_devpass"):           # (all at once: ~80 s each)
            src = ROOT / "multi_modal_normative_modeling_amd" / "csrc" / f"{name}.hip"
            out = Path(d) / f"{name}.s"
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                   f"-I{ROOT / 'include'}", str(src), "-o", str(out)]
            procs.append((out, cmd, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
        for out, cmd, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        return "\n".join(out.read_text() for out, _, _ in procs)


def main(listings=()):
    # ISA listings handed in (build() compiles them beside the objects), else the translation units are compiled here
    text = "\n".join(Path(a).read_text() for a in listings) if listings else compile_isa()
    bad = scan(text)
    print(f"blocks with scratch traffic ahead of the EXEC restore: {len(bad)}")
    for label, ln, sc in bad:
        print(" ", label, "line", ln, sc[:4])
    rc = 1 if bad else 0
    for name, r in sorted(resources(text).items()):
        short = name.split("N_1")[-1][:34]
        print(f"  {short:36s} vgpr {r.get('vgpr_count', -1):3d}  vgpr spills {r.get('vgpr_spill_count', -1):3d}  "
              f"sgpr spills {r.get('sgpr_spill_count', -1):3d}  scratch {r.get('private_segment_fixed_size', -1):4d} B")
        for fam, limits in LIMITS.items():
            if fam in name:
                for k, lim in limits.items():
                    if r.get(k, 0) > lim:
                        print(f"  !! {name}: {k} = {r[k]} exceeds the ceiling {lim}")
                        rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
