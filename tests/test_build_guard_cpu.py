"""The compiled ISA of csrc/nmhip.hip must not contain register-spill traffic ahead of an EXEC-mask restore
(tools/check_spill_exec.py explains the hazard: a lane-divergent branch + spills = lanes that never store).
Cross-compiles for gfx950 on the CPU box; no GPU needed."""
import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _mod():
    spec = importlib.util.spec_from_file_location("check_spill_exec", ROOT / "tools" / "check_spill_exec.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_scanner_flags_the_known_bad_pattern():
    bad_asm = """
.LBB0_1:
\ts_and_saveexec_b64 s[16:17], s[10:11]
\ts_cbranch_execz .LBB0_3
.LBB0_3:
\ts_mov_b32 s95, s63
\tscratch_store_dword off, v153, off offset:124 ; 4-byte Folded Spill
\ts_or_b64 exec, exec, s[16:17]
\tv_mov_b32 v0, v1
.LBB0_4:
\ts_or_b64 exec, exec, s[18:19]
\tscratch_store_dword off, v1, off offset:8
"""
    bad = _mod().scan(bad_asm)
    assert [b[0] for b in bad] == [".LBB0_3"]


def test_kernel_isa_has_no_spill_ahead_of_exec_restore_and_step_kernel_does_not_spill():
    """main() compiles the kernel once and checks both the EXEC hazard and the resource ceilings: nm_step_kernel (every
    instantiation) and the general-shape kernel 0 VGPR spills and 0 bytes of scratch, SGPR spills bounded; the head
    kernels 0 VGPR spills, 0 bytes of scratch and SGPR spills bounded (they were printed, not gated)."""
    m = _mod()
    assert m.main() == 0
    for fam in ("nm_step_kernel", "nm_wide_step_kernel", "nm_rs_kernel"):
        assert m.LIMITS[fam]["vgpr_spill_count"] == 0 and m.LIMITS[fam]["private_segment_fixed_size"] == 0
        assert m.LIMITS[fam]["sgpr_spill_count"] <= 540
    for fam in ("nm_head_step_kernel", "nm_clshead_kernel", "nm_reghead_kernel"):
        assert m.LIMITS[fam]["vgpr_spill_count"] == 0 and m.LIMITS[fam]["private_segment_fixed_size"] == 0


def test_resource_parser():
    txt = """
    .name:           _ZN12_GLOBAL__N_114nm_step_kernelILb0EEEvPK6nm_jobiii
    .private_segment_fixed_size: 268
    .sgpr_spill_count: 201
    .vgpr_count:     256
    .vgpr_spill_count: 88
"""
    r = _mod().resources(txt)
    assert list(r.values())[0] == {"private_segment_fixed_size": 268, "sgpr_spill_count": 201, "vgpr_count": 256,
                                   "vgpr_spill_count": 88}


def test_build_tracks_included_files():
    """build() must notice an edit of a file the library's sources #include (nm_wide.inc): the staleness check once looked
    at the .hip files only and kept a stale library after such an edit."""
    import inspect
    import re
    import __graft_entry__ as ge
    inc = set()
    for src in ge.SOURCES:
        inc |= set(re.findall(r'#include "([^"]+\.inc)"', src.read_text()))
    assert inc and all((ge.PKG / "csrc" / i).exists() for i in inc)
    assert '"*.inc"' in inspect.getsource(ge.build)
