"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/nmhip.h
declares, and its host-only helpers behave (no compute calls: there is no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    if not _lib.LIB_PATH.exists():
        import __graft_entry__ as ge
        ge.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    header = (ROOT / "include" / "nmhip.h").read_text()
    declared = set(re.findall(r"\b(nm_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_abi_struct_sizes_match(lib):
    sj, sm = C.c_int64(0), C.c_int64(0)
    assert lib.nm_abi_sizes(C.byref(sj), C.byref(sm)) == 0
    assert sj.value == C.sizeof(_lib.NmJob)
    assert sm.value == C.sizeof(_lib.NmModality)


def _probe(M=1, L=2, Z=10, C_=29, H=(110, 110), D=379):
    j = _lib.NmJob()
    j.M, j.L, j.Z, j.C = M, L, Z, C_
    for i, h in enumerate(H):
        j.H[i] = h
    for m in range(min(M, _lib.NM_MAX_MOD)):
        j.mod[m].D = D
        j.mod[m].Kx = (D + C_ + 1 + 31) // 32 * 32
        j.mod[m].x_pitch = (D + 3) // 4 * 4
        j.mod[m].Cz = (C_ + 1 + 7) // 8 * 8
    j.n_rows, j.loss_cap, j.eps_cap = 256, 1, 1
    j.n_params = 118479    # config A's parameter count
    j.wsh = 4096           # any non-null address: validation does not dereference it
    return j


def test_validate_job_limits(lib):
    assert lib.nm_validate_job(C.byref(_probe())) == 0
    assert lib.nm_validate_job(C.byref(_probe(M=5))) == -2
    assert lib.nm_validate_job(C.byref(_probe(L=9, H=(8, 8, 8)))) == -3
    assert lib.nm_validate_job(C.byref(_probe(L=5, H=(90, 90, 90, 90, 90)))) == 0     # "-H 90 90 90 90 90 10", commands_list9_endtoend.sh:24
    assert lib.nm_validate_job(C.byref(_probe(H=(128, 110)))) == -4
    assert lib.nm_validate_job(C.byref(_probe(Z=65))) == -5
    assert lib.nm_validate_job(C.byref(_probe(Z=64, C_=64))) == -6
    bad = _probe()
    bad.mod[0].Kx = 400
    assert lib.nm_validate_job(C.byref(bad)) == -7
    assert b"Kx" in lib.nm_status_string(-7)
    for field in ("n_rows", "loss_cap", "eps_cap"):            # modulo divisors inside the kernel
        bad = _probe()
        setattr(bad, field, 0)
        assert lib.nm_validate_job(C.byref(bad)) == -14
    bad = _probe()
    bad.wsh = None
    assert lib.nm_validate_job(C.byref(bad)) == -15
    for n in (0, 1 << 30):                                      # 32-bit byte offsets into params / adam_m / adam_v
        bad = _probe()
        bad.n_params = n
        assert lib.nm_validate_job(C.byref(bad)) == -21
    # deviation-pass kernel: one expert with the bypass, first hidden width <= 112, latent <= 32, Gaussian output
    ok = _probe()
    ok.w_off, ok.single_bypass = -1, 1
    assert lib.nm_devpass_ok(C.byref(ok)) == 0
    for field, val in (("M", 2), ("single_bypass", 0), ("n_private", 1), ("tc_weight", 1e-4), ("w_off", 0), ("out_kind", 1), ("wide", 1), ("Z", 33)):
        bad = _probe()
        bad.w_off, bad.single_bypass = -1, 1
        setattr(bad, field, val)
        assert lib.nm_devpass_ok(C.byref(bad)) == -22, field
    bad = _probe(H=(113, 110))
    bad.w_off, bad.single_bypass = -1, 1
    assert lib.nm_devpass_ok(C.byref(bad)) == -22
    # row-split launch: plain multimodal models with a partial-gradient buffer only
    ok = _probe(M=3)
    ok.w_off, ok.gpart, ok.gpart_stride = -1, 4096, 118528
    assert lib.nm_rowsplit_ok(C.byref(ok)) == 0
    for field, val in (("gpart", None), ("gpart_stride", 118479), ("tc_weight", 1e-4), ("w_off", 0), ("n_private", 2),
                       ("out_kind", 1), ("M_enc", 2), ("reg_head", 1), ("wide", 1)):
        bad = _probe(M=3)
        bad.w_off, bad.gpart, bad.gpart_stride = -1, 4096, 118528
        setattr(bad, field, val)
        assert lib.nm_rowsplit_ok(C.byref(bad)) == -20, field
    bad = _probe()
    bad.mod[0].enc_w[0] = 8                                     # weight matrices start on a tile boundary
    assert lib.nm_validate_job(C.byref(bad)) == -10
    # the general-shape path (nm_job_t.wide): the widths / latents of the reference's sweeps the fused tile cannot hold
    for kw in (dict(H=(1024, 512, 256), L=3, Z=32), dict(H=(300, 300), Z=30), dict(H=(110, 110), Z=100), dict(H=(2048,), L=1, Z=10)):
        j = _probe(**kw)
        assert lib.nm_validate_job(C.byref(j)) in (-4, -5, -6)      # not a fused-kernel shape ...
        j.wide, j.w_off = 1, -1
        j.wsh = None                                                # ... and the wide path needs no shadow images
        assert lib.nm_validate_job(C.byref(j)) == 0
        assert lib.nm_workspace_bytes(C.byref(j)) > 0
    # what the general-shape path serves: the DMVAE family's switches too; mvtCAE's total correlation needs
    # experts x latent <= 256 (its log-sum-exps sit in 256 floats of LDS); the regression head is its own kernel
    for field, val in (("n_private", 2), ("w_off", 0)):
        j = _probe(H=(300, 300), Z=30)
        j.wide, j.w_off = 1, -1
        setattr(j, field, val)
        assert lib.nm_validate_job(C.byref(j)) == 0, field
    j = _probe(H=(300, 300), Z=30, M=3)
    j.wide, j.w_off, j.tc_weight = 1, -1, 3e-4
    assert lib.nm_validate_job(C.byref(j)) == 0
    j = _probe(H=(300, 300), Z=100, M=3)
    j.wide, j.w_off, j.tc_weight = 1, -1, 3e-4
    assert lib.nm_validate_job(C.byref(j)) == -19
    j = _probe(H=(300, 300), Z=30)
    j.wide, j.reg_head, j.w_off = 1, 1, -1
    assert lib.nm_validate_job(C.byref(j)) == -11                   # (regression head: its own buffers are checked as ever)
    n = lib.nm_fill_shadow(C.byref(j))                              # ... and its first-layer images are the job's only shadow
    assert 0 < n < 1 << 20 and j.reg_s > 0 and j.mod[0].enc_s[0] == 0 and j.mod[0].out_s == 0
    assert lib.nm_validate_job(C.byref(_probe(H=(5000, 10)))) == -4


def test_shadow_layout(lib):
    j = _probe(M=3)
    n = lib.nm_fill_shadow(C.byref(j))
    assert n > 0 and n % 1024 == 0
    offs = []
    for m in range(3):
        md = j.mod[m]
        offs += [md.enc_s[0], md.enc_s[1], md.heads_s, md.dec_s[0], md.dec_s[1], md.out_s]
    assert offs == sorted(offs) and len(set(offs)) == len(offs) and all(o % 1024 == 0 for o in offs)
    # the first encoder layer of a 379 + 29 + 1 wide input: the matrix compact, [112][416] bf16, + 1 KiB of vectors
    assert j.mod[0].enc_s[1] - j.mod[0].enc_s[0] == 112 * 416 * 2 + 1024
    # ... a hidden 110 x 110 layer: [112][112] bf16 rounded to 1 KiB + 1 KiB of vectors (round 2: a 35 KiB padded image)
    assert j.mod[0].heads_s - j.mod[0].enc_s[1] == 25 * 1024 + 1024
    # 6 output chunk blobs of 18 KiB for 379 ROI
    assert j.mod[1].enc_s[0] - j.mod[0].out_s == 6 * 18432
    assert lib.nm_xb_elems(1024, 416) == 1024 * 7 * 72


def test_workspace_bytes(lib):
    w1 = lib.nm_workspace_bytes(C.byref(_probe()))
    w3 = lib.nm_workspace_bytes(C.byref(_probe(M=3)))
    assert w1 > 0 and w3 > w1 and w1 % 256 == 0
    # the classifier head's region: one [256][128] tile per activation, or tiles of blocks wider than 128 (nmhip.h: NM_MAX_CLS_WIDTH)
    p = _probe(M=3)
    p.cls_layers, p.cls_classes = 3, 2
    for i, w in enumerate((128, 64, 32)):
        p.cls_width[i] = w
    narrow = lib.nm_workspace_bytes(C.byref(p))
    p.cls_width[0] = 256
    tiled = lib.nm_workspace_bytes(C.byref(p))
    assert narrow > w3 and tiled > narrow + 2 * 256 * 128 * 2


def test_param_layout_matches_reference_names():
    from tests.golden_util import Golden
    for name in ("mm3_gpoe", "cfgA_T1w", "mm1_h1"):
        g = Golden(name)
        lay = nm.ParamLayout(nm.ModelSpec(g.dims, g.hidden, g.Z, g.c_dim))
        w = g.weights("w0")
        assert lay.names == list(w.keys())              # reference state_dict() order
        flat = lay.flatten(w)
        for k, v in lay.unflatten(flat).items():
            assert tuple(v.shape) == tuple(w[k].shape)
            assert (v == w[k]).all()
        assert all(o % 4 == 0 for o in lay.offsets.values())
        # weight matrices: whole 16 x 16 tiles on tile boundaries, zero outside the matrix
        for k, (nt, kt) in lay.tiles.items():
            n, kk = lay.shapes[k]
            o = lay.offsets[k]
            assert o % 256 == 0
            full = flat[o:o + nt * kt * 256].view(nt, kt, 16, 16).permute(0, 2, 1, 3).reshape(nt * 16, kt * 16)
            assert (full[:n, :kk] == w[k]).all() and (full[n:] == 0).all() and (full[:, kk:] == 0).all()
            assert float(flat[o + 16 * 1 + 3]) == float(w[k][1, 3])          # wt_off(1, 3) inside tile (0, 0)
        # natural <-> kernel buffer (the eager facade's parameter views)
        nat = lay.nat_flatten(w)
        import torch
        assert torch.equal(lay.nat_to_kernel(nat, torch.zeros(lay.total)), flat)
        assert torch.equal(lay.kernel_to_nat(flat, torch.zeros(lay.nat_total)), nat)
    g = Golden("cfgA_T1w")
    assert nm.ParamLayout(nm.ModelSpec(g.dims, g.hidden, g.Z, g.c_dim)).n_params == 118479


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.NmError):
        nm.Table(torch.zeros(4, 3), torch.zeros(4, 2))


def test_spec_limits_raise():
    assert nm.ModelSpec([10], [200], 5, 2).wide and not nm.ModelSpec([10], [127], 64, 29).wide
    assert nm.ModelSpec([10], [110, 110], 100, 29).wide                      # latent + c_dim > 127
    nm.ParamLayout(nm.ModelSpec([10], [200], 5, 2))                           # general-shape path: constructs
    with pytest.raises(ValueError):
        nm.ParamLayout(nm.ModelSpec([10], [5000], 5, 2))
    with pytest.raises(ValueError):
        nm.ParamLayout(nm.ModelSpec([10], [20], 129, 2))
    nm.ParamLayout(nm.ModelSpec([10, 10], [200], 5, 2, True, "regression"))   # regression / mvtCAE: general-shape path too
    nm.ParamLayout(nm.ModelSpec([10, 10], [200], 5, 2, True, "mvtcae"))
    nm.ParamLayout(nm.ModelSpec([10, 10], [20], 5, 2, True, "endtoend", (256, 128, 64), 2))    # classifier blocks up to 512 wide
    with pytest.raises(ValueError):
        nm.ParamLayout(nm.ModelSpec([10, 10], [20], 5, 2, True, "endtoend", (1024, 64), 2))
    nm.ParamLayout(nm.ModelSpec([10, 10], [200, 100], 5, 2, True, "dmvae"))   # ... and the DMVAE family
    with pytest.raises(ValueError):                                           # mvtCAE there: experts x latent <= 256
        nm.ParamLayout(nm.ModelSpec([10, 10, 10], [200, 100], 100, 2, True, "mvtcae"))
    with pytest.raises(ValueError):
        nm.ParamLayout(nm.ModelSpec([10] * 5, [20], 5, 2))
