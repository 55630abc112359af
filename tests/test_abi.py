"""CPU: the C-ABI shared library builds, loads and exports every symbol include/benerf_hip.h
declares (no compute calls without a GPU), and the product path refuses to run on CPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from benerf_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from benerf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "benerf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(benerf_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == list(_lib.EXPORTED_SYMBOLS), "ctypes table and header disagree"
    for name in declared:
        assert hasattr(lib, name), name


def test_size_queries(lib):
    from benerf_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "benerf_hip.h")).read()
    assert lib.benerf_version() == L.ABI_VERSION == int(re.search(r"#define BENERF_ABI_VERSION (\d+)", hdr).group(1))
    # sized for every arithmetic mode.  f32: rows of 2528 / 2432 floats per point + 8 layers of ReLU sign-bit words per
    # 64-point tile; split: points padded to 128, PE rows (64 + 32 floats) + f16 arrays (9 x 256 + 128 halfs), 9 mask layers,
    # 16 info words, then the twin arrays of 8-bit residual codes (one byte per saved value: the fp32-equivalent backward)
    def f32_act(m):
        return m * ((64 + 32) + 8 * 256 + 256 + 128) + 8 * ((m + 63) // 64) * 256 * 2

    def split_act(m):
        mp = (m + 127) // 128 * 128
        return mp * (64 + 32) + mp * (9 * 256 + 128) // 2 + 9 * (mp // 64) * 256 * 2 + 16 + mp * (9 * 256 + 128) // 4

    for m in (640, 641, 100000):
        assert lib.benerf_mlp_act_floats(m) == max(f32_act(m), split_act(m))
        mp = (m + 127) // 128 * 128
        assert lib.benerf_mlp_dact_floats(m) == max(m * (8 * 256 + 256 + 128), mp * (9 * 256 + 128) // 2 + 16 + mp * (9 * 256 + 128) // 4)
        # per-mode sizes (precision codes of include/benerf_hip.h: F32 0, SPLIT 1, AUTO 2, SPLIT_F16BWD 3)
        assert lib.benerf_mlp_act_floats_for(m, 0) == f32_act(m) and lib.benerf_mlp_act_floats_for(m, 1) == split_act(m)
        assert lib.benerf_mlp_act_floats_for(m, 3) == split_act(m) - mp * (9 * 256 + 128) // 4
        assert lib.benerf_mlp_dact_floats_for(m, 0) == m * (8 * 256 + 256 + 128)
        assert lib.benerf_mlp_dact_floats_for(m, 1) == mp * (9 * 256 + 128) // 2 + 16 + mp * (9 * 256 + 128) // 4
        assert lib.benerf_mlp_dact_floats_for(m, 3) == mp * (9 * 256 + 128) // 2 + 16
        assert lib.benerf_mlp_act_floats_for(m, 2) == 0 and lib.benerf_mlp_dact_floats_for(m, 7) == 0
    assert lib.benerf_mlp_dact_floats_per_point() == 8 * 256 + 256 + 128
    assert lib.benerf_mlp_packed_floats() > 2 * 593920 - 200000
    assert lib.benerf_mlp_dw_workspace_floats(1000) > 0


def test_bad_arguments_are_reported(lib):
    rc = lib.benerf_spline_poses_fwd(None, None, None, 2, 0, 0, None, None)
    assert rc == -1 and b"null" in lib.benerf_last_error()
    rc = lib.benerf_composite_fwd(None, None, None, None, 0.0, 0, 0, 5, 1, 1, None, None, None, None, None, None, None)
    assert rc == -1


def test_no_cpu_fallback():
    from benerf_amd import _lib, kernels
    with pytest.raises(_lib.BenerfHipError):
        kernels.spline_poses_fwd(torch.zeros(4, 6), None, torch.zeros(2), 2, 0)
