"""CPU suite: the C-ABI shared library loads without a GPU and exports every symbol include/dpsx.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dpsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpsx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    assert len(names) >= 25
    for must in ("dpsx_posterior_fwd_f32", "dpsx_posterior_bwd_f32", "dpsx_op_create_blur", "dpsx_op_create_resize",
                 "dpsx_op_create_mask", "dpsx_op_create_phase", "dpsx_op_forward_f32", "dpsx_op_adjoint_f32",
                 "dpsx_residual_norm_f32", "dpsx_norm_bwd_f32", "dpsx_step_fwd_f32", "dpsx_step_bwd_f32",
                 "dpsx_step_bwd_extra_f32",
                 "dpsx_step_update_f32", "dpsx_update_f32", "dpsx_score_f32", "dpsx_argmin_f32", "dpsx_gather_f32",
                 "dpsx_replicate_f32"):
        assert must in names


def test_library_loads_and_exports_every_declared_symbol():
    from dps_ttc_amd import _lib
    assert os.path.exists(_lib.SO_PATH), "build with: python -c 'import __graft_entry__ as g; g.build()'"
    raw = ctypes.CDLL(_lib.SO_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} is declared in include/dpsx.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in dps_ttc_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared_symbols()      # and nothing undeclared is bound


def test_no_compute_entry_points_work_without_a_gpu():
    from dps_ttc_amd import _lib
    lib = _lib.lib()
    assert lib.dpsx_abi_version() == _lib.ABI_VERSION == 3
    assert lib.dpsx_strerror(_lib.OK) == b"ok"
    assert b"workspace" in lib.dpsx_strerror(_lib.EWORKSPACE)
    with pytest.raises(_lib.DpsxError, match="invalid argument"):
        _lib.check(_lib.EINVAL, "probe")


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under dps_ttc_amd/ may reference it"""
    pkg = os.path.join(ROOT, "dps_ttc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "dps_oracle" not in src, f
