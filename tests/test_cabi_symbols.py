"""The C-ABI library loads (no GPU needed) and exports every symbol include/ectrans_mi.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ectrans_mi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(emi_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def hiplib():
    import ectrans_amd
    path = ectrans_amd.build()
    return ctypes.CDLL(path)


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("emi_init", "emi_setup", "emi_inv_trans", "emi_dir_trans", "emi_specnorm", "emi_inq_int",
                 "emi_inq_int_array", "emi_inq_real_array", "emi_release", "emi_finalize", "emi_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(hiplib):
    for s in declared_symbols():
        assert hasattr(hiplib, s), "libectrans_mi.so does not export %s" % s


def test_product_has_no_cpu_path(hiplib):
    """Without a GPU, emi_init must fail loudly (and never fall back to a CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    hiplib.emi_last_error.restype = ctypes.c_char_p
    rc = hiplib.emi_init(None)
    assert rc != 0
    assert b"no HIP device" in hiplib.emi_last_error()


def test_product_package_does_not_import_oracle():
    """ectrans_amd/ must not reference oracle/ or the emulator build."""
    pkg = os.path.join(ROOT, "ectrans_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".c", ".F90", ".f90")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn
