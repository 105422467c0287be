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


# ---- the Fortran drop-in: the precision-suffixed entry points of ecTrans 1.7.0 (src/trans/CMakeLists.txt:43-93, sedrenames.txt)
RENAMED = ["setup_trans", "trans_inq", "specnorm", "gath_grid", "dist_grid", "gath_spec", "dist_spec", "trans_release", "trans_end", "inv_transad",
           "dir_transad", "inv_trans", "dir_trans", "gpnorm_trans", "vordiv_to_uv", "trans_pnm", "dist_grid_32", "gath_grid_32", "gpnorm_transtl",
           "gpnorm_transad"]
COMMON = ["setup_trans0", "get_current", "ini_spec_dist"]  # ectrans_common_includes: one copy, plain names


def _dynsyms(lib):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_fortran_libraries_export_the_suffixed_entry_points():
    """libectrans_mi_f.so exports NAME_dp_, libectrans_mi_f_sp.so NAME_sp_ (flang's external-name mangling of NAME_DP / NAME_SP), neither
    exports the other's, the module procedures of the two do not collide (emi_shim_mod_dp / _sp), and SETUP_TRANS0 / GET_CURRENT /
    INI_SPEC_DIST exist once, in libectrans_mi_f_common.so -- so both precisions link into one executable.  The unsuffixed names of
    pre-1.6 callers remain as aliases in each precision library."""
    import subprocess
    d = os.path.join(ROOT, "ectrans_amd", "fortran")
    subprocess.check_call(["make", "-s", "-C", d, "libectrans_mi_f_common.so", "libectrans_mi_f.so", "libectrans_mi_f_sp.so"])
    dp, sp, cm = (_dynsyms(os.path.join(d, n)) for n in ("libectrans_mi_f.so", "libectrans_mi_f_sp.so", "libectrans_mi_f_common.so"))
    for n in RENAMED:
        assert n + "_dp_" in dp and n + "_sp_" in sp, n
        assert n + "_sp_" not in dp and n + "_dp_" not in sp, n
        assert n + "_" in dp and n + "_" in sp, "unsuffixed alias of %s missing" % n
        assert n + "_" not in cm and n + "_dp_" not in cm, n
    for n in COMMON:
        assert n + "_" in cm and n + "_" not in dp and n + "_" not in sp, n
    mod_dp = {x for x in dp if x.startswith("_QM")}
    mod_sp = {x for x in sp if x.startswith("_QM")}
    assert mod_dp and mod_sp and not (mod_dp & mod_sp), sorted(mod_dp & mod_sp)
    assert all("_dp" in x for x in mod_dp) and all("_sp" in x for x in mod_sp)


def test_fortran_compat_headers_are_generated_for_every_routine():
    """ectrans_amd/fortran/include: NAME_dp.h / NAME_sp.h (interface blocks) and trans_dp/NAME.h, trans_sp/NAME.h (the `#define NAME NAME_DP`
    back-compat headers of src/trans/CMakeLists.txt:76-85) for every renamed routine; plain headers for the three common ones."""
    inc = os.path.join(ROOT, "ectrans_amd", "fortran", "include")
    for n in RENAMED:
        for tag in ("dp", "sp"):
            assert ("SUBROUTINE %s_%s(" % (n.upper(), tag.upper())) in open(os.path.join(inc, "%s_%s.h" % (n, tag))).read(), (n, tag)
            txt = open(os.path.join(inc, "trans_" + tag, n + ".h")).read()
            assert "#define %s %s_%s" % (n.upper(), n.upper(), tag.upper()) in txt and '#include "../%s_%s.h"' % (n, tag) in txt, (n, tag)
    for n in COMMON:
        assert ("SUBROUTINE %s(" % n.upper()) in open(os.path.join(inc, n + ".h")).read()
