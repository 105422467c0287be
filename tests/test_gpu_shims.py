"""The Fortran drop-in shim and the transi-style C layer, driven by small native callers on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(subdir, target, marker):
    d = os.path.join(ROOT, "ectrans_amd", subdir)
    subprocess.check_call(["make", "-s", "-C", d, target])
    p = subprocess.run([os.path.join(d, target)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert marker in p.stdout


def test_fortran_shim_roundtrip():
    """SETUP_TRANS0/SETUP_TRANS/TRANS_INQ/INV_TRANS/DIR_TRANS/SPECNORM with the reference's keyword
    interfaces (tests/fortran/test_shim.F90): benchmark harmonic, norm drift <= 100 eps."""
    _run("fortran", "test_shim", "FORTRAN SHIM OK")


def test_transi_c_api():
    """trans_new/trans_setup/trans_inquire/trans_dirtrans/trans_invtrans/trans_specnorm
    (tests/transi/transi_test.c, modelled on the reference's transi_test_program.c)."""
    _run("transi", "transi_test", "TRANSI API OK")
