"""Checksum dumps of the benchmark harness (src/programs/ectrans-benchmark.F90:1455-1600, `dump_checksums`).

Every field of every array is gathered to task 1 (GATH_GRID / GATH_SPEC) and a CRC-64 of the global field is
appended to a text file, one line per field -- `zgpuv (lev, var) = 0123456789ABCDEF` -- in the reference's
formats.  The reference's own test then requires the dump of one decomposition (mpi0_omp1) to be byte-identical
to the dump of every other one (tests/compare_checksums.py:11-60); `compare_checksums` below is that comparison.
Host-side harness code: arrays are brought to numpy; nothing here is on the transform path.
"""
import ctypes as C
import os

import numpy as np

from . import TransError, _chk, _np, gath_grid, gath_spec, lib, real_dtype, trans_inq


def crc64(a, crc=0):
    """CRC-64 (emi_crc64) of the bytes of array `a`, continued from `crc`."""
    a = np.ascontiguousarray(a)
    c = C.c_ulonglong(crc)
    _chk(lib().emi_crc64(a.ctypes.data, a.nbytes, C.byref(c)))
    return c.value


def dump_checksums(filename, jstep, kresol, kproma=None, zgp=None, zgpuv=None, zgp3a=None, zgp2=None,
                   zspvor=None, zspdiv=None, zspscalar=None, zspsc3a=None, zspsc2=None):
    """One `iteration` block of the dump.  Argument combinations as the reference (iconfig 1-4): call mode 1 grid
    (zgp), call mode 2 grid (zgpuv, zgp3a, zgp2), call mode 1 spectral (zspvor, zspdiv, zspscalar), call mode 2
    spectral (zspvor, zspdiv, zspsc3a, zspsc2).  Arrays use this package's index order (Fortran order reversed).
    Every task calls; task 1 writes."""
    me = trans_inq(kresol, "myproc")
    grid1 = zgp is not None
    grid2 = zgpuv is not None and zgp3a is not None and zgp2 is not None
    spec1 = zspvor is not None and zspdiv is not None and zspscalar is not None
    spec2 = zspvor is not None and zspdiv is not None and zspsc3a is not None and zspsc2 is not None
    if not (grid1 or grid2 or spec1 or spec2):
        raise TransError("dump_checksums: invalid argument combination")
    dt = np.dtype(real_dtype(kresol))
    lines = []

    def gfield(blocked):  # (ngpblks, nproma) of one field -> global (ngptotg,)
        g = gath_grid(kresol, np.ascontiguousarray(_np(blocked))[:, None, :], 1, kto=1)
        return None if g is None else np.ascontiguousarray(g[0], dtype=dt)

    def sfield(col):  # (nspec2,) of one field -> global (nspec2g,)
        g = gath_spec(kresol, np.ascontiguousarray(_np(col))[:, None], 1, kto=1)
        return None if g is None else np.ascontiguousarray(g[:, 0], dtype=dt)

    def run(name, fields, fmt):
        crc = 0  # carried through the fields of one array, as the reference's icrc
        for idx, get in fields:
            g = get()
            if me == 1:
                crc = crc64(g, crc)
                lines.append(fmt % ((name,) + idx + (crc,)))

    one, two = "%s (%d) = %016X", "%s (%d, %d) = %016X"
    if grid1:
        a = _np(zgp)
        run("zgp", [((f + 1,), (lambda f=f: gfield(a[:, f]))) for f in range(a.shape[1])], one)
    elif grid2:
        a, b, c = _np(zgpuv), _np(zgp3a), _np(zgp2)
        run("zgpuv", [((l + 1, v + 1), (lambda l=l, v=v: gfield(a[:, v, l]))) for v in range(a.shape[1]) for l in range(a.shape[2])], two)
        run("zgp3a", [((l + 1, v + 1), (lambda l=l, v=v: gfield(b[:, v, l]))) for v in range(b.shape[1]) for l in range(b.shape[2])], two)
        run("zgp2", [((f + 1,), (lambda f=f: gfield(c[:, f]))) for f in range(c.shape[1])], one)
    else:
        v, d = _np(zspvor), _np(zspdiv)
        run("zspvor", [((f + 1,), (lambda f=f: sfield(v[:, f]))) for f in range(v.shape[1])], one)
        run("zspdiv", [((f + 1,), (lambda f=f: sfield(d[:, f]))) for f in range(d.shape[1])], one)
        if spec1 and not spec2:
            s = _np(zspscalar)
            run("zspscalar", [((f + 1,), (lambda f=f: sfield(s[:, f]))) for f in range(s.shape[1])], one)
        else:
            s3, s2 = _np(zspsc3a), _np(zspsc2)
            run("zspsc3a", [((l + 1, v + 1), (lambda l=l, v=v: sfield(s3[v, :, l]))) for v in range(s3.shape[0]) for l in range(s3.shape[2])], two)
            run("zspsc2", [((f + 1,), (lambda f=f: sfield(s2[:, f]))) for f in range(s2.shape[1])], one)
    if me == 1:
        # list-directed WRITE(unit,*) of the reference: a leading blank, integers right-adjusted in 12 columns
        head = [" ====================", " iteration%12d" % jstep, " ===================="]
        mode = "a" if (jstep > 1 and os.path.exists(filename)) else "w"
        with open(filename, mode) as f:
            f.write("\n".join(head + lines) + "\n")


def compare_checksums(file_a, file_b):
    """tests/compare_checksums.py:36: filecmp of two dumps -- byte identity."""
    import filecmp
    return filecmp.cmp(file_a, file_b, shallow=False)
