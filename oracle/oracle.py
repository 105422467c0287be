"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product (ectrans_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.orc_setup.restype = C.c_void_p
        L.orc_setup.argtypes = [C.c_int, C.c_int, ip, C.c_int, C.c_double]
        L.orc_setup_lazy.restype = C.c_void_p
        L.orc_setup_lazy.argtypes = [C.c_int, C.c_int, ip, C.c_double]
        L.orc_set_sp_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        for f in ("orc_nspec2", "orc_ngptot"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_rmu", "orc_rw"):
            getattr(L, f).restype = dp
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_nmen", "orc_ndglu", "orc_nasm0"):
            getattr(L, f).restype = ip
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_rpnma", "orc_rpnms"):
            getattr(L, f).restype = dp
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, ip, ip]
        L.orc_inv_trans.restype = C.c_int
        L.orc_inv_trans.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, C.c_int, C.c_int, C.c_int,
                                    C.c_int, dp]
        L.orc_dir_trans.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, dp]
        L.orc_specnorm.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orc_vordiv_to_uv.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp]
        L.orc_gpnorm.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp]
        L.orc_legpol.argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        L.orc_fft_r2c.argtypes = [C.c_int, dp, dp]
        L.orc_fft_c2r.argtypes = [C.c_int, dp, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """Mirror of SETUP_TRANS / INV_TRANS / DIR_TRANS / SPECNORM for one task.

    Array conventions (numpy, C-contiguous): spectral (nspec2, nfld); grid (nfld, ngptot).
    """

    def __init__(self, nsmax, nloen, belusov=False, ra=6371229.0, lazy=False):
        self.L = lib()
        nloen = np.ascontiguousarray(nloen, dtype=np.int32)
        self.nsmax, self.ndgl = int(nsmax), int(nloen.size)
        self.nloen = nloen
        # lazy: no stored Legendre panels (rebuilt per wavenumber inside the transforms; same values)
        if lazy:
            self.h = self.L.orc_setup_lazy(self.nsmax, self.ndgl, nloen.ctypes.data_as(C.POINTER(C.c_int)), float(ra))
        else:
            self.h = self.L.orc_setup(self.nsmax, self.ndgl, nloen.ctypes.data_as(C.POINTER(C.c_int)),
                                      int(belusov), float(ra))
        if not self.h:
            raise ValueError("orc_setup failed")
        self.nspec2 = self.L.orc_nspec2(self.h)
        self.ngptot = self.L.orc_ngptot(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    def _arr(self, fn, n, dt):
        return np.ctypeslib.as_array(fn(self.h), shape=(n,)).astype(dt).copy()

    @property
    def rmu(self):
        return self._arr(self.L.orc_rmu, self.ndgl, np.float64)

    @property
    def rw(self):
        return self._arr(self.L.orc_rw, self.ndgl, np.float64)

    @property
    def nmen(self):
        return self._arr(self.L.orc_nmen, self.ndgl, np.int32)

    @property
    def ndglu(self):
        return self._arr(self.L.orc_ndglu, self.nsmax + 1, np.int32)

    @property
    def nasm0(self):
        return self._arr(self.L.orc_nasm0, self.nsmax + 1, np.int32)

    def rpnm(self, m, sym):
        r, c = C.c_int(), C.c_int()
        fn = self.L.orc_rpnms if sym else self.L.orc_rpnma
        p = fn(self.h, m, C.byref(r), C.byref(c))
        if r.value * c.value == 0:
            return np.zeros((c.value, r.value))
        # column-major (rows=lat, cols=n desc) -> numpy [col][row]
        return np.ctypeslib.as_array(p, shape=(c.value, r.value)).copy()

    def inv_trans(self, spvor=None, spdiv=None, spsc=None, scders=False, vorgp=False, divgp=False,
                  uvder=False):
        nuv = 0 if spvor is None else spvor.shape[1]
        nsc = 0 if spsc is None else spsc.shape[1]
        if vorgp:
            divgp = True
        ngp = 2 * nuv + nsc + (2 * nsc if scders and nsc else 0) + (nuv if vorgp and nuv else 0) + \
            (nuv if divgp and nuv else 0) + (2 * nuv if uvder and nuv else 0)
        gp = np.zeros((ngp, self.ngptot))
        c = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (spvor, spdiv, spsc)]
        n = self.L.orc_inv_trans(self.h, nuv, nsc, _dp(c[0]), _dp(c[1]), _dp(c[2]), int(scders), int(vorgp),
                                 int(divgp), int(uvder), _dp(gp))
        assert n == ngp, (n, ngp)
        return gp

    def dir_trans(self, gp, nuv=0, nsc=0):
        gp = np.ascontiguousarray(gp, dtype=np.float64)
        assert gp.shape == (2 * nuv + nsc, self.ngptot)
        vor = np.zeros((self.nspec2, nuv)) if nuv else None
        div = np.zeros((self.nspec2, nuv)) if nuv else None
        sc = np.zeros((self.nspec2, nsc)) if nsc else None
        self.L.orc_dir_trans(self.h, nuv, nsc, _dp(gp), _dp(vor), _dp(div), _dp(sc))
        return vor, div, sc

    def legpol(self, m, jgl):
        """P_n^m(mu(jgl)), n = m .. nsmax (jgl 1-based): the values the Legendre panels hold"""
        out = np.zeros(self.nsmax - m + 1)
        self.L.orc_legpol(self.h, int(m), int(jgl), _dp(out))
        return out

    def set_sp_mode(self, on=True):
        """dir_trans computes LEDIR as libtrans_sp does: float operands, SGEMM, m = 0 in double (ledir_mod.F90:133-171)"""
        self.L.orc_set_sp_mode(self.h, int(on))

    def vordiv_to_uv(self, spvor, spdiv):
        """VORDIV_TO_UV: spectral (vor, div) -> spectral (U, V) = (u, v) cos(theta), n <= NSMAX"""
        vor, div = (np.ascontiguousarray(a, dtype=np.float64) for a in (spvor, spdiv))
        u, v = np.zeros_like(vor), np.zeros_like(vor)
        self.L.orc_vordiv_to_uv(self.h, vor.shape[1], _dp(vor), _dp(div), _dp(u), _dp(v))
        return u, v

    def gpnorm(self, gp):
        """GPNORM_TRANS: (average, minimum, maximum) of every grid field gp[f, point]"""
        gp = np.ascontiguousarray(gp, dtype=np.float64)
        nf = gp.shape[0]
        ave, mn, mx = np.zeros(nf), np.zeros(nf), np.zeros(nf)
        self.L.orc_gpnorm(self.h, nf, _dp(gp), _dp(ave), _dp(mn), _dp(mx))
        return ave, mn, mx

    def specnorm(self, sp):
        sp = np.ascontiguousarray(sp, dtype=np.float64)
        out = np.zeros(sp.shape[1])
        self.L.orc_specnorm(self.h, sp.shape[1], _dp(sp), _dp(out))
        return out


def fft_r2c(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(2 * (x.size // 2 + 1))
    lib().orc_fft_r2c(x.size, _dp(x), _dp(out))
    return out[0::2] + 1j * out[1::2]


def fft_c2r(X, n):
    buf = np.zeros(2 * (n // 2 + 1))
    buf[0::2], buf[1::2] = X.real, X.imag
    out = np.zeros(n)
    lib().orc_fft_c2r(n, _dp(buf), _dp(out))
    return out
