"""Pins the CPU oracle against the reference's own golden vectors.

Fixtures = data files held by the reference's tests
(/root/reference/tests/test_ectrans4py/data, used by test_ectrans4py.py:89-158):
T148 on a 150-latitude reduced Gaussian grid, tolerance 1e-10 absolute in both directions,
sum of Gaussian weights == 1, NMEN per latitude == zonal_wavenumbers.npy.
"""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle, fft_c2r, fft_r2c
from tests.common import closed_form_errors, octahedral

EPSILON = 1e-10  # test_ectrans4py.py:16


@pytest.fixture(scope="module")
def tl149(golden_dir):
    d = os.path.join(golden_dir, "tl149")
    nloen = np.load(os.path.join(d, "lon_number_by_lat.npy"))
    zw = np.load(os.path.join(d, "zonal_wavenumbers.npy"))
    sp = np.load(os.path.join(d, "tl149-c24-s1t@sp.npy"))
    gpll = np.load(os.path.join(d, "tl149-c24-s1t@sp2gp.npy"))
    # pack lat-lon padded data onto the reduced grid (test_ectrans4py.py:100-106)
    gp = np.concatenate([gpll[i, : nloen[i]] for i in range(nloen.size)])
    return nloen, zw, sp, gp


@pytest.mark.parametrize("belusov", [True, False])
def test_tl149_golden(tl149, belusov):
    nloen, zw, sp, gp = tl149
    o = Oracle(148, nloen, belusov=belusov)
    assert (o.ngptot, o.nspec2 // 2) == (33052, 11175)  # test_ectrans4py.py:94-97
    assert abs(o.rw.sum() - 1.0) < EPSILON  # test_ectrans4py.py:119-121
    np.testing.assert_array_equal(o.nmen, zw)  # test_ectrans4py.py:123-131
    g = o.inv_trans(spsc=sp.reshape(-1, 1))[0]
    assert np.abs(g - gp).max() < EPSILON  # test_sp2gp
    _, _, s = o.dir_trans(gp.reshape(1, -1), nsc=1)
    assert np.abs(s[:, 0] - sp).max() < EPSILON  # test_gp2sp


def test_belusov_vs_supolf_panels(tl149):
    nloen = tl149[0]
    a, b = Oracle(148, nloen, belusov=True), Oracle(148, nloen, belusov=False)
    for m in (0, 1, 2, 17, 74, 147, 148):
        for sym in (False, True):
            pa, pb = a.rpnm(m, sym), b.rpnm(m, sym)
            assert pa.shape == pb.shape
            if pa.size:
                assert np.abs(pa - pb).max() < 1e-12


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 18, 20, 30, 36, 50, 74, 94, 97, 101, 300, 2 * 1283])
def test_fft_semantics(n):
    """FFTW r2c/c2r semantics (tpm_fftw.F90:294-321) against numpy's DFT."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    X = np.fft.rfft(x)
    assert np.abs(fft_r2c(x) - X).max() < 1e-12 * max(1, n)
    assert np.abs(fft_c2r(X, n) - np.fft.irfft(X, n) * n).max() < 1e-12 * max(1, n)


def test_benchmark_harmonic_roundtrip():
    """ectrans-benchmark input (ectrans-benchmark.F90:1381-1419): Re(m=4,n=19)=1, T47/O48,
    2 inv+dir iterations, norm drift <= 100 eps (tests/CMakeLists.txt:252-299)."""
    N = 47
    H = N + 1
    nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))])
    o = Oracle(N, nloen, belusov=False)
    nlev = 2
    sp = np.zeros((o.nspec2, nlev))
    idx = o.nasm0[4] - 1 + 2 * (19 - 4)
    sp[idx, :] = 1.0
    vor, div, sc = sp.copy(), sp.copy(), sp.copy()
    n0 = o.specnorm(sc)
    for _ in range(2):
        gp = o.inv_trans(spvor=vor, spdiv=div, spsc=sc)
        vor, div, sc = o.dir_trans(gp, nuv=nlev, nsc=nlev)
    eps = np.finfo(np.float64).eps
    for a in (vor, div, sc):
        assert np.abs(n0 / o.specnorm(a) - 1.0).max() <= 100 * eps


@pytest.mark.parametrize("grid", ["full", "octahedral"])
def test_dense_roundtrip_with_winds(grid):
    """Dense random spectrum: inv then dir returns the input.

    On the full Gaussian grid the quadrature is exact (errors ~1e-14).  On the octahedral
    reduced grid the (lat, m > NMEN(lat)) corner is dropped by design
    (setup_geom_mod.F90:64-78), which leaves a ~1e-12 truncation error in the reference too.
    """
    N = 63
    H = N + 1
    if grid == "full":
        nloen = np.full(2 * H, 4 * H + 16)
        tol_sc, tol_vd = 1e-14, 2e-13
    else:
        nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))])
        tol_sc, tol_vd = 1e-11, 5e-11
    o = Oracle(N, nloen, belusov=False)
    rng = np.random.default_rng(20251114)
    nasm0 = o.nasm0

    def rand_spec(nf, zero00):
        sp = np.zeros((o.nspec2, nf))
        for m in range(N + 1):
            for n in range(m, N + 1):
                i = nasm0[m] - 1 + 2 * (n - m)
                sp[i] = rng.uniform(-0.5, 0.5, nf) / (n + 1)
                sp[i + 1] = 0.0 if m == 0 else rng.uniform(-0.5, 0.5, nf) / (n + 1)
        if zero00:
            sp[nasm0[0] - 1] = 0.0
        return sp

    vor, div, sc = rand_spec(2, True), rand_spec(2, True), rand_spec(3, False)
    gp = o.inv_trans(spvor=vor, spdiv=div, spsc=sc)
    v2, d2, s2 = o.dir_trans(gp, nuv=2, nsc=3)
    assert np.abs(s2 - sc).max() < tol_sc
    assert np.abs(v2 - vor).max() < tol_vd
    assert np.abs(d2 - div).max() < tol_vd


@pytest.mark.parametrize("grid", ["octahedral", "regular"])
def test_closed_form_winds_and_derivatives(grid):
    """The reference holds no golden vector for the wind / derivative operators (its own checks are norm round trips,
    tests/CMakeLists.txt:274-290).  Closed forms pin them to numbers no restatement shares: solid-body rotation
    (zeta = 2U/(a sqrt 3) P_1^0 => u = U cos(theta), v = 0), single harmonics of vorticity, divergence and a scalar with
    their analytic u, v, d/dlambda, d/dtheta fields -- vdtuv_mod.F90:97-143, spnsde_mod.F90:95-114, fsc_mod.F90:138-187 in
    INV_TRANS and uvtvd_mod.F90:91-139 in DIR_TRANS (observed 3e-14; 1e-12 asserted)."""
    nloen = octahedral(21) if grid == "octahedral" else np.full(44, 96, dtype=np.int32)
    o = Oracle(21, nloen)
    inv = lambda v, d, s: o.inv_trans(spvor=v, spdiv=d, spsc=s, scders=True, vorgp=True, divgp=True, uvder=True)
    dirt = lambda g: o.dir_trans(g, nuv=1, nsc=1)
    e_inv, e_dir = closed_form_errors(inv, dirt, 21, nloen, o.rmu, o.nasm0, o.nspec2)
    assert max(e_inv) < 1e-12 and max(e_dir) < 1e-12, (e_inv, e_dir)


def test_closed_form_vordiv_to_uv_and_gpnorm():
    """VORDIV_TO_UV (vd2uv_mod.F90:79-120) and GPNORM_TRANS (gpnorm_trans_ctl_mod.F90) of the oracle against closed forms: solid-body
    rotation gives U = u cos(theta) = U0 (1 - mu^2) = 2 U0 / 3 P_0 - 2 U0 / (3 sqrt 5) P_2 and V = 0 in the package's normalisation;
    a constant field has average = minimum = maximum; the zonal harmonic P_2^0 has average 0 (Gaussian quadrature is exact) and its
    maximum sqrt(5) P_2(mu) on the first latitude."""
    N = 21
    nloen = octahedral(N)
    o = Oracle(N, nloen)
    U, a = 30.0, 6371229.0
    vor, div = np.zeros((o.nspec2, 1)), np.zeros((o.nspec2, 1))
    vor[o.nasm0[0] - 1 + 2, 0] = 2 * U / (a * np.sqrt(3.0))
    u, v = o.vordiv_to_uv(vor, div)
    want = np.zeros_like(u)
    want[o.nasm0[0] - 1, 0], want[o.nasm0[0] - 1 + 4, 0] = 2 * U / 3, -2 * U / (3 * np.sqrt(5.0))
    assert np.abs(u - want).max() < 1e-13 * U and np.abs(v).max() < 1e-13 * U
    sc = np.zeros((o.nspec2, 2))
    sc[o.nasm0[0] - 1, 0], sc[o.nasm0[0] - 1 + 4, 1] = 3.5, 1.0
    ave, mn, mx = o.gpnorm(o.inv_trans(spsc=sc))
    assert abs(ave[0] - 3.5) < 1e-13 and abs(mn[0] - 3.5) < 1e-13 and abs(mx[0] - 3.5) < 1e-13
    assert abs(ave[1]) < 1e-13 and abs(mx[1] - np.sqrt(5.0) * (3 * o.rmu[0] ** 2 - 1) / 2) < 1e-12 and mn[1] < -1.0
