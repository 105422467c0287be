"""CPU-side debugging aid: the HIP kernels compiled for the thread-per-lane functional emulator
(tests/emu, see ectrans_amd/csrc/emi_rt.h) against the oracle on tiny cases.  This validates the
host logic (field tables, batching, index tables, FFT plans) and the kernel index arithmetic without
a GPU; the GPU parity tests proper are tests/test_gpu_parity.py."""
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import Oracle
from tests.common import adjoint_case, closed_form_errors, legpol_io_case, octahedral, run_case, utility_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-12  # fp64: observed ~1e-15


@pytest.fixture(scope="module")
def et():
    os.environ.setdefault("OMP_NUM_THREADS", "256")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    import ectrans_amd
    ectrans_amd._use_library_for_tests(os.path.join(ROOT, "tests", "emu", "libectrans_mi_emu.so"))
    ectrans_amd.setup_trans0(kmax_resol=4)
    yield ectrans_amd
    ectrans_amd.trans_end()
    ectrans_amd._L = None


XP = (lambda a: a, lambda a: a)
H = 9
SMOOTH = [20 + 4 * i for i in range(H)]
CASES = {
    "octahedral_winds": (8, SMOOTH + SMOOTH[::-1], 1, 2, {}, None),
    "bluestein_even": (8, [22, 26, 28, 30, 34, 38, 46, 58, 62] + [22, 26, 28, 30, 34, 38, 46, 58, 62][::-1], 1, 1, {}, None),
    "odd_lengths": (8, [19, 21, 23, 25, 27, 29, 33, 35, 37] + [19, 21, 23, 25, 27, 29, 33, 35, 37][::-1], 0, 2, {}, None),
    "derivatives": (8, SMOOTH + SMOOTH[::-1], 1, 1, dict(scders=True, vorgp=True, divgp=True, uvder=True), None),
    "nproma_blocks": (8, SMOOTH + SMOOTH[::-1], 1, 1, dict(scders=True), 37),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_emulated_kernels_match_oracle(et, name):
    nsmax, nloen, nuv, nsc, flags, nproma = CASES[name]
    e_inv, e_dir = run_case(et, Oracle, XP, nsmax, nloen, nuv, nsc, flags, nproma)
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


@pytest.mark.skipif(not os.environ.get("EMI_EMU_LARGE"), reason="six minutes in the emulator: EMI_EMU_LARGE=1 (the GPU tier covers these tiles at TCo399 / TCo1279)")
def test_emulated_legendre_tiles_of_a_long_wavenumber(et):
    """N = 260: the low wavenumbers have more than 128 (n - m) pairs per parity, so k_leg_dir runs its full one-parity tiles of 128
    pairs, the partial ones and the two-parity tile of the last <= 64 pairs (the cases above only reach the latter)."""
    n, h = 260, 6
    nl = [min(20 + 4 * i, 2 * n + 4) for i in range(h)]
    e_inv, e_dir = run_case(et, Oracle, XP, n, nl + nl[::-1], 0, 1, {}, None)
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


@pytest.mark.parametrize("name", ["octahedral_winds", "bluestein_even", "odd_lengths", "derivatives"])
def test_emulated_fp32_library_matches_oracle(et, name):
    """precision=4 (the reference's libtrans_sp arithmetic): same kernels instantiated for float with
    v_mfma_f32_16x16x4_f32's accumulator layout.  Tolerance: a few float epsilons x log-ish growth."""
    nsmax, nloen, nuv, nsc, flags, nproma = CASES[name]
    e_inv, e_dir = run_case(et, Oracle, XP, nsmax, nloen, nuv, nsc, flags, nproma, precision=4)
    assert e_inv < 2e-5 and e_dir < 2e-5, (e_inv, e_dir)
    assert e_inv > 1e-9  # really computed in float


HOT_A = [1028, 1284, 1540, 2052, 2564, 3076, 4100, 4612]  # first row length of every specialised Bluestein work length
HOT_B = [1276, 1532, 2044, 2556, 3068, 4092, 4604, 5116]  # ... and the last one


HOT_C = [5124, 6140, 6148, 7676, 7684, 8188, 8196, 10236]  # TCo2559 rows: work lengths 6144 ... 10240
HOT_D = [194, 258, 322, 386, 514, 578, 642, 770, 962, 1026]  # short rows, 2 ... 8 fields per workgroup: work lengths 256 ... 1280
HOT_E = [254, 318, 382, 510, 574, 638, 766, 958, 1022, 1278]  # ... and the last row length of each


@pytest.mark.parametrize("half,precision", [(HOT_A, 8), (HOT_B, 8), (HOT_C, 4), (HOT_D, 8)])  # (HOT_A, 4), (HOT_E, 8): GPU tier only
def test_specialised_fft_kernels_match_oracle(et, half, precision, monkeypatch):
    """k_fft_inv_hot / k_fft_dir_hot (work lengths 1280 ... 5120, the rows that carry TCo1279): a
    16-latitude grid whose rows select each of them, against the oracle."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    nuv, nsc = (1, 1) if half[0] > 1000 else (2, 7)  # short rows: 13 Fourier fields = ragged chunks of 2, 4 and 8
    e_inv, e_dir = run_case(et, Oracle, XP, 15, half + half[::-1], nuv, nsc, dict(scders=True), None, precision=precision)
    tol = TOL if precision == 8 else 2e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


R16_ROWS = [1540, 2044, 2052, 2556, 2564, 3068, 3076, 4092]  # first and last row length of the work lengths 256 R1, R1 = 8, 10, 12, 16


# k_fft_*_mr (direct mixed radix): row lengths 2 A B C that exercise every radix of EMI_MR_RADICES, one-, two- and three-pass plans,
# one and several fields per workgroup
MR_SHORT = [20, 24, 28, 32, 34, 38, 44, 46, 52, 68, 92, 286, 646, 480]            # 10, 12, 14, 16, 17, 19, 2*11, 23, 2*13, 2*17, 2*23, 11*13, 17*19, 6*8*5
MR_MID = [512, 1058, 1890, 1430, 1938, 2244, 1716, 1292]                          # 16*16, 23*23, 9*15*7, 10*11*13, 3*17*19, 6*11*17, 6*11*13, 2*17*19
MR_LONG = [4004, 4096, 5060, 5120, 4522, 4394, 4800, 4862, 2916, 4000, 4116]      # 14*11*13, 16*16*8, 10*11*23, 16*16*10, 7*17*19, 13^3, 16*15*10, 11*13*17, 18*9*9, 20*10*10, 21*14*7


@pytest.mark.parametrize("rows,precision,nproma", [(MR_SHORT, 8, None), (MR_MID, 8, None), (MR_LONG, 8, None), (MR_LONG, 4, None), (MR_SHORT, 8, 37)])
def test_direct_mixed_radix_fft_kernels_match_oracle(et, rows, precision, nproma):
    """k_fft_dir_mr / k_fft_inv_mr against the oracle: winds, scalars and derivatives (13 Fourier fields for the short rows: ragged
    field chunks), with and without NPROMA blocks that cut the rows."""
    nuv, nsc = (1, 1) if rows[0] > 1000 else (2, 7)
    e_inv, e_dir = run_case(et, Oracle, XP, 15, rows + rows[::-1], nuv, nsc, dict(scders=True, uvder=True), nproma, precision=precision)
    tol = TOL if precision == 8 else 2e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_direct_mixed_radix_fft_kernels_adjoints(et):
    """the `adj` scalings of k_fft_*_mr: dot-product identity of INV_TRANSAD / DIR_TRANSAD"""
    e_inv, e_dir = adjoint_case(et, XP, 15, MR_MID + MR_MID[::-1], 1, 1, nproma=3000)
    assert e_inv < 1e-12 and e_dir < 1e-12, (e_inv, e_dir)


@pytest.mark.parametrize("precision", [8])  # fp32: GPU tier
def test_register_resident_fft_kernels_blocked_rows(et, precision, monkeypatch):
    """k_fft_dir_r16 / k_fft_inv_r16 with NPROMA blocks that cut the rows (the element-wise grid path; the unblocked
    specialised-kernel cases above take the row-as-one-buffer path), winds and derivatives."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    e_inv, e_dir = run_case(et, Oracle, XP, 15, R16_ROWS + R16_ROWS[::-1], 1, 1, dict(scders=True, uvder=True), 1000, precision=precision)
    tol = TOL if precision == 8 else 2e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_register_resident_fft_kernels_adjoints(et, monkeypatch):
    """the `adj` scalings of the register-resident kernels: dot-product identity of INV_TRANSAD / DIR_TRANSAD"""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    e_inv, e_dir = adjoint_case(et, XP, 15, R16_ROWS + R16_ROWS[::-1], 1, 1, nproma=3000)
    assert e_inv < 1e-12 and e_dir < 1e-12, (e_inv, e_dir)


R16S_ROWS = [4100, 5116, 5124]  # k_fft_*_r16p<10> (first, last but one row length), <12>; the GPU tier runs the full list and R1 = 16


@pytest.mark.parametrize("nproma", [None, 1000])
def test_split_register_resident_fft_kernels(et, nproma, monkeypatch):
    """k_fft_dir_r16p / k_fft_inv_r16p (round 6): rows of more than 4096 points as two register-resident convolutions of a quarter of the
    row length (one after the other, the idle half parked in LDS) joined by one decimation step; whole rows (32 contiguous bytes per
    lane) and NPROMA blocks that cut them (element path)."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # keep rows with a 23-smooth half-length off the direct mixed-radix kernels
    e_inv, e_dir = run_case(et, Oracle, XP, 15, R16S_ROWS + R16S_ROWS[::-1], 1, 1, dict(scders=True, uvder=True), nproma)
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


def test_split_register_resident_fft_kernels_adjoints(et, monkeypatch):
    """the `adj` scalings of the split kernels: dot-product identity of INV_TRANSAD / DIR_TRANSAD"""
    monkeypatch.setenv("EMI_FFT_MR", "0")
    e_inv, e_dir = adjoint_case(et, XP, 15, R16S_ROWS[:2] + R16S_ROWS[1::-1], 1, 1, nproma=3000)
    assert e_inv < 1e-12 and e_dir < 1e-12, (e_inv, e_dir)


@pytest.mark.parametrize("env", [("EMI_FFT_NO_HOT", "1"), ("EMI_FFT_MR", "0")])
def test_ab_switches_keep_parity(et, monkeypatch, env):
    """The two switches the tests use to send the same rows through another kernel family: the generic FFT kernels instead of the
    specialised ones, the convolution kernels instead of the direct mixed-radix ones."""
    monkeypatch.setenv(*env)
    n, nloen, nuv, nsc, flags, nproma = CASES["nproma_blocks"]
    e_inv, e_dir = run_case(et, Oracle, XP, n, nloen, nuv, nsc, flags, nproma)
    assert e_inv < TOL and e_dir < TOL, (env, e_inv, e_dir)


def test_dist_and_gath_routines_single_task(et):
    """DIST_SPEC/GATH_SPEC/DIST_GRID/GATH_GRID with one task are pure re-layouts."""
    N = 10
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        rng = np.random.default_rng(3)
        g = rng.standard_normal((ns2, 4))
        loc = et.dist_spec(r, g, 4)
        assert np.array_equal(loc, g) and np.array_equal(et.gath_spec(r, loc, 4), g)
        gg = rng.standard_normal((4, ng))
        blk = et.dist_grid(r, gg, 4, kproma=100)
        assert blk.shape == ((ng - 1) // 100 + 1, 4, 100)
        assert np.array_equal(et.gath_grid(r, blk, 4), gg)
        with pytest.raises(et.TransError, match="task numbers"):
            et.dist_spec(r, g, 4, kfrom=2)
    finally:
        et.trans_release(r)


@pytest.mark.parametrize("nuv,nsc,nproma", [(0, 2, None), (2, 1, None), (1, 1, 37)])
def test_adjoint_transforms_dot_product(et, nuv, nsc, nproma):
    """INV_TRANSAD / DIR_TRANSAD: <A x, y> = <x, A* y> (tests/trans/test_invtrans_adjoint.F90,
    test_dirtrans_adjoint.F90; the reference accepts 2000 eps ... 20000 eps)."""
    e_inv, e_dir = adjoint_case(et, XP, 10, octahedral(10), nuv, nsc, nproma)
    assert e_inv < 1e-13 and e_dir < 1e-13, (e_inv, e_dir)


def test_adjoint_with_derivative_options_matches_transposed_oracle(et):
    """INV_TRANSAD with LDSCDERS / LDVORGP / LDDIVGP / LDUVDER (ltinvad_mod.F90:149-225: the adjoints of SPNSDE and of FSC's
    derivative outputs, spnsdead_mod.F90, fscad_mod.F90): against the weighted transpose of the oracle's forward INV_TRANS
    with the same options, formed column by column at T6/O7 (tests/common.py::adjoint_options_case)."""
    from tests.common import adjoint_options_case
    for flags in (dict(scders=True), dict(uvder=True), dict(vorgp=True), dict(divgp=True), dict(scders=True, vorgp=True, uvder=True)):
        e = adjoint_options_case(et, Oracle, XP, nsmax=6, flags=flags)
        assert e < 1e-12, (flags, e)


@pytest.mark.parametrize("seed", range(4))
def test_random_reduced_grids_match_oracle(et, seed):
    """Random reduced grids (row lengths of any parity and factorisation), truncations, field counts, options
    and NPROMA against the oracle (the GPU tier runs 40 of these)."""
    rng = np.random.default_rng(2000 + seed)
    nh = int(rng.integers(3, 7))
    half = np.sort(rng.integers(8, 120, nh))
    nloen = np.concatenate([half, half[::-1]]).astype(np.int32)
    nsmax = int(rng.integers(2, 2 * nh))
    nuv, nsc = int(rng.integers(0, 2)), int(rng.integers(1, 3))
    flags = dict(scders=bool(rng.integers(2)), uvder=bool(rng.integers(2)) and nuv > 0, vorgp=bool(rng.integers(2)) and nuv > 0)
    nproma = [None, 17, 100][int(rng.integers(3))]
    e_inv, e_dir = run_case(et, Oracle, XP, nsmax, nloen, nuv, nsc, flags, nproma, seed=seed)
    assert e_inv < TOL and e_dir < TOL, (nloen.tolist(), nsmax, nuv, nsc, flags, nproma, e_inv, e_dir)


def test_fp32_library_rejects_double_arrays(et):
    nloen = octahedral(7)
    r = et.setup_trans(7, len(nloen), nloen, precision=4)
    try:
        with pytest.raises(et.TransError, match="float32"):
            et.inv_trans(r, pspscalar=np.zeros((et.trans_inq(r, "nspec2"), 1)), pgp=np.zeros((1, 1, et.trans_inq(r, "ngptot"))))
    finally:
        et.trans_release(r)


def test_setup_tables_match_oracle(et):
    N = 15
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    o = Oracle(N, nloen)
    assert np.array_equal(et.trans_inq(r, "nmen"), o.nmen)
    assert np.array_equal(et.trans_inq(r, "ndglu"), o.ndglu)
    assert np.array_equal(et.trans_inq(r, "nasm0"), o.nasm0)
    assert np.abs(et.trans_inq(r, "rmu") - o.rmu).max() < 1e-15
    assert np.abs(et.trans_inq(r, "rgw") - o.rw).max() < 1e-16
    assert abs(et.trans_inq(r, "rgw").sum() - 1.0) < 1e-10  # test_ectrans4py.py:119-121
    for m in (0, 1, 2, 7, N):
        for sym in (False, True):
            assert np.abs(et.legendre_panel(r, m, sym) - o.rpnm(m, sym)).max(initial=0.0) < 1e-14
    assert (et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")) == (o.nspec2, o.ngptot)
    et.trans_release(r)


@pytest.mark.parametrize("precision", [8, 4])
def test_legendre_polynomial_file_io(et, tmp_path, precision):
    """CDIO_LEGPOL = writef / readf / membuf in the reference's file format (write_legpol_mod.F90)."""
    legpol_io_case(et, Oracle, XP, tmp_path, nsmax=15, precision=precision)


def test_device_legendre_setup_with_rescaling(et, monkeypatch):
    """k_legpol (SUPOLF on the device, m >= 2) on a full grid F64 / T127: every wavenumber at every latitude,
    so sin^m(theta) underflows 1e-100 near the poles and the 1e+-100 rescaling path runs (corr3 = 2
    at m = 127).  Panels against the oracle, and against the host path bit for bit."""
    N, ndgl = 127, 128
    r = et.setup_trans(N, ndgl, None, kdlon=256)
    o = Oracle(N, np.full(ndgl, 256, dtype=np.int32))
    monkeypatch.setenv("EMI_LEGPOL_HOST", "1")
    rh = et.setup_trans(N, ndgl, None, kdlon=256)
    try:
        for m in (2, 3, 40, 100, 126, 127):
            for sym in (False, True):
                dev = et.legendre_panel(r, m, sym)
                ref = o.rpnm(m, sym)
                assert dev.shape == ref.shape
                assert np.abs(dev - ref).max(initial=0.0) <= 1e-14 * max(1.0, np.abs(ref).max(initial=0.0))
                assert np.array_equal(dev, et.legendre_panel(rh, m, sym))
    finally:
        et.trans_release(r)
        et.trans_release(rh)


def test_call_mode_2_arrays_and_batches(et):
    """PGPUV/PGP3A/PGP2 + PSPSC3A/PSPSC2 (ectrans-benchmark call mode 2) and field batching."""
    N = 8
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    o = Oracle(N, nloen)
    rng = np.random.default_rng(3)
    from tests.common import random_spectrum
    nlev, nvar = 2, 2
    vor = random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, True)
    div = random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, True)
    sc3 = np.stack([random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, False) for _ in range(nvar)])
    sc2 = random_spectrum(rng, o.nasm0, N, o.nspec2, 1, False)
    # oracle sees the flat scalar list in the reference order: sc2, then sc3a (var outer, level inner)
    scflat = np.concatenate([sc2] + [sc3[v] for v in range(nvar)], axis=1)
    gref = o.inv_trans(spvor=vor, spdiv=div, spsc=scflat)
    ng = o.ngptot
    gpuv, gp3a, gp2 = np.zeros((1, 2, nlev, ng)), np.zeros((1, nvar, nlev, ng)), np.zeros((1, 1, ng))
    et.set_max_batch(0)
    et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    got = np.concatenate([gpuv[0].reshape(2 * nlev, ng), gp2[0], gp3a[0].reshape(nvar * nlev, ng)])
    assert (np.abs(got - gref).max(axis=1) / np.abs(gref).max(axis=1)).max() < TOL
    v2, d2, s3, s2 = np.zeros_like(vor), np.zeros_like(div), np.zeros_like(sc3), np.zeros_like(sc2)
    et.dir_trans(r, pspvor=v2, pspdiv=d2, pspsc3a=s3, pspsc2=s2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    vr, dr, sr = o.dir_trans(gref, nuv=nlev, nsc=1 + nvar * nlev)
    got_sc = np.concatenate([s2] + [s3[v] for v in range(nvar)], axis=1)
    for a, b in ((v2, vr), (d2, dr), (got_sc, sr)):
        assert np.abs(a - b).max() / np.abs(b).max() < TOL
    et.trans_release(r)


def test_host_output_arrays_keep_unwritten_elements(et):
    """Host arrays are staged through device memory: what INV_TRANS does not write -- the padding of the last
    NPROMA block, fields of PGP beyond those the call produces -- must come back as the caller left it (the
    reference never touches those elements)."""
    N = 8
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    npr = 37
    nb = (ng - 1) // npr + 1
    assert nb * npr > ng
    rng = np.random.default_rng(11)
    sp = rng.uniform(-1, 1, (ns2, 2))
    junk = [np.full(50000, np.nan) for _ in range(4)]  # poison the heap the staging buffers come from
    del junk
    gp = np.full((nb, 3, npr), -7.25)  # one field more than the call produces
    et.inv_trans(r, pspscalar=sp, pgp=gp, kproma=npr)
    tail = ng - (nb - 1) * npr
    assert np.all(gp[-1, :, tail:] == -7.25) and np.all(gp[:, 2, :] == -7.25)
    assert np.all(np.isfinite(gp)) and np.all(gp[:-1, :2, :] != -7.25)
    et.trans_release(r)


def test_host_staging_buffers_are_reused_between_calls(et):
    from tests.common import staging_pool_case
    staging_pool_case(et, Oracle, TOL, combos=((3, None), (1, 37), (5, None)))


def test_argument_errors_mirror_abort_trans(et):
    N = 8
    nloen = octahedral(N)
    with pytest.raises(et.TransError, match="KDGL IS NOT A POSITIVE, EVEN NUMBER"):
        et.setup_trans(N, 17, nloen[:17])
    with pytest.raises(et.TransError, match="KLOEN INVALID"):
        bad = nloen.copy()
        bad[3] = 0
        et.setup_trans(N, len(bad), bad)
    with pytest.raises(et.TransError, match="LDUSEFLT"):
        et.setup_trans(N, len(nloen), nloen, lduseflt=True)
    r = et.setup_trans(N, len(nloen), nloen)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    with pytest.raises(et.TransError, match="SECOND DIMENSION OF PGP TOO SMALL"):
        et.inv_trans(r, pspscalar=np.zeros((ns2, 3)), pgp=np.zeros((1, 2, ng)))
    with pytest.raises(et.TransError, match="PGP AND PGPUV"):
        et.inv_trans(r, pspscalar=np.zeros((ns2, 1)), pgp=np.zeros((1, 1, ng)), pgpuv=np.zeros((1, 1, 1, ng)))
    with pytest.raises(et.TransError, match="BOTH PRESENT"):
        et.inv_trans(r, pspscalar=np.zeros((ns2, 1)), pspsc2=np.zeros((ns2, 1)), pgp=np.zeros((1, 2, ng)))
    with pytest.raises(et.TransError, match="unknown resolution"):
        et.inv_trans(r + 1, pspscalar=np.zeros((ns2, 1)), pgp=np.zeros((1, 1, ng)))
    et.trans_release(r)
    with pytest.raises(et.TransError, match="unknown resolution"):
        et.trans_inq(r, "nspec2")


def test_array_extent_checks_mirror_abort_trans(et):
    """The extent checks of inv_trans.F90:476-600 / dir_trans.F90:370-491 on the shapes the caller really passes
    (emi_extents_t): before this the library derived the middle extents from the flags and a PGPUV with too few
    variables or a PGP2 of the wrong width was overrun silently."""
    N = 8
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    z = np.zeros
    uv = dict(pspvor=z((ns2, 2)), pspdiv=z((ns2, 2)))
    try:
        with pytest.raises(et.TransError, match="THIRD DIMENSION OF PGPUV TOO SMALL"):  # LDVORGP needs 4 variables
            et.inv_trans(r, pgpuv=z((1, 2, 2, ng)), ldvorgp=True, **uv)
        with pytest.raises(et.TransError, match="SEC. DIMENSION OF PGPUV INCONSISTENT"):
            et.inv_trans(r, pgpuv=z((1, 2, 3, ng)), **uv)
        with pytest.raises(et.TransError, match="FOURTH DIMENSION OF PGPUV TOO SMALL"):
            et.inv_trans(r, pgpuv=z((1, 2, 2, 100)), kproma=100, **uv)
        with pytest.raises(et.TransError, match="FIRST DIMENSION OF PGPUV TOO SMALL"):
            et.inv_trans(r, pgpuv=z((1, 2, 2, ng - 1)), **uv)
        with pytest.raises(et.TransError, match="SEC. DIMENSION OF PGP2 INCONSISTENT"):  # LDSCDERS triples IF_SC2_G
            et.inv_trans(r, pspsc2=z((ns2, 2)), pgp2=z((1, 2, ng)), ldscders=True)
        with pytest.raises(et.TransError, match="THIRD DIMENSION OF PGP3A INCONSISTENT"):
            et.inv_trans(r, pspsc3a=z((2, ns2, 3)), pgp3a=z((1, 3, 3, ng)))
        with pytest.raises(et.TransError, match="SEC. DIMENSION OF PGP3A INCONSISTENT"):
            et.inv_trans(r, pspsc3a=z((2, ns2, 3)), pgp3a=z((1, 2, 4, ng)))
        with pytest.raises(et.TransError, match="SPECTRAL ARRAY TOO SMALL|nspec2"):
            et.inv_trans(r, pspscalar=z((ns2 - 2, 1)), pgp=z((1, 1, ng)))
        with pytest.raises(et.TransError, match="DIR_TRANS:SEC. DIMENSION OF PGPUV INCONSISTENT"):
            et.dir_trans(r, pgpuv=z((1, 2, 3, ng)), **uv)
        with pytest.raises(et.TransError, match="DIR_TRANS:THIRD DIMENSION OF PGP TOO SMALL"):
            et.dir_trans(r, pspscalar=z((ns2, 1)), pgp=z((1, 1, 50)), kproma=50)
        # a PGPUV with MORE variables than the call produces is legal (inv_trans.F90:497: only '<' aborts) and is
        # addressed with its real extent: u, v land in variables 0, 1 of every block, the rest stays untouched
        o = Oracle(N, nloen)
        rng = np.random.default_rng(3)
        from tests.common import random_spectrum, rel_err
        vor, div = (random_spectrum(rng, o.nasm0, N, ns2, 2, True) for _ in range(2))
        npr = 100
        nb = (ng - 1) // npr + 1
        gpuv = np.full((nb, 4, 2, npr), -3.5)
        et.inv_trans(r, pspvor=vor, pspdiv=div, pgpuv=gpuv, kproma=npr)
        gref = o.inv_trans(spvor=vor, spdiv=div)
        got = np.concatenate([gpuv[b, :2].reshape(4, npr) for b in range(nb)], axis=1)[:, :ng]
        assert rel_err(got, gref, axis=1) < TOL and np.all(gpuv[:, 2:] == -3.5)
    finally:
        et.trans_release(r)


def test_fp32_library_mean_wavenumber_in_double(et):
    """The reference's single-precision library computes the Legendre transform of zonal wavenumber 0 in double ("DGEM for
    the mean to improve mass conservation", ledir_mod.F90:133-171).  The fp32 library does the same on the fp64 matrix
    cores (LegAcc<true>); against the oracle's sp mode (float operands, double accumulation for m = 0) the m = 0 column
    agrees to half a float ulp of its maximum and the global mean (0, 0) to one ulp."""
    N = 21
    nloen = octahedral(N)
    o = Oracle(N, nloen)
    rng = np.random.default_rng(1)
    from tests.common import random_spectrum
    s = random_spectrum(rng, o.nasm0, N, o.nspec2, 3, False)
    s[0, :] = 250.0  # a 250 K-like field
    g = o.inv_trans(spsc=s).astype(np.float32)
    r = et.setup_trans(N, len(nloen), nloen, precision=4)
    try:
        out = np.zeros((o.nspec2, 3), dtype=np.float32)
        et.dir_trans(r, pspscalar=out, pgp=g.reshape(1, 3, -1).copy())
        o.set_sp_mode(True)
        _, _, ref = o.dir_trans(g.astype(np.float64), nsc=3)
        m0 = slice(0, 2 * (N + 1), 2)
        ulp = float(np.finfo(np.float32).eps)
        assert np.abs(out[m0] - ref[m0]).max() <= 0.5 * ulp * np.abs(ref[m0]).max()
        assert np.abs(out[0] / ref[0] - 1.0).max() <= ulp
    finally:
        et.trans_release(r)


def test_adjoints_match_transposed_oracle_matrices(et):
    """INV_TRANSAD / DIR_TRANSAD against the explicit weighted transposes of the ORACLE's forward operators at T8/O9
    (tests/common.py::adjoint_matrix_case) -- an adjoint oracle that does not involve the HIP forward transform."""
    from tests.common import adjoint_matrix_case
    e_inv, e_dir = adjoint_matrix_case(et, Oracle, XP, nsmax=8)
    assert e_inv < 1e-12 and e_dir < 1e-12, (e_inv, e_dir)


def test_rows_longer_than_the_lds(et):
    """Rows whose FFT work array exceeds the 160 KiB of LDS (fp64: more than 10240 complex points) run the same passes on a
    global scratch buffer (k_fft_*_gm) -- the reference takes any KLOEN (ftdir_mod.F90:67-84).  Row lengths 20484 ... 20500
    (Bluestein work length 24576 = 384 KiB), mixed with an ordinary row, NPROMA blocks."""
    half = [20484, 20492, 300, 20500]
    nloen = np.array(half + half[::-1], dtype=np.int32)
    r = et.setup_trans(5, len(nloen), nloen)
    assert list(et.trans_inq(r, "fftwork")[:4]) == [24576, 24576, 150, 24576]
    et.trans_release(r)
    e_inv, e_dir = run_case(et, Oracle, XP, 5, nloen, 1, 1, dict(scders=True), 3000)
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


def test_belousov_generator_lduserpnm(et):
    """SETUP_TRANS(LDUSERPNM=.TRUE.), the default of the Fortran API: Belousov's generator SUPOL (supol_mod.F90:86-167,
    tpm_pol.F90:31-99) instead of the per-wavenumber SUPOLF recurrence.  Panels against the oracle's restatement of SUPOL
    (1e-14) -- and measurably different from the SUPOLF panels (1e-13 ... 1e-12), so the switch is real -- and a
    transform through them."""
    N = 47
    nloen = octahedral(N)
    ob, of = Oracle(N, nloen, belusov=True), Oracle(N, nloen)
    r = et.setup_trans(N, len(nloen), nloen, lduserpnm=True)
    try:
        e_b = e_f = 0.0
        for m in range(N + 1):
            for sym in (False, True):
                a, b, c = et.legendre_panel(r, m, sym), ob.rpnm(m, sym), of.rpnm(m, sym)
                if a.size:
                    e_b, e_f = max(e_b, np.abs(a - b).max()), max(e_f, np.abs(a - c).max())
        assert e_b <= 1e-14 and 1e-14 < e_f < 1e-11, (e_b, e_f)
    finally:
        et.trans_release(r)
    N = 15  # the transform through Belousov panels on a smaller grid: the emulator is slow
    e_inv, e_dir = run_case(et, lambda *a, **k: Oracle(*a, belusov=True, **k), XP, N, octahedral(N), 1, 1, setup_kw=dict(lduserpnm=True))
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


def test_adjoint_options_through_call_mode2_arrays(et):
    from tests.common import adjoint_options_call_mode2_case
    assert adjoint_options_call_mode2_case(et, XP) < 1e-14


def test_closed_form_winds_and_derivatives(et):
    """the kernels against analytic fields (tests/common.py::closed_form_case)"""
    to, back = XP
    nloen = octahedral(21)
    r = et.setup_trans(21, len(nloen), nloen)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")

        def inv(v, d, s):
            gp = to(np.zeros((1, 9, ng)))
            et.inv_trans(r, pspvor=to(v), pspdiv=to(d), pspscalar=to(s), pgp=gp, ldscders=True, ldvorgp=True, lddivgp=True, lduvder=True)
            return back(gp)[0]

        def dirt(g):
            v2, d2, s2 = (to(np.zeros((ns2, 1))) for _ in range(3))
            et.dir_trans(r, pspvor=v2, pspdiv=d2, pspscalar=s2, pgp=to(np.ascontiguousarray(g[None])))
            return back(v2), back(d2), back(s2)

        e_inv, e_dir = closed_form_errors(inv, dirt, 21, nloen, et.trans_inq(r, "rmu"), et.trans_inq(r, "nasm0"), ns2)
    finally:
        et.trans_release(r)
    assert max(e_inv) < 1e-12 and max(e_dir) < 1e-12, (e_inv, e_dir)


def test_inquiry_of_the_initialisation(et):
    """emi_inq_init: KMAX_RESOL and PRAD of SETUP_TRANS0 -- SETUP_TRANS0 / trans_init of a host whose transport initialised
    the library first abort on a different radius instead of computing with the transport's."""
    kmax, ra = et.inq_init()
    assert kmax >= 1 and ra == 6371229.0  # setup_trans0.F90:129 default


@pytest.mark.parametrize("precision", [8, 4])
def test_utility_routines_vordiv_to_uv_and_gpnorm(et, precision):
    """VORDIV_TO_UV (k_vd2uv) and GPNORM_TRANS (k_gpnorm) against the oracle and closed forms (tests/common.py::utility_case)"""
    e_uv, e_sb, e_gp = utility_case(et, Oracle, XP, 15, precision, 37)
    tol = 1e-12 if precision == 8 else 3e-6
    assert e_uv < tol and e_sb < tol and e_gp < tol, (e_uv, e_sb, e_gp)
