"""GPU parity tests proper: HIP path through the C-ABI vs the CPU oracle / golden fixtures.

Tolerances: the north-star asks for <= 1e-10 relative on spectral norms; the fp64 kernels are
observed at 1e-15..1e-13, tests assert 1e-11 (relative to the field maximum) unless noted.
"""
import os

import numpy as np
import pytest

from tests.common import adjoint_case, block, closed_form_errors, legpol_io_case, octahedral, random_spectrum, rel_err, run_case, unblock

pytestmark = pytest.mark.gpu
TOL = 1e-11
EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def et():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import ectrans_amd
    ectrans_amd.lib()  # fails loudly if the HIP library is missing
    ectrans_amd.setup_trans0(kmax_resol=6, device=0)
    yield ectrans_amd
    ectrans_amd.trans_end()


@pytest.fixture(scope="module")
def dev():
    import torch
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    back = lambda t: t.cpu().numpy()
    return to, back


HOST = (lambda a: np.ascontiguousarray(a), lambda a: a)


def Oracle(*a, **k):
    from oracle.oracle import Oracle as O
    return O(*a, **k)


def test_tl149_golden_vectors(et, dev, golden_dir):
    """The reference's own known-answer test (test_ectrans4py.py:133-158): 1e-10 absolute."""
    to, back = dev
    d = os.path.join(golden_dir, "tl149")
    nloen = np.load(os.path.join(d, "lon_number_by_lat.npy")).astype(np.int32)
    zw = np.load(os.path.join(d, "zonal_wavenumbers.npy"))
    sp = np.load(os.path.join(d, "tl149-c24-s1t@sp.npy"))
    gpll = np.load(os.path.join(d, "tl149-c24-s1t@sp2gp.npy"))
    gp_ref = np.concatenate([gpll[i, : nloen[i]] for i in range(nloen.size)])
    r = et.setup_trans(148, 150, nloen)
    assert (et.trans_inq(r, "ngptot"), et.trans_inq(r, "nspec2") // 2) == (33052, 11175)
    np.testing.assert_array_equal(et.trans_inq(r, "nmen"), zw)
    assert abs(et.trans_inq(r, "rgw").sum() - 1.0) < 1e-10
    gp = to(np.zeros((1, 1, gp_ref.size)))
    et.inv_trans(r, pspscalar=to(sp.reshape(-1, 1)), pgp=gp)
    assert np.abs(back(gp)[0, 0] - gp_ref).max() < 1e-10
    s2 = to(np.zeros((sp.size, 1)))
    et.dir_trans(r, pspscalar=s2, pgp=to(gp_ref.reshape(1, 1, -1)))
    assert np.abs(back(s2)[:, 0] - sp).max() < 1e-10
    et.trans_release(r)


H9 = [20 + 4 * i for i in range(9)]
BLUE = [22, 26, 28, 30, 34, 38, 46, 58, 62]
ODD = [19, 21, 23, 25, 27, 29, 33, 35, 37]
CASES = {
    "O22_winds": (21, octahedral(21), 2, 3, {}, None),
    "O64_winds": (63, octahedral(63), 3, 5, {}, None),
    "O160_winds": (159, octahedral(159), 2, 3, {}, None),
    "full_grid_F64": (63, np.full(128, 256), 1, 2, {}, None),
    "linear_grid_T47_regular": (47, np.full(48, 96), 1, 1, {}, None),
    "bluestein_even": (8, BLUE + BLUE[::-1], 1, 1, {}, None),
    "odd_lengths": (8, ODD + ODD[::-1], 1, 2, {}, None),
    "derivatives": (31, octahedral(31), 2, 2, dict(scders=True, vorgp=True, divgp=True, uvder=True), None),
    "nproma_blocks": (31, octahedral(31), 1, 2, dict(scders=True), 1000),
    "scalars_only": (31, octahedral(31), 0, 4, {}, None),
    "winds_only": (31, octahedral(31), 3, 0, dict(uvder=True), None),
    "many_fields_two_tiles": (21, octahedral(21), 40, 70, {}, None),
    # truncation finer than the grid: the wavenumbers 52 ... 63 lie above every latitude's NMEN -- no latitudes, coefficients zero (the fp32
    # k_leg_dir used row numbers nobody had staged for them until round 5: memory fault)
    "truncation_above_grid": (63, octahedral(21), 2, 3, {}, None),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_arrays_match_oracle(et, dev, name):
    nsmax, nloen, nuv, nsc, flags, nproma = CASES[name]
    from oracle.oracle import Oracle as O
    e_inv, e_dir = run_case(et, O, dev, nsmax, nloen, nuv, nsc, flags, nproma)
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


# n-pairs of the mean wavenumber (N + 2) // 2 = 64, 65, 96, 127, 128, 129, 192, 193, 256: every rest class of k_leg_dir's row tiling at
# the head of the wavenumber list, and every class below it as m rises
TILE_EDGE_N = [126, 128, 190, 252, 254, 256, 382, 384, 510]


@pytest.mark.parametrize("nsmax", TILE_EDGE_N)
@pytest.mark.parametrize("precision", [8, 4])
def test_direct_legendre_row_tiles(et, dev, nsmax, precision):
    """k_leg_dir's row tiles (emi_kernels_body.h): per wavenumber 2 floor(nk / 128) one-parity tiles of 128 (n - m) pairs, then for the rest
    r = nk mod 128 nothing, ONE two-parity tile of 64 pairs (r <= 64) or two partial one-parity tiles (fp32: two-parity tiles
    throughout).  A few latitudes per hemisphere keep the oracle cheap; the truncation makes the Legendre side long."""
    nh = 6
    half = np.array([min(20 + 4 * i, 2 * nsmax + 4) for i in range(nh)], dtype=np.int32)
    nloen = np.concatenate([half, half[::-1]])
    from oracle.oracle import Oracle as O
    e_inv, e_dir = run_case(et, O, dev, nsmax, nloen, 1, 2, {}, None, precision=precision)
    tol = TOL if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (nsmax, e_inv, e_dir)


FP32_CASES = ["O64_winds", "O160_winds", "full_grid_F64", "bluestein_even", "odd_lengths", "derivatives", "nproma_blocks",
              "many_fields_two_tiles", "truncation_above_grid"]


@pytest.mark.parametrize("name", FP32_CASES)
def test_fp32_library_matches_oracle(et, dev, name):
    """precision=4: the reference's single-precision library (libtrans_sp, JPRB=JPRM; BASELINE's
    AIFS configuration computes in it).  Same kernels instantiated for float / v_mfma_f32_16x16x4_f32;
    checked against the fp64 oracle on float32-rounded inputs.  Tolerance 3e-5 of the field maximum
    (= 250 float epsilons; observed <= 3e-6)."""
    nsmax, nloen, nuv, nsc, flags, nproma = CASES[name]
    from oracle.oracle import Oracle as O
    e_inv, e_dir = run_case(et, O, dev, nsmax, nloen, nuv, nsc, flags, nproma, precision=4)
    assert e_inv < 3e-5 and e_dir < 3e-5, (e_inv, e_dir)


HOT_A = [1028, 1284, 1540, 2052, 2564, 3076, 4100, 4612]  # first row length of every specialised Bluestein work length
HOT_B = [1276, 1532, 2044, 2556, 3068, 4092, 4604, 5116]  # ... and the last one


HOT_C = [5124, 6140, 6148, 7676, 7684, 8188, 8196, 10236]  # TCo2559 rows: work lengths 6144 ... 10240
HOT_D = [194, 258, 322, 386, 514, 578, 642, 770, 962, 1026]  # short rows, 2 ... 8 fields per workgroup: work lengths 256 ... 1280
HOT_E = [254, 318, 382, 510, 574, 638, 766, 958, 1022, 1278]  # ... and the last row length of each


@pytest.mark.parametrize("half,precision", [(HOT_A, 8), (HOT_B, 8), (HOT_A, 4), (HOT_B, 4), (HOT_C, 4), (HOT_C, 8), (HOT_D, 8), (HOT_E, 8),
                                            (HOT_D, 4)])
def test_specialised_fft_kernels_match_oracle(et, dev, half, precision, monkeypatch):
    """k_fft_inv_hot / k_fft_dir_hot (the Bluestein work lengths 1280 ... 5120 that carry TCo1279): a
    16-latitude grid whose rows select each of them, against the oracle."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    monkeypatch.setenv("EMI_FFT_R16S", "0")  # ... and the rows of more than 4096 points off the split kernels (round 6): the in-place kernels of the work
    # lengths 4608 ... 8192 still carry every such row whose half-length is odd (NLOEN = 2 mod 4: not on an octahedral grid, but any caller grid)
    from oracle.oracle import Oracle as O
    nsc = 3 if half[0] > 1000 else 9  # short rows: 13 Fourier fields + derivatives = ragged chunks of 2, 4 and 8 fields
    e_inv, e_dir = run_case(et, O, dev, 15, half + half[::-1], 2, nsc, dict(scders=True, uvder=True), None, precision=precision)
    tol = TOL if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_specialised_and_generic_fft_kernels_agree(et, dev, monkeypatch):
    """The specialised kernels run the same passes in the same order; only the compiler's fma
    contraction may differ: agreement to a few ulp."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    monkeypatch.setenv("EMI_FFT_R16S", "0")  # (rows of more than 4096 points: the in-place kernels, not the split ones)
    to, back = dev
    nloen = np.array(HOT_A + HOT_A[::-1], dtype=np.int32)
    rng = np.random.default_rng(5)
    outs = []
    for no_hot in (False, True):
        if no_hot:
            monkeypatch.setenv("EMI_FFT_NO_HOT", "1")
        r = et.setup_trans(15, len(nloen), nloen)
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        if not outs:
            sp = random_spectrum(rng, et.trans_inq(r, "nasm0"), 15, ns2, 3, False)
        gp = to(np.zeros((1, 3, ng)))
        et.inv_trans(r, pspscalar=to(sp), pgp=gp)
        s2 = to(np.zeros((ns2, 3)))
        et.dir_trans(r, pspscalar=s2, pgp=gp)
        outs.append((back(gp).copy(), back(s2).copy()))
        et.trans_release(r)
    assert rel_err(outs[0][0], outs[1][0]) < 1e-14 and rel_err(outs[0][1], outs[1][1]) < 1e-14
    assert not np.array_equal(outs[0][0], np.zeros_like(outs[0][0]))


# k_fft_*_mr (direct mixed radix, round 3): row lengths 2 A B C that exercise every radix of EMI_MR_RADICES, one-, two- and three-pass
# plans, one and several fields per workgroup
MR_SHORT = [20, 24, 28, 32, 34, 38, 44, 46, 52, 68, 92, 286, 646, 480]            # 10, 12, 14, 16, 17, 19, 2*11, 23, 2*13, 2*17, 2*23, 11*13, 17*19, 6*8*5
MR_MID = [512, 1058, 1890, 1430, 1938, 2244, 1716, 1292]                          # 16*16, 23*23, 9*15*7, 10*11*13, 3*17*19, 6*11*17, 6*11*13, 2*17*19
MR_LONG = [4004, 4096, 5060, 5120, 4522, 4394, 4800, 4862, 2916, 4000, 4116]      # 14*11*13, 16*16*8, 10*11*23, 16*16*10, 7*17*19, 13^3, 16*15*10, 11*13*17, 18*9*9, 20*10*10, 21*14*7
MR_XL = [8192, 8398, 9728, 9826, 10336, 9120, 10488, 9568, 9216, 10240, 9408, 8704]    # TCo2559-sized rows: 16^3, 13*17*19, 16*16*19, 17^3, 16*17*19, 15*16*19, 12*19*23, 16*13*23, 16*16*18, 16*16*20, 14*16*21, 16*16*17 (fp64: more than 64 KiB of LDS per row, set_lds_attrs)


@pytest.mark.parametrize("rows,precision,nproma", [(MR_SHORT, 8, None), (MR_MID, 8, None), (MR_LONG, 8, None), (MR_SHORT, 4, None), (MR_MID, 4, None),
                                                   (MR_LONG, 4, None), (MR_SHORT, 8, 37), (MR_MID, 8, 1000), (MR_LONG, 8, 4094), (MR_LONG, 4, 1000), (MR_XL, 8, None), (MR_XL, 4, None)])
def test_direct_mixed_radix_fft_kernels_match_oracle(et, dev, rows, precision, nproma):
    """k_fft_dir_mr / k_fft_inv_mr (rows whose half-length is a product of at most three radices from 2..16, 17, 19, 23: butterflies
    in registers, no chirp-z convolution) against the oracle: winds, scalars and all derivatives (ragged field chunks for the
    short rows), both precisions, rows inside one NPROMA block (grid rows read and written by the first / last pass) and
    rows cut by NPROMA blocks (copied through the LDS)."""
    from oracle.oracle import Oracle as O
    nsc = 3 if rows[0] > 1000 else 9
    e_inv, e_dir = run_case(et, O, dev, 15, rows + rows[::-1], 2, nsc, dict(scders=True, uvder=True, vorgp=True, divgp=True), nproma, precision=precision)
    tol = TOL if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_direct_mixed_radix_and_convolution_fft_kernels_agree(et, dev, monkeypatch):
    """EMI_FFT_MR=0 sends the same rows through the chirp-z kernels (or, the short 7-smooth ones, the generic mixed-radix
    kernels): both are the same transform to rounding, and really two different kernels."""
    to, back = dev
    rows = MR_SHORT[-3:] + MR_MID[:5] + MR_LONG
    nloen = np.array(rows + rows[::-1], dtype=np.int32)
    rng = np.random.default_rng(5)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("EMI_FFT_MR", "0")
        r = et.setup_trans(15, len(nloen), nloen)
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        if not outs:
            sp = random_spectrum(rng, et.trans_inq(r, "nasm0"), 15, ns2, 3, False)
        gp = to(np.zeros((1, 3, ng)))
        et.inv_trans(r, pspscalar=to(sp), pgp=gp)
        s2 = to(np.zeros((ns2, 3)))
        et.dir_trans(r, pspscalar=s2, pgp=gp)
        outs.append((back(gp).copy(), back(s2).copy()))
        et.trans_release(r)
    assert rel_err(outs[0][0], outs[1][0]) < 1e-13 and rel_err(outs[0][1], outs[1][1]) < 1e-13
    assert not np.array_equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("precision", [8, 4])
def test_direct_mixed_radix_fft_kernels_adjoints(et, dev, precision):
    """INV_TRANSAD / DIR_TRANSAD on rows that take k_fft_*_mr (their `adj` scalings): the reference's dot-product identity"""
    e_inv, e_dir = adjoint_case(et, dev, 15, MR_MID + MR_MID[::-1], 1, 2, nproma=3000, precision=precision)
    tol = 1e-12 if precision == 8 else 2e-4
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


R16_ROWS = [1540, 2044, 2052, 2556, 2564, 3068, 3076, 4092]  # first and last row length of the work lengths 256 R1, R1 = 8, 10, 12, 16


@pytest.mark.parametrize("precision,nproma", [(8, 1000), (4, 1000), (8, 4094)])
def test_register_resident_fft_kernels_blocked_rows(et, dev, precision, nproma, monkeypatch):
    """k_fft_dir_r16 / k_fft_inv_r16 (round 3: rows whose Bluestein work length is 256 R1, points in registers, LDS as the
    exchange medium) with NPROMA blocks that cut the rows: the element-wise grid path instead of the row-as-one-buffer
    path the unblocked cases take; winds, derivatives and both precisions."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    from oracle.oracle import Oracle as O
    e_inv, e_dir = run_case(et, O, dev, 15, R16_ROWS + R16_ROWS[::-1], 2, 3, dict(scders=True, uvder=True, vorgp=True, divgp=True), nproma, precision=precision)
    tol = TOL if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_register_resident_and_lds_fft_kernels_agree(et, dev, monkeypatch):
    """EMI_FFT_R16=0 sends the same rows through the in-place LDS kernels (other work lengths for some of them, another
    factorisation for all): both are the same transform to rounding."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    to, back = dev
    nloen = np.array(R16_ROWS + R16_ROWS[::-1], dtype=np.int32)
    rng = np.random.default_rng(5)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("EMI_FFT_R16", "0")
        r = et.setup_trans(15, len(nloen), nloen)
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        if not outs:
            sp = random_spectrum(rng, et.trans_inq(r, "nasm0"), 15, ns2, 3, False)
        gp = to(np.zeros((1, 3, ng)))
        et.inv_trans(r, pspscalar=to(sp), pgp=gp)
        s2 = to(np.zeros((ns2, 3)))
        et.dir_trans(r, pspscalar=s2, pgp=gp)
        outs.append((back(gp).copy(), back(s2).copy()))
        et.trans_release(r)
    assert rel_err(outs[0][0], outs[1][0]) < 1e-13 and rel_err(outs[0][1], outs[1][1]) < 1e-13
    assert not np.array_equal(outs[0][0], outs[1][0])  # really two different kernels


@pytest.mark.parametrize("precision", [8, 4])
def test_register_resident_fft_kernels_adjoints(et, dev, precision, monkeypatch):
    """INV_TRANSAD / DIR_TRANSAD on rows that take the register-resident kernels (their `adj` scalings): the reference's
    dot-product identity (tests/trans/test_invtrans_adjoint.F90: 2000 epsilon)."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    e_inv, e_dir = adjoint_case(et, dev, 15, R16_ROWS + R16_ROWS[::-1], 1, 2, nproma=3000, precision=precision)
    tol = 1e-12 if precision == 8 else 2e-4
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


# first / last row length of every class of the split kernels: k_fft_*_r16p<10> 4100 .. 5120, <12> 5124 .. 6144, <16> 6148 .. 8192; 4102 and 6146 have an
# odd half-length (no split: the in-place kernels beside the split ones in one call)
R16S_ROWS = [4100, 4102, 4612, 5116, 5120, 5124, 5136, 6144, 6146, 6148, 7000, 8188, 8192]


@pytest.mark.parametrize("precision,nproma", [(8, None), (4, None), (8, 1000), (4, 4094)])
def test_split_register_resident_fft_kernels_match_oracle(et, dev, precision, nproma, monkeypatch):
    """k_fft_dir_r16p / k_fft_inv_r16p (round 6: a row of more than 4096 points as TWO register-resident chirp-z convolutions of half its
    half-length, one after the other with the idle half parked in LDS, joined by one decimation step -- at TCo1279 the rows of 4100 .. 5136
    points that ran on k_fft_*_hot<23 | 24>) against the oracle: every class boundary, whole rows and NPROMA-cut rows, winds and
    derivatives (all FSC modes), both precisions."""
    monkeypatch.setenv("EMI_FFT_MR", "0")  # some of these rows have a 23-smooth half-length: keep them off the direct mixed-radix kernels
    from oracle.oracle import Oracle as O
    e_inv, e_dir = run_case(et, O, dev, 15, R16S_ROWS + R16S_ROWS[::-1], 2, 3, dict(scders=True, uvder=True, vorgp=True, divgp=True), nproma, precision=precision)
    tol = TOL if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


@pytest.mark.parametrize("precision", [8, 4])
def test_long_row_fft_kernels_through_the_exchange_order_tables(et, dev, precision, monkeypatch):
    """The register-resident, split, specialised in-place and direct mixed-radix kernels with the Fourier rows addressed through the
    exchange-order table (g.fftrow: what several tasks use; EMI_TEST_PATHS bit 0 switches it on for one task): the clamped look-ups and
    selects of FOURIER_IN, the table rows of FOURIER_OUT.  Rows of every kernel family in one grid."""
    monkeypatch.setenv("EMI_TEST_PATHS", "1")
    from oracle.oracle import Oracle as O
    rows = [1540, 2052, 3076, 4092, 4100, 4102, 5120, 5124, 6146, 2048, 3840, 4800]  # r16<8..16>, r16p<10 | 12>, hot<23>, hot<9>, mr (2^a 3 5 half-lengths)
    e_inv, e_dir = run_case(et, O, dev, 15, rows + rows[::-1], 1, 2, dict(scders=True, uvder=True), None, precision=precision)
    tol = TOL if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_split_and_lds_fft_kernels_agree(et, dev, monkeypatch):
    """EMI_FFT_R16S=0 sends the same rows through the in-place LDS kernels (one convolution of twice the work length): the same
    transform to rounding, and really another kernel."""
    monkeypatch.setenv("EMI_FFT_MR", "0")
    to, back = dev
    nloen = np.array(R16S_ROWS + R16S_ROWS[::-1], dtype=np.int32)
    rng = np.random.default_rng(6)
    outs = []
    for off in (False, True):
        monkeypatch.setenv("EMI_FFT_R16S", "0" if off else "1")
        r = et.setup_trans(15, len(nloen), nloen)
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        if not outs:
            sp = random_spectrum(rng, et.trans_inq(r, "nasm0"), 15, ns2, 3, False)
        gp = to(np.zeros((1, 3, ng)))
        et.inv_trans(r, pspscalar=to(sp), pgp=gp)
        s2 = to(np.zeros((ns2, 3)))
        et.dir_trans(r, pspscalar=s2, pgp=gp)
        outs.append((back(gp).copy(), back(s2).copy()))
        et.trans_release(r)
    assert rel_err(outs[0][0], outs[1][0]) < 1e-13 and rel_err(outs[0][1], outs[1][1]) < 1e-13
    assert not np.array_equal(outs[0][0], outs[1][0])  # really two different kernels


@pytest.mark.parametrize("precision", [8, 4])
def test_split_register_resident_fft_kernels_adjoints(et, dev, precision, monkeypatch):
    """INV_TRANSAD / DIR_TRANSAD on rows that take the split kernels (their `adj` scalings): the reference's dot-product identity
    (tests/trans/test_invtrans_adjoint.F90: 2000 epsilon)."""
    monkeypatch.setenv("EMI_FFT_MR", "0")
    e_inv, e_dir = adjoint_case(et, dev, 15, R16S_ROWS + R16S_ROWS[::-1], 1, 2, nproma=3000, precision=precision)
    tol = 1e-12 if precision == 8 else 2e-4
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_device_legendre_setup_matches_host_and_oracle(et, monkeypatch):
    """k_legpol (SUPOLF on the GPU, fp contraction off) against the host recurrence bit for bit and
    against the oracle: a full grid F64 / T127 (1e+-100 rescaling near the poles) and TCo639 rows."""
    from oracle.oracle import Oracle as O
    for N, nloen, ms in ((127, np.full(128, 256, dtype=np.int32), (2, 3, 40, 100, 126, 127)),
                         (639, octahedral(639), (2, 211, 500, 639))):
        monkeypatch.delenv("EMI_LEGPOL_HOST", raising=False)
        r = et.setup_trans(N, len(nloen), nloen)
        monkeypatch.setenv("EMI_LEGPOL_HOST", "1")
        rh = et.setup_trans(N, len(nloen), nloen)
        o = O(N, nloen)
        try:
            for m in ms:
                for sym in (False, True):
                    dev, ref = et.legendre_panel(r, m, sym), o.rpnm(m, sym)
                    assert dev.shape == ref.shape
                    assert np.abs(dev - ref).max(initial=0.0) <= 1e-14 * max(1.0, np.abs(ref).max(initial=0.0))
                    assert np.array_equal(dev, et.legendre_panel(rh, m, sym))
        finally:
            et.trans_release(r)
            et.trans_release(rh)


@pytest.mark.parametrize("precision,nsmax", [(8, 47), (4, 47), (8, 159)])
def test_legendre_polynomial_file_io(et, dev, tmp_path, precision, nsmax):
    """CDIO_LEGPOL = writef / readf / membuf in the reference's file format (write_legpol_mod.F90:66-158):
    written panels against an image assembled from the oracle's, panels read back bit for bit, transforms
    from a read set-up against the oracle, the reference's header checks.  T159 has device-computed
    panels with rescaling (m >= 2 runs k_legpol) behind the written file."""
    legpol_io_case(et, Oracle, dev, tmp_path, nsmax=nsmax, precision=precision)


@pytest.mark.parametrize("case", [(21, 0, 3, None, 8), (63, 2, 2, None, 8), (63, 1, 1, 1000, 8), (159, 2, 3, None, 8), (63, 2, 2, None, 4)])
def test_adjoint_transforms_dot_product(et, dev, case):
    """INV_TRANSAD / DIR_TRANSAD: <A x, y> = <x, A* y> with the inner products of the reference's
    tests/trans/test_invtrans_adjoint.F90 and test_dirtrans_adjoint.F90 (tolerance there: 2000 ...
    20000 machine epsilons), scalars and wind fields."""
    nsmax, nuv, nsc, nproma, prec = case
    e_inv, e_dir = adjoint_case(et, dev, nsmax, octahedral(nsmax), nuv, nsc, nproma, precision=prec)
    tol = 2000 * (np.finfo(np.float32).eps if prec == 4 else np.finfo(np.float64).eps)
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


@pytest.mark.parametrize("seed", range(40))
def test_random_reduced_grids_match_oracle(et, dev, seed):
    """Random reduced grids (row lengths of any parity and factorisation, 8 ... 700 points, symmetric
    about the equator), truncations, field counts, options and NPROMA against the oracle."""
    from oracle.oracle import Oracle as O
    rng = np.random.default_rng(1000 + seed)
    nh = int(rng.integers(4, 14))
    half = np.sort(rng.integers(8, 700, nh))
    nloen = np.concatenate([half, half[::-1]]).astype(np.int32)
    nsmax = int(rng.integers(2, 2 * nh))
    nuv, nsc = int(rng.integers(0, 3)), int(rng.integers(0, 4))
    if nuv + nsc == 0:
        nsc = 1
    flags = dict(scders=bool(rng.integers(2)) and nsc > 0, uvder=bool(rng.integers(2)) and nuv > 0,
                 vorgp=bool(rng.integers(2)) and nuv > 0, divgp=bool(rng.integers(2)) and nuv > 0)
    nproma = [None, 17, 100, 1000][int(rng.integers(4))]
    e_inv, e_dir = run_case(et, O, dev, nsmax, nloen, nuv, nsc, flags, nproma, seed=seed)
    assert e_inv < TOL and e_dir < TOL, (nloen.tolist(), nsmax, nuv, nsc, flags, nproma, e_inv, e_dir)


def test_host_arrays_match_oracle(et):
    """EMI_MEM_HOST: numpy arrays staged over PCIe, as a Fortran/C caller would pass them."""
    from oracle.oracle import Oracle as O
    e_inv, e_dir = run_case(et, O, HOST, 31, octahedral(31), 2, 3, dict(scders=True), 500)
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


def test_host_output_arrays_keep_unwritten_elements(et):
    """What INV_TRANS does not write (padding of the last NPROMA block, surplus fields of PGP) comes back
    from the device staging as the caller left it."""
    N = 21
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        npr = 97
        nb = (ng - 1) // npr + 1
        assert nb * npr > ng
        sp = np.random.default_rng(11).uniform(-1, 1, (ns2, 2))
        gp = np.full((nb, 3, npr), -7.25)
        et.inv_trans(r, pspscalar=sp, pgp=gp, kproma=npr)
        tail = ng - (nb - 1) * npr
        assert np.all(gp[-1, :, tail:] == -7.25) and np.all(gp[:, 2, :] == -7.25)
        assert np.all(np.isfinite(gp)) and np.all(gp[:-1, :2, :] != -7.25)
    finally:
        et.trans_release(r)


def test_host_staging_buffers_are_reused_between_calls(et):
    """The staging pool of csrc/emi_stage.h: calls of different sizes reuse device buffers and see only their own data."""
    from oracle.oracle import Oracle as O
    from tests.common import staging_pool_case
    staging_pool_case(et, O, TOL)


@pytest.mark.parametrize("precision", [8, 4])
def test_many_plain_copy_fields_and_odd_level_counts(et, dev, precision):
    """200 scalar fields of PSPSCALAR (imaginary parts of zonal wavenumber 0 poisoned: they are never read), then call-mode-2
    arrays with 65 levels x 3 variables + winds, in one batch and in batches of 128 fields."""
    from oracle.oracle import Oracle as O
    from tests.common import direct_spectral_tiles_case
    direct_spectral_tiles_case(et, O, dev, TOL if precision == 8 else 3e-5, nsmax=63, precision=precision)


def test_field_batching_is_invisible(et, dev):
    """NPROMATR-like field packets (dir_trans_ctl_mod.F90:128-175): results must not depend on
    the batch size."""
    from oracle.oracle import Oracle as O
    try:
        et.set_max_batch(64)
        e_inv, e_dir = run_case(et, O, dev, 21, octahedral(21), 50, 90, dict(scders=True, uvder=True))
    finally:
        et.set_max_batch(0)
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)


@pytest.mark.parametrize("paths", [1, 2, 4, 7])
def test_multi_task_code_paths_on_one_task(et, dev, paths, monkeypatch):
    """EMI_TEST_PATHS (test-only switch of csrc/ectrans_mi.hip): bit 0 = the exchange-order row tables (legN / legS / fftrow) on one task,
    bit 1 = the 4-batch three-stream pipeline on one task (needs >= 256 Fourier fields), bit 2 = the scalar fields of DIR_TRANS through
    k_postpack_dir instead of the epilogue of k_leg_dir.  Several tasks (and the adjoint options) run exactly this code; the switch keeps it
    covered on a box with one GPU.  Same bound as every other case."""
    from oracle.oracle import Oracle as O
    monkeypatch.setenv("EMI_TEST_PATHS", str(paths))
    e_inv, e_dir = run_case(et, O, dev, 31, octahedral(31), 40, 200, dict(vorgp=True))
    assert e_inv < TOL and e_dir < TOL, (paths, e_inv, e_dir)


def test_call_mode_2(et, dev):
    """PSPSC3A/PSPSC2 + PGPUV/PGP3A/PGP2 (the arrays ectrans-benchmark uses, :450-479)."""
    to, back = dev
    N = 31
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    o = Oracle(N, nloen)
    rng = np.random.default_rng(5)
    nlev, nvar = 3, 2
    vor = random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, True)
    div = random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, True)
    sc3 = np.stack([random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, False) for _ in range(nvar)])
    sc2 = random_spectrum(rng, o.nasm0, N, o.nspec2, 1, False)
    scflat = np.concatenate([sc2] + [sc3[v] for v in range(nvar)], axis=1)
    gref = o.inv_trans(spvor=vor, spdiv=div, spsc=scflat, scders=True)
    ng = o.ngptot
    gpuv, gp3a, gp2 = to(np.zeros((1, 2, nlev, ng))), to(np.zeros((1, 3 * nvar, nlev, ng))), to(np.zeros((1, 3, ng)))
    et.inv_trans(r, pspvor=to(vor), pspdiv=to(div), pspsc3a=to(sc3), pspsc2=to(sc2), pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2,
                 ldscders=True)
    nsc = 1 + nvar * nlev
    uv = back(gpuv)[0].reshape(2 * nlev, ng)
    a3, a2 = back(gp3a)[0], back(gp2)[0]
    parts = []
    for k in range(3):  # value, N-S derivative, E-W derivative (trltog_mod.F90:632-690)
        parts.append(np.concatenate([a2[k:k + 1]] + [a3[k * nvar + v] for v in range(nvar)]))
    got = np.concatenate([uv, parts[0], parts[1], parts[2]])
    ref = np.concatenate([gref[:2 * nlev], gref[2 * nlev:2 * nlev + nsc], gref[2 * nlev + nsc:2 * nlev + 2 * nsc],
                          gref[2 * nlev + 2 * nsc:]])
    assert rel_err(got, ref, axis=1) < TOL
    et.trans_release(r)


@pytest.mark.parametrize("nfld", [0, 10])
@pytest.mark.parametrize("opts", ["", "scders uvders", "vordiv", "scders uvders vordiv"])
@pytest.mark.parametrize("nproma", [0, 16])
@pytest.mark.parametrize("callmode", [1, 2])
def test_benchmark_option_matrix(et, dev, nfld, opts, nproma, callmode):
    """The reference's CTest matrix of ectrans-benchmark at T47/O48 (tests/CMakeLists.txt:219-326): nfld 0 / 10 x 20 levels,
    --scders --uvders, --vordiv, --nproma 16, call modes 1 and 2; harmonic Re(4,19) = 1 in every field, two iterations of
    INV_TRANS + DIR_TRANS, spectral-norm drift <= 100 eps (ectrans-benchmark.F90:743-756, 847-871)."""
    to, back = dev
    N, nlev = 47, 20
    scders, uvders, vordiv = "scders" in opts, "uvders" in opts, "vordiv" in opts
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        npr = nproma or ng
        nb = (ng - 1) // npr + 1
        i419 = int(et.trans_inq(r, "nasm0")[4] - 1 + 2 * 15)
        nsc3 = nfld * nlev
        vor, div, sc2 = to(np.zeros((ns2, nlev))), to(np.zeros((ns2, nlev))), to(np.zeros((ns2, 1)))
        sc3 = to(np.zeros((max(nfld, 1), ns2, nlev)))  # PSPSC3A(nlev, nspec2, nfld), C order
        for a in (vor, div, sc2):
            a[i419] = 1.0
        sc3[:, i419] = 1.0
        nuvvar = 2 + (2 if vordiv else 0) + (2 if uvders else 0)
        dmul = 3 if scders else 1
        flags = dict(ldscders=scders, ldvorgp=vordiv, lddivgp=vordiv, lduvder=uvders, kproma=npr)
        if callmode == 2:
            gpuv = to(np.zeros((nb, nuvvar, nlev, npr)))
            gp3a = to(np.zeros((nb, max(nfld, 1) * dmul, nlev, npr)))
            gp2 = to(np.zeros((nb, dmul, npr)))
            kw3 = dict(pspsc3a=sc3, pgp3a=gp3a) if nfld else {}
            u0 = 2 if vordiv else 0  # PGPUV variables: [vor, div,] u, v [, u_EW, v_EW]
        else:
            nsc = nsc3 + 1
            spsc = to(np.zeros((ns2, nsc)))
            spsc[i419] = 1.0
            gp = to(np.zeros((nb, nuvvar * nlev + nsc * dmul, npr)))
            gd = to(np.zeros((nb, 2 * nlev + nsc, npr)))
            u0 = 2 * nlev if vordiv else 0
        n0 = [et.specnorm(r, vor), et.specnorm(r, div), et.specnorm(r, sc2)]
        for _ in range(2):
            if callmode == 2:
                et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc2=sc2, pgpuv=gpuv, pgp2=gp2, **kw3, **flags)
                kd = dict(pspsc3a=sc3, pgp3a=gp3a[:, :nfld].contiguous()) if nfld else {}
                et.dir_trans(r, pspvor=vor, pspdiv=div, pspsc2=sc2, pgpuv=gpuv[:, u0:u0 + 2].contiguous(), pgp2=gp2[:, :1].contiguous(),
                             kproma=npr, **kd)
            else:
                et.inv_trans(r, pspvor=vor, pspdiv=div, pspscalar=spsc, pgp=gp, **flags)
                gd[:, :2 * nlev] = gp[:, u0:u0 + 2 * nlev]  # u, v
                s0 = u0 + 2 * nlev
                gd[:, 2 * nlev:] = gp[:, s0:s0 + nsc]
                et.dir_trans(r, pspvor=vor, pspdiv=div, pspscalar=spsc, pgp=gd, kproma=npr)
        n1 = [et.specnorm(r, vor), et.specnorm(r, div), et.specnorm(r, sc2)]
        if callmode == 1:
            n0.append(np.full(nsc, n0[2][0]))
            n1.append(et.specnorm(r, spsc))
        elif nfld:
            n0.append(np.full(nlev, n0[2][0]))
            n1.append(et.specnorm(r, sc3[nfld - 1]))
        for a, b in zip(n0, n1):
            assert np.abs(np.asarray(a) / np.asarray(b) - 1.0).max() <= 100 * np.finfo(np.float64).eps
    finally:
        et.trans_release(r)


@pytest.mark.parametrize("precision", [8, 4])
def test_benchmark_harmonic_round_trips(et, dev, precision):
    """ectrans-benchmark semantics at T47/O48 (tests/CMakeLists.txt:219-326): Re(4,19)=1 in every
    field, 2 x (inv, dir), spectral-norm drift <= 100 eps (ectrans-benchmark.F90:847-871), eps being
    that of the library's working precision (epsilon(1.0_jprb))."""
    to, back = dev
    EPS = float(np.finfo(np.float32 if precision == 4 else np.float64).eps)
    if precision == 4:
        to = lambda a, _to=dev[0]: _to(a.astype(np.float32))
    N = 47
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen, precision=precision)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    nlev, nfld = 20, 10
    i419 = int(et.trans_inq(r, "nasm0")[4] - 1 + 2 * 15)
    vor, div, sc2 = to(np.zeros((ns2, nlev))), to(np.zeros((ns2, nlev))), to(np.zeros((ns2, 1)))
    sc3 = to(np.zeros((nfld, ns2, nlev)))
    for a in (vor, div, sc2):
        a[i419] = 1.0
    sc3[:, i419] = 1.0
    n0 = [et.specnorm(r, vor), et.specnorm(r, div), et.specnorm(r, sc2), et.specnorm(r, sc3[0])]
    gpuv, gp3a, gp2 = to(np.zeros((1, 2, nlev, ng))), to(np.zeros((1, nfld, nlev, ng))), to(np.zeros((1, 1, ng)))
    for _ in range(2):
        et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
        et.dir_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    n1 = [et.specnorm(r, vor), et.specnorm(r, div), et.specnorm(r, sc2), et.specnorm(r, sc3[0])]
    for a, b in zip(n0, n1):
        assert np.abs(a / b - 1.0).max() <= 100 * EPS
    et.trans_release(r)


def test_full_size_properties_tco399(et, dev):
    """BASELINE config[1] size (TCo399) through size-independent properties: linearity, the
    harmonic round trip, and agreement of spectral norms with the oracle on a few fields."""
    import torch
    to, back = dev
    N = 399
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    nasm0 = et.trans_inq(r, "nasm0")
    rng = np.random.default_rng(20251114)
    nf = 6
    a = random_spectrum(rng, nasm0, N, ns2, nf, False)
    b = random_spectrum(rng, nasm0, N, ns2, nf, False)
    ga, gb, gab = (to(np.zeros((1, nf, ng))) for _ in range(3))
    et.inv_trans(r, pspscalar=to(a), pgp=ga)
    et.inv_trans(r, pspscalar=to(b), pgp=gb)
    et.inv_trans(r, pspscalar=to(2.0 * a - 3.0 * b), pgp=gab)
    lin = (2.0 * ga - 3.0 * gb - gab).abs().max().item() / gab.abs().max().item()
    assert lin < 1e-13
    # inverse followed by direct returns the input up to the octahedral truncation error
    s2 = to(np.zeros((ns2, nf)))
    et.dir_trans(r, pspscalar=s2, pgp=ga)
    assert rel_err(back(s2), a) < 1e-9
    assert np.abs(et.specnorm(r, s2) / et.specnorm(r, to(a)) - 1.0).max() < 1e-10
    # oracle on the same inputs (2 fields keep the CPU time at a few seconds)
    o = Oracle(N, nloen)
    gref = o.inv_trans(spsc=a[:, :2])
    assert rel_err(back(ga)[0, :2], gref, axis=1) < TOL
    _, _, sref = o.dir_trans(gref, nsc=2)
    s3 = to(np.zeros((ns2, 2)))
    et.dir_trans(r, pspscalar=s3, pgp=to(gref.reshape(1, 2, -1)))
    assert rel_err(back(s3), sref) < TOL
    assert np.abs(et.specnorm(r, s3) / o.specnorm(sref) - 1.0).max() < 1e-10  # north-star tolerance
    et.trans_release(r)


def test_two_resolutions_coexist(et, dev):
    to, back = dev
    r1 = et.setup_trans(21, 44, octahedral(21))
    r2 = et.setup_trans(31, 64, octahedral(31))
    assert r1 != r2
    for r, N in ((r1, 21), (r2, 31)):
        o = Oracle(N, octahedral(N))
        sp = random_spectrum(np.random.default_rng(N), o.nasm0, N, o.nspec2, 2, False)
        gp = to(np.zeros((1, 2, o.ngptot)))
        et.inv_trans(r, pspscalar=to(sp), pgp=gp)
        assert rel_err(back(gp)[0], o.inv_trans(spsc=sp), axis=1) < TOL
    et.trans_release(r1)
    et.trans_release(r2)


def test_no_device_memory_leak_over_setup_release_cycles(et, dev):
    """Device memory after N cycles of SETUP_TRANS / transforms (device and host arrays, two field counts) / TRANS_RELEASE
    equals the level after the first cycle (the reference's tests/transi/transi_test_memory.c does this for the heap): panels,
    tables, work buffers, tile maps and descriptors all go back; the staging pool of host arrays is bounded by the largest call."""
    import torch
    to, back = dev
    N = 63
    nloen = octahedral(N)
    rng = np.random.default_rng(0)

    def cycle(nf):
        r = et.setup_trans(N, len(nloen), nloen)
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        sp = rng.uniform(-1, 1, (ns2, nf))
        gp = np.zeros((1, nf, ng))
        et.inv_trans(r, pspscalar=sp, pgp=gp)  # host arrays
        et.dir_trans(r, pspscalar=sp, pgp=gp)
        tsp, tgp = to(sp), to(gp)
        et.inv_trans(r, pspscalar=tsp, pgp=tgp)  # device arrays
        et.dir_trans(r, pspscalar=tsp, pgp=tgp)
        torch.cuda.synchronize()
        et.trans_release(r)
        del tsp, tgp
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    levels = [cycle(70 if i % 2 else 7) for i in range(16)]
    # a leak shrinks the free memory cycle after cycle; the HIP runtime's own pools may still grow once or twice early on
    assert max(levels[8:]) - min(levels[8:]) == 0 and levels[0] - levels[-1] <= 64 << 20, levels


def test_errors_on_gpu(et, dev):
    to, _ = dev
    r = et.setup_trans(21, 44, octahedral(21))
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    with pytest.raises(et.TransError, match="SECOND DIMENSION OF PGP TOO SMALL"):
        et.inv_trans(r, pspscalar=to(np.zeros((ns2, 3))), pgp=to(np.zeros((1, 2, ng))))
    with pytest.raises(et.TransError, match="same memory space"):
        et.inv_trans(r, pspscalar=np.zeros((ns2, 1)), pgp=to(np.zeros((1, 1, ng))))
    et.trans_release(r)


def test_alltoallv_hook_over_rccl_single_rank(dev):
    """The torch.distributed hook the multi-GPU path uses (ectrans_amd/dist.py), exercised on this
    one-GPU box with a 1-rank RCCL group: raw device pointers -> tensors -> all_to_all_single."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ectrans_amd import dist as edist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    hook = edist.make_alltoallv_hook(None, "cuda:0")
    src = torch.arange(1000, dtype=torch.float64, device="cuda:0")
    dst = torch.zeros_like(src)
    cnt = (C.c_longlong * 1)(src.numel() * 8)
    dsp = (C.c_longlong * 1)(0)
    rc = hook(None, src.data_ptr(), cnt, dsp, dst.data_ptr(), cnt, dsp, 1, None)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(src, dst)
    # a caller-supplied HIP stream: the collective must be ordered on it (producer kernel before, consumer after)
    side = torch.cuda.Stream(device="cuda:0")
    with torch.cuda.stream(side):
        src2 = torch.zeros(1 << 22, dtype=torch.float64, device="cuda:0")
        for _ in range(20):
            src2 += 1.0  # still running when the hook is called
        dst2 = torch.zeros_like(src2)
        cnt2 = (C.c_longlong * 1)(src2.numel() * 8)
        rc = hook(None, src2.data_ptr(), cnt2, dsp, dst2.data_ptr(), cnt2, dsp, 1, side.cuda_stream)
        out = dst2 * 2.0
    side.synchronize()
    assert rc == 0 and float(out.min()) == 40.0 and float(out.max()) == 40.0
    dist.destroy_process_group()


def test_calls_on_different_streams_are_serialised(et, dev):
    """Calls on one resolution share its descriptor and work buffers whatever stream each names: the library
    orders every call behind the previous one of that resolution (a device-side event wait), and SPECNORM,
    which has no stream argument, behind the last transform.  Two alternating non-blocking streams, no host
    synchronisation in between, against the same calls run one by one."""
    import torch
    N = 399
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        nasm0 = et.trans_inq(r, "nasm0")
        rng = np.random.default_rng(17)
        nf = 96
        to, back = dev
        a, b = (to(random_spectrum(rng, nasm0, N, ns2, nf, False)) for _ in range(2))
        ga, gb, ga0, gb0 = (torch.zeros((1, nf, ng), dtype=torch.float64, device="cuda:0") for _ in range(4))
        sa, sb = torch.zeros_like(a), torch.zeros_like(b)
        et.inv_trans(r, pspscalar=a, pgp=ga0)
        torch.cuda.synchronize()
        et.inv_trans(r, pspscalar=b, pgp=gb0)
        torch.cuda.synchronize()
        n0 = et.specnorm(r, a)
        s1, s2 = torch.cuda.Stream(device="cuda:0"), torch.cuda.Stream(device="cuda:0")
        torch.cuda.synchronize()
        for _ in range(3):
            et.inv_trans(r, pspscalar=a, pgp=ga, stream=s1.cuda_stream)
            et.inv_trans(r, pspscalar=b, pgp=gb, stream=s2.cuda_stream)
        et.dir_trans(r, pspscalar=sa, pgp=ga, stream=s1.cuda_stream)
        n1 = et.specnorm(r, sa)  # null stream, right behind the direct transform on s1
        torch.cuda.synchronize()
        assert torch.equal(ga, ga0) and torch.equal(gb, gb0)
        assert np.abs(n1 / n0 - 1.0).max() < 1e-10
    finally:
        et.trans_release(r)


def test_fp32_library_mean_wavenumber_in_double(et, dev, monkeypatch):
    """fp32 library, zonal wavenumber 0 on the fp64 matrix cores with promoted operands (the reference's sp build:
    "DGEM for the mean to improve mass conservation", cpu/internal/ledir_mod.F90:133-171): the global mean -- coefficient
    (0, 0) -- of TCo399 fields with a 250 K mean matches the oracle's sp mode (float operands, double accumulation for
    m = 0) to one float ulp and the whole m = 0 column to one ulp of its maximum."""
    to, back = dev
    N = 399
    nloen = octahedral(N)
    o = Oracle(N, nloen, lazy=True)
    rng = np.random.default_rng(2)
    nf = 4
    s = random_spectrum(rng, o.nasm0, N, o.nspec2, nf, False)
    s[0, :] = 250.0 + np.arange(nf)
    g = o.inv_trans(spsc=s).astype(np.float32)
    o.set_sp_mode(True)
    _, _, ref = o.dir_trans(g.astype(np.float64), nsc=nf)
    ulp = float(np.finfo(np.float32).eps)
    m0 = slice(0, 2 * (N + 1), 2)
    err = {}
    for single in (False,):
        r = et.setup_trans(N, len(nloen), nloen, precision=4)
        try:
            out = to(np.zeros((o.nspec2, nf), dtype=np.float32))
            et.dir_trans(r, pspscalar=out, pgp=to(g.reshape(1, nf, -1)))
            got = back(out).astype(np.float64)
            err[single] = (np.abs(got[0] / ref[0] - 1.0).max() / ulp, np.abs(got[m0] - ref[m0]).max() / np.abs(ref[m0]).max() / ulp)
            if not single:  # the inverse transform of m = 0 runs in double too (as the reference's GPU back-end, leinv_mod.F90:273)
                gp = to(np.zeros((1, nf, o.ngptot), dtype=np.float32))
                et.inv_trans(r, pspscalar=to(s.astype(np.float32)), pgp=gp)
                assert rel_err(back(gp)[0].astype(np.float64), o.inv_trans(spsc=s.astype(np.float32).astype(np.float64)), axis=1) < 3e-5
        finally:
            et.trans_release(r)
    assert err[False][0] <= 1.0 and err[False][1] <= 1.0, err


@pytest.mark.parametrize("nsmax,precision", [(10, 8), (21, 8), (10, 4)])
def test_adjoints_match_transposed_oracle_matrices(et, dev, nsmax, precision):
    """INV_TRANSAD / DIR_TRANSAD against an oracle adjoint: the dense matrices of the ORACLE's forward INV_TRANS /
    DIR_TRANS (vor, div, scalar <-> u, v, scalar), formed column by column at T10/O11 and T21/O22 and transposed with
    the inner products of the reference's adjoint tests (tests/common.py::adjoint_matrix_case).  Unlike the dot-product
    identity this cannot be satisfied by a wrong-but-adjoint pair of HIP kernels."""
    from tests.common import adjoint_matrix_case
    e_inv, e_dir = adjoint_matrix_case(et, Oracle, dev, nsmax=nsmax, precision=precision)
    tol = 1e-12 if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


@pytest.mark.parametrize("half,precision", [([10244, 10248, 10252, 10256, 5136, 20484], 8), ([20484, 20500, 40964, 1284], 4)])
def test_rows_longer_than_the_lds(et, dev, half, precision):
    """Any KLOEN (ftdir_mod.F90:67-84): rows whose Bluestein work array exceeds the 160 KiB of LDS -- in fp64 everything
    beyond 10240 complex points, i.e. the four longest rows of TCo2559 (10244 ... 10256) -- run the passes on a global
    scratch buffer (k_fft_inv_gm / k_fft_dir_gm) instead of returning EMI_ERR_UNSUPPORTED; mixed with rows that use
    the specialised LDS kernels, derivatives on, against the oracle."""
    from oracle.oracle import Oracle as O
    nloen = np.array(half + half[::-1], dtype=np.int32)
    e_inv, e_dir = run_case(et, O, dev, 7, nloen, 2, 3, dict(scders=True, uvder=True), 10000, precision=precision)
    tol = TOL if precision == 8 else 3e-5
    assert e_inv < tol and e_dir < tol, (e_inv, e_dir)


def test_belousov_generator_lduserpnm(et, dev, golden_dir):
    """LDUSERPNM=.TRUE. (the Fortran API's default, setup_trans.F90:231): Belousov's SUPOL as the generator of the Legendre
    panels.  Against the oracle's SUPOL at T159 (panels 1e-14, transforms 1e-11) and against the reference's golden
    vectors (1e-10, as test_ectrans4py.py:133-158)."""
    to, back = dev
    N = 159
    nloen = octahedral(N)
    ob = Oracle(N, nloen, belusov=True)
    r = et.setup_trans(N, len(nloen), nloen, lduserpnm=True)
    try:
        for m in (0, 1, 2, 57, 158, 159):
            for sym in (False, True):
                a, b = et.legendre_panel(r, m, sym), ob.rpnm(m, sym)
                assert a.shape == b.shape and np.abs(a - b).max(initial=0.0) <= 1e-14
    finally:
        et.trans_release(r)
    e_inv, e_dir = run_case(et, lambda *a, **k: Oracle(*a, belusov=True, **k), dev, N, nloen, 2, 3, setup_kw=dict(lduserpnm=True))
    assert e_inv < TOL and e_dir < TOL, (e_inv, e_dir)
    d = os.path.join(golden_dir, "tl149")
    nl = np.load(os.path.join(d, "lon_number_by_lat.npy")).astype(np.int32)
    sp = np.load(os.path.join(d, "tl149-c24-s1t@sp.npy"))
    gpll = np.load(os.path.join(d, "tl149-c24-s1t@sp2gp.npy"))
    gp_ref = np.concatenate([gpll[i, : nl[i]] for i in range(nl.size)])
    r = et.setup_trans(148, 150, nl, lduserpnm=True)
    try:
        gp = to(np.zeros((1, 1, gp_ref.size)))
        et.inv_trans(r, pspscalar=to(sp.reshape(-1, 1)), pgp=gp)
        assert np.abs(back(gp)[0, 0] - gp_ref).max() < 1e-10
        s2 = to(np.zeros((sp.size, 1)))
        et.dir_trans(r, pspscalar=s2, pgp=to(gp_ref.reshape(1, 1, -1)))
        assert np.abs(back(s2)[:, 0] - sp).max() < 1e-10
    finally:
        et.trans_release(r)


@pytest.mark.parametrize("flags", [dict(scders=True), dict(uvder=True), dict(vorgp=True), dict(scders=True, vorgp=True, uvder=True)])
def test_adjoint_with_derivative_options_matches_transposed_oracle(et, dev, flags):
    """INV_TRANSAD with LDSCDERS / LDVORGP / LDDIVGP / LDUVDER (ltinvad_mod.F90:149-225, spnsdead_mod.F90, fscad_mod.F90)
    against the weighted transpose of the ORACLE's forward INV_TRANS with the same options (T8/O9, NPROMA blocks)."""
    from tests.common import adjoint_options_case
    e = adjoint_options_case(et, Oracle, dev, nsmax=8, flags=flags, nproma=61)
    assert e < 1e-12, (flags, e)


def test_adjoint_options_through_call_mode2_arrays(et, dev):
    from tests.common import adjoint_options_call_mode2_case
    assert adjoint_options_call_mode2_case(et, dev, nsmax=21) < 1e-14


@pytest.mark.parametrize("grid,precision", [("octahedral", 8), ("regular", 8), ("octahedral", 4)])
def test_closed_form_winds_and_derivatives(et, dev, grid, precision):
    """HIP path against analytic fields (solid-body rotation, single harmonics of vorticity, divergence and a scalar with
    all derivative outputs): numbers neither the oracle nor the reference's tests hold -- see tests/common.py::closed_form_case."""
    to0, back0 = dev
    dt = np.float32 if precision == 4 else np.float64
    to, back = (lambda a: to0(np.ascontiguousarray(a, dtype=dt))), (lambda a: np.asarray(back0(a), dtype=np.float64))
    nloen = octahedral(21) if grid == "octahedral" else np.full(44, 96, dtype=np.int32)
    r = et.setup_trans(21, len(nloen), nloen, precision=precision)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")

        def inv(v, d, s):
            gp = to(np.zeros((1, 9, ng)))
            et.inv_trans(r, pspvor=to(v), pspdiv=to(d), pspscalar=to(s), pgp=gp, ldscders=True, ldvorgp=True, lddivgp=True, lduvder=True)
            return back(gp)[0]

        def dirt(g):
            v2, d2, s2 = (to(np.zeros((ns2, 1))) for _ in range(3))
            et.dir_trans(r, pspvor=v2, pspdiv=d2, pspscalar=s2, pgp=to(g[None]))
            return back(v2), back(d2), back(s2)

        e_inv, e_dir = closed_form_errors(inv, dirt, 21, nloen, et.trans_inq(r, "rmu"), et.trans_inq(r, "nasm0"), ns2)
    finally:
        et.trans_release(r)
    tol = 1e-12 if precision == 8 else 2e-5
    assert max(e_inv) < tol and max(e_dir) < tol, (e_inv, e_dir)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [8, 4])
def test_utility_routines_vordiv_to_uv_and_gpnorm(et, dev, precision):
    """VORDIV_TO_UV (k_vd2uv: vd2uv_mod.F90:79-120) and GPNORM_TRANS (k_gpnorm: gpnorm_trans_ctl_mod.F90) on device arrays against the
    oracle and closed forms (tests/common.py::utility_case), NPROMA blocks, both precisions"""
    from oracle.oracle import Oracle as O
    from tests.common import utility_case
    e_uv, e_sb, e_gp = utility_case(et, O, dev, 63, precision, 1000)
    tol = 1e-12 if precision == 8 else 3e-6
    assert e_uv < tol and e_sb < tol and e_gp < tol, (e_uv, e_sb, e_gp)
