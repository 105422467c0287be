"""Shared helpers of the parity tests (oracle = checker, never the thing under test)."""
import numpy as np


def octahedral(nsmax):
    h = nsmax + 1
    return np.array([20 + 4 * i for i in range(h)] + [20 + 4 * i for i in reversed(range(h))], dtype=np.int32)


def n_of_index(nasm0, nsmax, nspec2):
    n_of = np.zeros(nspec2)
    for m in range(nsmax + 1):
        i0 = nasm0[m] - 1
        n_of[i0:i0 + 2 * (nsmax - m + 1)] = np.repeat(np.arange(m, nsmax + 1), 2)
    return n_of


def random_spectrum(rng, nasm0, nsmax, nspec2, nf, zero00):
    """Dense case of SURVEY 8d: U(-0.5,0.5)/(n+1), imag(m=0)=0, (0,0)=0 for vor/div."""
    sp = rng.uniform(-0.5, 0.5, (nspec2, nf)) / (n_of_index(nasm0, nsmax, nspec2)[:, None] + 1.0)
    sp[1:2 * (nsmax + 1):2] = 0.0
    if zero00:
        sp[0] = 0.0
    return sp


def rel_err(a, b, axis=None):
    a, b = np.asarray(a), np.asarray(b)
    if axis is None:
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    return (np.abs(a - b).max(axis=axis) / np.maximum(np.abs(b).max(axis=axis), 1e-300)).max()


def block(g, nproma):
    """(nfld, ngptot) -> (ngpblks, nfld, nproma) zero padded (PGP layout)."""
    nf, ng = g.shape
    nb = (ng - 1) // nproma + 1
    out = np.zeros((nb, nf, nproma))
    for b in range(nb):
        w = min(nproma, ng - b * nproma)
        out[b, :, :w] = g[:, b * nproma:b * nproma + w]
    return out


def unblock(gp, ngptot):
    nb, nf, npr = gp.shape
    return np.concatenate([gp[b] for b in range(nb)], axis=1)[:, :ngptot]


def run_case(et, Oracle, xp, nsmax, nloen, nuv, nsc, flags=None, nproma=None, seed=1, precision=8, setup_kw=None):
    """inverse + direct through the C-ABI (`et`) against the oracle; returns (e_inv, e_dir).
    xp(a) moves a numpy array to the memory space under test and back: (to, back).
    precision=4 runs the fp32 library on float32 copies of the same inputs (the oracle stays fp64)."""
    flags = flags or {}
    to, back = xp
    if precision == 4:
        to0, back0 = xp
        to, back = (lambda a: to0(a.astype(np.float32))), (lambda a: np.asarray(back0(a), dtype=np.float64))
    nloen = np.asarray(nloen, dtype=np.int32)
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision, **(setup_kw or {}))
    try:
        o = Oracle(nsmax, nloen)
        rng = np.random.default_rng(seed)
        vor = random_spectrum(rng, o.nasm0, nsmax, o.nspec2, nuv, True) if nuv else None
        div = random_spectrum(rng, o.nasm0, nsmax, o.nspec2, nuv, True) if nuv else None
        sc = random_spectrum(rng, o.nasm0, nsmax, o.nspec2, nsc, False) if nsc else None
        gref = o.inv_trans(spvor=vor, spdiv=div, spsc=sc, **flags)
        ngp, ng = gref.shape
        npr = nproma or ng
        gp = to(np.zeros(((ng - 1) // npr + 1, ngp, npr)))
        kw = {k2: flags.get(k1, False) for k1, k2 in (("scders", "ldscders"), ("vorgp", "ldvorgp"),
                                                      ("divgp", "lddivgp"), ("uvder", "lduvder"))}
        et.inv_trans(r, pspvor=None if vor is None else to(vor), pspdiv=None if div is None else to(div),
                     pspscalar=None if sc is None else to(sc), pgp=gp, kproma=npr, **kw)
        e_inv = rel_err(unblock(back(gp), ng), gref, axis=1)
        off = (nuv if flags.get("vorgp") else 0) + (nuv if (flags.get("divgp") or flags.get("vorgp")) else 0)
        gdir = gref[off:off + 2 * nuv + nsc]
        v2 = to(np.zeros_like(vor)) if nuv else None
        d2 = to(np.zeros_like(div)) if nuv else None
        s2 = to(np.zeros_like(sc)) if nsc else None
        et.dir_trans(r, pspvor=v2, pspdiv=d2, pspscalar=s2, pgp=to(block(gdir, npr)), kproma=npr)
        vr, dr, sr = o.dir_trans(gdir, nuv=nuv, nsc=nsc)
        e_dir = max(rel_err(back(a), b) for a, b in ((v2, vr), (d2, dr), (s2, sr)) if b is not None)
        return e_inv, e_dir
    finally:
        et.trans_release(r)


def spec_weights(nasm0, nsmax, nspec2):
    """Weights of the spectral inner product of the reference's adjoint tests
    (tests/trans/test_invtrans_adjoint.F90:276-315): 1 for m = 0 real parts, 0 for m = 0 imaginary
    parts, 2 for m > 0."""
    w = np.full(nspec2, 2.0)
    w[0:2 * (nsmax + 1):2] = 1.0
    w[1:2 * (nsmax + 1):2] = 0.0
    return w


def adjoint_case(et, xp, nsmax, nloen, nuv, nsc, nproma=None, seed=7, precision=8):
    """Dot-product tests of INV_TRANSAD and DIR_TRANSAD against INV_TRANS and DIR_TRANS (the
    reference's test_invtrans_adjoint.F90 / test_dirtrans_adjoint.F90): returns the two errors
    |<A x, y> - <x, A* y>| / (|A x| |y|)."""
    to0, back0 = xp
    dt = np.float32 if precision == 4 else np.float64
    to = lambda a: to0(np.ascontiguousarray(a, dtype=dt))
    back = lambda a: np.asarray(back0(a), dtype=np.float64)
    nloen = np.asarray(nloen, dtype=np.int32)
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        nasm0 = et.trans_inq(r, "nasm0")
        w = spec_weights(nasm0, nsmax, ns2)[:, None]
        rng = np.random.default_rng(seed)
        npr = nproma or ng
        nb = (ng - 1) // npr + 1
        nf = 2 * nuv + nsc
        kw = lambda v, d, s: dict(pspvor=v if nuv else None, pspdiv=d if nuv else None, pspscalar=s if nsc else None)
        rs = lambda n: rng.uniform(-1, 1, (ns2, max(n, 1)))
        rg = lambda: block(rng.uniform(-1, 1, (nf, ng)), npr)
        dot_sp = lambda a, b: sum(float((w * back(x) * back(y)).sum()) for x, y in zip(a, b) if x is not None)
        dot_gp = lambda a, b: float((back(a) * back(b)).sum())
        # ---- INV_TRANS vs INV_TRANSAD
        x = [to(rs(nuv)), to(rs(nuv)), to(rs(nsc))]
        for i in (0, 1):  # the reference's forward model ignores vor/div (0,0)
            t = back(x[i]); t[0:2] = 0.0; x[i] = to(t)
        y = to(rg())
        ax = to(np.zeros((nb, nf, npr)))
        et.inv_trans(r, pgp=ax, kproma=npr, **kw(*x))
        aty = [to(np.zeros((ns2, max(nuv, 1)))), to(np.zeros((ns2, max(nuv, 1)))), to(np.zeros((ns2, max(nsc, 1))))]
        et.inv_transad(r, pgp=y, kproma=npr, **kw(*aty))
        sel = lambda l: [l[0] if nuv else None, l[1] if nuv else None, l[2] if nsc else None]
        lhs, rhs = dot_gp(ax, y), dot_sp(sel(x), sel(aty))
        # relative to the size of the terms, not to the (possibly cancelling) sum
        e_inv = abs(lhs - rhs) / np.sqrt(dot_gp(ax, ax) * dot_gp(y, y))
        # ---- DIR_TRANS vs DIR_TRANSAD
        xg = to(rg())
        ys = [to(rs(nuv)), to(rs(nuv)), to(rs(nsc))]
        bx = [to(np.zeros((ns2, max(nuv, 1)))), to(np.zeros((ns2, max(nuv, 1)))), to(np.zeros((ns2, max(nsc, 1))))]
        et.dir_trans(r, pgp=xg, kproma=npr, **kw(*bx))
        bty = to(np.zeros((nb, nf, npr)))
        et.dir_transad(r, pgp=bty, kproma=npr, **kw(*ys))
        lhs, rhs = dot_sp(sel(bx), sel(ys)), dot_gp(xg, bty)
        e_dir = abs(lhs - rhs) / np.sqrt(dot_gp(xg, xg) * dot_gp(bty, bty))
        return e_inv, e_dir
    finally:
        et.trans_release(r)


def legpol_image(o, nsmax, nloen):
    """The reference's Legendre-polynomial file (write_legpol_mod.F90:66-158, non-FLT, no lat-lon part)
    assembled from the ORACLE's panels: 'LEGPOL  ', NSMAX, NDGNH; (NLOEN, NMEN) per northern latitude;
    per wavenumber RPNMA then RPNMS, column-major 8-byte reals."""
    ndgnh = len(nloen) // 2
    parts = [b"LEGPOL  ", np.array([nsmax, ndgnh], dtype="<i4").tobytes(),
             np.stack([np.asarray(nloen)[:ndgnh], np.asarray(o.nmen)[:ndgnh]], axis=1).astype("<i4").tobytes()]
    for m in range(nsmax + 1):
        for sym in (False, True):
            parts.append(np.ascontiguousarray(o.rpnm(m, sym), dtype="<f8").tobytes())  # [col][row] = column-major
    return b"".join(parts)


def legpol_io_case(et, Oracle, xp, tmpdir, nsmax=21, precision=8):
    """CDIO_LEGPOL of SETUP_TRANS: writef / readf / membuf in the reference's byte format."""
    import os
    nloen = octahedral(nsmax)
    o = Oracle(nsmax, nloen)
    ref = legpol_image(o, nsmax, nloen)
    head = 16 + 8 * (len(nloen) // 2)
    fw = os.path.join(str(tmpdir), "legpol_w.bin")
    # writef: integers exact, panels as the oracle's
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision, cdio_legpol="writef", cdlegpolfname=fw)
    mine = open(fw, "rb").read()
    assert len(mine) == len(ref) and mine[:head] == ref[:head]
    a, b = np.frombuffer(mine[head:], dtype="<f8"), np.frombuffer(ref[head:], dtype="<f8")
    assert np.abs(a - b).max() <= (1e-14 if precision == 8 else 1e-6) * np.abs(b).max()
    computed = {(m, s): et.legendre_panel(r, m, s) for m in range(nsmax + 1) for s in (False, True)}
    et.trans_release(r)
    # readf of a file made from the oracle's panels: the library holds exactly those values ...
    fr = os.path.join(str(tmpdir), "legpol_r.bin")
    open(fr, "wb").write(ref)
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision, cdio_legpol="READF", cdlegpolfname=fr)
    for m in range(nsmax + 1):
        for s in (False, True):
            want = o.rpnm(m, s)
            assert np.array_equal(et.legendre_panel(r, m, s), want if precision == 8 else want.astype(np.float32).astype(np.float64))
    et.trans_release(r)
    # ... and transforms with them
    tol = 1e-12 if precision == 8 else 3e-5
    e = run_case(et, Oracle, xp, nsmax, nloen, 1, 2, precision=precision, setup_kw=dict(cdio_legpol="readf", cdlegpolfname=fr))
    assert max(e) < tol, e
    # membuf: the image this library wrote gives back the computed panels bit for bit
    for seg in (mine, np.frombuffer(mine, dtype=np.uint8)):
        r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision, cdio_legpol="membuf", klegpolptr=seg)
        assert all(np.array_equal(et.legendre_panel(r, m, s), v) for (m, s), v in computed.items())
        et.trans_release(r)
    # the reference's checks (read_legpol_mod.F90:88-118)
    bad = bytearray(ref)
    bad[0:8] = b"LEGPOLBF"
    other = octahedral(nsmax).copy()
    other[3] += 4
    other[-4] += 4
    cases = [(dict(klegpolptr=bytes(bad)), nsmax, nloen, "WRONG LABEL"),
             (dict(klegpolptr=ref), nsmax - 1, nloen, "WRONG SPECTRAL TRUNCATION"),
             (dict(klegpolptr=ref), nsmax, octahedral(nsmax + 1), "WRONG NO OF GAUSSIAN LATITUDES"),
             (dict(klegpolptr=ref), nsmax, other, "WRONG NLOEN"),
             (dict(klegpolptr=ref[:len(ref) - 8]), nsmax, nloen, "BYTES_IO_READ FAILED"),
             (dict(klegpolptr=None), nsmax, nloen, "KLEGPOLPTR")]
    for kw, n, nl, msg in cases:
        try:
            et.setup_trans(n, len(nl), nl, precision=precision, cdio_legpol="membuf", **kw)
        except et.TransError as err:
            assert msg in str(err), (msg, str(err))
        else:
            raise AssertionError("no error for " + msg)
    for kw, msg in ((dict(cdio_legpol="readf", cdlegpolfname=os.path.join(str(tmpdir), "absent.bin")), "BYTES_IO_OPEN FAILED"),
                    (dict(cdio_legpol="readf"), "CDLEGPOLFNAME"), (dict(cdio_legpol="mmap"), "UNKNOWN METHOD")):
        try:
            et.setup_trans(nsmax, len(nloen), nloen, **kw)
        except et.TransError as err:
            assert msg in str(err), (msg, str(err))
        else:
            raise AssertionError("no error for " + msg)


def fp32_rows_inverse(o, nsmax, nloen, spec, lats):
    """A plain float32 CPU evaluation of INV_TRANS for one scalar field at the latitudes `lats` (1-based, northern): float32
    Legendre functions (the oracle's SUPOLF values rounded once), float32 dot products (numpy: BLAS SDOT) for
    F_m = sum_n x_n^m P_n^m, and scipy's single-precision complex-to-real FFT -- the arithmetic class of the reference's
    libtrans_sp (SGEMM + FFTW in float, leinv_mod.F90:133-166, tpm_fftw.F90:294-316).  The yardstick of the fp32 library's
    tolerances: {lat: row of NLOEN(lat) float32 values}."""
    import scipy.fft
    nasm0, nmen = o.nasm0, o.nmen
    sp32 = np.asarray(spec, dtype=np.float32)
    out = {}
    for jgl in lats:
        n = int(nloen[jgl - 1])
        F = np.zeros(n // 2 + 1, dtype=np.complex64)
        for m in range(0, int(nmen[jgl - 1]) + 1):
            p = o.legpol(m, jgl).astype(np.float32)
            i0 = nasm0[m] - 1
            re, im = sp32[i0:i0 + 2 * (nsmax - m + 1):2], sp32[i0 + 1:i0 + 2 * (nsmax - m + 1):2]
            F[m] = np.float32(np.dot(re, p)) + 1j * (np.float32(np.dot(im, p)) if m else np.float32(0.0))
        out[jgl] = scipy.fft.irfft(F, n) * np.float32(n)
        assert out[jgl].dtype == np.float32
    return out


def fp32_columns_direct(o, nsmax, nloen, grid, ms):
    """The same for DIR_TRANS of one scalar grid field, for the zonal wavenumbers `ms`: single-precision real-to-complex FFT of
    every latitude row, then x_n^m = sum_lat w P_n^m(mu) F_m(lat) in float32 (m = 0 in double on the float operands, as
    ledir_mod.F90:133-171 does).  {m: complex coefficients for n = m .. nsmax}."""
    import scipy.fft
    ndgl, H = len(nloen), len(nloen) // 2
    nmen, rw = o.nmen, o.rw
    off = np.concatenate([[0], np.cumsum(nloen)])
    g32 = np.asarray(grid, dtype=np.float32)
    mmax = max(ms)
    Fm = np.zeros((ndgl, mmax + 1), dtype=np.complex64)
    for j in range(ndgl):
        n = int(nloen[j])
        f = scipy.fft.rfft(g32[off[j]:off[j + 1]]) / np.float32(n)
        k = min(mmax, int(nmen[j]), n // 2)
        Fm[j, :k + 1] = f[:k + 1]
    out = {}
    for m in ms:
        lat_n = [j for j in range(H) if nmen[j] >= m]  # northern latitudes that carry m (0-based)
        P = np.stack([o.legpol(m, j + 1) for j in lat_n], axis=1)  # (n - m, lat)
        par = (-1.0) ** np.arange(nsmax - m + 1)[:, None]  # P_n^m(-mu) = (-1)^(n-m) P_n^m(mu)
        w = rw[lat_n]
        fn, fs = Fm[lat_n, m], Fm[[ndgl - 1 - j for j in lat_n], m]
        if m == 0:
            A = P.astype(np.float32).astype(np.float64)
            wf = lambda f: (w.astype(np.float32) * f.real.astype(np.float32)).astype(np.float64)
            out[m] = ((A @ wf(fn)) + (par * A) @ wf(fs)).astype(np.float32).astype(np.complex64)
        else:
            A, As = P.astype(np.float32), (par * P).astype(np.float32)
            w32 = w.astype(np.float32)
            xr = A @ (w32 * fn.real) + As @ (w32 * fs.real)
            xi = A @ (w32 * fn.imag) + As @ (w32 * fs.imag)
            out[m] = (xr + 1j * xi).astype(np.complex64)
    return out


def full_size_call_mode2(et, Oracle, nsmax, nlev, nfld, precision=8, tol=1e-11, tol_norm=1e-10, chunk=16, report=None, tol_rms=None,
                         tol_group=None, yardstick=None):
    """BASELINE-size parity through linearity (the reference's benchmark checks norms at the size it times,
    ectrans-benchmark.F90:743-756, 847-871): the call-mode-2 arrays of bench.py / ectrans-benchmark.F90:450-479
    (vor/div x nlev, nfld x nlev 3-D scalars, one surface field => KF = 2 nlev + nfld nlev + 1) are filled with
    c_f x base field, c_f distinct per field, on the dense seed-20251114 spectrum; the oracle transforms the
    1 vor/div pair + 3 scalar base fields, and EVERY field of the HIP result -- every column tile, every batch --
    is compared with c_f x the oracle's field, in both directions, plus the per-field spectral norms.
    Errors are relative to the field maximum: the largest over all elements of all fields (`inv`, `dir`) and the
    largest per-field root-mean-square (`inv_rms`, `dir_rms`; an indexing error would show in both, fp32 rounding at
    N = 2559 only in the tails of the first).  `inv_row` / `dir_n`: the same differences relative to the maximum of the
    field's own LATITUDE ROW (inverse) and of its own TOTAL WAVENUMBER n (direct; the spectrum falls like 1/(n+1)), so that the
    short polar rows and the high-n coefficients are not judged against a maximum set elsewhere (bound: tol_group).
    Returns a dict of the observed errors."""
    import torch
    dev = torch.device("cuda:0")
    tdt = torch.float64 if precision == 8 else torch.float32
    ndt = np.float64 if precision == 8 else np.float32
    nloen = octahedral(nsmax)
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision)
    res = {}
    try:
        o = Oracle(nsmax, nloen, lazy=True)
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        assert (ns2, ng) == (o.nspec2, o.ngptot)
        rng = np.random.default_rng(20251114)
        nb = 3
        rnd = lambda a: a.astype(ndt).astype(np.float64)  # the oracle sees exactly the library's inputs
        V, D = rnd(random_spectrum(rng, o.nasm0, nsmax, ns2, 1, True)), rnd(random_spectrum(rng, o.nasm0, nsmax, ns2, 1, True))
        S = rnd(random_spectrum(rng, o.nasm0, nsmax, ns2, nb, False))
        gref = o.inv_trans(spvor=V, spdiv=D, spsc=S)  # u, v, s0, s1, s2
        gin = rnd(gref)
        vr, dr, sr = o.dir_trans(gin, nuv=1, nsc=nb)
        # distinct coefficients, both signs, O(1)
        cuv = torch.tensor([(1.0 + 0.37 * l / nlev) * (-1) ** l for l in range(nlev)], dtype=tdt, device=dev)
        c3 = torch.tensor([[(0.5 + (v * nlev + l + 1) / (nfld * nlev)) * (-1) ** (v + l) for l in range(nlev)] for v in range(nfld)],
                          dtype=tdt, device=dev)
        c2 = -1.75
        base3 = torch.tensor([[(v * nlev + l) % nb for l in range(nlev)] for v in range(nfld)], device=dev)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=ndt)).to(dev)
        tV, tD, tS = t(V[:, 0]), t(D[:, 0]), t(S.T)  # tS: (nb, ns2)
        spvor, spdiv = tV[:, None] * cuv[None, :], tD[:, None] * cuv[None, :]
        spsc3a = torch.empty((nfld, ns2, nlev), dtype=tdt, device=dev)
        for v in range(nfld):
            spsc3a[v] = tS[base3[v]].T * c3[v][None, :]
        spsc2 = (c2 * tS[0])[:, None].contiguous()
        gpuv = torch.zeros((1, 2, nlev, ng), dtype=tdt, device=dev)
        gp3a = torch.zeros((1, nfld, nlev, ng), dtype=tdt, device=dev)
        gp2 = torch.zeros((1, 1, ng), dtype=tdt, device=dev)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()  # the library sizes its field batches from the free HBM
        et.inv_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
        torch.cuda.synchronize()

        def group_max(x, grp):
            """x (k, n) >= 0 -> (k, ngroups): maxima over the entries of equal group index (grp = (index, count)): the latitude
            row of a grid point, the total wavenumber of a spectral entry"""
            out = torch.zeros((x.shape[0], grp[1]), dtype=x.dtype, device=x.device)
            return out.scatter_reduce_(1, grp[0][None, :].expand(x.shape[0], -1), x, "amax", include_self=True)

        gworst = [0.0]

        def cmp_rows(got, coef, refs, idx, grp=None):
            """got (nf, n) [any strides], coef (nf,), refs (k, n), idx (nf,): max_f max|got_f - c_f ref_idx(f)| / max|c_f ref_idx(f)|
            in chunks of `chunk` fields with one temporary"""
            rmax = refs.abs().amax(dim=1)
            gmax = group_max(refs.abs(), grp) if grp is not None else None
            worst = 0.0
            for i in range(0, got.shape[0], chunk):
                c, ix = coef[i:i + chunk], idx[i:i + chunk]
                d = refs[ix]
                d.mul_(c[:, None]).sub_(got[i:i + chunk]).abs_()
                if grp is not None:
                    gworst[0] = max(gworst[0], float((group_max(d, grp) / (c.abs()[:, None] * gmax[ix])).max()))
                worst = max(worst, float((d.amax(dim=1) / (c.abs() * rmax[ix])).max()))
                rms[0] = max(rms[0], float((d.square_().mean(dim=1).sqrt() / (c.abs() * rmax[ix])).max()))
                del d
            return worst

        rms = [0.0]

        zero = lambda n: torch.zeros(n, dtype=torch.long, device=dev)
        tg = t(gref)  # (5, ng)
        rows = (torch.from_numpy(np.repeat(np.arange(len(nloen), dtype=np.int64), nloen)).to(dev), len(nloen))
        e_inv = max(cmp_rows(gpuv[0, 0], cuv, tg[0:1], zero(nlev), rows), cmp_rows(gpuv[0, 1], cuv, tg[1:2], zero(nlev), rows),
                    cmp_rows(gp2[0], torch.tensor([c2], dtype=tdt, device=dev), tg[2:3], zero(1), rows))
        for v in range(nfld):
            e_inv = max(e_inv, cmp_rows(gp3a[0, v], c3[v], tg[2:], base3[v], rows))
        res["inv"], res["inv_rms"], res["inv_row"] = e_inv, rms[0], gworst[0]
        rms[0] = gworst[0] = 0.0
        if yardstick:
            # fp32 library: the HIP rows of the surface field (c2 x base field 0) against a plain float32 CPU evaluation of the
            # same rows -- both measured against the fp64 oracle, relative to the field maximum
            off = np.concatenate([[0], np.cumsum(nloen)])
            fmax = float(np.abs(gref[2]).max())
            # the sampled rows plus the three northern rows on which the HIP result of this field is worst
            dsurf = (gp2[0, 0].double() / c2 - tg[2].double()).abs_()
            per_lat = group_max(dsurf[None], rows)[0][:len(nloen) // 2]
            worst_lats = [int(j) + 1 for j in torch.topk(per_lat, 3).indices.cpu().numpy()]
            res["inv_fp32_worst_rows"] = worst_lats
            del dsurf
            y32 = fp32_rows_inverse(o, nsmax, nloen, S[:, 0], sorted(set(list(yardstick["lats"]) + worst_lats)))
            worst_ratio, e32s, ehs = 0.0, [], []
            for jgl, row32 in y32.items():
                ref = gref[2][off[jgl - 1]:off[jgl]]
                hip = gp2[0, 0, off[jgl - 1]:off[jgl]].double().cpu().numpy() / c2
                e32, eh = float(np.abs(row32 - ref).max() / fmax), float(np.abs(hip - ref).max() / fmax)
                e32s.append(e32), ehs.append(eh)
                worst_ratio = max(worst_ratio, eh / max(e32, 4 * np.finfo(np.float32).eps))
            res["inv_fp32_cpu"], res["inv_fp32_hip"], res["inv_fp32_ratio"] = max(e32s), max(ehs), worst_ratio
        # ---- direct: exact images of the oracle's grid fields in, every spectral field against c_f x oracle
        tgi = t(gin)
        for l0 in range(0, nlev, chunk):
            sl = slice(l0, min(nlev, l0 + chunk))
            gpuv[0, 0, sl] = cuv[sl, None] * tgi[0][None, :]
            gpuv[0, 1, sl] = cuv[sl, None] * tgi[1][None, :]
            for v in range(nfld):
                gp3a[0, v, sl] = c3[v, sl, None] * tgi[2 + base3[v, sl]]
        gp2[0, 0] = c2 * tgi[2]
        for a in (spvor, spdiv, spsc3a, spsc2):
            a.zero_()
        del tg, tgi
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        et.dir_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
        torch.cuda.synchronize()
        tvr, tdr, tsr = t(vr[:, 0]), t(dr[:, 0]), t(sr.T)
        byn = (torch.from_numpy(n_of_index(o.nasm0, nsmax, ns2).astype(np.int64)).to(dev), nsmax + 1)
        e_dir = max(cmp_rows(spvor.T, cuv, tvr[None], zero(nlev), byn), cmp_rows(spdiv.T, cuv, tdr[None], zero(nlev), byn),
                    cmp_rows(spsc2.T, torch.tensor([c2], dtype=tdt, device=dev), tsr[0:1], zero(1), byn))
        for v in range(nfld):
            e_dir = max(e_dir, cmp_rows(spsc3a[v].T, c3[v], tsr, base3[v], byn))
        res["dir"], res["dir_rms"], res["dir_n"] = e_dir, rms[0], gworst[0]
        if yardstick:
            smax = float(np.abs(sr[:, 0]).max())
            hip2 = spsc2[:, 0].double().cpu().numpy() / c2
            # ... plus the three zonal wavenumbers on which the HIP result of this field is worst
            m_of = np.zeros(ns2, dtype=np.int64)
            for m in range(nsmax + 1):
                m_of[o.nasm0[m] - 1:o.nasm0[m] - 1 + 2 * (nsmax - m + 1)] = m
            per_m = np.zeros(nsmax + 1)
            np.maximum.at(per_m, m_of, np.abs(hip2 - sr[:, 0]))
            worst_ms = [int(m) for m in np.argsort(per_m)[-3:]]
            res["dir_fp32_worst_ms"] = worst_ms
            y32 = fp32_columns_direct(o, nsmax, nloen, gin[2], sorted(set(list(yardstick["ms"]) + worst_ms)))
            worst_ratio, e32s, ehs = 0.0, [], []
            for m, x32 in y32.items():
                i0 = o.nasm0[m] - 1
                ref = sr[i0:i0 + 2 * (nsmax - m + 1):2, 0] + 1j * sr[i0 + 1:i0 + 2 * (nsmax - m + 1):2, 0]
                hip = hip2[i0:i0 + 2 * (nsmax - m + 1):2] + 1j * hip2[i0 + 1:i0 + 2 * (nsmax - m + 1):2]
                e32, eh = float(np.abs(x32 - ref).max() / smax), float(np.abs(hip - ref).max() / smax)
                e32s.append(e32), ehs.append(eh)
                worst_ratio = max(worst_ratio, eh / max(e32, 4 * np.finfo(np.float32).eps))
            res["dir_fp32_cpu"], res["dir_fp32_hip"], res["dir_fp32_ratio"] = max(e32s), max(ehs), worst_ratio
        # ---- spectral norms of every field against |c_f| x the oracle's norm (north star: <= 1e-10)
        nv, nd, nsr = o.specnorm(vr)[0], o.specnorm(dr)[0], o.specnorm(sr)
        e_n = max(np.abs(et.specnorm(r, spvor) / (cuv.abs().cpu().numpy() * nv) - 1.0).max(),
                  np.abs(et.specnorm(r, spdiv) / (cuv.abs().cpu().numpy() * nd) - 1.0).max(),
                  abs(et.specnorm(r, spsc2)[0] / (abs(c2) * nsr[0]) - 1.0))
        for v in range(nfld):
            want = c3[v].abs().cpu().numpy() * nsr[base3[v].cpu().numpy()]
            e_n = max(e_n, np.abs(et.specnorm(r, spsc3a[v]) / want - 1.0).max())
        res["norm"] = float(e_n)
        # the reference benchmark's own criterion on the dense case: |norm(x0) / norm(dir(inv(x0))) - 1| -- here x0 and
        # the direct transform of the oracle's inverse; on octahedral grids this carries the grid's truncation error
        res["spectral_norm_rel_error"] = float(abs(o.specnorm(S[:, :1])[0] / nsr[0] - 1.0))
        if report is not None:
            report.update(res)
        assert res["inv"] < tol and res["dir"] < tol and res["norm"] < tol_norm, res
        assert tol_rms is None or (res["inv_rms"] < tol_rms and res["dir_rms"] < tol_rms), res
        assert tol_group is None or (res["inv_row"] < tol_group and res["dir_n"] < tol_group), res
        assert not yardstick or (res["inv_fp32_ratio"] <= yardstick["factor"] and res["dir_fp32_ratio"] <= yardstick["factor"]), res
        return res
    finally:
        et.trans_release(r)
        torch.cuda.empty_cache()


def adjoint_matrix_case(et, Oracle, xp, nsmax=10, seed=3, precision=8):
    """INV_TRANSAD / DIR_TRANSAD against an ORACLE adjoint (the dot-product test above only shows that the HIP forward
    and adjoint are a pair).  The dense matrices of the oracle's forward transforms are formed column by column on a tiny
    grid -- A: (vor, div, scalar) -> (u, v, scalar) for INV_TRANS, B: (u, v, scalar) -> (vor, div, scalar) for DIR_TRANS --
    and transposed with the inner products of the reference's adjoint tests (test_invtrans_adjoint.F90:243-315: plain
    sum over grid points; SPECNORM weights w in spectral space):  A* = W^+ A^T,  B* = B^T W.  Returns the two errors
    relative to the largest element."""
    to0, back0 = xp
    dt = np.float32 if precision == 4 else np.float64
    to = lambda a: to0(np.ascontiguousarray(a, dtype=dt))
    back = lambda a: np.asarray(back0(a), dtype=np.float64)
    nloen = octahedral(nsmax)
    o = Oracle(nsmax, nloen)
    ns2, ng = o.nspec2, o.ngptot
    w = spec_weights(o.nasm0, nsmax, ns2)
    winv = np.where(w > 0, 1.0 / np.maximum(w, 1e-300), 0.0)
    eye_s, z_s = np.eye(ns2), np.zeros((ns2, ns2))
    # A: columns = (vor basis | div basis | scalar basis); rows = (u | v | scalar) grid points
    g = o.inv_trans(spvor=np.concatenate([eye_s, z_s], axis=1), spdiv=np.concatenate([z_s, eye_s], axis=1), spsc=eye_s)
    nuv = 2 * ns2  # oracle grid field order: u(nuv) v(nuv) scalars(ns2)
    A = np.zeros((3 * ng, 3 * ns2))
    A[0:ng, 0:2 * ns2] = g[0:nuv].T          # u from (vor | div)
    A[ng:2 * ng, 0:2 * ns2] = g[nuv:2 * nuv].T  # v
    A[2 * ng:, 2 * ns2:] = g[2 * nuv:].T     # scalar
    # B: columns = (u basis | v basis | scalar basis) grid; rows = (vor | div | scalar) spectral
    eye_g, z_g = np.eye(ng), np.zeros((ng, ng))
    gp = np.concatenate([np.concatenate([eye_g, z_g]), np.concatenate([z_g, eye_g]), eye_g])  # u(2ng) v(2ng) scalars(ng)
    vr, dr, sr = o.dir_trans(gp, nuv=2 * ng, nsc=ng)
    B = np.zeros((3 * ns2, 3 * ng))
    B[0:ns2, 0:2 * ng] = vr
    B[ns2:2 * ns2, 0:2 * ng] = dr
    B[2 * ns2:, 2 * ng:] = sr
    W3, Winv3 = np.tile(w, 3), np.tile(winv, 3)
    rng = np.random.default_rng(seed)
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision)
    try:
        nf = 2  # two independent right-hand sides per field kind
        # ---- INV_TRANSAD: y on the grid (u, v, scalar) -> (vor, div, scalar) = W^+ A^T y
        y = rng.uniform(-1, 1, (3, nf, ng))
        want = Winv3[:, None] * (A.T @ y.transpose(0, 2, 1).reshape(3 * ng, nf))
        pgp = to(np.concatenate([y[0], y[1], y[2]])[None])  # PGP fields: u(nf) v(nf) scalars(nf)
        v, d, s = (to(np.zeros((ns2, nf))) for _ in range(3))
        et.inv_transad(r, pspvor=v, pspdiv=d, pspscalar=s, pgp=pgp)
        got = np.concatenate([back(v), back(d), back(s)])
        e_inv = np.abs(got - want).max() / np.abs(want).max()
        # ---- DIR_TRANSAD: y in spectral space (vor, div, scalar) -> grid (u, v, scalar) = B^T W y
        ys = rng.uniform(-1, 1, (3, ns2, nf))
        want = B.T @ (W3[:, None] * ys.reshape(3 * ns2, nf))
        gout = to(np.zeros((1, 3 * nf, ng)))
        et.dir_transad(r, pspvor=to(ys[0]), pspdiv=to(ys[1]), pspscalar=to(ys[2]), pgp=gout)
        got = back(gout)[0].reshape(3, nf, ng).transpose(0, 2, 1).reshape(3 * ng, nf)
        e_dir = np.abs(got - want).max() / np.abs(want).max()
        return e_inv, e_dir
    finally:
        et.trans_release(r)


def adjoint_options_case(et, Oracle, xp, nsmax=6, flags=None, seed=5, nproma=None, precision=8):
    """INV_TRANSAD with LDSCDERS / LDVORGP / LDDIVGP / LDUVDER against the ORACLE: the dense matrix A of the oracle's
    forward INV_TRANS with the same options -- columns (vor | div | scalar) basis vectors, rows every grid field the
    options produce, in INV_TRANS's order (inv_trans.h:66-76) -- and A* = W^+ A^T (weights: spec_weights).  Returns the
    error relative to the largest element."""
    flags = flags or {}
    to0, back0 = xp
    dt = np.float32 if precision == 4 else np.float64
    to = lambda a: to0(np.ascontiguousarray(a, dtype=dt))
    back = lambda a: np.asarray(back0(a), dtype=np.float64)
    nloen = octahedral(nsmax)
    o = Oracle(nsmax, nloen)
    ns2, ng = o.nspec2, o.ngptot
    w = spec_weights(o.nasm0, nsmax, ns2)
    winv = np.where(w > 0, 1.0 / np.maximum(w, 1e-300), 0.0)
    eye, z = np.eye(ns2), np.zeros((ns2, ns2))
    # forward on the basis: nuv = 2 ns2 wind "fields" (vor basis | div basis), nsc = ns2 scalar fields
    g = o.inv_trans(spvor=np.concatenate([eye, z], axis=1), spdiv=np.concatenate([z, eye], axis=1), spsc=eye, **flags)
    nuv, nsc = 2 * ns2, ns2
    lv = flags.get("vorgp", False)
    ld = flags.get("divgp", False) or lv
    # grid field groups in INV_TRANS's order with their widths (in basis fields)
    groups = ([("vor", nuv)] if lv else []) + ([("div", nuv)] if ld else []) + [("u", nuv), ("v", nuv), ("sc", nsc)] + \
        ([("ns", nsc)] if flags.get("scders") else []) + ([("uew", nuv), ("vew", nuv)] if flags.get("uvder") else []) + \
        ([("scew", nsc)] if flags.get("scders") else [])
    assert g.shape[0] == sum(n for _, n in groups)
    # one test field of each kind: A maps (vor, div, scalar) coefficients to the stacked grid fields
    blocks, off = [], 0
    for name, n in groups:
        blk = np.zeros((ng, 3 * ns2))
        if n == nuv:
            blk[:, :2 * ns2] = g[off:off + n].T
        else:
            blk[:, 2 * ns2:] = g[off:off + n].T
        blocks.append(blk)
        off += n
    A = np.concatenate(blocks)  # (ngroups * ng, 3 ns2)
    rng = np.random.default_rng(seed)
    y = rng.uniform(-1, 1, (len(groups), ng))
    want = np.tile(winv, 3)[:, None] * (A.T @ y.reshape(-1, 1))
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision)
    try:
        npr = nproma or ng
        v, d, s = (to(np.zeros((ns2, 1))) for _ in range(3))
        et.inv_transad(r, pspvor=v, pspdiv=d, pspscalar=s, pgp=to(block(y, npr)), kproma=npr,
                       ldscders=flags.get("scders", False), ldvorgp=lv, lddivgp=ld, lduvder=flags.get("uvder", False))
        got = np.concatenate([back(v), back(d), back(s)])
        return float(np.abs(got - want).max() / np.abs(want).max())
    finally:
        et.trans_release(r)


def adjoint_options_call_mode2_case(et, xp, nsmax=8, seed=9, nproma=53):
    """INV_TRANSAD with all options through the call-mode-2 arrays (PGPUV, PGP3A, PGP2 in INV_TRANS's layouts:
    variables [vor][div] u v [u_EW v_EW]; value / N-S / E-W blocks of the scalar arrays, trltog_mod.F90:632-690) must
    give what the same inputs give through one PGP array.  Returns the largest difference."""
    to, back = xp
    nloen = octahedral(nsmax)
    r = et.setup_trans(nsmax, len(nloen), nloen)
    try:
        ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
        nlev, nvar = 2, 2
        nsc = 1 + nvar * nlev  # sc2 (1 field) + sc3a
        nb = (ng - 1) // nproma + 1
        rng = np.random.default_rng(seed)
        gpuv = rng.uniform(-1, 1, (nb, 6, nlev, nproma))       # vor div u v uew vew
        gp3a = rng.uniform(-1, 1, (nb, 3 * nvar, nlev, nproma))  # [value vars][N-S vars][E-W vars]
        gp2 = rng.uniform(-1, 1, (nb, 3, nproma))             # value, N-S, E-W of the one surface field
        # the same fields in PGP order: vor div u v | scalars (sc2, then sc3a var-major) | N-S | uew vew | E-W
        sc = lambda k: [gp2[:, k:k + 1]] + [gp3a[:, k * nvar + v] for v in range(nvar)]
        parts = [gpuv[:, 0], gpuv[:, 1], gpuv[:, 2], gpuv[:, 3]] + sc(0) + sc(1) + [gpuv[:, 4], gpuv[:, 5]] + sc(2)
        pgp = np.concatenate(parts, axis=1)
        outs = []
        for mode in (1, 2):
            v, d = to(np.zeros((ns2, nlev))), to(np.zeros((ns2, nlev)))
            kw = dict(ldscders=True, ldvorgp=True, lddivgp=True, lduvder=True, kproma=nproma)
            if mode == 1:
                s = to(np.zeros((ns2, nsc)))
                et.inv_transad(r, pspvor=v, pspdiv=d, pspscalar=s, pgp=to(pgp), **kw)
                outs.append(np.concatenate([back(v), back(d), back(s)], axis=1))
            else:
                s2, s3 = to(np.zeros((ns2, 1))), to(np.zeros((nvar, ns2, nlev)))
                et.inv_transad(r, pspvor=v, pspdiv=d, pspsc2=s2, pspsc3a=s3, pgpuv=to(gpuv), pgp3a=to(gp3a), pgp2=to(gp2), **kw)
                outs.append(np.concatenate([back(v), back(d), back(s2)] + [back(s3)[k] for k in range(nvar)], axis=1))
        return float(np.abs(outs[0] - outs[1]).max() / np.abs(outs[0]).max())
    finally:
        et.trans_release(r)


def staging_pool_case(et, Oracle, tol, combos=((3, None), (1, 37), (5, None), (2, 100), (5, 37))):
    """The device-side staging buffers of host arrays are kept between calls (csrc/emi_stage.h): calls with more,
    fewer and again more fields, with and without NPROMA padding, must each see only their own data -- a reused, larger
    buffer holds the previous call's values behind the part this call fills."""
    N = 8
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    o = Oracle(N, nloen)
    ns2, ng = o.nspec2, o.ngptot
    rng = np.random.default_rng(5)
    try:
        for nf, npr in combos:
            npr = npr or ng
            sp = random_spectrum(rng, o.nasm0, N, ns2, nf, False)
            nb = (ng - 1) // npr + 1
            gp = np.full((nb, nf, npr), -7.25)
            et.inv_trans(r, pspscalar=sp, pgp=gp, kproma=npr)
            got = np.concatenate([gp[b] for b in range(nb)], axis=1)[:, :ng]
            assert rel_err(got, o.inv_trans(spsc=sp), axis=1) < tol
            if nb * npr > ng:
                assert np.all(gp[-1, :, ng - (nb - 1) * npr:] == -7.25)
            back = np.full((ns2, nf), np.nan)
            et.dir_trans(r, pspscalar=back, pgp=gp, kproma=npr)
            assert np.abs(back - sp).max() < 1e3 * tol
    finally:
        et.trans_release(r)


def direct_spectral_tiles_case(et, Oracle, xp, tol, nsmax=39, precision=8, nf=200, nlev=65, nvar=3, mbs=(0, 128)):
    """Many plain-copy fields: PSPSCALAR with 200 fields (several 64-field column tiles of the Legendre kernels), then
    call-mode-2 arrays with an odd level count (65 levels x 3 variables + winds: tiles that straddle variables, field
    pairs that do not start on a 16-byte boundary, a tile mixing winds and scalars), in one batch and in batches.  The
    imaginary parts of zonal wavenumber 0 are poisoned: they must not be read (prfi1b_mod.F90).  (Written for the
    experiment in which k_leg_inv read such tiles in the caller's arrays; kept as a parity case.)"""
    to, back = xp
    nloen = octahedral(nsmax)
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision)
    o = Oracle(nsmax, nloen)
    ns2, ng = o.nspec2, o.ngptot
    rng = np.random.default_rng(77)
    dt = np.float64 if precision == 8 else np.float32
    try:
        sp = random_spectrum(rng, o.nasm0, nsmax, ns2, nf, False).astype(dt)
        ref = o.inv_trans(spsc=sp.astype(np.float64))
        bad = sp.copy()
        bad[1:2 * (nsmax + 1):2] = 1e30  # imag(m = 0): defined to be zero, never read (prfi1b_mod.F90)
        gp = to(np.zeros((1, nf, ng), dtype=dt))
        et.inv_trans(r, pspscalar=to(bad), pgp=gp)
        assert rel_err(np.asarray(back(gp), dtype=np.float64)[0], ref, axis=1) < tol
        vor, div = (random_spectrum(rng, o.nasm0, nsmax, ns2, nlev, True).astype(dt) for _ in range(2))
        s3 = random_spectrum(rng, o.nasm0, nsmax, ns2, nlev * nvar, False).astype(dt)
        s2 = random_spectrum(rng, o.nasm0, nsmax, ns2, 1, False).astype(dt)
        sc3a = np.ascontiguousarray(s3.reshape(ns2, nvar, nlev).transpose(1, 0, 2))  # C view of PSPSC3A(nlev, nspec2, nvar)
        ref = o.inv_trans(spvor=vor.astype(np.float64), spdiv=div.astype(np.float64),
                          spsc=np.concatenate([s3, s2], axis=1).astype(np.float64))
        for mb in mbs:
            et.set_max_batch(mb)
            gpuv = to(np.zeros((1, 2, nlev, ng), dtype=dt))
            gp3a = to(np.zeros((1, nvar, nlev, ng), dtype=dt))
            gp2 = to(np.zeros((1, 1, ng), dtype=dt))
            et.inv_trans(r, pspvor=to(vor), pspdiv=to(div), pspsc3a=to(sc3a), pspsc2=to(s2), pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
            got = np.concatenate([np.asarray(back(gpuv), dtype=np.float64)[0].reshape(2 * nlev, ng),
                                  np.asarray(back(gp3a), dtype=np.float64)[0].reshape(nvar * nlev, ng),
                                  np.asarray(back(gp2), dtype=np.float64)[0]], axis=0)
            assert rel_err(got, ref, axis=1) < tol, mb
    finally:
        et.set_max_batch(0)
        et.trans_release(r)


def closed_form_case(nsmax, nloen, rmu, nasm0, nspec2, ra=6371229.0):
    """Fields whose transforms are known in closed form -- numbers that no restatement of the reference shares:
    solid-body rotation (zeta = 2U/a sin(theta) = 2U/(a sqrt 3) P_1^0) plus one vorticity harmonic c P_3^2 e^{2 i lambda}, one
    divergence harmonic d P_2^1 e^{i lambda} and a scalar s00 + g P_3^2 e^{2 i lambda}, with the reference's normalisation
    (1/2 int P^2 dmu = 1, no Condon-Shortley phase: P_1^0 = sqrt(3) mu, P_2^1 = sqrt(15/2) mu cos(theta),
    P_3^2 = sqrt(105/8) mu cos^2(theta)) and its conventions: a real field is x_0 + 2 Re sum_{m>0}; psi = -a^2/(n(n+1)) zeta,
    chi likewise; u = -(1/a) dpsi/dtheta + (1/(a cos)) dchi/dlambda, v = (1/(a cos)) dpsi/dlambda + (1/a) dchi/dtheta.
    Returns (vor, div, sc) spectral columns of shape (nspec2, 1) and the nine grid fields INV_TRANS yields with
    LDVORGP, LDDIVGP, LDSCDERS, LDUVDER: vor, div, u, v, s, ds/dtheta / a, du/dlambda / (a cos), dv/dlambda / (a cos),
    ds/dlambda / (a cos) -- pins vdtuv_mod.F90:97-143, spnsde_mod.F90:95-114, fsc_mod.F90:138-187 (inverse) and, read the
    other way, uvtvd_mod.F90:91-139 (direct)."""
    a = float(ra)
    U, c, d, g, s00 = 37.5, 3.0e-5 - 1.25e-5j, -2.0e-5 + 0.5e-5j, 1.5 + 0.75j, 250.0
    vor, div, sc = (np.zeros((nspec2, 1)) for _ in range(3))

    def put(arr, m, n, z):
        i = nasm0[m] - 1 + 2 * (n - m)  # NASM0 is 1-based
        arr[i, 0], arr[i + 1, 0] = z.real, (z.imag if m else 0.0)

    put(vor, 0, 1, complex(2.0 * U / (a * np.sqrt(3.0))))
    put(vor, 2, 3, c)
    put(div, 1, 2, d)
    put(sc, 0, 0, complex(s00))
    put(sc, 2, 3, g)
    lam = np.concatenate([2.0 * np.pi * np.arange(n) / n for n in nloen])
    mu = np.concatenate([np.full(n, rmu[j]) for j, n in enumerate(nloen)])
    ct = np.sqrt(1.0 - mu * mu)
    p32, dp32 = np.sqrt(105.0 / 8.0) * mu * ct * ct, np.sqrt(105.0 / 8.0) * (1.0 - 3.0 * mu * mu)
    p21, dp21 = np.sqrt(7.5) * mu * ct, np.sqrt(7.5) * (1.0 - 2.0 * mu * mu) / ct
    e2, e1, eg = c * np.exp(2j * lam), d * np.exp(1j * lam), g * np.exp(2j * lam)
    re2 = lambda z: 2.0 * np.real(z)
    zeta_h, dvg = re2(e2) * p32, re2(e1) * p21
    # psi = -a U mu - a^2/12 zeta_h, chi = -a^2/6 D
    dpsi_dmu, dpsi_dlam = -a * U - (a * a / 12.0) * re2(e2) * dp32, -(a * a / 12.0) * re2(2j * e2) * p32
    dchi_dmu, dchi_dlam = -(a * a / 6.0) * re2(e1) * dp21, -(a * a / 6.0) * re2(1j * e1) * p21
    u = -(ct / a) * dpsi_dmu + dchi_dlam / (a * ct)
    v = dpsi_dlam / (a * ct) + (ct / a) * dchi_dmu
    du_dlam = (ct / a) * (a * a / 12.0) * re2(2j * e2) * dp32 + (-(a * a / 6.0) * re2(-e1) * p21) / (a * ct)
    dv_dlam = (-(a * a / 12.0) * re2(-4.0 * e2) * p32) / (a * ct) + (ct / a) * (-(a * a / 6.0)) * re2(1j * e1) * dp21
    s = s00 + re2(eg) * p32
    gp = np.stack([2.0 * U / a * mu + zeta_h, dvg, u, v, s, (ct / a) * re2(eg) * dp32, du_dlam / (a * ct), dv_dlam / (a * ct),
                   re2(2j * eg) * p32 / (a * ct)])
    return vor, div, sc, gp


def closed_form_errors(inv, dirt, nsmax, nloen, rmu, nasm0, nspec2, ra=6371229.0):
    """inv(vor, div, sc) -> (9, ngptot) grid fields with all derivative options; dirt(gp3) with gp3 = (u, v, s) ->
    (vor, div, sc).  Returns the largest error of every grid field relative to its own maximum and of the three spectral
    fields relative to their largest coefficient."""
    vor, div, sc, gp = closed_form_case(nsmax, nloen, rmu, nasm0, nspec2, ra)
    got = inv(vor, div, sc)
    e_inv = [float(np.abs(got[i] - gp[i]).max() / np.abs(gp[i]).max()) for i in range(9)]
    v2, d2, s2 = dirt(gp[2:5])
    e_dir = [float(np.abs(x - y).max() / np.abs(y).max()) for x, y in ((v2, vor), (d2, div), (s2, sc))]
    return e_inv, e_dir


def utility_case(et, oracle_cls, dev, nsmax=21, precision=8, nproma=37):
    """VORDIV_TO_UV and GPNORM_TRANS of the product against the oracle (random dense fields) and against closed forms: solid-body
    rotation (vor = 2U/(a sqrt 3) P_1^0 => U = u cos(theta) = 2U/3 P_0 - 2U/(3 sqrt 5) P_2, V = 0), a constant field and the
    zonal harmonic P_2^0 (mean 0, maximum sqrt(5) P_2(mu_1) on the first Gaussian latitude).  Returns the relative errors."""
    to, back = dev
    dt = np.float64 if precision == 8 else np.float32
    nloen = octahedral(nsmax)
    o = oracle_cls(nsmax, nloen)
    rng = np.random.default_rng(5)
    vor = random_spectrum(rng, o.nasm0, nsmax, o.nspec2, 3, True)
    div = random_spectrum(rng, o.nasm0, nsmax, o.nspec2, 3, True)
    ur, vr = o.vordiv_to_uv(vor, div)
    u, v = et.vordiv_to_uv(to(vor.astype(dt)), to(div.astype(dt)), nsmax)
    e_uv = max(rel_err(back(u), ur), rel_err(back(v), vr))
    U, a = 30.0, 6371229.0
    sb = np.zeros((o.nspec2, 1))
    sb[o.nasm0[0] - 1 + 2, 0] = 2 * U / (a * np.sqrt(3.0))
    u, v = (back(x) for x in et.vordiv_to_uv(to(sb.astype(dt)), to(np.zeros_like(sb).astype(dt)), nsmax))
    want = np.zeros_like(sb)
    want[o.nasm0[0] - 1, 0], want[o.nasm0[0] - 1 + 4, 0] = 2 * U / 3, -2 * U / (3 * np.sqrt(5.0))
    e_sb = max(np.abs(u - want).max(), np.abs(v).max()) / U
    # grid-point norms
    r = et.setup_trans(nsmax, len(nloen), nloen, precision=precision)
    g = o.inv_trans(spvor=vor, spdiv=div)
    sc = np.zeros((o.nspec2, 2))
    sc[o.nasm0[0] - 1, 0], sc[o.nasm0[0] - 1 + 4, 1] = 3.5, 1.0
    g = np.concatenate([g, o.inv_trans(spsc=sc)])
    ar, mnr, mxr = o.gpnorm(g)
    assert abs(ar[-2] - 3.5) < 1e-13 and abs(ar[-1]) < 1e-13 and abs(mxr[-1] - np.sqrt(5.0) * (3 * o.rmu[0] ** 2 - 1) / 2) < 1e-12  # the oracle itself
    ng, nf = o.ngptot, g.shape[0]
    nb = (ng - 1) // nproma + 1
    pad = np.zeros((nf, nb * nproma))
    pad[:, :ng] = g
    pgp = np.ascontiguousarray(pad.reshape(nf, nb, nproma).transpose(1, 0, 2)).astype(dt)
    a1, mn1, mx1 = et.gpnorm_trans(r, to(pgp), kproma=nproma)
    scale = np.abs(g).max(axis=1)
    e_gp = max((np.abs(a1 - ar) / scale).max(), (np.abs(mn1 - mnr) / scale).max(), (np.abs(mx1 - mxr) / scale).max())
    # LDAVE_ONLY: the caller's extrema come back (one task), only the averages are computed; fewer fields than the array holds
    a2, mn2, mx2 = et.gpnorm_trans(r, to(pgp), kfields=2, kproma=nproma, ldave_only=True, pmin=[-1.0, -2.0], pmax=[3.0, 4.0])
    assert list(mn2) == [-1.0, -2.0] and list(mx2) == [3.0, 4.0] and np.abs(a2 - a1[:2]).max() == 0.0
    et.trans_release(r)
    return e_uv, e_sb, e_gp
