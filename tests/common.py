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
