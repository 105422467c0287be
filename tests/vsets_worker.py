"""One task of the V-set tests (tests/test_dist_gloo.py::test_vsets_*, tests/test_gpu_shims.py): NPRTRW x NPRTRV tasks.

Spectral arrays hold the wavenumbers of the task's W-set and the fields of its V-set (KVSETUV / KVSETSC / KVSETSC2 / KVSETSC3A,
inv_trans.F90:212-300); grid arrays hold ALL fields on the task's own latitudes.  Every local piece is checked against the
oracle's global result, in both call modes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402
import faulthandler  # noqa: E402
faulthandler.dump_traceback_later(int(os.environ.get("EMI_TEST_WATCHDOG", "840")), exit=True)  # a deadlock between the tasks shows its stack

import ectrans_amd as et  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.common import octahedral, random_spectrum, rel_err  # noqa: E402

PREC = int(os.environ.get("EMI_TEST_PRECISION", "8"))
DT = np.float32 if PREC == 4 else np.float64


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nprv = int(os.environ["EMI_TEST_NPRTRV"])
    nprw = world // nprv
    dist.init_process_group("gloo", rank=rank, world_size=world)
    on_gpu = os.environ.get("EMI_TEST_DEVICE", "cpu") == "cuda"
    if on_gpu:
        import torch
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=DT)).to("cuda:0")
        back = lambda t: t.cpu().numpy().astype(np.float64)
        et.setup_trans0(kmax_resol=2, kprtrw=nprw, kprtrv=nprv, myproc=rank + 1, device=0)
    else:
        to, back = (lambda a: np.ascontiguousarray(a, dtype=DT)), (lambda a: np.asarray(a, dtype=np.float64))
        et._use_library_for_tests(os.path.join(ROOT, "tests", "emu", "libectrans_mi_emu.so"))
        et.setup_trans0(kmax_resol=2, kprtrw=nprw, kprtrv=nprv, myproc=rank + 1, device=None)
    N = int(os.environ.get("EMI_TEST_NSMAX", "15"))
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen, precision=PREC)
    assert (et.trans_inq(r, "nproc"), et.trans_inq(r, "nprtrw"), et.trans_inq(r, "nprtrv")) == (world, nprw, nprv)
    myw, myv = et.trans_inq(r, "mysetw"), et.trans_inq(r, "mysetv")
    assert (myw, myv) == (rank // nprv + 1, rank % nprv + 1) and et.trans_inq(r, "myproc") == rank + 1  # pe2set_mod.F90:111-112
    o = Oracle(N, nloen)
    rng = np.random.default_rng(11)  # same global fields on every task
    nuv, nsc = 3, 5
    vor = random_spectrum(rng, o.nasm0, N, o.nspec2, nuv, True)
    div = random_spectrum(rng, o.nasm0, N, o.nspec2, nuv, True)
    sc = random_spectrum(rng, o.nasm0, N, o.nspec2, nsc, False)
    kvuv = np.array([(i % nprv) + 1 for i in range(nuv)], dtype=np.int32)            # dealt round-robin
    kvsc = np.array([((i * 2 + 1) % nprv) + 1 for i in range(nsc)], dtype=np.int32)  # another pattern (a V-set may get none)
    luv, lsc = np.flatnonzero(kvuv == myv), np.flatnonzero(kvsc == myv)
    gref = o.inv_trans(spvor=vor, spdiv=div, spsc=sc, scders=True, uvder=True, vorgp=True, divgp=True)
    # ---- local pieces: wavenumbers of the W-set, latitudes of the task
    myms = et.trans_inq(r, "myms")
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    gidx = np.concatenate([np.arange(o.nasm0[m] - 1, o.nasm0[m] - 1 + 2 * (N - m + 1)) for m in myms])
    lat0, lat1 = et.trans_inq(r, "nfrstlat") - 1, et.trans_inq(r, "nlstlat")
    latlo = et.trans_inq(r, "latlo")
    assert latlo[rank] == lat0 and latlo[rank + 1] == lat1 and latlo[0] == 0 and latlo[-1] == len(nloen)
    gp0 = int(nloen[:lat0].sum())
    assert ng == int(nloen[lat0:lat1].sum())
    loc = lambda a, cols: np.ascontiguousarray(a[gidx][:, cols])
    sel = lambda a: to(a) if a.shape[1] else None
    # ---- INV_TRANS, single PGP, all options, NPROMA blocks
    npr = 37
    nb = (ng - 1) // npr + 1
    gp = to(np.zeros((nb, gref.shape[0], npr)))
    et.inv_trans(r, pspvor=sel(loc(vor, luv)), pspdiv=sel(loc(div, luv)), pspscalar=sel(loc(sc, lsc)), pgp=gp, kproma=npr,
                 ldscders=True, lduvder=True, ldvorgp=True, lddivgp=True, kvsetuv=kvuv, kvsetsc=kvsc)
    got = np.concatenate(list(back(gp)), axis=1)[:, :ng]
    e_inv = rel_err(got, gref[:, gp0:gp0 + ng], axis=1)
    # ---- DIR_TRANS of u, v, scalars (all fields on my points) -> my V-set's fields on my wavenumbers
    gdir = gref[2 * nuv:4 * nuv + nsc]
    v2, d2, s2 = to(np.zeros((ns2, max(len(luv), 1)))), to(np.zeros((ns2, max(len(luv), 1)))), to(np.zeros((ns2, max(len(lsc), 1))))
    et.dir_trans(r, pspvor=v2 if len(luv) else None, pspdiv=d2 if len(luv) else None, pspscalar=s2 if len(lsc) else None,
                 pgp=to(gdir[None, :, gp0:gp0 + ng]), kvsetuv=kvuv, kvsetsc=kvsc)
    vr, dr, sr = o.dir_trans(gdir, nuv=nuv, nsc=nsc)
    errs = []
    if len(luv):
        errs += [rel_err(back(v2), vr[gidx][:, luv]), rel_err(back(d2), dr[gidx][:, luv])]
    norms = et.specnorm(r, to(loc(sc, lsc)))  # every task calls: the sums run over the tasks of the V-set (a V-set may hold no field)
    e_norm = 0.0
    if len(lsc):
        errs.append(rel_err(back(s2), sr[gidx][:, lsc]))
        e_norm = np.abs(norms / o.specnorm(sc)[lsc] - 1.0).max()
    # SPECNORM with KVSET (specnorm.F90:82-101): the norms of ALL fields on every task
    e_norm = max(e_norm, np.abs(et.specnorm(r, to(loc(sc, lsc)), kvset=kvsc) / o.specnorm(sc) - 1.0).max())
    e_dir = max(errs) if errs else 0.0
    # ---- call mode 2: PGPUV / PGP3A / PGP2 with levels dealt to the V-sets (KVSETSC3A per level)
    nlev, nvar = int(os.environ.get("EMI_TEST_NLEV", "4")), 2  # NLEV < NPRTRV: a V-set that owns no level passes no PSPSC3A at all
    kv3 = np.array([(l % nprv) + 1 for l in range(nlev)], dtype=np.int32)
    kv2 = np.array([nprv], dtype=np.int32)
    l3 = np.flatnonzero(kv3 == myv)
    vor3, div3 = random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, True), random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, True)
    sc3 = [random_spectrum(rng, o.nasm0, N, o.nspec2, nlev, False) for _ in range(nvar)]
    sc2 = random_spectrum(rng, o.nasm0, N, o.nspec2, 1, False)
    g3 = o.inv_trans(spvor=vor3, spdiv=div3, spsc=np.concatenate([sc2] + sc3, axis=1))  # u(nlev) v(nlev) sc2 sc3a[var][lev]
    gpuv, gp3a, gp2 = to(np.zeros((1, 2, nlev, ng))), to(np.zeros((1, nvar, nlev, ng))), to(np.zeros((1, 1, ng)))
    sp3a = to(np.stack([s[gidx][:, l3] for s in sc3])) if len(l3) else None
    own2 = myv == nprv
    et.inv_trans(r, pspvor=sel(loc(vor3, l3)), pspdiv=sel(loc(div3, l3)), pspsc3a=sp3a, pspsc2=to(loc(sc2, [0])) if own2 else None,
                 pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, kvsetuv=kv3, kvsetsc3a=kv3, kvsetsc2=kv2)
    sl = slice(gp0, gp0 + ng)
    e_m2 = max(rel_err(back(gpuv)[0, 0], g3[0:nlev, sl], axis=1), rel_err(back(gpuv)[0, 1], g3[nlev:2 * nlev, sl], axis=1),
               rel_err(back(gp2)[0], g3[2 * nlev:2 * nlev + 1, sl], axis=1),
               max(rel_err(back(gp3a)[0, v], g3[2 * nlev + 1 + v * nlev:2 * nlev + 1 + (v + 1) * nlev, sl], axis=1) for v in range(nvar)))
    # and back
    v3o, d3o = (to(np.zeros((ns2, max(len(l3), 1)))) for _ in range(2))
    s3o = to(np.zeros((nvar, ns2, max(len(l3), 1))))
    s2o = to(np.zeros((ns2, 1)))
    et.dir_trans(r, pspvor=v3o if len(l3) else None, pspdiv=d3o if len(l3) else None, pspsc3a=s3o if len(l3) else None,
                 pspsc2=s2o if own2 else None, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, kvsetuv=kv3, kvsetsc3a=kv3, kvsetsc2=kv2)
    vr3, dr3, sr3 = o.dir_trans(g3, nuv=nlev, nsc=1 + nvar * nlev)
    e_m2d = 0.0
    if len(l3):
        e_m2d = max(rel_err(back(v3o), vr3[gidx][:, l3]), rel_err(back(d3o), dr3[gidx][:, l3]),
                    max(rel_err(back(s3o)[v], sr3[gidx][:, 1 + v * nlev + l3]) for v in range(nvar)))
    if own2:
        e_m2d = max(e_m2d, rel_err(back(s2o), sr3[gidx][:, 0:1]))
    # ---- a PGP3A with room for one variable more than PSPSC3A names: the reference aborts (`IUBOUND(3) /= IF_SC3A_G3`, inv_trans.F90:557-561,
    # dir_trans.F90:451-455) and so must every task here, together and before any exchange (only checked when every V-set owns a level: a
    # task without PSPSC3A takes the count from the grid array and has nothing to compare it with, as with the reference's zero-level array)
    if nlev >= nprv:
        big = to(np.zeros((1, nvar + 1, nlev, ng)))
        for fn, kw in ((et.inv_trans, dict(pspsc3a=sp3a)), (et.dir_trans, dict(pspsc3a=s3o))):
            try:
                fn(r, pgp3a=big, kvsetsc3a=kv3, **kw)
                raise SystemExit("rank %d: a PGP3A with a spare variable was accepted" % rank)
            except RuntimeError as e:
                assert "THIRD DIMENSION OF PGP3A INCONSISTENT" in str(e), str(e)
    # ---- a call in which only the last V-set holds a field (the benchmark's surface field on its own): the other V-sets make no
    # TRMTOL / TRLTOM exchange at all -- transports must not need them (point-to-point blocks, no collective over all tasks)
    gp1 = to(np.zeros((1, 1, ng)))
    et.inv_trans(r, pspsc2=to(loc(sc2, [0])) if own2 else None, pgp2=gp1, kvsetsc2=kv2)
    e_one = rel_err(back(gp1)[0], g3[2 * nlev:2 * nlev + 1, sl], axis=1)
    s1o = to(np.zeros((ns2, 1)))
    et.dir_trans(r, pspsc2=s1o if own2 else None, pgp2=gp1, kvsetsc2=kv2)
    if own2:
        e_one = max(e_one, rel_err(back(s1o), sr3[gidx][:, 0:1]))
    e_m2 = max(e_m2, e_one)
    print("rank %d/%d (W-set %d, V-set %d): nump %d lats %d..%d uv %s sc %s  e_inv %.2e e_dir %.2e e_norm %.2e mode2 %.2e %.2e" % (
        rank, world, myw, myv, len(myms), lat0 + 1, lat1, list(luv), list(lsc), e_inv, e_dir, e_norm, e_m2, e_m2d), flush=True)
    tol = (1e-12, 1e-13) if PREC == 8 else (3e-5, 1e-5)
    assert max(e_inv, e_dir, e_m2, e_m2d) < tol[0] and e_norm < tol[1], (e_inv, e_dir, e_norm, e_m2, e_m2d)
    et.trans_release(r)
    et.trans_end()
    dist.barrier()
    dist.destroy_process_group()
    print("VSETS OK rank %d" % rank, flush=True)


if __name__ == "__main__":
    main()
