"""One task of the world_size-2 gloo test (launched by tests/test_dist_gloo.py).

Runs the real host logic + kernels (CPU functional emulator build) with the W-set sharding and the
all-to-all-v hook over gloo, and checks every local piece against the oracle's global result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402

import ectrans_amd as et  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.common import octahedral, random_spectrum, rel_err  # noqa: E402


PREC = int(os.environ.get("EMI_TEST_PRECISION", "8"))
DT = np.float32 if PREC == 4 else np.float64


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    on_gpu = os.environ.get("EMI_TEST_DEVICE", "cpu") == "cuda"
    if os.environ.get("EMI_TEST_DEVICE", "cpu") == "cuda_per_rank":
        # one GPU per task and the NATIVE RCCL transport (ectrans_amd/rccl: grouped ncclSend / ncclRecv over xGMI) -- the configuration of
        # `bench.py --gpus N`; gloo only carries the 128-byte unique id and SPECNORM's partial sums
        import torch
        on_gpu = True
        dev_ = "cuda:%d" % rank
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=DT)).to(dev_)
        back = lambda t: t.cpu().numpy().astype(np.float64)
        torch.cuda.set_device(rank)
        et.setup_trans0(kmax_resol=2, kprtrw=world, myproc=rank + 1, device=rank, transport=os.environ.get("EMI_TEST_TRANSPORT", "rccl"))
    elif on_gpu:  # all ranks share cuda:0; the hook stages the exchange through gloo (ectrans_amd/dist.py)
        import torch
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=DT)).to("cuda:0")
        back = lambda t: t.cpu().numpy().astype(np.float64)
        et.setup_trans0(kmax_resol=2, kprtrw=world, myproc=rank + 1, device=0)
    else:
        to, back = (lambda a: np.ascontiguousarray(a, dtype=DT)), (lambda a: np.asarray(a, dtype=np.float64))
        et._use_library_for_tests(os.path.join(ROOT, "tests", "emu", "libectrans_mi_emu.so"))
        et.setup_trans0(kmax_resol=2, kprtrw=world, myproc=rank + 1, device=None)
    N = int(os.environ.get("EMI_TEST_NSMAX", "10"))
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen, precision=PREC)
    o = Oracle(N, nloen)
    rng = np.random.default_rng(11)  # same global fields on every task
    nuv, nsc = 1, int(os.environ.get("EMI_TEST_NSC", "2"))
    vor = random_spectrum(rng, o.nasm0, N, o.nspec2, nuv, True)
    div = random_spectrum(rng, o.nasm0, N, o.nspec2, nuv, True)
    sc = random_spectrum(rng, o.nasm0, N, o.nspec2, nsc, False)
    gref = o.inv_trans(spvor=vor, spdiv=div, spsc=sc, scders=True, uvder=True)
    # ---- local pieces
    myms = et.trans_inq(r, "myms")
    nasm0 = et.trans_inq(r, "nasm0")
    procm = et.trans_inq(r, "procm")
    assert all(procm[m] == rank + 1 for m in myms) and sorted(np.flatnonzero(procm == rank + 1)) == sorted(myms)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    assert ns2 == sum(2 * (N - m + 1) for m in myms)
    gidx = np.concatenate([np.arange(o.nasm0[m] - 1, o.nasm0[m] - 1 + 2 * (N - m + 1)) for m in myms])
    assert all(nasm0[m] >= 1 for m in myms) and all(nasm0[m] == -99 for m in range(N + 1) if m not in set(myms))
    lat0, lat1 = et.trans_inq(r, "nfrstlat") - 1, et.trans_inq(r, "nlstlat")
    gp0 = int(nloen[:lat0].sum())
    assert ng == int(nloen[lat0:lat1].sum())
    loc = lambda a: np.ascontiguousarray(a[gidx])
    gp = to(np.zeros((1, gref.shape[0], ng)))
    et.inv_trans(r, pspvor=to(loc(vor)), pspdiv=to(loc(div)), pspscalar=to(loc(sc)), pgp=gp, ldscders=True, lduvder=True)
    e_inv = rel_err(back(gp)[0], gref[:, gp0:gp0 + ng], axis=1)
    gdir = gref[:2 * nuv + nsc]
    v2, d2, s2 = to(np.zeros((ns2, nuv))), to(np.zeros((ns2, nuv))), to(np.zeros((ns2, nsc)))
    et.dir_trans(r, pspvor=v2, pspdiv=d2, pspscalar=s2, pgp=to(gdir[None, :, gp0:gp0 + ng]))
    vr, dr, sr = o.dir_trans(gdir, nuv=nuv, nsc=nsc)
    e_dir = max(rel_err(back(a), b[gidx]) for a, b in ((v2, vr), (d2, dr), (s2, sr)))
    e_norm = np.abs(et.specnorm(r, to(loc(sc))) / o.specnorm(sc) - 1.0).max()
    # utility routines with several tasks: VORDIV_TO_UV on this task's wavenumbers, GPNORM_TRANS = the global norms on every task
    ur, vr = o.vordiv_to_uv(vor, div)
    u2, w2 = et.vordiv_to_uv(to(loc(vor)), to(loc(div)), N)
    e_uv = max(rel_err(back(u2), ur[gidx]), rel_err(back(w2), vr[gidx]))
    ave, gmn, gmx = et.gpnorm_trans(r, gp)
    ar, mnr, mxr = o.gpnorm(gref)
    gsc = np.abs(gref).max(axis=1)
    e_gpn = max((np.abs(ave - ar) / gsc).max(), (np.abs(gmn - mnr) / gsc).max(), (np.abs(gmx - mxr) / gsc).max())
    print("rank %d/%d: nump %d nlat %d e_inv %.2e e_dir %.2e e_norm %.2e e_uv %.2e e_gpnorm %.2e" % (
        rank, world, len(myms), lat1 - lat0, e_inv, e_dir, e_norm, e_uv, e_gpn), flush=True)
    assert e_uv < (1e-12 if PREC == 8 else 3e-6) and e_gpn < (1e-12 if PREC == 8 else 3e-5), (e_uv, e_gpn)
    tol = (1e-12, 1e-13) if PREC == 8 else (3e-5, 1e-5)  # fp32 library: as tests/test_gpu_parity.py
    assert e_inv < tol[0] and e_dir < tol[0] and e_norm < tol[1], (e_inv, e_dir, e_norm)
    if PREC != 8:  # the re-layout helpers below are precision independent; exact-equality checks in fp64 only
        et.trans_release(r)
        et.trans_end()
        dist.barrier()
        dist.destroy_process_group()
        print("DIST OK rank %d" % rank, flush=True)
        return
    # ---- DIST_SPEC / GATH_SPEC / DIST_GRID / GATH_GRID: fields 0,1 live on the last task, field 2 on task 1
    roots = np.array([world, world, 1])
    glob = np.concatenate([vor, sc[:, :2]], axis=1)  # (nspec2g, 3), identical on every task by construction
    src = glob.copy()
    src[:, roots != rank + 1] = np.nan  # a task only has to provide the fields it is the source of
    locsp = et.dist_spec(r, src, 3, kfrom=roots)
    assert np.array_equal(locsp, glob[gidx])
    back_g = et.gath_spec(r, locsp, 3, kto=roots)
    tgt = np.flatnonzero(roots == rank + 1)
    if tgt.size:
        assert np.array_equal(back_g, glob[:, tgt])
    else:
        assert back_g is None
    ggrid = gref[:3]
    srcg = ggrid.copy()
    srcg[roots != rank + 1] = np.nan
    locgp = et.dist_grid(r, srcg, 3, kfrom=roots, kproma=37)
    assert locgp.shape == ((ng - 1) // 37 + 1, 3, 37)
    assert np.array_equal(np.concatenate(list(locgp), axis=1)[:, :ng], ggrid[:, gp0:gp0 + ng])
    back_gp = et.gath_grid(r, locgp, 3, kto=roots)
    if tgt.size:
        assert np.array_equal(back_gp, ggrid[tgt])
    else:
        assert back_gp is None
    et.trans_release(r)
    et.trans_end()
    dist.barrier()
    dist.destroy_process_group()
    print("DIST OK rank %d" % rank, flush=True)


if __name__ == "__main__":
    main()
