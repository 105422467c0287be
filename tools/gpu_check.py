"""Quick GPU sanity run: parity vs the oracle at a few sizes + rough timings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ectrans_amd as et
from oracle.oracle import Oracle

dev = torch.device("cuda:0")
et.setup_trans0(kmax_resol=8, device=0)
rng = np.random.default_rng(1)

def octa(N):
    H = N + 1
    return np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))], dtype=np.int32)

def spec(o, N, nf, zero00):
    sp = rng.uniform(-0.5, 0.5, (o.nspec2, nf))
    n_of = np.zeros(o.nspec2)
    nasm0 = o.nasm0
    for m in range(N + 1):
        i0 = nasm0[m] - 1
        n_of[i0:i0 + 2 * (N - m + 1)] = np.repeat(np.arange(m, N + 1), 2)
    sp /= (n_of[:, None] + 1.0)
    sp[1:2 * (N + 1):2] = 0.0
    if zero00:
        sp[0] = 0.0
    return sp

def parity(N, nloen, nuv, nsc, label):
    t0 = time.time(); r = et.setup_trans(N, len(nloen), nloen); ts = time.time() - t0
    t0 = time.time(); o = Oracle(N, nloen); to = time.time() - t0
    vor, div, sc = spec(o, N, nuv, True), spec(o, N, nuv, True), spec(o, N, nsc, False)
    tv, td, tsc = (torch.from_numpy(x).to(dev) for x in (vor, div, sc))
    gp = torch.zeros((1, 2 * nuv + nsc, o.ngptot), dtype=torch.float64, device=dev)
    et.inv_trans(r, pspvor=tv, pspdiv=td, pspscalar=tsc, pgp=gp); torch.cuda.synchronize()
    gref = o.inv_trans(spvor=vor, spdiv=div, spsc=sc)
    g = gp[0].cpu().numpy()
    e_inv = (np.abs(g - gref).max(axis=1) / np.abs(gref).max(axis=1)).max()
    v2, d2, s2 = torch.zeros_like(tv), torch.zeros_like(td), torch.zeros_like(tsc)
    gin = torch.from_numpy(gref.reshape(1, 2 * nuv + nsc, -1).copy()).to(dev)
    et.dir_trans(r, pspvor=v2, pspdiv=d2, pspscalar=s2, pgp=gin); torch.cuda.synchronize()
    vr, dr, sr = o.dir_trans(gref, nuv=nuv, nsc=nsc)
    e_dir = max(np.abs(a.cpu().numpy() - b).max() / np.abs(b).max() for a, b in ((v2, vr), (d2, dr), (s2, sr)))
    print("%-10s N=%4d setup %.2fs (oracle %.2fs) inv rel %.2e dir rel %.2e" % (label, N, ts, to, e_inv, e_dir), flush=True)
    et.trans_release(r)

parity(21, octa(21), 2, 3, "O22")
d = os.path.join(ROOT, "tests", "golden", "tl149")
nl = np.load(os.path.join(d, "lon_number_by_lat.npy")).astype(np.int32)
parity(148, nl, 1, 2, "tl149")
parity(159, octa(159), 2, 3, "O160")
parity(63, np.full(128, 200, dtype=np.int32), 1, 1, "F64x200")

def timing(N, nlev, nfld, iters=3):
    nloen = octa(N)
    t0 = time.time(); r = et.setup_trans(N, len(nloen), nloen); print("setup N=%d %.2fs" % (N, time.time() - t0), flush=True)
    nspec2, ngptot = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    vor = torch.zeros((nspec2, nlev), dtype=torch.float64, device=dev)
    i = et.trans_inq(r, "nasm0")[4] - 1 + 2 * (19 - 4)
    vor[i] = 1.0
    div = vor.clone(); sc3 = torch.zeros((nfld, nspec2, nlev), dtype=torch.float64, device=dev); sc3[:, i] = 1.0
    sc2 = torch.zeros((nspec2, 1), dtype=torch.float64, device=dev); sc2[i] = 1.0
    gpuv = torch.zeros((1, 2, nlev, ngptot), dtype=torch.float64, device=dev)
    gp3a = torch.zeros((1, nfld, nlev, ngptot), dtype=torch.float64, device=dev)
    gp2 = torch.zeros((1, 1, ngptot), dtype=torch.float64, device=dev)
    kf = 2 * nlev + nfld * nlev + 1
    n0 = et.specnorm(r, sc2)
    for it in range(iters + 1):
        torch.cuda.synchronize(); t0 = time.time()
        et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
        torch.cuda.synchronize(); t1 = time.time()
        et.dir_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
        torch.cuda.synchronize(); t2 = time.time()
        ph = et.last_phase_ms()
        print("  N=%d KF=%d it%d inv %.1f ms dir %.1f ms  phases(dir) %s" % (N, kf, it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ["%.1f" % x for x in ph]), flush=True)
    n1 = et.specnorm(r, sc2)
    wm = et.work_model(r, kf)
    print("  norm drift %.2e; LT flops/dir %.3e" % (abs(n0[0] / n1[0] - 1), wm["legendre_flops"]), flush=True)
    et.trans_release(r)

timing(399, 20, 1)
timing(399, 137, 4)
if len(sys.argv) > 1:
    timing(1279, 20, 1)
