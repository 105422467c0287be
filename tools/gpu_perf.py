"""Phase timings on the GPU: python tools/gpu_perf.py NSMAX NLEV NFLD [iters] [precision]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ectrans_amd as et
if os.environ.get("EMI_LIB"):  # A/B runs of two builds on the same box
    et._use_library_for_tests(os.environ["EMI_LIB"])
N, nlev, nfld = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev = torch.device("cuda:0")
et.setup_trans0(kmax_resol=2, device=0)
H = N + 1
nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))], dtype=np.int32)
t0 = time.time(); r = et.setup_trans(N, 2 * H, nloen, precision=prec); print("setup %.2fs" % (time.time() - t0), flush=True)
ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
z = lambda *s: torch.zeros(s, dtype=torch.float32 if prec == 4 else torch.float64, device=dev)
vor, div, sc3, sc2 = z(ns2, nlev), z(ns2, nlev), z(nfld, ns2, nlev), z(ns2, 1)
g = torch.Generator(device=dev); g.manual_seed(1)
for a in (vor, div, sc3, sc2):
    a.uniform_(-0.5, 0.5, generator=g)
vor[1:2 * (N + 1):2] = 0; div[1:2 * (N + 1):2] = 0; sc2[1:2 * (N + 1):2] = 0; sc3[:, 1:2 * (N + 1):2] = 0
vor[0] = 0; div[0] = 0
gpuv, gp3a, gp2 = z(1, 2, nlev, ng), z(1, nfld, nlev, ng), z(1, 1, ng)
kf = 2 * nlev + nfld * nlev + 1
wm = et.work_model(r, kf)
et.set_profile(True)
n0 = et.specnorm(r, sc2)
ref = sc2.clone()
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.time()
    et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    pi = et.last_phase_ms(); torch.cuda.synchronize(); t1 = time.time()
    et.dir_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    pd = et.last_phase_ms(); torch.cuda.synchronize(); t2 = time.time()
    print("KF=%d it%d inv %.1f ms [pack %.1f leg %.1f (%.1f TF) fft %.1f] dir %.1f ms [pack %.1f leg %.1f (%.1f TF) fft %.1f]" % (
        kf, it, (t1 - t0) * 1e3, pi[0], pi[1], wm["legendre_flops"] / pi[1] / 1e9, pi[2],
        (t2 - t1) * 1e3, pd[0], pd[1], wm["legendre_flops"] / pd[1] / 1e9, pd[2]), flush=True)
print("roundtrip max rel err %.2e" % ((sc2 - ref).abs().max().item() / ref.abs().max().item()))
