"""Compute time of ONE task of a several-task job on one GPU: python tools/gpu_rank_perf.py NSMAX NLEV NFLD NPROC MYPROC [iters] [precision]

The all-to-all-v hook is replaced by a no-op, so the kernels run on whatever the exchange buffers hold
(timings of the Legendre / FFT / pack kernels do not depend on the data) and nothing is communicated: what is
printed is the per-task compute time of the W-set decomposition -- kernel efficiency at the smaller per-task
sizes, launch tails of the pipelined field batches (EMI_PIPELINE_DIST) -- not a job time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ectrans_amd as et
from ectrans_amd import dist as edist
if os.environ.get("EMI_LIB"):
    et._use_library_for_tests(os.environ["EMI_LIB"])
N, nlev, nfld, nproc, myproc = (int(a) for a in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
prec = int(sys.argv[7]) if len(sys.argv) > 7 else 8
dev = torch.device("cuda:0")
noop = edist._A2A(lambda *a: 0)
et.setup_trans0(kmax_resol=2, device=0, kprtrw=nproc, myproc=myproc, alltoallv=noop)
H = N + 1
nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))], dtype=np.int32)
t0 = time.time(); r = et.setup_trans(N, 2 * H, nloen, precision=prec); print("setup %.2fs" % (time.time() - t0), flush=True)
ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
z = lambda *s: torch.zeros(s, dtype=torch.float32 if prec == 4 else torch.float64, device=dev)
vor, div, sc3, sc2 = z(ns2, nlev), z(ns2, nlev), z(nfld, ns2, nlev), z(ns2, 1)
gpuv, gp3a, gp2 = z(1, 2, nlev, ng), z(1, nfld, nlev, ng), z(1, 1, ng)
kf = 2 * nlev + nfld * nlev + 1
wm = et.work_model(r, kf)
et.set_profile(True)
print("task %d/%d: nspec2 %d ngptot %d, Legendre flops/direction %.3e" % (myproc, nproc, ns2, ng, wm["legendre_flops"]))
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.time()
    et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    pi = et.last_phase_ms(); torch.cuda.synchronize(); t1 = time.time()
    et.dir_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    pd = et.last_phase_ms(); torch.cuda.synchronize(); t2 = time.time()
    print("KF=%d it%d inv %.1f ms [pack %.1f leg %.1f (%.1f TF) fft %.1f] dir %.1f ms [pack %.1f leg %.1f (%.1f TF) fft %.1f]  pair %.1f ms" % (
        kf, it, (t1 - t0) * 1e3, pi[0], pi[1], wm["legendre_flops"] / max(pi[1], 1e-9) / 1e9, pi[2],
        (t2 - t1) * 1e3, pd[0], pd[1], wm["legendre_flops"] / max(pd[1], 1e-9) / 1e9, pd[2], (t2 - t0) * 1e3), flush=True)
# the same pairs queued back to back, nothing resolved or synchronised in between (what bench.py times)
et.set_profile(0)
torch.cuda.synchronize(); t0 = time.time(); K = 10
for it in range(K):
    et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    et.dir_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
torch.cuda.synchronize()
print("back to back: %.1f ms per pair" % ((time.time() - t0) / K * 1e3))
