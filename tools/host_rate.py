"""PCIe-inclusive rate: the same call with host (numpy) arrays through EMI_MEM_HOST.
   python tools/host_rate.py NSMAX NLEV NFLD"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ectrans_amd as et
N, nlev, nfld = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
et.setup_trans0(kmax_resol=1, device=0)
H = N + 1
nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))], dtype=np.int32)
r = et.setup_trans(N, 2 * H, nloen)
ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
kf = 2 * nlev + nfld * nlev + 1
z = lambda *s: np.zeros(s)
vor, div, sc3, sc2 = z(ns2, nlev), z(ns2, nlev), z(nfld, ns2, nlev), z(ns2, 1)
i419 = int(et.trans_inq(r, "nasm0")[4] - 1 + 2 * 15)
for a in (vor, div, sc2):
    a[i419] = 1.0
sc3[:, i419] = 1.0
gpuv, gp3a, gp2 = z(1, 2, nlev, ng), z(1, nfld, nlev, ng), z(1, 1, ng)
nbytes = 2 * 8.0 * (vor.size + div.size + sc3.size + sc2.size + gpuv.size + gp3a.size + gp2.size)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    et.inv_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    et.dir_trans(r, pspvor=vor, pspdiv=div, pspsc3a=sc3, pspsc2=sc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("KF=%d host arrays: pair %.1f ms, %.1f GB over PCIe -> %.1f GB/s" % (kf, dt * 1e3, nbytes / 1e9, nbytes / 1e9 / dt), flush=True)
print("Re(4,19) after the round trips: %.15f" % sc2[i419, 0])
