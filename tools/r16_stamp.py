"""Stage clocks of k_fft_inv_r16<16> (experiment build with the RST stamps, -DEMI_MR_STAMP, in $EMI_LIB): python tools/r16_stamp.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ectrans_amd as et
et._use_library_for_tests(os.environ["EMI_LIB"])
N, nlev = 1279, 137
dev = torch.device("cuda:0")
et.setup_trans0(kmax_resol=2, device=0)
H = N + 1
nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))], dtype=np.int32)
r = et.setup_trans(N, 2 * H, nloen)
ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
sc3 = torch.rand((10, ns2, nlev), dtype=torch.float64, device=dev)
gp3 = torch.zeros((1, 10, nlev, ng), dtype=torch.float64, device=dev)
L = et.lib()
out = (C.c_ulonglong * 16)()
L.emi_debug_mr_stamps(out)
for it in range(2):
    et.inv_trans(r, pspsc3a=sc3, pgp3a=gp3)
    torch.cuda.synchronize()
    L.emi_debug_mr_stamps(out)
    v = np.array(list(out), dtype=np.float64)
    names = ["stage 1 (FOURIER_IN gather + pairing -> LDS)", "barrier", "LDS -> registers + barrier", "convolution chain", "stage 3 (chirp + grid store)"]
    tot = v[:5].sum()
    print("it %d: %d workgroups, %.0f clocks per workgroup (wave 0)" % (it, v[7], tot / max(v[7], 1)))
    for n_, x in zip(names, v[:5]):
        print("  %-48s %5.1f %%  %8.0f clocks" % (n_, 100 * x / tot, x / max(v[7], 1)))
    print("  inside stage 1: set-up + issue of the Fourier-row loads %.0f clocks, until they arrived %.0f, tables + pairing + LDS stores %.0f" % (
        v[9] / max(v[7], 1), v[10] / max(v[7], 1), (v[0] - v[9] - v[10]) / max(v[7], 1)))
