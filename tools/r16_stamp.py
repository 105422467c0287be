"""Stage clocks of k_fft_inv_r16<16> / k_fft_dir_r16<16> (wave 0 of every workgroup; R16_STAMP points of emi_kernels_body.h):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEMI_MR_STAMP=1 -o ectrans_amd/libectrans_mi.so.stamp ectrans_amd/csrc/ectrans_mi.hip
    EMI_LIB=$PWD/ectrans_amd/libectrans_mi.so.stamp python tools/r16_stamp.py inv|dir          (through gpurun)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ectrans_amd as et
et._use_library_for_tests(os.environ["EMI_LIB"])
which = sys.argv[1] if len(sys.argv) > 1 else "inv"
N, nlev = 1279, 137
dev = torch.device("cuda:0")
et.setup_trans0(kmax_resol=2, device=0)
H = N + 1
nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))], dtype=np.int32)
r = et.setup_trans(N, 2 * H, nloen)
ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
sc3 = torch.rand((10, ns2, nlev), dtype=torch.float64, device=dev)
gp3 = torch.rand((1, 10, nlev, ng), dtype=torch.float64, device=dev)
L = et.lib()
out = (C.c_ulonglong * 16)()
L.emi_debug_mr_stamps(out)
names = {"inv": ["stage 1 (FOURIER_IN gather + pairing -> LDS)", "LDS -> registers + barriers", "convolution chain", "stage 3 (chirp + grid store)"],
         "dir": ["stage 1 (grid row + chirp)", "convolution chain", "Z to LDS + barriers", "FOURIER_OUT (pairing + scatter)"]}[which]
for it in range(2):
    if which == "inv":
        et.inv_trans(r, pspsc3a=sc3, pgp3a=gp3)
    else:
        et.dir_trans(r, pspsc3a=sc3, pgp3a=gp3)
    torch.cuda.synchronize()
    L.emi_debug_mr_stamps(out)
    v = np.array(list(out), dtype=np.float64)
    tot = v[:4].sum()
    print("it %d (%s): %d workgroups, %.0f clocks per workgroup (wave 0)" % (it, which, v[7], tot / max(v[7], 1)))
    for n_, x in zip(names, v[:4]):
        print("  %-48s %5.1f %%  %8.0f clocks" % (n_, 100 * x / tot, x / max(v[7], 1)))
