"""Stage clocks of k_leg_inv / k_leg_dir (wave 0 of every workgroup; LEG_STAMP points of emi_kernels_body.h):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEMI_MR_STAMP=1 -DEMI_LEG_STAMP=1 -o ectrans_amd/libectrans_mi.so.lst1 ectrans_amd/csrc/ectrans_mi.hip   (2: k_leg_dir)
    EMI_LIB=$PWD/ectrans_amd/libectrans_mi.so.lst1 python tools/leg_stamp.py inv|dir          (through gpurun)"""
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
sc3 = torch.rand((12, ns2, nlev), dtype=torch.float64, device=dev)
gp3 = torch.rand((1, 12, nlev, ng), dtype=torch.float64, device=dev)
L = et.lib()
out = (C.c_ulonglong * 16)()
L.emi_debug_mr_stamps(out)
names = ["end of matrix phase -> first barrier passed", "operand waits (vmcnt) + LDS writes issued", "next stage's loads issued",
         "LDS writes done + second barrier passed", "matrix phase issued (dir: + sums / differences)"]
et.set_profile(True)
for it in range(3):
    if which == "inv":
        et.inv_trans(r, pspsc3a=sc3, pgp3a=gp3)
    else:
        et.dir_trans(r, pspsc3a=sc3, pgp3a=gp3)
    ph = et.last_phase_ms()
    torch.cuda.synchronize()
    L.emi_debug_mr_stamps(out)
    v = np.array(list(out), dtype=np.float64)
    tot, nst, ntile = v[:5].sum(), max(v[6], 1), max(v[7], 1)
    print("it %d (%s): Legendre %.1f ms; %d tiles, %.1f stages per tile, %.0f clocks per stage (wave 0)" % (it, which, ph[1], ntile, nst / ntile, tot / nst))
    for n_, x in zip(names, v[:5]):
        print("  %-48s %5.1f %%  %8.0f clocks per stage" % (n_, 100 * x / tot, x / nst))
    print("  per tile: start-up %.0f clocks, stage loop %.0f, drain %.0f until the last store is issued, %.0f until the stores are acknowledged" % (
        v[5] / ntile, tot / ntile, v[8] / ntile, v[9] / ntile))
