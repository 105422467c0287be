"""Stage clocks of k_fft_dir_mr (library built with -DEMI_MR_STAMP into $EMI_LIB): python tools/mr_stamp.py [nlev]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ectrans_amd as et
et._use_library_for_tests(os.environ["EMI_LIB"])
N, nlev = 1279, int(sys.argv[1]) if len(sys.argv) > 1 else 137
dev = torch.device("cuda:0")
et.setup_trans0(kmax_resol=2, device=0)
H = N + 1
nloen = np.array([20 + 4 * i for i in range(H)] + [20 + 4 * i for i in reversed(range(H))], dtype=np.int32)
r = et.setup_trans(N, 2 * H, nloen)
ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
sc3 = torch.zeros((10, ns2, nlev), dtype=torch.float64, device=dev)
gp3 = torch.rand((1, 10, nlev, ng), dtype=torch.float64, device=dev)
L = et.lib()
out = (C.c_ulonglong * 16)()
for it in range(2):
    et.dir_trans(r, pspsc3a=sc3, pgp3a=gp3)
    torch.cuda.synchronize()
    L.emi_debug_mr_stamps(out)
    v = np.array(list(out), dtype=np.float64)
    names = ["pass 1 (grid loads)", "barrier 1", "pass 2", "barrier 2", "pass 3", "barrier 3", "FOURIER_OUT"]
    tot = v[:7].sum() + v[8] + v[9]
    print("it %d: %d workgroups, %.0f clocks per workgroup (wave 0)" % (it, v[7], tot / max(v[7], 1)))
    for n_, x in zip(names, v[:7]):
        print("  %-22s %5.1f %%  %8.0f clocks" % (n_, 100 * x / tot, x / max(v[7], 1)))
    print("  inside FOURIER_OUT: until the loads arrived %.0f clocks, compute + stores issued %.0f clocks" % (v[8] / max(v[7], 1), v[9] / max(v[7], 1)))
