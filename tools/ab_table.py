import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
import ectrans_amd as et
from tests.common import octahedral, random_spectrum
et.setup_trans0(kmax_resol=4, device=0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 639
nloen = octahedral(N)
outs = []
for tab in ("0", "1"):
    os.environ.pop("EMI_FB_TABLE", None)
    if tab == "1":
        os.environ["EMI_FB_TABLE"] = "1"
    r = et.setup_trans(N, len(nloen), nloen)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    if not outs:
        rng = np.random.default_rng(3)
        nasm0 = et.trans_inq(r, "nasm0")
        sp = random_spectrum(rng, nasm0, N, ns2, 3, False)
        vor = random_spectrum(rng, nasm0, N, ns2, 2, True); div = random_spectrum(rng, nasm0, N, ns2, 2, True)
    gp = torch.zeros((1, 4 + 3 * 3 + 4, ng), dtype=torch.float64, device="cuda:0")
    et.inv_trans(r, pspvor=torch.from_numpy(vor).cuda(), pspdiv=torch.from_numpy(div).cuda(), pspscalar=torch.from_numpy(sp).cuda(), pgp=gp, ldscders=True, lduvder=True)
    outs.append(gp.cpu().numpy()[0].copy())
    et.trans_release(r)
a, b = outs
d = np.abs(a - b)
print("max abs diff", d.max(), "rel", d.max() / np.abs(a).max(), "fields differing", np.flatnonzero(d.max(axis=1) > 0))
off = np.concatenate([[0], np.cumsum(nloen)])
bad = [(int(nloen[j]), float(d[:, off[j]:off[j+1]].max())) for j in range(len(nloen)) if d[:, off[j]:off[j+1]].max() > 0]
print("latitudes differing:", len(bad), "of", len(nloen), bad[:12])
