import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ectrans_amd as et
from oracle.oracle import Oracle
from tests.common import run_case
if os.environ.get("EMI_LIB"): et._use_library_for_tests(os.environ["EMI_LIB"])
et.setup_trans0(kmax_resol=4, device=0)
dev = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0"), lambda t: t.cpu().numpy())
N=int(sys.argv[1]); prec=int(sys.argv[2]); nh=int(sys.argv[3]); nuv=int(sys.argv[4]); nsc=int(sys.argv[5])
half=np.array([min(20+4*i, 2*N+4) for i in range(nh)],dtype=np.int32)
print(N,prec,nh,nuv,nsc, run_case(et, Oracle, dev, N, np.concatenate([half,half[::-1]]), nuv, nsc, {}, None, precision=prec), flush=True)
