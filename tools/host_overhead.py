import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import ectrans_amd as et
et.setup_trans0(kmax_resol=2, device=0)
N=21; H=N+1
nloen=np.array([20+4*i for i in range(H)]+[20+4*i for i in reversed(range(H))],dtype=np.int32)
r=et.setup_trans(N,2*H,nloen)
ns2,ng=et.trans_inq(r,"nspec2"),et.trans_inq(r,"ngptot")
dev=torch.device("cuda:0"); nlev,nfld=137,10
z=lambda *s: torch.zeros(s,dtype=torch.float64,device=dev)
vor,div,sc3,sc2=z(ns2,nlev),z(ns2,nlev),z(nfld,ns2,nlev),z(ns2,1)
gpuv,gp3a,gp2=z(1,2,nlev,ng),z(1,nfld,nlev,ng),z(1,1,ng)
for _ in range(5):
    et.inv_trans(r,pspvor=vor,pspdiv=div,pspsc3a=sc3,pspsc2=sc2,pgpuv=gpuv,pgp3a=gp3a,pgp2=gp2)
    et.dir_trans(r,pspvor=vor,pspdiv=div,pspsc3a=sc3,pspsc2=sc2,pgpuv=gpuv,pgp3a=gp3a,pgp2=gp2)
torch.cuda.synchronize(); t=time.perf_counter(); K=50
for _ in range(K):
    et.inv_trans(r,pspvor=vor,pspdiv=div,pspsc3a=sc3,pspsc2=sc2,pgpuv=gpuv,pgp3a=gp3a,pgp2=gp2)
    et.dir_trans(r,pspvor=vor,pspdiv=div,pspsc3a=sc3,pspsc2=sc2,pgpuv=gpuv,pgp3a=gp3a,pgp2=gp2)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/K
print("T21 x 1645 fields: %.3f ms per pair wall (host + tiny GPU work)" % (dt*1e3))
