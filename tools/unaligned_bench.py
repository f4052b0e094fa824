import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np, spmv_acc_amd
from spmv_acc_amd import synth
for name in ("boneS10","Hardesty3"):
    if name=="Hardesty3": m,n,nnz,rp,ci,v = synth.hardesty3_like_torch(device="cuda")
    else: m,n,nnz,rp,ci,v = synth.large_set_like_torch(name, device="cuda")
    x=torch.rand(n,dtype=torch.float64,device='cuda'); y=torch.zeros(m,dtype=torch.float64,device='cuda')
    pc=torch.zeros(nnz+1,dtype=torch.int32,device='cuda'); pv=torch.zeros(nnz+1,dtype=torch.float64,device='cuda')
    pc[1:]=ci; pv[1:]=v
    for tag,(c_,v_) in (("aligned",(ci,v)),("offset by one element",(pc[1:],pv[1:]))):
        for s in ("adaptive","flat","adaptive_plus"):
            ms=spmv_acc_amd.time_spmv(s,25,1.0,1.0,m,n,nnz,rp,c_,v_,x,y)[5:]
            print(name,tag,s,"%.1f us"%(np.median(ms)*1e3))
        spmv_acc_amd.release_plans(rp)
