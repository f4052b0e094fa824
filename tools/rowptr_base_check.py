#!/usr/bin/env python3
"""A row shard passed WITHOUT rebasing (rowptr + r0, rowptr[0] > 0, colindex / value = the whole arrays): every strategy against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np, spmv_acc_amd, oracle_lib
from spmv_acc_amd import synth
rowptr, cols, vals = synth.random_csr(30000, 30000, 11, seed=4, kind="powerlaw")
rng=np.random.default_rng(1); x=rng.standard_normal(30000); y0=rng.standard_normal(30000)
ref = oracle_lib.host_spmv(1.0,1.0,rowptr,cols,vals,x,y0)
drp,dci,dv,dx = (torch.from_numpy(a).cuda() for a in (rowptr,cols,vals,x))
r0,r1=12345,27001
for s in spmv_acc_amd.STRATEGIES:
    dy=torch.from_numpy(y0).cuda()
    try:
        spmv_acc_amd.csr_spmv(1.0,1.0,r1-r0,30000,int(rowptr[r1]),drp[r0:],dci,dv,dx,dy[r0:],strategy=s)
        torch.cuda.synchronize()
        got=dy.cpu().numpy()
        ok = np.allclose(got[r0:r1],ref[r0:r1],rtol=1e-12,atol=1e-12) and np.array_equal(got[:r0],y0[:r0]) and np.array_equal(got[r1:],y0[r1:])
        print(s, "OK" if ok else "MISMATCH", float(np.abs(got[r0:r1]-ref[r0:r1]).max()))
        if not ok:
            bad_in = np.nonzero(~np.isclose(got[r0:r1], ref[r0:r1], rtol=1e-9, atol=1e-9))[0]
            bad_out = np.nonzero(np.concatenate([got[:r0] != y0[:r0], got[r1:] != y0[r1:]]))[0]
            print("   rows off inside the shard:", bad_in[:8], len(bad_in), " rows touched outside:", bad_out[:8], len(bad_out))
    except Exception as e: print(s,"ERR",e)
    spmv_acc_amd.release_plans(drp[r0:])
