#!/usr/bin/env python3
"""Host->device staging rate of spmv_acc_stage_csr (pinned double-buffered hipMemcpyAsync) vs a plain pageable copy."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth
lib = spmv_acc_amd.load_library()
m, n, nnz, rp, ci, v = synth.hardesty3_like_torch(device="cuda")
hrp, hci, hv = rp.cpu().numpy(), ci.cpu().numpy(), v.cpu().numpy()
hx, hy = np.ones(n), np.ones(m)
nbytes = hrp.nbytes + hci.nbytes + hv.nbytes + hx.nbytes + hy.nbytes
outs = [ctypes.c_void_p() for _ in range(5)]
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.spmv_acc_stage_csr(m, n, nnz, hrp.ctypes.data, hci.ctypes.data, hv.ctypes.data, hx.ctypes.data, hy.ctypes.data,
                                *[ctypes.byref(o) for o in outs])
    dt = time.perf_counter() - t0
    assert rc == 0
    chk = torch.empty(16, dtype=torch.float64, device="cuda")
    for o in outs:
        lib.spmv_acc_free_device(o)
    print(f"stage_csr: {nbytes/1e6:.0f} MB in {dt*1e3:.1f} ms = {nbytes/dt/1e9:.1f} GB/s")
for rep in range(3):
    t0 = time.perf_counter()
    ts = [torch.from_numpy(a).cuda() for a in (hrp, hci, hv, hx, hy)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pageable torch .cuda(): {nbytes/dt/1e9:.1f} GB/s")
    del ts
