#!/usr/bin/env python3
"""Which array's position matters?  All five arrays of a stand-in in raw hipMalloc allocations (2 MB-aligned), then ONE of them shifted inside a
padded allocation by an odd amount (1 MB + 68 KB + 256 B), line_enhance with every choice pinned, per-launch protocol, us."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, spmv_acc_amd
from spmv_acc_amd import synth
lib = spmv_acc_amd.load_library()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipFree.argtypes = [ctypes.c_void_p]
SHIFT = (1 << 20) + (68 << 10) + 256
def dev_copy(arr, shift):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), arr.nbytes + (8 << 20)) == 0
    assert hip.hipMemcpy(ctypes.c_void_p(p.value + shift), arr.ctypes.data, arr.nbytes, 1) == 0
    return p.value, p.value + shift
name = sys.argv[1] if len(sys.argv) > 1 else "af_shell10"
m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
h = {"rowptr": rp.cpu().numpy(), "colindex": ci.cpu().numpy(), "values": v.cpu().numpy(), "x": x.cpu().numpy(), "y": y0.cpu().numpy()}
del rp, ci, v, x
torch.cuda.empty_cache()
lib.spmv_acc_set_tunable(b"deterministic", 1)
cases = [(None, 0), ("rowptr", SHIFT), ("colindex", SHIFT), ("values", SHIFT), ("x", SHIFT), ("y", SHIFT), ("all", SHIFT)] + [("values", s) for s in (256, 4096, 65536, 1 << 20, (1 << 20) + 256, 3 << 20)]
for moved, amount in cases:
    bases, d = [], {}
    for i, (k, arr) in enumerate(h.items()):
        shift = (amount * (i + 1) if moved == "all" else amount) if moved in (k, "all") else 0
        b, p = dev_copy(arr, shift)
        bases.append(b); d[k] = p
    ts = []
    for rnd in range(2):
        spmv_acc_amd.release_plans(d["rowptr"])
        for _ in range(6):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, d["rowptr"], d["colindex"], d["values"], d["x"], d["y"], strategy="line_enhance")
        ts.append(float(np.median(spmv_acc_amd.time_spmv("line_enhance", 40, 1.0, 1.0, m, n, nnz, d["rowptr"], d["colindex"], d["values"], d["x"], d["y"], y0=y0))) * 1e3)
    spmv_acc_amd.release_plans(d["rowptr"])
    torch.cuda.synchronize()
    for b in bases:
        hip.hipFree(ctypes.c_void_p(b))
    print(f"{name}: shifted {str(moved):9s} by {amount:8d} B {min(ts):7.1f} us   (colindex % 2 MB = {d['colindex'] % (1 << 21)}, values % 2 MB = {d['values'] % (1 << 21)})", flush=True)
