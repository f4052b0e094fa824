import sys, time, torch
sys.path.insert(0, "/root/repo")
import spmv_acc_amd
from spmv_acc_amd import synth
m, n, nnz, rp, ci, v = synth.rmat_torch(25, device="cuda", seed=0xC4)
x = torch.rand(n, device="cuda", dtype=torch.float64)
y = torch.zeros(m, device="cuda", dtype=torch.float64)
tiny = torch.arange(1025, dtype=torch.int32, device="cuda")
ty = torch.zeros(1024, dtype=torch.float64, device="cuda")
spmv_acc_amd.csr_spmv(1.0, 1.0, 1024, 1024, 1024, tiny, tiny[:1024].contiguous(), torch.ones(1024, dtype=torch.float64, device="cuda"), torch.ones(1024, dtype=torch.float64, device="cuda"), ty, strategy="line_enhance")
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy="line_enhance")
    torch.cuda.synchronize()
    print(f"call {k+1}: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
