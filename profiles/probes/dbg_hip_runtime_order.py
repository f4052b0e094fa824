import ctypes, sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "libfirst":
    l = ctypes.CDLL("spmv_acc_amd/lib/libspmv_acc.so")
    import torch
else:
    import torch
    l = ctypes.CDLL("spmv_acc_amd/lib/libspmv_acc.so")
print("cuda", torch.cuda.is_available())
t = torch.zeros(4, device="cuda")
maps = open("/proc/self/maps").read()
print(sorted({ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln or "libhsa-runtime" in ln}))
hip = ctypes.CDLL("libamdhip64.so.7")
d = ctypes.c_int(-1); print("hipGetDevice rc", hip.hipGetDevice(ctypes.byref(d)), d.value)
