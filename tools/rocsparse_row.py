"""rocSPARSE dcsrmv as a COMPARISON ROW only (the reference's benchmark/benchmark_rocsparse.hpp:19-84 does the same:
csrmv without and with the analysis step).  Never part of the product path.  Uses the librocsparse.so bundled with
the torch wheel so that it shares torch's HIP runtime."""
import ctypes
import os

import numpy as np


class RocsparseCsrmv:
    def __init__(self):
        import torch

        self.torch = torch
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librocsparse.so")
        self.lib = ctypes.CDLL(path)
        self.handle = ctypes.c_void_p()
        self.descr = ctypes.c_void_p()
        assert self.lib.rocsparse_create_handle(ctypes.byref(self.handle)) == 0
        assert self.lib.rocsparse_create_mat_descr(ctypes.byref(self.descr)) == 0
        vp, ci = ctypes.c_void_p, ctypes.c_int
        self.lib.rocsparse_dcsrmv.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        self.lib.rocsparse_dcsrmv_analysis.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]

    def time(self, m, n, nnz, rp, ci, v, x, y, analysis, iters=20):
        """Median microseconds of y = 1*A*x + 1*y; `analysis` adds rocsparse_dcsrmv_analysis (timed separately)."""
        torch = self.torch
        info = ctypes.c_void_p()
        t_analysis = 0.0
        if analysis:
            assert self.lib.rocsparse_create_mat_info(ctypes.byref(info)) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = self.lib.rocsparse_dcsrmv_analysis(self.handle, 111, m, n, nnz, self.descr, v.data_ptr(), rp.data_ptr(),
                                                    ci.data_ptr(), info)
            e1.record()
            e1.synchronize()
            assert rc == 0, rc
            t_analysis = e0.elapsed_time(e1) * 1e3
        one = ctypes.c_double(1.0)
        call = lambda: self.lib.rocsparse_dcsrmv(self.handle, 111, m, n, nnz, ctypes.addressof(one), self.descr, v.data_ptr(),
                                                 rp.data_ptr(), ci.data_ptr(), info, x.data_ptr(), ctypes.addressof(one),
                                                 y.data_ptr())
        for _ in range(3):
            assert call() == 0
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(iters):
            e0.record()
            call()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        if analysis:
            self.lib.rocsparse_destroy_mat_info(info)
        return float(np.median(ts)), t_analysis
