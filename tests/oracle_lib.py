"""ctypes wrapper over oracle/liboracle.so (our CPU restatement) and, when present,
oracle/_ref/libref_analyze.so (the reference's own analysis translation unit).
TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_analyze.so")
REF_READERS_SO = os.path.join(ORACLE_DIR, "_ref", "libref_readers.so")

_ip = ctypes.POINTER(ctypes.c_int)
_dp = ctypes.POINTER(ctypes.c_double)


def _i(a):
    return a.ctypes.data_as(_ip)


def _d(a):
    return a.ctypes.data_as(_dp)


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True, stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = ctypes.CDLL(ORACLE_SO)
        cd, ci = ctypes.c_double, ctypes.c_int
        L.oracle_host_spmv.argtypes = [cd, cd, _dp, _ip, _ip, ci, ci, ci, _dp, _dp]
        L.oracle_host_spmv.restype = None
        L.oracle_host_spmv_plain.argtypes = [_dp, _ip, _ip, ci, ci, ci, _dp, _dp]
        L.oracle_host_spmv_plain.restype = None
        L.oracle_host_spmv_omp.argtypes = [cd, cd, _dp, _ip, _ip, ci, _dp, _dp, ci]
        L.oracle_host_spmv_omp.restype = None
        L.oracle_host_spmv_bench.argtypes = [cd, cd, _dp, _ip, _ip, ci, ci, _dp, _dp, _dp, ci, ci, _dp]
        L.oracle_stream_triad_gbs.argtypes = [ctypes.c_longlong, ci, ci]
        L.oracle_stream_triad_gbs.restype = cd
        L.oracle_verify.argtypes = [_dp, _dp, ci]
        L.oracle_verify_y.argtypes = [_dp, _dp, ci, _dp, _ip, _ip]
        L.oracle_verify_y.restype = None
        L.oracle_rand_vector.argtypes = [ci, _dp]
        L.oracle_rand_vector.restype = None
        L.oracle_srand.argtypes = [ctypes.c_uint]
        L.oracle_ref_mem_bytes.argtypes = [ci, ci]
        L.oracle_ref_mem_bytes.restype = cd
        L.oracle_ref_gibps.argtypes = [ci, ci, cd]
        L.oracle_ref_gibps.restype = cd
        L.oracle_ref_gflops.argtypes = [ci, cd]
        L.oracle_ref_gflops.restype = cd
        L.oracle_break_points.argtypes = [_ip, ci, ci, _ip, ci]
        L.oracle_break_points.restype = None
        L.oracle_break_points_v2.argtypes = [_ip, ci, ci, _ip, ci]
        L.oracle_break_points_v2.restype = None
        L.oracle_break_points_len.argtypes = [ci, ci]
        L.oracle_adaptive_plus_analyze.argtypes = [ci, ci, ci, ci, ci, _ip, _ip, ci, _ip]
        L.oracle_adaptive_plus_vec.argtypes = [ci, ci]
        L.oracle_adaptive_pick.argtypes = [ci, _ip]
        L.oracle_adaptive_line_params.argtypes = [ci, ci, _ip, _ip]
        L.oracle_adaptive_enhance_params.argtypes = [ci, ci, _ip, _ip, _ip]
        L.oracle_adaptive_flat_vec.argtypes = [ci, _ip]
        L.oracle_adaptive_vec_row_bp.argtypes = [ci, ci]
        _lib = L
    return _lib


def ref():
    """The compiled reference analysis (None when oracle/_ref was not built: no /root/reference)."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        R = ctypes.CDLL(REF_SO)
        ci = ctypes.c_int
        R.ref_adaptive_plus_analyze.argtypes = [ci, ci, ci, ci, ci, _ip, _ip, ci, _ip]
        _ref = R
    return _ref


_ref_readers = None


def ref_readers():
    """The reference's own matrix readers (None when oracle/_ref was not built)."""
    global _ref_readers
    if _ref_readers is None and os.path.exists(REF_READERS_SO):
        R = ctypes.CDLL(REF_READERS_SO)
        R.ref_read_matrix.argtypes = [ctypes.c_char_p, ctypes.c_int, _ip, _ip, _ip, ctypes.POINTER(_ip), ctypes.POINTER(_ip),
                                      ctypes.POINTER(_dp), ctypes.POINTER(_dp), _ip]
        R.ref_free.argtypes = [ctypes.c_void_p]
        _ref_readers = R
    return _ref_readers


def ref_read_matrix(path, fmt):
    """(rows, cols, nnz, rowptr, colidx, values, x) as the REFERENCE's reader parses `path` (fmt: csr | bin2 | mtx)."""
    R = ref_readers()
    rows, cols, nnz, xlen = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rp, ci, v, x = _ip(), _ip(), _dp(), _dp()
    rc = R.ref_read_matrix(path.encode(), {"csr": 0, "bin2": 1, "mtx": 2}[fmt], ctypes.byref(rows), ctypes.byref(cols),
                           ctypes.byref(nnz), ctypes.byref(rp), ctypes.byref(ci), ctypes.byref(v), ctypes.byref(x),
                           ctypes.byref(xlen))
    assert rc == 0, "reference reader raised"
    out = (rows.value, cols.value, nnz.value, np.ctypeslib.as_array(rp, (rows.value + 1,)).copy(),
           np.ctypeslib.as_array(ci, (max(nnz.value, 1),))[: nnz.value].copy(),
           np.ctypeslib.as_array(v, (max(nnz.value, 1),))[: nnz.value].copy(),
           np.ctypeslib.as_array(x, (xlen.value,)).copy() if xlen.value else np.zeros(0))
    for p in (rp, ci, v, x):
        R.ref_free(ctypes.cast(p, ctypes.c_void_p))
    return out


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def host_spmv(alpha, beta, rowptr, cols, vals, x, y0):
    """y = alpha*A*x + beta*y0 (sequential oracle).  Returns a new array."""
    rowptr, cols, vals, x = _c(rowptr, np.int32), _c(cols, np.int32), _c(vals, np.float64), _c(x, np.float64)
    y = np.array(y0, dtype=np.float64, copy=True)
    m = rowptr.size - 1
    lib().oracle_host_spmv(alpha, beta, _d(vals), _i(rowptr), _i(cols), m, x.size, vals.size, _d(x), _d(y))
    return y


def host_spmv_omp(alpha, beta, rowptr, cols, vals, x, y, threads):
    """In-place multi-threaded form (for timing)."""
    m = rowptr.size - 1
    lib().oracle_host_spmv_omp(alpha, beta, _d(vals), _i(rowptr), _i(cols), m, _d(x), _d(y), threads)


def host_spmv_bench(alpha, beta, rowptr, cols, vals, x, y0, threads, reps):
    """The timed form of the CPU baseline (first-touch placement by the reading threads, y restored before every run): returns
    (seconds per run, y of the last run)."""
    m = len(rowptr) - 1
    secs = np.zeros(reps, dtype=np.float64)
    y = np.empty(m, dtype=np.float64)
    rc = lib().oracle_host_spmv_bench(alpha, beta, _d(vals), _i(rowptr), _i(cols), m, len(x), _d(x), _d(y0), _d(y), threads, reps, _d(secs))
    if rc != 0:
        raise MemoryError("oracle_host_spmv_bench: allocation failed")
    return secs, y


def stream_triad_gbs(elems, threads, reps=5):
    return float(lib().oracle_stream_triad_gbs(int(elems), threads, reps))


def host_spmv_inplace(alpha, beta, rowptr, cols, vals, x, y):
    m = rowptr.size - 1
    lib().oracle_host_spmv(alpha, beta, _d(vals), _i(rowptr), _i(cols), m, x.size, vals.size, _d(x), _d(y))


def max_threads():
    return lib().oracle_max_threads()


def verify(dy, hy):
    dy, hy = _c(dy, np.float64), _c(hy, np.float64)
    return lib().oracle_verify(_d(dy), _d(hy), dy.size)


def verify_y(dy, hy):
    dy, hy = _c(dy, np.float64), _c(hy, np.float64)
    me = ctypes.c_double(0)
    ff = ctypes.c_int(0)
    fc = ctypes.c_int(0)
    lib().oracle_verify_y(_d(dy), _d(hy), dy.size, ctypes.byref(me), ctypes.byref(ff), ctypes.byref(fc))
    return me.value, ff.value, fc.value


def break_points(rowptr, stride, v2=False):
    rowptr = _c(rowptr, np.int32)
    m = rowptr.size - 1
    n = lib().oracle_break_points_len(int(rowptr[m]), stride)
    bp = np.zeros(n, dtype=np.int32)
    (lib().oracle_break_points_v2 if v2 else lib().oracle_break_points)(_i(rowptr), m, stride, _i(bp), n)
    return bp


def _analyze(fn, rowptr, min_nnz, threads, vec, ref_order):
    rowptr = _c(rowptr, np.int32)
    m = rowptr.size - 1
    nnz = int(rowptr[m])
    cap = m + 2 + 2 * (nnz // max(min_nnz, 1))
    bp = np.zeros(cap, dtype=np.int32)
    fbr = np.zeros(m + 1, dtype=np.int32)
    if ref_order:
        blocks = fn(threads, vec, m, nnz, min_nnz, _i(rowptr), _i(bp), cap, _i(fbr))
    else:
        blocks = fn(m, nnz, min_nnz, threads, vec, _i(rowptr), _i(bp), cap, _i(fbr))
    assert blocks >= 0, blocks
    return blocks, bp[: blocks + 1].copy(), fbr


def adaptive_plus_analyze(rowptr, min_nnz=2048, threads=512, vec=1):
    return _analyze(lib().oracle_adaptive_plus_analyze, rowptr, min_nnz, threads, vec, False)


def ref_adaptive_plus_analyze(rowptr, min_nnz=2048, threads=512, vec=1):
    return _analyze(ref().ref_adaptive_plus_analyze, rowptr, min_nnz, threads, vec, True)


def adaptive_pick(rowptr):
    rowptr = _c(rowptr, np.int32)
    return lib().oracle_adaptive_pick(rowptr.size - 1, _i(rowptr))


def scaled_error(d, h, alpha, beta, rowptr, cols, vals, x, y0):
    """max_i |d_i - h_i| / (|alpha| * sum_j |a_ij x_j| + |beta y0_i|)  -- the robust gate of SURVEY.md 8(c):
    a relative error measured against the magnitude of the terms that were added, so rows that cancel
    to ~0 do not blow it up.  Rows whose scale is exactly 0 must match exactly."""
    import scipy.sparse as sp

    m = rowptr.size - 1
    A = sp.csr_matrix((np.abs(vals), cols, rowptr), shape=(m, x.size))
    scale = abs(alpha) * (A @ np.abs(x)) + np.abs(beta * np.asarray(y0))
    diff = np.abs(np.asarray(d) - np.asarray(h))
    zero = scale == 0
    if np.any(diff[zero] != 0):
        return np.inf
    return float(np.max(diff[~zero] / scale[~zero])) if np.any(~zero) else 0.0
