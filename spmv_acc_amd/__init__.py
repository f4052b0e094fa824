"""spmv_acc_amd -- Python plumbing over the C ABI of libspmv_acc.so (include/spmv_acc.h).

The product is the HIP library; this module only loads it with ctypes and passes raw device
pointers (``tensor.data_ptr()``).  There is no CPU fallback: every compute entry point raises if the
library is missing.  (The CPU oracle lives under ``oracle/`` and is test infrastructure only.)

Reference interface mirrored (names and argument meaning): ``sparse_spmv`` (src/acc/api/spmv.h:27-28),
``sparse_csr_spmv`` (api/spmv.h:20-21) and the KERNEL_STRATEGY names (src/configure.cmake:17-40).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspmv_acc.so")

STRATEGIES = (
    "default", "adaptive", "thread_row", "wf_row", "block_row_ordinary", "light",
    "vector_row", "line_enhance", "line", "flat", "adaptive_plus",
)
# the strategies BASELINE.json's north_star names (+ the row-block preprocessing entry)
HOT_STRATEGIES = ("default", "adaptive", "flat", "line_enhance", "line", "vector_row", "adaptive_plus")

_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_double_p = ctypes.POINTER(ctypes.c_double)

# every symbol include/spmv_acc.h declares (tests check the library exports all of them)
C_ABI_SYMBOLS = (
    "sparse_spmv", "spmv_acc_csr_spmv", "spmv_acc_csr_spmv_strategy", "spmv_acc_set_strategy",
    "spmv_acc_set_strategy_id", "spmv_acc_get_strategy", "spmv_acc_strategy_name", "spmv_acc_parse_strategy",
    "spmv_acc_break_points", "spmv_acc_break_points_len", "spmv_acc_adaptive_plus_analyze",
    "spmv_acc_adaptive_plus_vec", "spmv_acc_adaptive_branch", "spmv_acc_partition_rows", "spmv_acc_stage_csr",
    "spmv_acc_free_device", "spmv_acc_release_plans", "spmv_acc_cached_plans", "spmv_acc_query_plan",
    "spmv_acc_set_stream", "spmv_acc_get_stream", "spmv_acc_last_error", "spmv_acc_last_error_string",
    "spmv_acc_clear_error", "spmv_acc_time_spmv", "spmv_acc_version", "spmv_acc_set_tunable",
    "spmv_acc_get_tunable", "spmv_acc_reset_tunables", "spmv_acc_time_spmv_total", "spmv_acc_copy_ceiling_gbs", "spmv_acc_adaptive_plus_analyze_device", "spmv_acc_prepare",
    "spmv_acc_last_prepare_us", "spmv_acc_sharded_spmv", "spmv_acc_shard_prepare", "spmv_acc_csr_spmv_chunks", "spmv_acc_query_plan_settled", "spmv_acc_query_plan_col16", "spmv_acc_time_spmv_cold", "spmv_acc_csr_spmv_oop", "spmv_acc_check_plans",
    "spmv_acc_query_plan_beta0", "spmv_acc_query_plan_slab_passes", "spmv_acc_shard_create", "spmv_acc_shard_step", "spmv_acc_shard_pipeline",
    "spmv_acc_shard_destroy", "spmv_acc_rccl_comm_init_all", "spmv_acc_rccl_comm_destroy", "spmv_acc_set_tune_cache",
    "spmv_acc_prepare_beta", "spmv_acc_time_spmv_events", "spmv_acc_refresh_values", "spmv_acc_time_spmv_region", "spmv_acc_query_plan_last_kernel", "spmv_acc_time_spmv_kernels",
)

_lib = None


class SpmvAccError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Load libspmv_acc.so (built by ``__graft_entry__.build()`` / ``make -C spmv_acc_amd/csrc``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise SpmvAccError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C spmv_acc_amd/csrc`. There is no CPU fallback."
        )
    # One HIP runtime per process: torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64 and
    # initialise it for device memory.  If libspmv_acc.so were loaded first, its NEEDED libamdhip64.so.7 would
    # bring in /opt/rocm's copy, and the second runtime to initialise finds no device.  Importing torch first
    # makes the soname resolve to the runtime torch already loaded (kernels + tensors then share one context).
    try:
        import torch  # noqa: F401
    except ImportError:  # C/C++ consumers link the library directly; Python without torch still works
        pass
    lib = ctypes.CDLL(p)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.sparse_spmv.argtypes = [ci, cd, cd, ci, ci, vp, vp, vp, vp, vp]
    lib.sparse_spmv.restype = None
    lib.spmv_acc_csr_spmv.argtypes = [ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_csr_spmv.restype = None
    lib.spmv_acc_csr_spmv_strategy.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_csr_spmv_strategy.restype = None
    lib.spmv_acc_set_strategy.argtypes = [ctypes.c_char_p]
    lib.spmv_acc_set_strategy_id.argtypes = [ci]
    lib.spmv_acc_strategy_name.argtypes = [ci]
    lib.spmv_acc_strategy_name.restype = ctypes.c_char_p
    lib.spmv_acc_parse_strategy.argtypes = [ctypes.c_char_p]
    lib.spmv_acc_break_points.argtypes = [vp, ci, ci, ci, vp, ci]
    lib.spmv_acc_break_points_len.argtypes = [ci, ci]
    lib.spmv_acc_adaptive_plus_analyze.argtypes = [ci, ci, ci, ci, vp, vp, ci, vp]
    lib.spmv_acc_adaptive_plus_vec.argtypes = [ci, ci]
    lib.spmv_acc_adaptive_plus_analyze_device.argtypes = [ci, ci, ci, ci, vp, vp, ci, vp]
    lib.spmv_acc_adaptive_branch.argtypes = [ci, ci, ci, ci, ci]
    lib.spmv_acc_partition_rows.argtypes = [ci, ci, ci, vp, vp]
    lib.spmv_acc_stage_csr.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp] + [ctypes.POINTER(vp)] * 5
    lib.spmv_acc_free_device.argtypes = [vp]
    lib.spmv_acc_prepare.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.spmv_acc_release_plans.argtypes = [vp]
    lib.spmv_acc_release_plans.restype = None
    lib.spmv_acc_query_plan.argtypes = [vp, ci, vp]
    lib.spmv_acc_set_stream.argtypes = [vp]
    lib.spmv_acc_set_stream.restype = None
    lib.spmv_acc_get_stream.restype = vp
    lib.spmv_acc_last_error_string.restype = ctypes.c_char_p
    lib.spmv_acc_clear_error.restype = None
    lib.spmv_acc_time_spmv.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_time_spmv_events.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint]
    lib.spmv_acc_time_spmv_cold.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_longlong, vp]
    lib.spmv_acc_version.restype = ctypes.c_char_p
    lib.spmv_acc_set_tunable.argtypes = [ctypes.c_char_p, ci]
    lib.spmv_acc_get_tunable.argtypes = [ctypes.c_char_p]
    lib.spmv_acc_reset_tunables.restype = None
    lib.spmv_acc_time_spmv_total.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_time_spmv_kernels.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_time_spmv_region.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_copy_ceiling_gbs.argtypes = [vp, vp, ctypes.c_longlong, ci]
    lib.spmv_acc_copy_ceiling_gbs.restype = cd
    lib.spmv_acc_last_prepare_us.restype = cd
    lib.spmv_acc_sharded_spmv.argtypes = [vp, ci, cd, cd, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_csr_spmv_oop.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_csr_spmv_oop.restype = None
    lib.spmv_acc_query_plan_beta0.argtypes = [vp, ci]
    lib.spmv_acc_query_plan_slab_passes.argtypes = [vp, ci]
    lib.spmv_acc_query_plan_settled.argtypes = [vp, ci]
    lib.spmv_acc_query_plan_col16.argtypes = [vp, ci]
    lib.spmv_acc_query_plan_last_kernel.argtypes = [vp, ci]
    lib.spmv_acc_shard_create.argtypes = [ctypes.POINTER(vp), vp, ci, ci, ci, ci, ci, vp, vp, vp, ci]
    lib.spmv_acc_shard_step.argtypes = [vp, cd, cd, vp, vp, vp]
    lib.spmv_acc_shard_pipeline.argtypes = [vp]
    lib.spmv_acc_shard_destroy.argtypes = [vp]
    lib.spmv_acc_rccl_comm_init_all.argtypes = [ctypes.POINTER(vp), ci, vp]
    lib.spmv_acc_rccl_comm_destroy.argtypes = [vp]
    lib.spmv_acc_set_tune_cache.argtypes = [ctypes.c_char_p]
    lib.spmv_acc_set_tune_cache.restype = None
    lib.spmv_acc_refresh_values.argtypes = [vp]
    lib.spmv_acc_prepare_beta.argtypes = [ci, cd, ci, ci, ci, vp, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.spmv_acc_shard_prepare.argtypes = [vp, cd, vp]
    lib.spmv_acc_csr_spmv_chunks.argtypes = [ci, cd, cd, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    if path is None:
        _lib = lib
    return lib


def strategy_id(name_or_id) -> int:
    if isinstance(name_or_id, int):
        return name_or_id
    s = load_library().spmv_acc_parse_strategy(str(name_or_id).encode())
    if s < 0:
        raise SpmvAccError(f"unknown KERNEL_STRATEGY {name_or_id!r}")
    return s


def _check(lib) -> None:
    code = lib.spmv_acc_last_error()
    if code not in (0, 1):  # 1 = unsupported trans: reported, not fatal (reference ignores trans)
        msg = lib.spmv_acc_last_error_string().decode()
        lib.spmv_acc_clear_error()
        raise SpmvAccError(f"spmv_acc error {code}: {msg}")


def _ptr(t) -> int:
    """Raw pointer of a torch tensor / numpy array / int (0 for None)."""
    if t is None:
        return 0
    if isinstance(t, int):
        return t
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def _require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and hasattr(t, "is_cuda") and not t.is_cuda:
            raise SpmvAccError("device pointers required: tensor is not on the GPU (no CPU fallback)")


def _require(lib, **named) -> None:
    """Torch tensors handed to the C ABI are passed as raw pointers, so what the kernels assume is checked here: on the GPU,
    all on one device, contiguous, int32 indices / float64 values, and at least as many elements as the shape says.
    name -> (tensor, "i32" | "f64", minimum element count); raw integer pointers and None pass through unchecked.
    Also points the library stream at torch's current stream on that device, so the launches are ordered with the caller's
    other torch work (the default stream is HIP's NULL stream, the reference's behaviour)."""
    device = None
    for name, (t, kind, count) in named.items():
        if t is None or not hasattr(t, "is_cuda"):
            continue
        if not t.is_cuda:
            raise SpmvAccError(f"{name}: device pointers required, tensor is not on the GPU (no CPU fallback)")
        want = "torch.int32" if kind == "i32" else "torch.float64"
        if str(t.dtype) != want:
            raise SpmvAccError(f"{name}: dtype {t.dtype}, the library reads {want} (int32 indices, fp64 values)")
        if not t.is_contiguous():
            raise SpmvAccError(f"{name}: tensor is not contiguous")
        if t.numel() < count:
            raise SpmvAccError(f"{name}: {t.numel()} elements, the shape needs at least {count}")
        if device is None:
            device = t.device
        elif t.device != device:
            raise SpmvAccError(f"{name}: on {t.device}, other arguments on {device}")
    if device is not None:
        import torch

        lib.spmv_acc_set_stream(torch.cuda.current_stream(device).cuda_stream)


def _csr_args(lib, m, n, nnz, rowptr, colindex, value, x, y=None, y0=None) -> None:
    k = max(nnz, 0)
    _require(lib, rowptr=(rowptr, "i32", m + 1), colindex=(colindex, "i32", k), value=(value, "f64", k),
             x=(x, "f64", n), y=(y, "f64", m), y0=(y0, "f64", m))


def sparse_spmv(trans: int, alpha: float, beta: float, m: int, n: int, rowptr, colindex, value, x, y) -> None:
    """Ten-argument entry with the active strategy; all tensors on the GPU; y updated in place (async)."""
    lib = load_library()
    _csr_args(lib, m, n, -1, rowptr, colindex, value, x, y)
    lib.sparse_spmv(trans, alpha, beta, m, n, _ptr(rowptr), _ptr(colindex), _ptr(value), _ptr(x), _ptr(y))
    _check(lib)


def csr_spmv(alpha: float, beta: float, m: int, n: int, nnz: int, rowptr, colindex, value, x, y,
             strategy=None, h_rowptr=None, trans: int = 0, y_in=None) -> None:
    """Descriptor entry (sparse_csr_spmv flattened).  ``h_rowptr``: optional host (numpy int32) rowptr.
    ``y_in``: out-of-place form, y = alpha*A*x + beta*y_in (spmv_acc_csr_spmv_oop); None = in place."""
    lib = load_library()
    _csr_args(lib, m, n, nnz, rowptr, colindex, value, x, y, y_in)
    if y_in is not None:
        lib.spmv_acc_csr_spmv_oop(-1 if strategy is None else strategy_id(strategy), trans, alpha, beta, m, n, nnz,
                                  _ptr(h_rowptr), _ptr(rowptr), _ptr(colindex), _ptr(value), _ptr(x), _ptr(y_in), _ptr(y))
        _check(lib)
        return
    args = (trans, alpha, beta, m, n, nnz, _ptr(h_rowptr), _ptr(rowptr), _ptr(colindex), _ptr(value), _ptr(x), _ptr(y))
    if strategy is None:
        lib.spmv_acc_csr_spmv(*args)
    else:
        lib.spmv_acc_csr_spmv_strategy(strategy_id(strategy), *args)
    _check(lib)


def prepare(m: int, n: int, nnz: int, rowptr, colindex, value, x, strategy=None, h_rowptr=None, beta: float = 1.0) -> float:
    """Build the plan of ``strategy`` for this matrix (structural passes + per-matrix timings) without touching any y, for the
    beta class the caller will run in (beta == 0 / beta != 0).  Returns the device milliseconds it took."""
    lib = load_library()
    _csr_args(lib, m, n, nnz, rowptr, colindex, value, x)
    ms = ctypes.c_float(0.0)
    sid = lib.spmv_acc_get_strategy() if strategy is None else strategy_id(strategy)
    rc = lib.spmv_acc_prepare_beta(sid, beta, m, n, nnz, _ptr(h_rowptr), _ptr(rowptr), _ptr(colindex), _ptr(value), _ptr(x),
                                   ctypes.byref(ms))
    if rc != 0:
        _check(lib)
    return float(ms.value)


def break_points(rowptr, m: int, nnz: int, stride: int, out) -> None:
    """Device form of the row-block preprocessing pass into ``out`` (GPU int32, break_points_len entries)."""
    lib = load_library()
    _require(lib, rowptr=(rowptr, "i32", m + 1), out=(out, "i32", 1))
    rc = lib.spmv_acc_break_points(_ptr(rowptr), m, nnz, stride, _ptr(out), out.numel())
    if rc != 0:
        _check(lib)


def break_points_len(nnz: int, stride: int) -> int:
    return load_library().spmv_acc_break_points_len(nnz, stride)


def adaptive_plus_analyze(h_rowptr, m: int, min_nnz_per_block: int = 2048, threads_per_block: int = 512,
                          vec_size: int = 1):
    """Host form of the row-block preprocessing pass.  Returns (blocks, break_points, first_block_of_row)."""
    import numpy as np

    lib = load_library()
    h_rowptr = np.ascontiguousarray(h_rowptr, dtype=np.int32)
    nnz = int(h_rowptr[m])
    bp = np.zeros(m + 2 + nnz // (2 * min_nnz_per_block), dtype=np.int32)  # rows + long-row slices
    fbr = np.zeros(m + 1, dtype=np.int32)
    blocks = lib.spmv_acc_adaptive_plus_analyze(m, min_nnz_per_block, threads_per_block, vec_size, _ptr(h_rowptr),
                                                _ptr(bp), bp.size, _ptr(fbr))
    if blocks < 0:
        raise SpmvAccError(f"adaptive_plus_analyze failed ({blocks})")
    return blocks, bp[: blocks + 1].copy(), fbr


def adaptive_plus_analyze_device(rowptr, m: int, nnz: int, min_nnz_per_block: int = 1024, threads_per_block: int = 256,
                                 vec_size: int = 1):
    """Device form of the same pass (GPU int32 rowptr).  Returns (blocks, break_points, first_block_of_row) as GPU tensors."""
    import torch

    lib = load_library()
    _require(lib, rowptr=(rowptr, "i32", m + 1))
    cap = m + 2 + nnz // (2 * min_nnz_per_block)
    bp = torch.empty(cap, dtype=torch.int32, device=rowptr.device)
    fbr = torch.empty(m + 1, dtype=torch.int32, device=rowptr.device)
    blocks = lib.spmv_acc_adaptive_plus_analyze_device(m, min_nnz_per_block, threads_per_block, vec_size, _ptr(rowptr),
                                                       _ptr(bp), cap, _ptr(fbr))
    if blocks < 0:
        _check(lib)
        raise SpmvAccError(f"adaptive_plus_analyze_device failed ({blocks})")
    return blocks, bp[: blocks + 1], fbr


def adaptive_branch(m: int, h_rowptr) -> int:
    return load_library().spmv_acc_adaptive_branch(m, int(h_rowptr[m // 4]), int(h_rowptr[m // 2]),
                                                   int(h_rowptr[3 * m // 4]), int(h_rowptr[m]))


def partition_rows(m: int, parts: int, mode: int = 0, h_rowptr=None):
    import numpy as np

    lib = load_library()
    out = np.zeros(parts + 1, dtype=np.int32)
    if h_rowptr is not None:
        h_rowptr = np.ascontiguousarray(h_rowptr, dtype=np.int32)
    rc = lib.spmv_acc_partition_rows(m, parts, mode, _ptr(h_rowptr), _ptr(out))
    if rc != 0:
        _check(lib)
        raise SpmvAccError("partition_rows failed")
    return out


EVENT_DISABLE_SYSTEM_FENCE = 0x20000000  # hipEventDisableSystemFence


def time_spmv(strategy, iters: int, alpha: float, beta: float, m: int, n: int, nnz: int, rowptr, colindex, value,
              x, y, y0=None, h_rowptr=None, event_flags: int = 0) -> Sequence[float]:
    """Per-launch durations (ms) from hipEvents recorded on the library stream around each SpMV; y restored from y0 (device
    copy) before each launch, outside the event pair.  event_flags: hipEventCreateWithFlags flags (0 = hipEventDefault)."""
    lib = load_library()
    _csr_args(lib, m, n, nnz, rowptr, colindex, value, x, y, y0)
    out = (ctypes.c_float * iters)()
    rc = lib.spmv_acc_time_spmv_events(strategy_id(strategy), iters, alpha, beta, m, n, nnz, _ptr(h_rowptr), _ptr(rowptr),
                                       _ptr(colindex), _ptr(value), _ptr(x), _ptr(y), _ptr(y0),
                                       ctypes.cast(out, ctypes.c_void_p), event_flags)
    if rc != 0:
        msg = lib.spmv_acc_last_error_string().decode()
        raise SpmvAccError(f"time_spmv failed ({rc}): {msg}")
    return list(out)


def time_spmv_cold(strategy, iters: int, alpha: float, beta: float, m: int, n: int, nnz: int, rowptr, colindex, value, x, y, y0, flush,
                   h_rowptr=None) -> Sequence[float]:
    """The per-launch protocol with a cold cache hierarchy (spmv_acc_time_spmv_cold): before every timed launch, after y has been restored, the copy
    kernel moves `flush` (a device tensor of >= 2 x the 256 MB Infinity Cache; its second half is overwritten with its first) under the default
    cache policy.  Context for the fractions, never a gate."""
    lib = load_library()
    _csr_args(lib, m, n, nnz, rowptr, colindex, value, x, y, y0)
    out = (ctypes.c_float * iters)()
    rc = lib.spmv_acc_time_spmv_cold(strategy_id(strategy), iters, alpha, beta, m, n, nnz, _ptr(h_rowptr), _ptr(rowptr), _ptr(colindex),
                                     _ptr(value), _ptr(x), _ptr(y), _ptr(y0), _ptr(flush), int(flush.numel() * flush.element_size()),
                                     ctypes.cast(out, ctypes.c_void_p))
    if rc != 0:
        raise SpmvAccError(f"time_spmv_cold failed ({rc}): {lib.spmv_acc_last_error_string().decode()}")
    return list(out)


def time_spmv_total(strategy, iters: int, alpha: float, beta: float, m: int, n: int, nnz: int, rowptr, colindex, value,
                    x, y, h_rowptr=None) -> float:
    """Total milliseconds of `iters` back-to-back SpMVs between ONE hipEvent pair on the library stream."""
    lib = load_library()
    _csr_args(lib, m, n, nnz, rowptr, colindex, value, x, y)
    out = ctypes.c_float(0.0)
    rc = lib.spmv_acc_time_spmv_total(strategy_id(strategy), iters, alpha, beta, m, n, nnz, _ptr(h_rowptr), _ptr(rowptr),
                                      _ptr(colindex), _ptr(value), _ptr(x), _ptr(y), ctypes.addressof(out))
    if rc != 0:
        raise SpmvAccError(f"time_spmv_total failed ({rc}): {lib.spmv_acc_last_error_string().decode()}")
    return float(out.value)


def time_spmv_kernels(strategy, iters: int, alpha: float, beta: float, m: int, n: int, nnz: int, rowptr, colindex, value, x, y, y0=None):
    """The per-launch protocol with the library's kernel clock on: returns (event_ms, kernel_ms, launches) per call -- the event pair around
    the call (the reference harness's figure) and the sum of the call's own kernel durations (what rocprofv3 --kernel-trace reports)."""
    lib = load_library()
    _csr_args(lib, m, n, nnz, rowptr, colindex, value, x, y, y0)
    ev = (ctypes.c_float * iters)()
    kn = (ctypes.c_float * iters)()
    ln = (ctypes.c_int * iters)()
    rc = lib.spmv_acc_time_spmv_kernels(strategy_id(strategy), iters, alpha, beta, m, n, nnz, None, _ptr(rowptr), _ptr(colindex), _ptr(value),
                                        _ptr(x), _ptr(y), _ptr(y0), ctypes.cast(ev, ctypes.c_void_p), ctypes.cast(kn, ctypes.c_void_p),
                                        ctypes.cast(ln, ctypes.c_void_p))
    if rc != 0:
        raise SpmvAccError(f"time_spmv_kernels failed ({rc}): {lib.spmv_acc_last_error_string().decode()}")
    return list(ev), list(kn), list(ln)


def time_spmv_region(strategy, iters: int, alpha: float, beta: float, m: int, n: int, nnz: int, rowptr, colindex, value, x, y):
    """`iters` back-to-back SpMVs between ONE hipEvent pair and nothing else (no plan work, no allocation: settle the plan with prepare() first).
    Returns a closure: each call runs one region and returns its total milliseconds -- argument checking and pointer conversion happen HERE, once,
    so that a wall clock around the closure reads the region itself."""
    lib = load_library()
    _csr_args(lib, m, n, nnz, rowptr, colindex, value, x, y)
    out = ctypes.c_float(0.0)
    args = (strategy_id(strategy), iters, alpha, beta, m, n, nnz, None, _ptr(rowptr), _ptr(colindex), _ptr(value), _ptr(x), _ptr(y),
            ctypes.addressof(out))
    fn = lib.spmv_acc_time_spmv_region

    def run() -> float:
        rc = fn(*args)
        if rc != 0:
            raise SpmvAccError(f"time_spmv_region failed ({rc}): {lib.spmv_acc_last_error_string().decode()}")
        return float(out.value)

    return run


def copy_ceiling_gbs(dst, src, reps: int = 5) -> float:
    """Streaming-copy ceiling in GB/s (read + write) for two equally sized GPU tensors."""
    lib = load_library()
    _require_cuda(dst, src)
    nbytes = (src.numel() * src.element_size()) // 16 * 16
    return float(lib.spmv_acc_copy_ceiling_gbs(_ptr(dst), _ptr(src), nbytes, reps))


def release_plans(rowptr=None) -> None:
    load_library().spmv_acc_release_plans(_ptr(rowptr))


def set_tune_cache(path: Optional[str]) -> None:
    """Persist the per-matrix timed choices in ``path`` (None: off); see include/spmv_acc.h."""
    load_library().spmv_acc_set_tune_cache(path.encode() if path else None)


def refresh_values(rowptr) -> int:
    """After changing ``value`` in place while the opt-in column slabs (tunable col_slabs) are in use: re-copy the values into the
    plan's slabs.  Returns the number of plans refreshed."""
    return int(load_library().spmv_acc_refresh_values(_ptr(rowptr)))


def check_plans() -> int:
    """After a device synchronisation: drop every cached plan (any thread's) whose stale-plan guard has fired; returns how
    many.  ``_check`` / spmv_acc_last_error only ask the plan the calling thread used last."""
    return int(load_library().spmv_acc_check_plans())


KERNEL_NAMES = {0: "rowblock", 1: "rowblock_plus", 2: "flat_tile", 3: "slab_passes", 4: "vector_tile", 5: "vector_row", 6: "wave_row", 7: "light",
                8: "block_row", 9: "col_slabs", 10: "scale_only"}  # include/spmv_acc.h: enum spmv_acc_kernel


def query_plan(rowptr, m: int):
    import numpy as np

    out = np.zeros(9, dtype=np.int32)
    found = load_library().spmv_acc_query_plan(_ptr(rowptr), m, _ptr(out))
    if not found:
        return None
    keys = ("nnz", "adaptive_branch", "vec", "flat_tiles", "plus_blocks", "aligned16", "stream_policy", "flat_fixup", "adaptive_family")
    info = dict(zip(keys, (int(v) for v in out)))
    info["slab_passes"] = max(0, int(load_library().spmv_acc_query_plan_slab_passes(_ptr(rowptr), m)))
    info["settled"] = int(load_library().spmv_acc_query_plan_settled(_ptr(rowptr), m)) == 1
    info["col16"] = int(load_library().spmv_acc_query_plan_col16(_ptr(rowptr), m))
    info["last_kernel"] = KERNEL_NAMES.get(int(load_library().spmv_acc_query_plan_last_kernel(_ptr(rowptr), m)), "none")
    return info


def set_strategy(name: str) -> None:
    lib = load_library()
    if lib.spmv_acc_set_strategy(name.encode()) != 0:
        lib.spmv_acc_clear_error()
        raise SpmvAccError(f"unknown KERNEL_STRATEGY {name!r}")


def get_strategy() -> str:
    lib = load_library()
    return lib.spmv_acc_strategy_name(lib.spmv_acc_get_strategy()).decode()
