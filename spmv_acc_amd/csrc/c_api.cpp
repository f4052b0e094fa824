// c_api.cpp -- extern "C" entry points declared in include/spmv_acc.h.
// (The C++-linkage twins the reference's executables link against live in cxx_api.cpp: a function
// cannot be declared with both linkages in one translation unit.)
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/spmv_acc.h"
#include "engine.hpp"

#include <string>
#include <cstdint>

using namespace spmv_acc;

// ---- pinned host -> device staging -----------------------------------------------------------------------
// The caller's arrays are pinned IN PLACE (hipHostRegister) and sent with hipMemcpyAsync on a private copy
// stream, all five transfers in flight together; arrays that cannot be registered (e.g. read-only mappings) go
// through a pinned double buffer instead.  One synchronisation at the end.
namespace {
// hipHostRegister is per process, not per thread: N host threads staging N shards of ONE host matrix (spmv-cli --gpus N) all pin the same x, and
// neighbouring shards' slices of colindex / values share a page at their common edge.  Two threads that both "succeed" in registering one range --
// or one that unregisters what another still copies from -- abort inside the runtime ("Memobj map does not have ptr"; found in round 4 by the
// first run of the N > 1 driver, over the mock communicator of tests/cxx/mock_rccl.cpp).  So registrations go through one table: an identical
// range is shared (reference-counted), a range that overlaps another one page-wise is not registered at all (its owner takes the bounce buffers).
struct PinTable {
  std::mutex mu;
  struct Range { uintptr_t lo, hi; int refs; };
  std::vector<Range> ranges;
  static constexpr uintptr_t kPage = 4096;
  bool pin(void *p, size_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p), lo = a & ~(kPage - 1), hi = (a + bytes + kPage - 1) & ~(kPage - 1);
    std::lock_guard<std::mutex> lk(mu);
    for (auto &r : ranges) {
      if (r.lo == a && a + bytes <= r.hi) { // the same array or a true prefix of it: every byte was passed to hipHostRegister -- share
        // (round 5: a longer request that merely ends inside the registration's last PAGE is not shared; it falls to the bounce buffers below)
        ++r.refs;
        return true;
      }
      if (lo < ((r.hi + kPage - 1) & ~(kPage - 1)) && (r.lo & ~(kPage - 1)) < hi) return false; // shares a page with another registration
    }
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    ranges.push_back({a, a + bytes, 1});
    return true;
  }
  void unpin(void *p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < ranges.size(); ++i) {
      if (ranges[i].lo != a) continue;
      if (--ranges[i].refs == 0) {
        (void)hipHostUnregister(p);
        ranges.erase(ranges.begin() + static_cast<long>(i));
      }
      return;
    }
  }
};
PinTable g_pins;

struct Stager {
  static constexpr size_t kChunk = 32u << 20; // 32 MiB per pinned bounce buffer (fallback path only)
  void *pinned[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  hipStream_t copy = nullptr;
  std::vector<void *> registered;
  bool ok = true;
  Stager() { ok = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) == hipSuccess; }
  ~Stager() {
    for (void *p : registered) g_pins.unpin(p);
    for (int i = 0; i < 2; ++i) {
      if (done[i]) (void)hipEventDestroy(done[i]);
      if (pinned[i]) (void)hipHostFree(pinned[i]);
    }
    if (copy) (void)hipStreamDestroy(copy);
  }
  bool ensure_bounce() {
    for (int i = 0; i < 2; ++i) {
      if (!pinned[i] && (hipHostMalloc(&pinned[i], kChunk, hipHostMallocDefault) != hipSuccess ||
                         hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess))
        return false;
    }
    return true;
  }
  bool upload(void *dst, const void *src, size_t bytes) {
    if (g_pins.pin(const_cast<void *>(src), bytes)) {
      registered.push_back(const_cast<void *>(src));
      return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, copy) == hipSuccess;
    }
    // fallback: memcpy into one pinned buffer while the other one is in flight
    if (!ensure_bounce()) return false;
    size_t off = 0;
    int slot = 0;
    while (off < bytes) {
      const size_t n = std::min(kChunk, bytes - off);
      if (hipEventSynchronize(done[slot]) != hipSuccess) return false;
      std::memcpy(pinned[slot], static_cast<const char *>(src) + off, n);
      if (hipMemcpyAsync(static_cast<char *>(dst) + off, pinned[slot], n, hipMemcpyHostToDevice, copy) != hipSuccess)
        return false;
      if (hipEventRecord(done[slot], copy) != hipSuccess) return false;
      off += n;
      slot ^= 1;
    }
    return true;
  }
};

template <typename T> bool stage_one(Stager &st, const T *h, size_t count, T **d) {
  *d = nullptr;
  if (!h || count == 0) return true;
  if (hipMalloc(reinterpret_cast<void **>(d), sizeof(T) * count) != hipSuccess) return false;
  return st.upload(*d, h, sizeof(T) * count);
}
} // namespace

namespace {
__global__ void copy_doubles_kernel(double *dst, const double *src, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
} // namespace

// ---- RCCL, resolved at run time (the library links only the HIP runtime) ------------------------------------------------------
namespace {
typedef int (*nccl_all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
nccl_all_gather_t resolve_all_gather() {
  static std::once_flag once;
  static nccl_all_gather_t fn = nullptr;
  std::call_once(once, [] {
    void *lib = nullptr;
    if (const char *env = std::getenv("SPMV_ACC_RCCL_LIB")) lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    // the copy the process already uses (the caller made its communicator with it) before any other
    for (const char *name : {"librccl.so.1", "librccl.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (lib) fn = reinterpret_cast<nccl_all_gather_t>(dlsym(lib, "ncclAllGather"));
  });
  return fn;
}
constexpr int kNcclFloat64 = 8; // ncclDataType_t ncclFloat64 / ncclDouble (rccl.h)
} // namespace

extern "C" {

const char *spmv_acc_version(void) { return "spmv_acc_amd 0.6 (gfx950)"; }

int spmv_acc_set_strategy(const char *name) {
  const int s = parse_strategy(name);
  if (s < 0) {
    set_error(kErrUnknownStrategy, std::string("unknown KERNEL_STRATEGY: ") + (name ? name : "(null)"));
    return -1;
  }
  return set_active_strategy(s);
}
int spmv_acc_set_strategy_id(int strategy) { return set_active_strategy(strategy); }
int spmv_acc_get_strategy(void) { return active_strategy(); }
const char *spmv_acc_strategy_name(int strategy) { return strategy_name(strategy); }
int spmv_acc_parse_strategy(const char *name) { return parse_strategy(name); }

void sparse_spmv(int htrans, const double halpha, const double hbeta, int hm, int hn, const int *rowptr,
                 const int *colindex, const double *value, const double *x, double *y) {
  run_spmv(active_strategy(), htrans, halpha, hbeta, hm, hn, /*nnz=*/-1, /*h_rowptr=*/nullptr, rowptr, colindex, value,
           x, y);
}

void spmv_acc_csr_spmv(int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
                       const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                       double *dy) {
  run_spmv(active_strategy(), trans, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy);
}

void spmv_acc_csr_spmv_strategy(int strategy, int trans, double alpha, double beta, int m, int n, int nnz,
                                const int *h_rowptr, const int *d_rowptr, const int *d_colindex,
                                const double *d_value, const double *dx, double *dy) {
  run_spmv(strategy, trans, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy);
}

void spmv_acc_csr_spmv_oop(int strategy, int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
                           const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                           const double *dy_in, double *dy_out) {
  run_spmv(strategy < 0 ? active_strategy() : strategy, trans, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx,
           dy_out, dy_in);
}

int spmv_acc_csr_spmv_chunks(int strategy, double alpha, double beta, int n, int nchunks, const int *row_cuts, const int *nnz_ends,
                             const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                             const double *dy_in, double *dy_out, void *const *streams, void *const *events) {
  if (nchunks < 0 || (nchunks > 0 && (!row_cuts || !nnz_ends || !d_rowptr || !dy_out || !streams))) {
    set_error(kErrBadArgument, "spmv_acc_csr_spmv_chunks: bad argument");
    return kErrBadArgument;
  }
  const int sid = strategy < 0 ? active_strategy() : strategy;
  hipStream_t keep = get_stream();
  int first_err = kOk;
  std::string first_what;
  clear_error();
  for (int k = 0; k < nchunks; ++k) {
    const int a = row_cuts[k], b = row_cuts[k + 1];
    hipStream_t cs = static_cast<hipStream_t>(streams[k & 1]);
    if (b > a && first_err == kOk) {
      set_stream(cs);
      run_spmv(sid, 0, alpha, beta, b - a, n, nnz_ends[k], nullptr, d_rowptr + a, d_colindex, d_value, dx, dy_out + a, dy_in ? dy_in + a : nullptr);
      if (last_error() != kOk) {
        first_err = last_error();
        first_what = last_error_string();
      }
    }
    // (the event is recorded whatever happened: the caller's exchange of chunk k waits for it)
    if (events && events[k] && hipEventRecord(static_cast<hipEvent_t>(events[k]), cs) != hipSuccess && first_err == kOk) {
      first_err = kErrHip;
      first_what = "spmv_acc_csr_spmv_chunks: hipEventRecord failed";
    }
  }
  set_stream(keep);
  if (first_err != kOk) set_error(first_err, first_what);
  return first_err;
}

int spmv_acc_break_points_len(int nnz, int stride) {
  if (stride <= 0 || nnz < 0) return -1;
  return nnz / stride + (nnz % stride ? 1 : 0) + 1;
}

int spmv_acc_break_points(const int *d_rowptr, int m, int nnz, int stride, int *d_break_points, int bp_len) {
  if (!d_rowptr || !d_break_points || m < 0 || nnz < 0 || stride <= 0 || bp_len < 1) {
    set_error(kErrBadArgument, "spmv_acc_break_points: bad argument");
    return kErrBadArgument;
  }
  launch_break_points(get_stream(), d_rowptr, m, nnz, stride, d_break_points, bp_len);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(kErrHip, hipGetErrorString(e));
    return kErrHip;
  }
  return kOk;
}

int spmv_acc_adaptive_plus_analyze(int m, int min_nnz_per_block, int threads_per_block, int vec_size,
                                   const int *h_rowptr, int *h_break_points, int bp_cap,
                                   int *h_first_block_of_row) {
  if (m < 0 || !h_rowptr || !h_break_points || !h_first_block_of_row || min_nnz_per_block <= 0 || vec_size <= 0 ||
      threads_per_block < vec_size) {
    set_error(kErrBadArgument, "spmv_acc_adaptive_plus_analyze: bad argument");
    return -2;
  }
  std::vector<int> bp, fbr;
  const int blocks = plus_analyze_host(m, min_nnz_per_block, threads_per_block, vec_size, h_rowptr, bp, fbr);
  if (static_cast<int>(bp.size()) > bp_cap) return -1;
  std::copy(bp.begin(), bp.end(), h_break_points);
  std::copy(fbr.begin(), fbr.end(), h_first_block_of_row);
  return blocks;
}

int spmv_acc_adaptive_plus_analyze_device(int m, int min_nnz_per_block, int threads_per_block, int vec_size,
                                          const int *d_rowptr, int *d_break_points, int bp_cap,
                                          int *d_first_block_of_row) {
  if (m < 0 || !d_rowptr || !d_break_points || !d_first_block_of_row || min_nnz_per_block <= 0 || vec_size <= 0 ||
      threads_per_block < vec_size || bp_cap < 1) {
    set_error(kErrBadArgument, "spmv_acc_adaptive_plus_analyze_device: bad argument");
    return -2;
  }
  return plus_analyze_device(m, min_nnz_per_block, threads_per_block, vec_size, d_rowptr, d_break_points, bp_cap,
                             d_first_block_of_row);
}

int spmv_acc_adaptive_plus_vec(int m, int nnz) { return plus_pick_vec(m, nnz); }

int spmv_acc_adaptive_branch(int m, int rp_quarter, int rp_half, int rp_three_quarter, int rp_last) {
  if (m <= 0) return 0;
  RowptrSamples s;
  s.q1 = rp_quarter;
  s.half = rp_half;
  s.q3 = rp_three_quarter;
  s.last = rp_last;
  return adaptive_branch(m, s);
}

int spmv_acc_partition_rows(int m, int parts, int mode, const int *h_rowptr, int *row_begin) {
  if (m < 0 || parts <= 0 || !row_begin || (mode == 1 && !h_rowptr)) {
    set_error(kErrBadArgument, "spmv_acc_partition_rows: bad argument");
    return kErrBadArgument;
  }
  row_begin[0] = 0;
  row_begin[parts] = m;
  if (mode == 0) {
    // equal row counts: the first (m mod parts) ranks get one more row only if m is not divisible;
    // callers that need identical shard sizes for an allgather pad to ceil(m/parts)
    const long long per = (static_cast<long long>(m) + parts - 1) / parts;
    for (int p = 1; p < parts; ++p) row_begin[p] = static_cast<int>(std::min<long long>(per * p, m));
  } else {
    const long long nnz = h_rowptr[m];
    for (int p = 1; p < parts; ++p) {
      const long long target = nnz * p / parts;
      const int *it = std::lower_bound(h_rowptr, h_rowptr + m + 1, static_cast<int>(target));
      int r = static_cast<int>(it - h_rowptr);
      r = std::max(r, row_begin[p - 1]);
      row_begin[p] = std::min(r, m);
    }
  }
  return kOk;
}

int spmv_acc_sharded_spmv(void *nccl_comm, int strategy, double alpha, double beta, int m_local, int m_pad, int n, int nnz_local,
                          const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                          double *dy_local, double *dy_full) {
  if (!nccl_comm || m_local < 0 || m_pad < m_local || m_pad <= 0 || !dy_local || !dy_full) {
    set_error(kErrBadArgument, "spmv_acc_sharded_spmv: bad argument");
    return kErrBadArgument;
  }
  const nccl_all_gather_t all_gather = resolve_all_gather();
  if (!all_gather) {
    set_error(kErrNoDevice, "spmv_acc_sharded_spmv: no RCCL found in the process (librccl.so.1 / librccl.so / SPMV_ACC_RCCL_LIB)");
    return kErrNoDevice;
  }
  clear_error();
  if (m_local > 0) run_spmv(strategy, 0, alpha, beta, m_local, n, nnz_local, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy_local);
  if (last_error() != kOk) return last_error();
  // same stream: the collective reads y_local behind the kernels that wrote it
  const int rc = all_gather(dy_local, dy_full, static_cast<size_t>(m_pad), kNcclFloat64, nccl_comm, get_stream());
  if (rc != 0) {
    set_error(kErrHip, "spmv_acc_sharded_spmv: ncclAllGather failed (ncclResult " + std::to_string(rc) + ")");
    return kErrHip;
  }
  return kOk;
}

int spmv_acc_stage_csr(int m, int n, int nnz, const int *h_rowptr, const int *h_colindex, const double *h_value,
                       const double *h_x, const double *h_y, int **d_rowptr, int **d_colindex, double **d_value,
                       double **d_x, double **d_y) {
  if (m < 0 || n < 0 || nnz < 0 || !d_rowptr || !d_colindex || !d_value || !d_x || !d_y) {
    set_error(kErrBadArgument, "spmv_acc_stage_csr: bad argument");
    return kErrBadArgument;
  }
  Stager st;
  bool ok = st.ok;
  ok = ok && stage_one(st, h_rowptr, static_cast<size_t>(m) + 1, d_rowptr);
  ok = ok && stage_one(st, h_colindex, static_cast<size_t>(nnz), d_colindex);
  ok = ok && stage_one(st, h_value, static_cast<size_t>(nnz), d_value);
  ok = ok && stage_one(st, h_x, static_cast<size_t>(n), d_x);
  ok = ok && stage_one(st, h_y, static_cast<size_t>(m), d_y);
  ok = ok && hipStreamSynchronize(st.copy) == hipSuccess;
  if (!ok) {
    const hipError_t e = hipGetLastError();
    set_error(kErrHip, std::string("spmv_acc_stage_csr: ") + hipGetErrorString(e));
    return kErrHip;
  }
  return kOk;
}

int spmv_acc_free_device(void *p) {
  if (!p) return kOk;
  return hipFree(p) == hipSuccess ? kOk : kErrHip;
}

void spmv_acc_release_plans(const int *d_rowptr) { release_plans(d_rowptr); }
int spmv_acc_cached_plans(void) { return cached_plan_count(); }
int spmv_acc_check_plans(void) { return check_plans(); }
int spmv_acc_query_plan(const int *d_rowptr, int m, int *out) {
  PlanInfo info;
  if (!out || !query_plan(d_rowptr, m, &info)) return 0;
  out[0] = info.nnz;
  out[1] = info.adaptive_branch;
  out[2] = info.vec;
  out[3] = info.flat_tiles;
  out[4] = info.plus_blocks;
  out[5] = info.aligned16;
  out[6] = info.stream_policy;
  out[7] = info.flat_fixup;
  out[8] = info.adaptive_family;
  return 1;
}
int spmv_acc_query_plan_beta0(const int *d_rowptr, int m) {
  PlanInfo info;
  return query_plan(d_rowptr, m, &info) ? info.adaptive_family_beta0 : -2;
}
int spmv_acc_query_plan_slab_passes(const int *d_rowptr, int m) {
  PlanInfo info;
  return query_plan(d_rowptr, m, &info) ? info.slab_passes : -2;
}
int spmv_acc_query_plan_last_kernel(const int *d_rowptr, int m) {
  PlanInfo info;
  return query_plan(d_rowptr, m, &info) ? info.last_kernel : -2;
}
int spmv_acc_query_plan_col16(const int *d_rowptr, int m) {
  PlanInfo info;
  return query_plan(d_rowptr, m, &info) ? info.col16 : -2;
}
int spmv_acc_query_plan_settled(const int *d_rowptr, int m) {
  PlanInfo info;
  return query_plan(d_rowptr, m, &info) ? info.settled : -2;
}

void spmv_acc_set_stream(void *hip_stream) { set_stream(static_cast<hipStream_t>(hip_stream)); }
void *spmv_acc_get_stream(void) { return static_cast<void *>(get_stream()); }

int spmv_acc_set_tunable(const char *name, int value) { return name ? set_tunable(name, value) : -1; }
int spmv_acc_get_tunable(const char *name) { return name ? get_tunable(name) : -1; }
void spmv_acc_reset_tunables(void) { reset_tunables(); }

double spmv_acc_last_prepare_us(void) { return last_prepare_us(); }

int spmv_acc_last_error(void) { return last_error(); }
const char *spmv_acc_last_error_string(void) { return last_error_string(); }
void spmv_acc_clear_error(void) { clear_error(); }

void spmv_acc_set_tune_cache(const char *path) { set_tune_cache(path); }
int spmv_acc_refresh_values(const int *d_rowptr) { return refresh_values(d_rowptr); }

int spmv_acc_prepare(int strategy, int m, int n, int nnz, const int *h_rowptr, const int *d_rowptr, const int *d_colindex,
                     const double *d_value, const double *dx, float *ms_out) {
  return spmv_acc_prepare_beta(strategy, 1.0, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, ms_out);
}

int spmv_acc_prepare_beta(int strategy, double beta, int m, int n, int nnz, const int *h_rowptr, const int *d_rowptr,
                          const int *d_colindex, const double *d_value, const double *dx, float *ms_out) {
  if (m <= 0) return kOk;
  hipStream_t st = get_stream();
  double *scratch = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&scratch), sizeof(double) * static_cast<size_t>(m)) != hipSuccess) {
    set_error(kErrHip, "spmv_acc_prepare: hipMalloc failed");
    return kErrHip;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = ms_out && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  clear_error();
  if (timed) (void)hipEventRecord(e0, st);
  // the caller's y is not involved: the first call runs into a zeroed scratch, in the caller's beta CLASS (beta == 0: y is only
  // written; else it is read too) -- the choices that depend on the y read (cache policy, adaptive's kernel family, flat's cut-row
  // form) are timed and kept per class.  spmv_acc_prepare = beta 1, the reference's protocol.
  (void)hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(m), st);
  {
    // everything up front: no tuning budget, and the calls are repeated while they still find plan work to do (a family that won adaptive's
    // comparison refines its own sub-choices on its next call; a second look may be open) -- what ordinary calls spread over the first few SpMVs
    UnboundedTuningScope unbounded;
    for (int round = 0; round < 6; ++round) {
      run_spmv(strategy, 0, 1.0, beta != 0.0 ? 1.0 : 0.0, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, scratch);
      if (last_error() != kOk || last_prepare_us() <= 0.0) break;
    }
  }
  if (timed) (void)hipEventRecord(e1, st);
  int rc = hipStreamSynchronize(st) == hipSuccess ? kOk : kErrHip;
  if (timed && rc == kOk && hipEventElapsedTime(ms_out, e0, e1) != hipSuccess) rc = kErrHip;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(scratch);
  if (rc != kOk) set_error(kErrHip, "spmv_acc_prepare: HIP failure");
  if (rc == kOk && last_error() != kOk) rc = last_error();
  return rc;
}

int spmv_acc_time_spmv(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                       const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                       const double *dx, double *dy, const double *d_y0, float *ms_out) {
  return spmv_acc_time_spmv_events(strategy, iters, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy, d_y0,
                                   ms_out, 0u);
}

// d_flush / flush_bytes (spmv_acc_time_spmv_cold): after y has been restored and BEFORE the start event, the second half of the scratch buffer is
// overwritten with the first by the copy kernel under the DEFAULT cache policy -- flush_bytes of traffic that allocate in the L2s and the 256 MB
// Infinity Cache and displace whatever the previous launch left there: the timed launch starts cold.
static int time_spmv_per_launch(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                                const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                                const double *dx, double *dy, const double *d_y0, float *ms_out, unsigned event_flags, void *d_flush,
                                long long flush_bytes);
int spmv_acc_time_spmv_events(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                              const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                              const double *dx, double *dy, const double *d_y0, float *ms_out, unsigned event_flags) {
  return time_spmv_per_launch(strategy, iters, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy, d_y0, ms_out, event_flags,
                              nullptr, 0);
}
int spmv_acc_time_spmv_cold(int strategy, int iters, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
                            const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx, double *dy,
                            const double *d_y0, void *d_flush, long long flush_bytes, float *ms_out) {
  if (!d_flush || flush_bytes < 32 || reinterpret_cast<uintptr_t>(d_flush) % 16 != 0) {
    set_error(kErrBadArgument, "spmv_acc_time_spmv_cold: needs a 16-byte aligned scratch buffer");
    return kErrBadArgument;
  }
  return time_spmv_per_launch(strategy, iters, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy, d_y0, ms_out, 0u, d_flush,
                              flush_bytes);
}
static int time_spmv_per_launch(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                                const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                                const double *dx, double *dy, const double *d_y0, float *ms_out, unsigned event_flags, void *d_flush,
                                long long flush_bytes) {
  if (iters <= 0 || !ms_out) {
    set_error(kErrBadArgument, "spmv_acc_time_spmv: bad argument");
    return kErrBadArgument;
  }
  // a SETTLED plan is what gets timed (round 4): whatever the first-call budget left open is finished here, outside the timed launches
  if (spmv_acc_prepare_beta(strategy, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, nullptr) != kOk) return last_error();
  hipStream_t st = get_stream();
  std::vector<hipEvent_t> ev(2 * static_cast<size_t>(iters));
  for (auto &e : ev) {
    if (hipEventCreateWithFlags(&e, event_flags) != hipSuccess) {
      set_error(kErrHip, "hipEventCreate failed");
      return kErrHip;
    }
  }
  clear_error();
  // y reset: a copy KERNEL on the same queue as the SpMV (SPMV_ACC_RESET_MEMCPY=1: hipMemcpyAsync, which may run on a DMA engine
  // and hand over to the compute queue through a signal -- that hand-over then sits between the first event and the kernel)
  static const bool reset_memcpy = [] {
    const char *e = std::getenv("SPMV_ACC_RESET_MEMCPY");
    return e && *e && *e != '0';
  }();
  // SPMV_ACC_RESET_NT=0: the reset copy with default-policy loads / stores (rounds 1-4); default since round 5: non-temporal, like the DMA write of the
  // reference's own reset (hipMemcpy host -> device, benchmark/csr_spmv.hpp:68) it does not park 2 * 8 * m bytes of y in the L2s in front of the timed launch
  static const bool reset_nt = [] {
    const char *e = std::getenv("SPMV_ACC_RESET_NT");
    return !(e && *e == '0');
  }();
  const size_t ybytes = sizeof(double) * static_cast<size_t>(m);
  const bool kernel_copy = !reset_memcpy && reinterpret_cast<uintptr_t>(dy) % 16 == 0 && reinterpret_cast<uintptr_t>(d_y0) % 16 == 0;
  for (int i = 0; i < iters; ++i) {
    if (d_y0) {
      const size_t body = kernel_copy ? ybytes / 16 * 16 : 0;
      if (body) launch_stream_copy(st, dy, d_y0, static_cast<long long>(body), reset_nt);
      if (body < ybytes && kernel_copy) // (odd m: the last double)
        hipLaunchKernelGGL(copy_doubles_kernel, dim3(1), dim3(64), 0, st, dy + body / 8, d_y0 + body / 8, static_cast<int>((ybytes - body) / 8));
      else if (body < ybytes)
        (void)hipMemcpyAsync(dy, d_y0, ybytes, hipMemcpyDeviceToDevice, st);
    }
    if (d_flush) {
      const long long half = flush_bytes / 2 / 16 * 16;
      launch_stream_copy(st, static_cast<char *>(d_flush) + half, d_flush, half, /*non_temporal=*/false);
    }
    (void)hipEventRecord(ev[2 * i], st);
    run_spmv(strategy, 0, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy);
    (void)hipEventRecord(ev[2 * i + 1], st);
  }
  int rc = kOk;
  if (hipStreamSynchronize(st) != hipSuccess) rc = kErrHip;
  for (int i = 0; i < iters && rc == kOk; ++i) {
    if (hipEventElapsedTime(&ms_out[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = kErrHip;
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  if (rc != kOk) set_error(kErrHip, "spmv_acc_time_spmv: HIP failure while timing");
  if (rc == kOk && last_error() != kOk) rc = last_error();
  return rc;
}

// Same protocol with ONE event pair around all `iters` launches: total_ms / iters is the average launch
// duration including the back-to-back launch gaps (what a solver loop sees); no per-launch event markers
// sit between the kernels.
int spmv_acc_time_spmv_total(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                             const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                             const double *dx, double *dy, float *total_ms_out) {
  if (iters <= 0 || !total_ms_out) {
    set_error(kErrBadArgument, "spmv_acc_time_spmv_total: bad argument");
    return kErrBadArgument;
  }
  if (spmv_acc_prepare_beta(strategy, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, nullptr) != kOk) return last_error(); // (as above)
  hipStream_t st = get_stream();
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    set_error(kErrHip, "hipEventCreate failed");
    return kErrHip;
  }
  clear_error();
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i)
    run_spmv(strategy, 0, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy);
  (void)hipEventRecord(e1, st);
  int rc = kOk;
  if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(total_ms_out, e0, e1) != hipSuccess) rc = kErrHip;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != kOk) set_error(kErrHip, "spmv_acc_time_spmv_total: HIP failure while timing");
  if (rc == kOk && last_error() != kOk) rc = last_error();
  return rc;
}

// The per-launch protocol of spmv_acc_time_spmv_events with the KERNEL CLOCK on (round 5): besides the event pair around every call -- the
// reference harness's figure, which holds the marker packets and the dispatch latency of the protocol too -- every kernel the call launches carries
// its own start / stop events (hipExtLaunchKernelGGL: the dispatch's begin / end timestamps, what rocprofv3 --kernel-trace reports), and
// kernel_ms_out[i] is their SUM over call i's launches: the call's kernel time without the gaps.  launches_out (may be null): kernels per call.
static bool reset_nt_copy() { // (SPMV_ACC_RESET_NT, see spmv_acc_time_spmv_events)
  static const bool nt = [] {
    const char *e = std::getenv("SPMV_ACC_RESET_NT");
    return !(e && *e == '0');
  }();
  return nt;
}
int spmv_acc_time_spmv_kernels(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                               const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                               const double *dx, double *dy, const double *d_y0, float *event_ms_out, float *kernel_ms_out, int *launches_out) {
  if (iters <= 0 || !kernel_ms_out) {
    set_error(kErrBadArgument, "spmv_acc_time_spmv_kernels: bad argument");
    return kErrBadArgument;
  }
  if (spmv_acc_prepare_beta(strategy, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, nullptr) != kOk) return last_error();
  hipStream_t st = get_stream();
  std::vector<hipEvent_t> ev(2 * static_cast<size_t>(iters), nullptr);
  for (auto &e : ev) {
    if (hipEventCreate(&e) != hipSuccess) {
      e = nullptr;
      for (auto &made : ev)
        if (made) (void)hipEventDestroy(made);
      set_error(kErrHip, "hipEventCreate failed");
      return kErrHip;
    }
  }
  clear_error();
  const size_t ybytes = sizeof(double) * static_cast<size_t>(m);
  std::vector<size_t> first(static_cast<size_t>(iters) + 1, 0);
  kernel_clock_begin();
  for (int i = 0; i < iters; ++i) {
    if (d_y0) { // y reset outside both clocks (the copy is launched with the kernel clock off)
      if (ybytes % 16 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0 && reinterpret_cast<uintptr_t>(d_y0) % 16 == 0)
        launch_stream_copy(st, dy, d_y0, static_cast<long long>(ybytes), reset_nt_copy());
      else
        (void)hipMemcpyAsync(dy, d_y0, ybytes, hipMemcpyDeviceToDevice, st);
    }
    (void)hipEventRecord(ev[2 * i], st);
    first[i] = kernel_clock_used();
    kernel_clock_set(true);
    run_spmv(strategy, 0, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy);
    kernel_clock_set(false);
    (void)hipEventRecord(ev[2 * i + 1], st);
  }
  first[iters] = kernel_clock_used();
  int rc = hipStreamSynchronize(st) == hipSuccess ? kOk : kErrHip;
  if (kernel_clock_failed()) rc = kErrHip;
  for (int i = 0; i < iters && rc == kOk; ++i) {
    float ms = 0.f;
    if (event_ms_out && hipEventElapsedTime(&event_ms_out[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = kErrHip;
    double sum = 0.0;
    for (size_t k = first[i]; k + 1 < first[i + 1] && rc == kOk; k += 2) { // (pairs: [k] start, [k + 1] stop)
      if (hipEventElapsedTime(&ms, kernel_clock_event(k), kernel_clock_event(k + 1)) != hipSuccess) rc = kErrHip;
      sum += ms;
    }
    kernel_ms_out[i] = static_cast<float>(sum);
    if (launches_out) launches_out[i] = static_cast<int>((first[i + 1] - first[i]) / 2);
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  kernel_clock_release();
  if (rc != kOk) set_error(kErrHip, "spmv_acc_time_spmv_kernels: HIP failure while timing");
  if (rc == kOk && last_error() != kOk) rc = last_error();
  return rc;
}

// One timed REGION and nothing else (round 5; bench.py's `value`): `iters` back-to-back launches between one event pair -- no plan work (the caller
// has settled the plan: spmv_acc_prepare), no allocation, no event creation (one event pair per host thread, made on first use) -- so that a
// wall clock around this call, bracketed by device synchronisations, reads the launches' own time.  (spmv_acc_time_spmv_total settles the plan
// itself, i.e. allocates and frees a scratch y inside the call: 0.4 ms of host time, 14 % of a 20-launch region on the headline matrix.)
int spmv_acc_time_spmv_region(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                              const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                              const double *dx, double *dy, float *total_ms_out) {
  if (iters <= 0 || !total_ms_out) {
    set_error(kErrBadArgument, "spmv_acc_time_spmv_region: bad argument");
    return kErrBadArgument;
  }
  thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  thread_local int ev_device = -1;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error(kErrHip, "spmv_acc_time_spmv_region: hipGetDevice failed");
    return kErrHip;
  }
  if (!ev[0] || !ev[1] || ev_device != dev) { // (events belong to a device: a thread that moved on drops the old pair before it makes a new one)
    for (auto &e : ev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) {
      for (auto &e : ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
      }
      (void)hipGetLastError();
      set_error(kErrHip, "spmv_acc_time_spmv_region: hipEventCreate failed");
      return kErrHip;
    }
    ev_device = dev;
  }
  hipStream_t st = get_stream();
  clear_error();
  const unsigned work0 = plan_work_count();
  (void)hipEventRecord(ev[0], st);
  for (int i = 0; i < iters; ++i)
    run_spmv(strategy, 0, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy);
  (void)hipEventRecord(ev[1], st);
  int rc = kOk;
  if (hipEventSynchronize(ev[1]) != hipSuccess || hipEventElapsedTime(total_ms_out, ev[0], ev[1]) != hipSuccess) rc = kErrHip;
  if (rc != kOk) set_error(kErrHip, "spmv_acc_time_spmv_region: HIP failure while timing");
  if (rc == kOk && last_error() != kOk) rc = last_error();
  // The region is for SETTLED plans (the caller prepares first): plan work inside it -- a structural pass, a deferred timing phase with its trial
  // launches -- would sit between the two events and inflate the figure without a sign.  Reported as an error; the time is still written.
  if (rc == kOk && plan_work_count() != work0) {
    set_error(kErrBadArgument, "spmv_acc_time_spmv_region: the plan was not settled -- plan work ran inside the timed region (call spmv_acc_prepare first)");
    rc = kErrBadArgument;
  }
  return rc;
}

// Streaming-copy ceiling of the device in this process: dst[i] = src[i] over `bytes` (multiple of 16) with the
// same 16-B non-temporal loads the SpMV kernels use; returns GB/s counting read + write bytes, best of `reps`.
double spmv_acc_copy_ceiling_gbs(void *dst, const void *src, long long bytes, int reps) {
  if (!dst || !src || bytes < 16 || reps <= 0) return -1.0;
  hipStream_t st = get_stream();
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  float best = 1e30f;
  for (int nt = 1; nt >= 0 && best > 0.f; --nt) { // non-temporal and default-policy copies: the faster one is the ceiling (a tunable chose until round 5)
    for (int r = 0; r < reps + 1; ++r) {
      (void)hipEventRecord(e0, st);
      launch_stream_copy(st, dst, src, bytes, nt != 0);
      (void)hipEventRecord(e1, st);
      float ms = 0.f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
        best = -1.f;
        break;
      }
      if (r > 0 && ms < best) best = ms; // first repetition is a warm-up
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return best > 0.f ? 2.0 * static_cast<double>(bytes) / (best * 1e-3) / 1e9 : -1.0;
}

} // extern "C"
