// shard.cpp -- one rank's side of the row-range sharded SpMV behind the C boundary (include/spmv_acc.h, "row-sharded SpMV").
//
// Reference role: none -- the reference is single-GPU (hipSetDevice(0) at cli/main.cpp:89 and benchmark/main.cpp:96, no stream
// and no collective anywhere).  BASELINE's north_star adds the row-range partition with an RCCL allgather of the y sub-vectors
// over xGMI; spmv_acc_amd/dist.py does it with torch.distributed (one process per GPU), this file gives C / C++ consumers the
// same step (one process, one host thread per GPU: spmv-cli --gpus N).
//
// A shard is created once per rank and matrix.  A step computes this rank's rows STRAIGHT INTO their place in the gathered
// vector (the out-of-place kernels read the old slice wherever the caller keeps it) and moves the slice to every peer in place:
//   pipeline 1 : the kernels, then ONE ncclAllGather (in place: send buffer = this rank's slice of the receive buffer) on the
//                library stream;
//   pipeline C : the rows are cut into C chunks; chunk c's kernels go to the library stream, and as soon as they have finished
//                (event) the chunk travels on a second stream -- grouped ncclSend / ncclRecv with every peer, straight to its place
//                in their vectors -- while chunk c+1 computes.  xGMI is point to point, 7 links per GPU: the grouped fan-out
//                drives all of them at once.  (An allgather of a row sub-range would land chunk-major in the receive buffer, which
//                is why the chunks travel point to point.)  The library stream waits for the last chunk's arrival before anything
//                enqueued after the step runs.
// RCCL is resolved at run time (dlopen), so libspmv_acc.so keeps linking only the HIP runtime.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spmv_acc.h"
#include "engine.hpp"

using namespace spmv_acc;

namespace {

struct Rccl {
  typedef int (*all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
  typedef int (*send_t)(const void *, size_t, int, int, void *, hipStream_t);
  typedef int (*recv_t)(void *, size_t, int, int, void *, hipStream_t);
  typedef int (*group_t)(void);
  typedef int (*init_all_t)(void **, int, const int *);
  typedef int (*destroy_t)(void *);
  typedef int (*count_t)(void *, int *);
  all_gather_t all_gather = nullptr;
  send_t send = nullptr;
  recv_t recv = nullptr;
  group_t group_start = nullptr, group_end = nullptr;
  init_all_t init_all = nullptr;
  destroy_t destroy = nullptr;
  count_t count = nullptr, user_rank = nullptr;
  bool ok = false;
};

const Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void *lib = nullptr;
    if (const char *env = std::getenv("SPMV_ACC_RCCL_LIB")) lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    // the copy the process already uses (the caller made its communicator with it) before any other
    for (const char *name : {"librccl.so.1", "librccl.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    r.all_gather = reinterpret_cast<Rccl::all_gather_t>(dlsym(lib, "ncclAllGather"));
    r.send = reinterpret_cast<Rccl::send_t>(dlsym(lib, "ncclSend"));
    r.recv = reinterpret_cast<Rccl::recv_t>(dlsym(lib, "ncclRecv"));
    r.group_start = reinterpret_cast<Rccl::group_t>(dlsym(lib, "ncclGroupStart"));
    r.group_end = reinterpret_cast<Rccl::group_t>(dlsym(lib, "ncclGroupEnd"));
    r.init_all = reinterpret_cast<Rccl::init_all_t>(dlsym(lib, "ncclCommInitAll"));
    r.destroy = reinterpret_cast<Rccl::destroy_t>(dlsym(lib, "ncclCommDestroy"));
    r.count = reinterpret_cast<Rccl::count_t>(dlsym(lib, "ncclCommCount"));
    r.user_rank = reinterpret_cast<Rccl::count_t>(dlsym(lib, "ncclCommUserRank"));
    r.ok = r.all_gather && r.send && r.recv && r.group_start && r.group_end && r.init_all && r.destroy && r.count && r.user_rank;
  });
  return r;
}
constexpr int kNcclFloat64 = 8; // ncclDataType_t ncclFloat64 / ncclDouble (rccl.h)

__global__ void rebase_rowptr_kernel(const int *__restrict__ rp, int first, int count, int *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = rp[first + i] - rp[first];
}

int fail(int code, const std::string &what) {
  set_error(code, what);
  return code;
}

} // namespace

struct spmv_acc_shard {
  void *comm = nullptr;
  int rank = 0, world = 1, strategy = 0, m_local = 0, m_pad = 0, n = 0, nnz = 0, depth = 1, device = 0;
  struct Chunk {
    int a = 0, b = 0;   // rows of this rank's slice the chunk computes (clipped to m_local)
    int c0 = 0, c1 = 0; // rows of the PADDED slice the chunk's exchange moves (the same on every rank)
    const int *rp = nullptr;
    const int *ci = nullptr;
    const double *v = nullptr;
    int nnz = 0;
    int *owned_rp = nullptr; // rebased copy (pipeline > 1)
  };
  std::vector<Chunk> chunks;
  hipStream_t comm_stream = nullptr;
  std::vector<hipEvent_t> chunk_done;
  hipEvent_t step_begin = nullptr, arrived = nullptr;
};

extern "C" {

int spmv_acc_rccl_comm_init_all(void **comms, int ndev, const int *devices) {
  const Rccl &R = rccl();
  if (!R.ok) return fail(kErrNoDevice, "no RCCL found in the process (librccl.so.1 / librccl.so / SPMV_ACC_RCCL_LIB)");
  if (!comms || ndev <= 0) return fail(kErrBadArgument, "spmv_acc_rccl_comm_init_all: bad argument");
  const int rc = R.init_all(comms, ndev, devices);
  return rc == 0 ? kOk : fail(kErrHip, "ncclCommInitAll failed (ncclResult " + std::to_string(rc) + ")");
}

int spmv_acc_rccl_comm_destroy(void *comm) {
  const Rccl &R = rccl();
  if (!R.ok) return fail(kErrNoDevice, "no RCCL found in the process");
  return (!comm || R.destroy(comm) == 0) ? kOk : fail(kErrHip, "ncclCommDestroy failed");
}

int spmv_acc_shard_create(spmv_acc_shard_t *out, void *nccl_comm, int strategy, int m_local, int m_pad, int n, int nnz_local,
                          const int *d_rowptr, const int *d_colindex, const double *d_value, int pipeline) {
  if (!out) return fail(kErrBadArgument, "spmv_acc_shard_create: null handle pointer");
  *out = nullptr;
  if (!nccl_comm || m_local < 0 || m_pad < m_local || m_pad <= 0 || nnz_local < 0 || !d_rowptr)
    return fail(kErrBadArgument, "spmv_acc_shard_create: bad argument");
  const Rccl &R = rccl();
  if (!R.ok) return fail(kErrNoDevice, "spmv_acc_shard_create: no RCCL found in the process (librccl.so.1 / librccl.so / SPMV_ACC_RCCL_LIB)");
  spmv_acc_shard *S = new spmv_acc_shard();
  S->comm = nccl_comm;
  if (R.count(nccl_comm, &S->world) != 0 || R.user_rank(nccl_comm, &S->rank) != 0 || hipGetDevice(&S->device) != hipSuccess) {
    delete S;
    return fail(kErrHip, "spmv_acc_shard_create: cannot query the communicator");
  }
  S->strategy = strategy < 0 ? active_strategy() : strategy;
  S->m_local = m_local;
  S->m_pad = m_pad;
  S->n = n;
  S->nnz = nnz_local;
  S->depth = pipeline < 1 ? 1 : (pipeline > m_pad ? m_pad : pipeline);
  hipStream_t st = get_stream();
  bool ok = true;
  if (S->depth == 1) {
    spmv_acc_shard::Chunk c;
    c.a = 0, c.b = m_local, c.c0 = 0, c.c1 = m_pad, c.rp = d_rowptr, c.ci = d_colindex, c.v = d_value, c.nnz = nnz_local;
    S->chunks.push_back(c);
  } else {
    const int per = (m_pad + S->depth - 1) / S->depth;
    for (int k = 0; k < S->depth && k * per < m_pad; ++k) {
      spmv_acc_shard::Chunk c;
      c.c0 = k * per;
      c.c1 = (k + 1) * per < m_pad ? (k + 1) * per : m_pad;
      c.a = c.c0 < m_local ? c.c0 : m_local;
      c.b = c.c1 < m_local ? c.c1 : m_local;
      int ends[2] = {0, 0};
      ok = ok && hipMemcpy(&ends[0], d_rowptr + c.a, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(&ends[1], d_rowptr + c.b, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
      if (ok && c.b > c.a) {
        const int count = c.b - c.a + 1;
        ok = hipMalloc(reinterpret_cast<void **>(&c.owned_rp), sizeof(int) * static_cast<size_t>(count)) == hipSuccess;
        if (ok) hipLaunchKernelGGL(rebase_rowptr_kernel, dim3((count + 255) / 256), dim3(256), 0, st, d_rowptr, c.a, count, c.owned_rp);
      }
      c.rp = c.owned_rp;
      c.ci = d_colindex ? d_colindex + ends[0] : nullptr;
      c.v = d_value ? d_value + ends[0] : nullptr;
      c.nnz = ends[1] - ends[0];
      S->chunks.push_back(c);
    }
    ok = ok && hipStreamSynchronize(st) == hipSuccess &&
         hipStreamCreateWithFlags(&S->comm_stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&S->step_begin, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&S->arrived, hipEventDisableTiming) == hipSuccess;
    S->chunk_done.resize(S->chunks.size(), nullptr);
    for (auto &e : S->chunk_done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    spmv_acc_shard_destroy(S);
    return fail(kErrHip, "spmv_acc_shard_create: HIP failure while building the chunks");
  }
  *out = S;
  return kOk;
}

int spmv_acc_shard_destroy(spmv_acc_shard_t S) {
  if (!S) return kOk;
  for (auto &c : S->chunks) {
    if (c.owned_rp) {
      release_plans(c.owned_rp);
      (void)hipFree(c.owned_rp);
    }
  }
  for (auto &e : S->chunk_done)
    if (e) (void)hipEventDestroy(e);
  if (S->step_begin) (void)hipEventDestroy(S->step_begin);
  if (S->arrived) (void)hipEventDestroy(S->arrived);
  if (S->comm_stream) (void)hipStreamDestroy(S->comm_stream);
  delete S;
  return kOk;
}

int spmv_acc_shard_pipeline(spmv_acc_shard_t S) { return S ? static_cast<int>(S->chunks.size()) : 0; }

int spmv_acc_shard_step(spmv_acc_shard_t S, double alpha, double beta, const double *dx, const double *dy_in_local, double *dy_full) {
  if (!S || !dy_full) return fail(kErrBadArgument, "spmv_acc_shard_step: bad argument");
  const Rccl &R = rccl();
  // a shard belongs to the device that was current when it was made (its chunk arrays, events and second stream live there); a host
  // thread that drives several shards has to hipSetDevice before each step, and forgetting it would launch on the wrong card
  int current = -1;
  if (hipGetDevice(&current) != hipSuccess || current != S->device)
    return fail(kErrBadArgument, "spmv_acc_shard_step: the shard was created on device " + std::to_string(S->device) +
                                     ", the calling thread's current device is " + std::to_string(current));
  hipStream_t st = get_stream();
  double *own = dy_full + static_cast<size_t>(S->rank) * S->m_pad;
  clear_error();
  if (S->chunks.size() == 1) {
    const auto &c = S->chunks[0];
    if (S->m_local > 0) run_spmv(S->strategy, 0, alpha, beta, S->m_local, S->n, c.nnz, nullptr, c.rp, c.ci, c.v, dx, own, dy_in_local);
    if (last_error() != kOk) return last_error();
    // in place (send buffer = this rank's slice of the receive buffer), same stream: behind the kernels that wrote the slice
    const int rc = R.all_gather(own, dy_full, static_cast<size_t>(S->m_pad), kNcclFloat64, S->comm, st);
    return rc == 0 ? kOk : fail(kErrHip, "spmv_acc_shard_step: ncclAllGather failed (ncclResult " + std::to_string(rc) + ")");
  }
  // the exchange writes the peers' slices of dy_full: whatever was enqueued before this step (readers of the old vector) first
  if (hipEventRecord(S->step_begin, st) != hipSuccess || hipStreamWaitEvent(S->comm_stream, S->step_begin, 0) != hipSuccess)
    return fail(kErrHip, "spmv_acc_shard_step: event failure");
  for (size_t k = 0; k < S->chunks.size(); ++k) {
    const auto &c = S->chunks[k];
    if (c.b > c.a)
      run_spmv(S->strategy, 0, alpha, beta, c.b - c.a, S->n, c.nnz, nullptr, c.rp, c.ci, c.v, dx, own + c.a,
               dy_in_local ? dy_in_local + c.a : nullptr);
    if (last_error() != kOk) return last_error();
    if (hipEventRecord(S->chunk_done[k], st) != hipSuccess || hipStreamWaitEvent(S->comm_stream, S->chunk_done[k], 0) != hipSuccess)
      return fail(kErrHip, "spmv_acc_shard_step: event failure");
    if (S->world > 1) {
      const size_t count = static_cast<size_t>(c.c1 - c.c0);
      int rc = R.group_start();
      for (int d = 1; d < S->world && rc == 0; ++d) { // position d: send to rank + d, receive from rank - d
        const int dst = (S->rank + d) % S->world, src = (S->rank - d + S->world) % S->world;
        rc = R.send(own + c.c0, count, kNcclFloat64, dst, S->comm, S->comm_stream);
        if (rc == 0) rc = R.recv(dy_full + static_cast<size_t>(src) * S->m_pad + c.c0, count, kNcclFloat64, src, S->comm, S->comm_stream);
      }
      const int rc_end = R.group_end();
      if (rc != 0 || rc_end != 0)
        return fail(kErrHip, "spmv_acc_shard_step: grouped ncclSend / ncclRecv failed (ncclResult " + std::to_string(rc ? rc : rc_end) + ")");
    }
  }
  if (hipEventRecord(S->arrived, S->comm_stream) != hipSuccess || hipStreamWaitEvent(st, S->arrived, 0) != hipSuccess)
    return fail(kErrHip, "spmv_acc_shard_step: event failure");
  return kOk;
}

} // extern "C"
