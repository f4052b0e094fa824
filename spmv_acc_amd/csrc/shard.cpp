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
//   pipeline C : the rows are cut into C chunks -- row SUB-RANGES of the shard's own arrays (rowptr + a, the whole colindex / value arrays: the kernels
//                accept rowptr[0] > 0), nothing is copied; the chunks' kernels alternate over two compute streams of the shard's own (consecutive chunks are
//                independent: on ONE stream each kernel would wait for its predecessor's last wavefront, 8 launches cost 0.245 ms against 0.156 for one;
//                alternating, a chunk's tail overlaps the next one's start), and as soon as a chunk's kernels have finished
//                (event) the chunk travels on the exchange stream -- grouped ncclSend / ncclRecv with every peer, straight to its place
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
    const int *rp = nullptr; // rowptr + a: the chunk as a row sub-range of the shard's arrays (rp[0] = first non-zero of row a, not 0)
    int nnz_end = 0;         // rowptr[b]: what the library takes as `nnz` of an un-rebased view
  };
  const int *ci = nullptr;
  const double *v = nullptr;
  hipStream_t chunk_stream[2] = {nullptr, nullptr}; // pipeline > 1: the chunks' kernels alternate over these
  bool prepared[2] = {false, false};                // per beta class: every chunk's plan is built and tuned (spmv_acc_shard_prepare)
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
  S->ci = d_colindex;
  S->v = d_value;
  bool ok = true;
  const int per = (m_pad + S->depth - 1) / S->depth;
  std::vector<int> cut; // rows where chunks begin / end, clipped to m_local
  for (int k = 0; k < S->depth && k * per < m_pad; ++k) {
    spmv_acc_shard::Chunk c;
    c.c0 = k * per;
    c.c1 = (k + 1) * per < m_pad ? (k + 1) * per : m_pad;
    c.a = c.c0 < m_local ? c.c0 : m_local;
    c.b = c.c1 < m_local ? c.c1 : m_local;
    c.rp = d_rowptr + c.a;
    S->chunks.push_back(c);
  }
  if (S->chunks.size() == 1) {
    S->chunks[0].nnz_end = nnz_local; // (the whole shard: rowptr[0] = 0 by the entry's contract)
  } else {
    for (auto &c : S->chunks) // one 4-byte read per chunk, once per shard
      ok = ok && hipMemcpy(&c.nnz_end, d_rowptr + c.b, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&S->comm_stream, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&S->chunk_stream[0], hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithFlags(&S->chunk_stream[1], hipStreamNonBlocking) == hipSuccess &&
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
  if (S->chunks.size() > 1)
    for (auto &c : S->chunks) release_plans(c.rp, c.b - c.a); // (the chunk views' plans, by (pointer, rows): chunk 0 shares its pointer with the whole-shard plan, which is the caller's to release)
  for (auto &q : S->chunk_stream)
    if (q) (void)hipStreamDestroy(q);
  for (auto &e : S->chunk_done)
    if (e) (void)hipEventDestroy(e);
  if (S->step_begin) (void)hipEventDestroy(S->step_begin);
  if (S->arrived) (void)hipEventDestroy(S->arrived);
  if (S->comm_stream) (void)hipStreamDestroy(S->comm_stream);
  delete S;
  return kOk;
}

int spmv_acc_shard_pipeline(spmv_acc_shard_t S) { return S ? static_cast<int>(S->chunks.size()) : 0; }

// Build and tune every chunk's plan for the caller's beta class, so that a step only enqueues: plan building allocates, frees and
// synchronises (implicit device-wide syncs), which must not happen between the chunk exchanges of a step that other ranks are already
// in.  Collective-free; every rank calls it before its first step (a step on an unprepared shard still works: the first one then
// builds what is missing, as a plain SpMV call does).
int spmv_acc_shard_prepare(spmv_acc_shard_t S, double beta, const double *dx) {
  if (!S || !dx) return fail(kErrBadArgument, "spmv_acc_shard_prepare: bad argument");
  int current = -1;
  if (hipGetDevice(&current) != hipSuccess || current != S->device)
    return fail(kErrBadArgument, "spmv_acc_shard_prepare: the shard was created on device " + std::to_string(S->device) +
                                     ", the calling thread's current device is " + std::to_string(current));
  const int cls = beta != 0.0 ? 1 : 0;
  hipStream_t st = get_stream();
  int rc = kOk;
  for (size_t k = 0; k < S->chunks.size() && rc == kOk; ++k) {
    const auto &c = S->chunks[k];
    if (c.b <= c.a) continue;
    // on the stream the chunk's kernels will use (a plan orders itself behind its previous launches, whatever stream they were on)
    if (S->chunks.size() > 1) set_stream(S->chunk_stream[k & 1]);
    rc = spmv_acc_prepare_beta(S->strategy, beta, c.b - c.a, S->n, c.nnz_end, nullptr, c.rp, S->ci, S->v, dx, nullptr);
  }
  set_stream(st);
  if (rc == kOk) S->prepared[cls] = true;
  return rc;
}

int spmv_acc_shard_step(spmv_acc_shard_t S, double alpha, double beta, const double *dx, const double *dy_in_local, double *dy_full) {
  if (!S || !dy_full) return fail(kErrBadArgument, "spmv_acc_shard_step: bad argument");
  const Rccl &R = rccl();
  // a shard belongs to the device that was current when it was made (its events and streams live there); a host
  // thread that drives several shards has to hipSetDevice before each step, and forgetting it would launch on the wrong card
  int current = -1;
  if (hipGetDevice(&current) != hipSuccess || current != S->device)
    return fail(kErrBadArgument, "spmv_acc_shard_step: the shard was created on device " + std::to_string(S->device) +
                                     ", the calling thread's current device is " + std::to_string(current));
  hipStream_t st = get_stream();
  double *own = dy_full + static_cast<size_t>(S->rank) * S->m_pad;
  clear_error();
  // A rank whose local SpMV fails STILL takes part in every exchange of the step (its peers are already in them: leaving early would
  // leave them blocked in a collective for ever); the first local error is what the step returns, after the exchanges are enqueued.
  int first_err = kOk;
  std::string first_what;
  auto note = [&](int code, const std::string &what) {
    if (first_err == kOk) {
      first_err = code;
      first_what = what;
    }
  };
  auto note_engine = [&] {
    if (last_error() != kOk) note(last_error(), last_error_string());
  };
  if (S->chunks.size() == 1) {
    const auto &c = S->chunks[0];
    if (S->m_local > 0) run_spmv(S->strategy, 0, alpha, beta, S->m_local, S->n, c.nnz_end, nullptr, c.rp, S->ci, S->v, dx, own, dy_in_local);
    note_engine();
    // in place (send buffer = this rank's slice of the receive buffer), same stream: behind the kernels that wrote the slice
    const int rc = R.all_gather(own, dy_full, static_cast<size_t>(S->m_pad), kNcclFloat64, S->comm, st);
    if (rc != 0) note(kErrHip, "spmv_acc_shard_step: ncclAllGather failed (ncclResult " + std::to_string(rc) + ")");
    return first_err == kOk ? kOk : fail(first_err, first_what);
  }
  // the exchange writes the peers' slices of dy_full, the chunk kernels read x and the old slice: whatever was enqueued before this
  // step (the producer of x, readers of the old vector) comes first on all three streams
  bool ev_ok = hipEventRecord(S->step_begin, st) == hipSuccess && hipStreamWaitEvent(S->comm_stream, S->step_begin, 0) == hipSuccess &&
               hipStreamWaitEvent(S->chunk_stream[0], S->step_begin, 0) == hipSuccess &&
               hipStreamWaitEvent(S->chunk_stream[1], S->step_begin, 0) == hipSuccess;
  for (size_t k = 0; k < S->chunks.size(); ++k) {
    const auto &c = S->chunks[k];
    hipStream_t cs = S->chunk_stream[k & 1];
    if (c.b > c.a && first_err == kOk) {
      set_stream(cs);
      run_spmv(S->strategy, 0, alpha, beta, c.b - c.a, S->n, c.nnz_end, nullptr, c.rp, S->ci, S->v, dx, own + c.a,
               dy_in_local ? dy_in_local + c.a : nullptr);
      set_stream(st);
      note_engine();
    }
    ev_ok = ev_ok && hipEventRecord(S->chunk_done[k], cs) == hipSuccess && hipStreamWaitEvent(S->comm_stream, S->chunk_done[k], 0) == hipSuccess;
    if (S->world > 1) {
      const size_t count = static_cast<size_t>(c.c1 - c.c0);
      int rc = R.group_start();
      for (int d = 1; d < S->world && rc == 0; ++d) { // position d: send to rank + d, receive from rank - d
        const int dst = (S->rank + d) % S->world, src = (S->rank - d + S->world) % S->world;
        rc = R.send(own + c.c0, count, kNcclFloat64, dst, S->comm, S->comm_stream);
        if (rc == 0) rc = R.recv(dy_full + static_cast<size_t>(src) * S->m_pad + c.c0, count, kNcclFloat64, src, S->comm, S->comm_stream);
      }
      const int rc_end = R.group_end();
      if (rc != 0 || rc_end != 0) note(kErrHip, "spmv_acc_shard_step: grouped ncclSend / ncclRecv failed (ncclResult " + std::to_string(rc ? rc : rc_end) + ")");
    }
  }
  ev_ok = ev_ok && hipEventRecord(S->arrived, S->comm_stream) == hipSuccess && hipStreamWaitEvent(st, S->arrived, 0) == hipSuccess;
  if (!ev_ok) note(kErrHip, "spmv_acc_shard_step: event failure");
  return first_err == kOk ? kOk : fail(first_err, first_what);
}

} // extern "C"
