// mock_rccl.cpp -- TEST INFRASTRUCTURE: a single-process stand-in for the eight RCCL entries libspmv_acc.so binds at run time (shard.cpp::rccl()),
// so that the N > 1 logic of the C / C++ sharded step -- chunk bounds, the in-place gather's offsets, the pairing of the point-to-point fan-out, the
// failure paths, spmv-cli's agreement between phases -- can run with N host threads on ONE GPU (RCCL itself refuses two ranks on one device, and no
// box with more than one GPU was ever available).  It is NOT RCCL and proves nothing about RCCL: every rank is a host thread of this process, the
// "network" is hipMemcpyAsync between buffers of the same device, and a call blocks its host thread until the peers of the operation have arrived
// (a peer that never comes makes the call FAIL after a timeout instead of hanging -- which is what the failure-path tests look for).
// Loaded through SPMV_ACC_RCCL_LIB by tests/test_gpu_round3.py::test_cli_two_ranks_on_one_gpu_over_a_mock_rccl; never shipped, never linked.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

constexpr int kOk = 0, kFail = 1; // ncclSuccess / ncclUnhandledCudaError
constexpr size_t kF64 = 8;
double timeout_s() {
  const char *e = std::getenv("MOCK_RCCL_TIMEOUT_S");
  return e ? std::atof(e) : 20.0;
}

struct Posted {            // what a sender (or an allgather participant) publishes
  const void *ptr = nullptr;
  size_t bytes = 0;
  hipEvent_t ready = nullptr; // recorded on the poster's stream behind the producer of ptr
  int taken = 0;              // receivers that have enqueued their copy
  std::vector<hipEvent_t> copied; // recorded by each receiver behind its copy: the poster's stream waits for them (its buffer is free again)
};

struct World {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::tuple<int, int, long>, Posted> mail; // (src, dst or -1 for allgather, sequence) -> posting
  std::map<std::pair<int, int>, long> seq_send, seq_recv; // per (src, dst)
  std::vector<long> seq_gather;                           // per rank
  int alive = 0;
};
struct Comm {
  World *w = nullptr;
  int rank = 0;
  // an open group (ncclGroupStart .. ncclGroupEnd) of this rank's thread
  struct Op { bool send; void *ptr; size_t bytes; int peer; hipStream_t st; };
  std::vector<Op> group;
  bool grouping = false;
};
thread_local Comm *t_group_comm = nullptr; // the communicator of the thread's open group

hipEvent_t new_event() {
  hipEvent_t e = nullptr;
  (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  return e;
}

// receiver side of one point-to-point message: wait for the matching posting (the sender's ncclGroupEnd publishes it), copy behind the sender's
// "ready" event on the receiver's stream, hand the sender a "copied" event its stream then waits for (its buffer is free again behind that point)
int do_recv(Comm *c, void *ptr, size_t bytes, int src, hipStream_t st) {
  World *w = c->w;
  std::unique_lock<std::mutex> lk(w->mu);
  const long s = w->seq_recv[{src, c->rank}]++;
  const auto key = std::make_tuple(src, c->rank, s);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
  if (!w->cv.wait_until(lk, deadline, [&] { return w->mail.count(key) && w->mail[key].ready; })) return kFail; // the sender never came
  Posted &p = w->mail[key];
  if (p.bytes != bytes) return kFail; // mismatched counts: a pairing bug in the caller
  const void *from = p.ptr;
  const hipEvent_t ready = p.ready;
  lk.unlock();
  hipEvent_t done = new_event();
  const bool ok = done && hipStreamWaitEvent(st, ready, 0) == hipSuccess &&
                  hipMemcpyAsync(ptr, from, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess && hipEventRecord(done, st) == hipSuccess;
  lk.lock();
  p.copied.push_back(done);
  p.taken = 1;
  w->cv.notify_all();
  return ok ? kOk : kFail;
}

} // namespace

extern "C" {

int ncclCommInitAll(void **comms, int ndev, const int *) {
  if (!comms || ndev <= 0) return kFail;
  World *w = new World();
  w->n = ndev;
  w->alive = ndev;
  w->seq_gather.assign(ndev, 0);
  for (int r = 0; r < ndev; ++r) {
    Comm *c = new Comm();
    c->w = w, c->rank = r;
    comms[r] = c;
  }
  return kOk;
}
int ncclCommDestroy(void *comm) {
  Comm *c = static_cast<Comm *>(comm);
  if (!c) return kOk;
  World *w = c->w;
  bool last;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    last = --w->alive == 0;
  }
  delete c;
  if (last) delete w;
  return kOk;
}
int ncclCommCount(void *comm, int *n) {
  *n = static_cast<Comm *>(comm)->w->n;
  return kOk;
}
int ncclCommUserRank(void *comm, int *r) {
  *r = static_cast<Comm *>(comm)->rank;
  return kOk;
}
int ncclGroupStart(void) { // (the communicator is known at the first op of the group)
  t_group_comm = nullptr;
  return kOk;
}
int ncclSend(const void *ptr, size_t count, int /*dtype*/, int peer, void *comm, hipStream_t st) {
  Comm *c = static_cast<Comm *>(comm);
  c->grouping = true;
  t_group_comm = c;
  c->group.push_back({true, const_cast<void *>(ptr), count * kF64, peer, st});
  return kOk;
}
int ncclRecv(void *ptr, size_t count, int /*dtype*/, int peer, void *comm, hipStream_t st) {
  Comm *c = static_cast<Comm *>(comm);
  c->grouping = true;
  t_group_comm = c;
  c->group.push_back({false, ptr, count * kF64, peer, st});
  return kOk;
}
int ncclGroupEnd(void) {
  Comm *c = t_group_comm;
  t_group_comm = nullptr;
  if (!c) return kOk; // an empty group
  // Posting every send before blocking in any receive keeps a pair of ranks that both send first and then receive from deadlocking (as a real
  // group does); the blocking part of a send -- waiting for its receiver -- comes last.
  int rc = kOk;
  World *w = c->w;
  struct Pending { std::tuple<int, int, long> key; hipStream_t st; };
  std::vector<Pending> sends;
  for (auto &op : c->group) {
    if (!op.send) continue;
    hipEvent_t ready = new_event();
    if (!ready || hipEventRecord(ready, op.st) != hipSuccess) { rc = kFail; continue; }
    std::lock_guard<std::mutex> lk(w->mu);
    const long s = w->seq_send[{c->rank, op.peer}]++;
    Posted &p = w->mail[{c->rank, op.peer, s}];
    p.ptr = op.ptr, p.bytes = op.bytes, p.ready = ready;
    sends.push_back({std::make_tuple(c->rank, op.peer, s), op.st});
    w->cv.notify_all();
  }
  for (auto &op : c->group)
    if (!op.send && do_recv(c, op.ptr, op.bytes, op.peer, op.st) != kOk) rc = kFail;
  for (auto &sd : sends) {
    std::unique_lock<std::mutex> lk(w->mu);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
    if (!w->cv.wait_until(lk, deadline, [&] { return w->mail[sd.key].taken == 1; })) { rc = kFail; continue; }
    const hipEvent_t done = w->mail[sd.key].copied[0];
    w->mail.erase(sd.key);
    lk.unlock();
    if (hipStreamWaitEvent(sd.st, done, 0) != hipSuccess) rc = kFail;
  }
  c->group.clear();
  c->grouping = false;
  return rc;
}
// every rank contributes `count` doubles; rank r's land at recv + r * count in every rank's buffer (in place when send is that slice already)
int ncclAllGather(const void *send, void *recv, size_t count, int /*dtype*/, void *comm, hipStream_t st) {
  Comm *c = static_cast<Comm *>(comm);
  World *w = c->w;
  const size_t bytes = count * kF64;
  hipEvent_t ready = new_event();
  if (!ready || hipEventRecord(ready, st) != hipSuccess) return kFail;
  std::unique_lock<std::mutex> lk(w->mu);
  const long s = w->seq_gather[c->rank]++;
  {
    Posted &mine = w->mail[{c->rank, -1, s}];
    mine.ptr = send, mine.bytes = bytes, mine.ready = ready;
  }
  w->cv.notify_all();
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
  int rc = kOk;
  for (int p = 0; p < w->n; ++p) {
    const auto key = std::make_tuple(p, -1, s);
    if (!w->cv.wait_until(lk, deadline, [&] { return w->mail.count(key) && w->mail[key].ready; })) return kFail; // a peer never came
    Posted &from = w->mail[key];
    if (from.bytes != bytes) rc = kFail;
    char *dst = static_cast<char *>(recv) + static_cast<size_t>(p) * bytes;
    hipEvent_t done = new_event();
    const void *src = from.ptr;
    const hipEvent_t src_ready = from.ready;
    lk.unlock();
    bool ok = done && hipStreamWaitEvent(st, src_ready, 0) == hipSuccess;
    if (ok && src != dst) ok = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    ok = ok && hipEventRecord(done, st) == hipSuccess;
    lk.lock();
    w->mail[key].copied.push_back(done);
    w->mail[key].taken++;
    w->cv.notify_all();
    if (!ok) rc = kFail;
  }
  // my buffer is free again once every rank has copied it
  const auto mine = std::make_tuple(c->rank, -1, s);
  if (!w->cv.wait_until(lk, deadline, [&] { return w->mail[mine].taken == w->n; })) return kFail;
  const std::vector<hipEvent_t> copied = w->mail[mine].copied;
  w->mail.erase(mine);
  lk.unlock();
  for (hipEvent_t e : copied)
    if (hipStreamWaitEvent(st, e, 0) != hipSuccess) rc = kFail;
  return rc;
}

} // extern "C"
