// plan.cpp -- the stale-plan guard, the plan cache and the tune cache (split out of engine.cpp in round 4; no behaviour change).
// Reference role: none -- the reference re-runs its preprocessing on every call (hip-flat/flat.cpp:39-44); plans are what replaces that.
#include "engine_internal.hpp"

namespace spmv_acc {
using namespace detail;

// ---- plans ---------------------------------------------------------------------------------------------------------
namespace detail {

// ---- stale-plan guard -------------------------------------------------------------------------------------------
// The reference recomputes its preprocessing on every call (flat.cpp:39-44), so it can never act on a matrix that has
// changed; its callers therefore never announce a change.  Plans here are keyed by pointers and shape, and a caller that
// frees a matrix and gets the same addresses back for another one of the same shape (the norm for hipMalloc after hipFree),
// or rewrites the structure in place, would meet the old matrix' break points / row blocks.  Every plan therefore records
// kGuardSamples rowptr entries (device) and owns a sticky flag in pinned host memory; the first wave of block 0 of every
// SpMV kernel compares (device_utils.hpp::check_plan_guard) and raises the flag.  The host looks at the flag -- an
// ordinary memory read, no synchronisation -- when the plan is used again and in spmv_acc_last_error(): the plan is
// dropped, SPMV_ACC_ERR_BAD_ARGUMENT is recorded (the y of the call that raised the flag is not to be trusted) and the
// matrix gets a fresh plan.  Slots come from one pool per device, recycled first-in first-out so that a kernel of a
// dropped plan that is still in flight does not meet its slot's next owner.
constexpr int kGuardSlots = 4096;
struct GuardPool {
  int *d_guard = nullptr; // kGuardSlots * kGuardSamples ints
  int *h_flags = nullptr; // kGuardSlots ints, hipHostMalloc (coherent, device-visible)
  // free slots, oldest first.  A slot released by a plan that had launched kernels carries an event recorded behind the plan's
  // last launch: the slot gets a new owner only once that event has completed, so a kernel of the dropped plan that is still
  // in flight can never raise the flag of the slot's next owner (first-in first-out alone only made that unlikely).
  std::deque<std::pair<int, hipEvent_t>> free_slots;
  bool failed = false;
};
std::mutex g_guard_mu; // not g_mu: plans die (and return their slot) both under g_mu and outside it
std::map<int, GuardPool> g_guard_pools;

int guard_acquire(int device, const int **d_guard, int **h_flag) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  GuardPool &P = g_guard_pools[device];
  if (P.failed) return -1;
  if (!P.d_guard) {
    if (hipMalloc(reinterpret_cast<void **>(&P.d_guard), sizeof(int) * kGuardSlots * kGuardSamples) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&P.h_flags), sizeof(int) * kGuardSlots,
                      hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      if (P.d_guard) (void)hipFree(P.d_guard);
      P.d_guard = nullptr;
      P.failed = true; // plans of this device run unguarded
      return -1;
    }
    std::memset(P.h_flags, 0, sizeof(int) * kGuardSlots);
    for (int i = 0; i < kGuardSlots; ++i) P.free_slots.emplace_back(i, nullptr);
  }
  int slot = -1;
  for (size_t tries = P.free_slots.size(); tries > 0 && slot < 0; --tries) {
    const std::pair<int, hipEvent_t> cand = P.free_slots.front();
    P.free_slots.pop_front();
    if (cand.second && hipEventQuery(cand.second) == hipErrorNotReady) {
      P.free_slots.push_back(cand); // its last owner's kernels are still running: not yet
      continue;
    }
    (void)hipGetLastError();
    if (cand.second) (void)hipEventDestroy(cand.second);
    slot = cand.first;
  }
  if (slot < 0) return -1; // (this plan runs unguarded)
  __atomic_store_n(&P.h_flags[slot], 0, __ATOMIC_RELAXED);
  *d_guard = P.d_guard + static_cast<size_t>(slot) * kGuardSamples;
  *h_flag = P.h_flags + slot;
  return slot;
}
// `launched`: the plan has enqueued kernels, the last of them on `last_stream` (a stream of `device`)
void guard_release(int device, int slot, bool launched, hipStream_t last_stream) {
  if (slot < 0) return;
  hipEvent_t ev = nullptr;
  if (launched) {
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (ev && hipEventRecord(ev, last_stream) != hipSuccess) { // (e.g. the caller has destroyed that stream: its work is done)
      (void)hipEventDestroy(ev);
      ev = nullptr;
    }
    (void)hipGetLastError();
    if (switched) (void)hipSetDevice(cur);
  }
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guard_pools[device].free_slots.emplace_back(slot, ev);
}


// ---- tune cache: the per-matrix timed choices, kept across processes --------------------------------------------------------
// The first call on a matrix times a handful of choices (13 ms on the headline matrix, 85 SpMVs' worth) and every process pays
// again; the reference's choice is a pure function of its inputs (strategy_picker.cpp:19-65) and costs nothing.  With
// SPMV_ACC_TUNE_CACHE=<file> (or spmv_acc_set_tune_cache) the choices are appended to a text file, one line per matrix, keyed by a
// digest of (library version, device name, m, n, nnz, the 64 rowptr samples of the stale-plan guard); a later process that meets
// the same matrix on the same device adopts them and only runs the structural passes.  Opt-in; the last line for a key wins;
// a choice the current build cannot honour (a cut-row form that is not legal on this matrix) falls back to the safe one.
std::mutex g_tune_mu;
std::string g_tune_path;
bool g_tune_path_set = false, g_tune_loaded = false;
std::map<unsigned long long, TuneRecord> g_tune_db;

void tune_load_locked() {
  if (!g_tune_path_set) {
    if (const char *e = std::getenv("SPMV_ACC_TUNE_CACHE")) g_tune_path = e;
    g_tune_path_set = true;
  }
  if (g_tune_loaded || g_tune_path.empty()) return;
  g_tune_loaded = true;
  if (FILE *f = std::fopen(g_tune_path.c_str(), "r")) {
    // line by line: a line cut short (a writer killed mid-write) or written by another version is skipped by itself
    char line[1024];
    while (std::fgets(line, sizeof(line), f)) {
      char tag[32];
      unsigned long long key = 0;
      int used = 0;
      if (std::sscanf(line, "%31s %llx%n", tag, &key, &used) != 2 || std::strcmp(tag, "spmvacc6") != 0) continue;
      TuneRecord r;
      bool ok = true;
      const char *at = line + used;
      for (int i = 0; i < kTuneFields && ok; ++i) {
        int step = 0;
        ok = std::sscanf(at, "%d%n", &r.v[i], &step) == 1;
        at += step;
      }
      if (ok && std::strchr(at, '\n')) g_tune_db[key] = r;
    }
    std::fclose(f);
  }
}
bool tune_cache_enabled() {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  tune_load_locked();
  return !g_tune_path.empty();
}
// What the adopted choices depend on, all of it in the key (round 4; until then: version string, device, shape, 64 rowptr samples, so that a
// matrix with the same row structure and other columns, a changed tunable or a rebuilt library adopted stale choices -- correct, legality is
// re-checked, but not re-timed): + 64 colindex samples (the column distribution decides hints, slab passes, adaptive's family), + the
// tunables the timings were taken under, + the build's date and time.
unsigned long long tune_key_of(int dev, int m, int n, int nnz, const int *samples, const int *col_samples) {
  unsigned long long h = 1469598103934665603ULL; // FNV-1a
  auto mix = [&h](const void *p, size_t bytes) {
    const unsigned char *c = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ c[i]) * 1099511628211ULL;
  };
  static const char kVersion[] = "spmv_acc_amd 0.6 tune v8";
  mix(kVersion, sizeof(kVersion));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
    mix(prop.name, strnlen(prop.name, sizeof(prop.name)));
    mix(prop.gcnArchName, strnlen(prop.gcnArchName, sizeof(prop.gcnArchName)));
    mix(&prop.multiProcessorCount, sizeof(int));
  }
  (void)hipGetLastError();
  mix(&m, sizeof(int));
  mix(&n, sizeof(int));
  mix(&nnz, sizeof(int));
  mix(samples, sizeof(int) * kGuardSamples);
  if (col_samples) mix(col_samples, sizeof(int) * kGuardSamples);
  static const char kBuild[] = __DATE__ " " __TIME__;
  mix(kBuild, sizeof(kBuild));
  for (TunableId id : {kT_rowblock_target, kT_hint_budget_kb, kT_cache_ends_mb, kT_zigzag, kT_xcd_chunk, kT_slab_whole_below, kT_rowlen, kT_slab_kb,
                       kT_hint_min_x_mb, kT_flat_small_nnz_k, kT_max_grid_blocks, kT_strict_strategy, kT_col16}) {
    const int v = tun(id);
    mix(&v, sizeof(int));
  }
  return h ? h : 1;
}

void tune_log(const char *fmt, ...);
TuneRecord tune_snapshot(const Plan &p) {
  TuneRecord r;
  int k = 0;
  for (int f = 0; f < kFamilyCount; ++f)
    for (int c = 0; c < 2; ++c) r.v[k++] = p.stream_policy[f][c];
  // a PROVISIONAL family (the best of the families timed so far under the call budget, dispatch.cpp::run_adaptive_timed) is not a choice yet:
  // neither the flag nor the timings travel, so an adopting process would take it as final and never finish the comparison.  It is kept as
  // "not timed" until the comparison is complete (a short process of one to three calls leaves nothing half-measured behind).
  r.v[k++] = p.adaptive_provisional[0] ? -1 : p.adaptive_family[0];
  r.v[k++] = p.adaptive_provisional[1] ? -1 : p.adaptive_family[1];
  r.v[k++] = p.flat_npt_choice;
  r.v[k++] = p.flat_early_choice ? 1 : 0;
  r.v[k++] = p.flat_geometry_tuned ? 1 : 0;
  r.v[k++] = p.flat_mode_choice[0];
  r.v[k++] = p.flat_mode_choice[1];
  r.v[k++] = p.plus_tuned_min;
  r.v[k++] = p.hint_state == 0 ? 0 : -1; // only "the census found nothing to protect" is worth keeping: the bits themselves are rebuilt
  for (int f = 0; f < 3; ++f) r.v[k++] = p.hint_use[f];
  r.v[k++] = p.flat_rowblock_choice;
  r.v[k++] = p.seg_choice;
  r.v[k++] = p.c16_use[kFamRowblock];
  r.v[k++] = p.c16_use[kFamFlat];
  r.v[k++] = p.rb_target;
  r.v[k++] = tun(kT_col_slabs) == -2 ? -1 : p.slab_copy_choice; // (the tests' hook keeps the copy whatever the timing says: not a measured choice)
  return r;
}
} // namespace detail
void set_tune_cache(const char *path) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_tune_path = path ? path : "";
  g_tune_path_set = true;
  g_tune_loaded = false;
  g_tune_db.clear();
}
namespace detail {
// a fresh plan adopts what an earlier process (or an earlier plan of this process) measured on the same matrix
void tune_adopt(Plan &p) {
  TuneRecord r;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tune_db.find(p.tune_key);
    if (it == g_tune_db.end()) return;
    r = it->second;
  }
  auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
  int k = 0;
  for (int f = 0; f < kFamilyCount; ++f)
    for (int c = 0; c < 2; ++c, ++k) p.stream_policy[f][c] = in(r.v[k], 0, 3) ? r.v[k] : -1;
  p.adaptive_family[0] = in(r.v[k], 0, 2) ? r.v[k] : -1;
  ++k;
  p.adaptive_family[1] = in(r.v[k], 0, 2) ? r.v[k] : -1;
  ++k;
  p.flat_npt_choice = (r.v[k] == 4 || r.v[k] == 8) ? r.v[k] : 0;
  ++k;
  p.flat_early_choice = r.v[k++] == 1;
  p.flat_geometry_tuned = r.v[k++] == 1 && p.flat_npt_choice > 0;
  p.flat_mode_choice[0] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.flat_mode_choice[1] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.plus_tuned_min = (r.v[k] == 1024 || r.v[k] == 1536 || r.v[k] == 1920) ? r.v[k] : 0;
  ++k;
  if (r.v[k++] == 0) p.hint_state = 0;
  for (int f = 0; f < 3; ++f, ++k) p.hint_use[f] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  p.flat_rowblock_choice = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.seg_choice = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.c16_use[kFamRowblock] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.c16_use[kFamFlat] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.rb_target = (r.v[k] == kRowblockTargetRule || r.v[k] == kRowblockTargetAlt) ? r.v[k] : 0;
  ++k;
  p.slab_copy_choice = in(r.v[k], 0, 1) ? r.v[k] : -1;
  tune_log("m %d nnz %d: choices adopted from the tune cache (key %016llx)", p.A.m, p.A.nnz, p.tune_key);
}
// after a call that did plan work: keep what the plan now knows
void tune_store(const Plan &p) {
  if (!p.tune_key) return;
  const TuneRecord r = tune_snapshot(p);
  std::lock_guard<std::mutex> lk(g_tune_mu);
  if (g_tune_path.empty()) return;
  auto it = g_tune_db.find(p.tune_key);
  if (it != g_tune_db.end() && it->second == r) return;
  g_tune_db[p.tune_key] = r;
  if (FILE *f = std::fopen(g_tune_path.c_str(), "a")) { // one line, one write: concurrent processes interleave whole lines
    std::string line = "spmvacc6 ";
    char buf[32];
    std::snprintf(buf, sizeof(buf), "%016llx", p.tune_key);
    line += buf;
    for (int i = 0; i < kTuneFields; ++i) line += " " + std::to_string(r.v[i]);
    line += "\n";
    std::fwrite(line.data(), 1, line.size(), f);
    std::fclose(f);
  }
}

// Derived matrices (the slabs of the opt-in column-slab blocking) have plans of their own, keyed by pointers into their parent's
// arrays.  When the parent dies those plans must go too; a plan can die under g_mu, so the rowptrs are queued here and the entries
// are erased the next time the cache is touched (before any lookup: a re-used address never meets a dead plan).
std::mutex g_deferred_mu;
std::vector<const void *> g_deferred_rp;
void Plan::free_slabs() {
  if (!slab_crp.empty()) {
    std::lock_guard<std::mutex> lk(g_deferred_mu);
    for (int *crp : slab_crp)
      if (crp) g_deferred_rp.push_back(crp); // (the slabs' plans are keyed by their compact row pointers)
  }
  for (int *q : slab_rowid)
    if (q) (void)hipFree(q);
  for (int *q : slab_crp)
    if (q) (void)hipFree(q);
  slab_rowid.clear();
  slab_crp.clear();
  slab_rows.clear();
  if (d_slab_ys) (void)hipFree(d_slab_ys);
  d_slab_ys = nullptr;
  if (d_slab_rp) (void)hipFree(d_slab_rp);
  if (d_slab_ci) (void)hipFree(d_slab_ci);
  if (d_slab_v) (void)hipFree(d_slab_v);
  if (d_slab_off) (void)hipFree(d_slab_off);
  d_slab_off = nullptr;
  if (d_value_samples) (void)hipFree(d_value_samples);
  d_value_samples = nullptr;
  if (h_values_changed) (void)hipHostFree(h_values_changed);
  h_values_changed = nullptr;
  value_samples = 0;
  d_slab_rp = d_slab_ci = nullptr;
  d_slab_v = nullptr;
  slab_count = 0;
  slab_off.clear();
}

std::map<PlanKey, std::shared_ptr<Plan>> g_plans; // a running call keeps its plan alive through its own reference
void drain_deferred_locked() { // g_mu held
  std::vector<const void *> dead;
  {
    std::lock_guard<std::mutex> lk(g_deferred_mu);
    dead.swap(g_deferred_rp);
  }
  while (!dead.empty()) { // (erasing a plan may queue more)
    for (auto it = g_plans.begin(); it != g_plans.end();) {
      if (std::find(dead.begin(), dead.end(), std::get<1>(it->first)) != dead.end()) it = g_plans.erase(it);
      else ++it;
    }
    dead.clear();
    std::lock_guard<std::mutex> lk(g_deferred_mu);
    dead.swap(g_deferred_rp);
  }
}
// a plan and its rule twin (PlanKey's last member) live and die together: a stale matrix, a changed nnz, the LRU bound and the release entries take both
static void erase_with_sibling_locked(const PlanKey &k) { // g_mu held
  PlanKey a = k, b = k;
  std::get<6>(a) = 0;
  std::get<6>(b) = 1;
  g_plans.erase(a);
  g_plans.erase(b);
}
thread_local std::weak_ptr<Plan> t_last_plan;     // the plan this thread's latest run_spmv used (last_error asks it, and only it)

// Is the calling thread inside a stream capture (set by run_spmv)?  Plan work -- allocations, synchronisation, timings -- would
// invalidate the capture: required work is refused with an error that says so, optional work (timed choices) is skipped and the
// call runs with what the plan already holds.
thread_local bool t_capturing = false;
bool plan_work_allowed(const char *what) {
  if (!t_capturing) return true;
  set_error(kErrBadArgument, std::string("this call needs plan work (") + what +
                                 ") that allocates or synchronises and cannot run inside a stream capture: run the same call "
                                 "(or spmv_acc_prepare with this strategy) once outside the capture first; nothing was enqueued");
  return false;
}
constexpr size_t kMaxPlans = 1024; // beyond this the least recently used plan is dropped
unsigned long long g_use_clock = 0;

// Is p readable by the host?  The reference's sparse_spmv hands the SAME device pointer in as "host"
// rowptr (api/spmv_imp.cpp:14-17), which only works with host-visible device memory.
bool host_readable(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  std::memset(&attr, 0, sizeof(attr));
  const hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // plain malloc'ed memory is not known to HIP: that is a host pointer
    return true;
  }
  return attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeUnregistered ||
         attr.type == hipMemoryTypeManaged;
}

// h if the host may dereference it, else null (checked only on the once-per-matrix paths)
const int *host_view(const int *h) { return host_readable(h) ? h : nullptr; }

bool fetch_samples(Plan &p, const int *h_rowptr) {
  if (p.have_samples) return true;
  if (!plan_work_allowed("reading the rowptr samples")) return false;
  ++t_plan_work;
  h_rowptr = host_view(h_rowptr);
  const int m = p.A.m;
  const int idx[4] = {m / 4, m / 2, static_cast<int>(3LL * m / 4), m};
  int out[4];
  if (h_rowptr) {
    for (int i = 0; i < 4; ++i) out[i] = h_rowptr[idx[i]];
  } else {
    for (int i = 0; i < 4; ++i) {
      if (!hip_ok(hipMemcpy(&out[i], p.A.rp + idx[i], sizeof(int), hipMemcpyDeviceToHost), "read rowptr sample"))
        return false;
    }
  }
  p.samples.q1 = out[0];
  p.samples.half = out[1];
  p.samples.q3 = out[2];
  p.samples.last = out[3];
  p.have_samples = true;
  return true;
}

std::shared_ptr<Plan> get_plan(int m, int n, int nnz, const int *h_rowptr, const int *rp, const int *ci, const double *v) {
  int dev = 0;
  if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return nullptr;
  const PlanKey key(dev, rp, ci, v, m, n, t_rule_twin ? 1 : 0);
  std::lock_guard<std::mutex> lk(g_mu);
  drain_deferred_locked();
  auto it = g_plans.find(key);
  if (it != g_plans.end() && it->second->is_stale()) {
    if (!plan_work_allowed("rebuilding a stale plan")) return nullptr;
    set_error(kErrBadArgument,
              "the matrix behind a cached plan changed (same pointers and shape, different rowptr) without "
              "spmv_acc_release_plans: the results of the EARLIER calls made on it since the change are invalid; the plan has "
              "been rebuilt, so the call that reports this ran on the matrix as it is now and its y is valid");
    erase_with_sibling_locked(key);
    it = g_plans.end();
  }
  if (it != g_plans.end()) {
    if (nnz < 0 || nnz == it->second->A.nnz) {
      it->second->last_use = ++g_use_clock;
      return it->second;
    }
    // same buffers, different nnz: the caller rebuilt the matrix in place
    if (!plan_work_allowed("rebuilding the plan of a matrix whose nnz changed")) return nullptr;
    erase_with_sibling_locked(key);
  }
  if (!plan_work_allowed("building the plan of a matrix seen for the first time")) return nullptr;
  if (!t_rule_twin && g_plans.size() >= kMaxPlans) { // (the bound counts matrices: a twin neither counts nor evicts)
    size_t primaries = 0;
    auto oldest = g_plans.end();
    for (auto jt = g_plans.begin(); jt != g_plans.end(); ++jt) {
      if (std::get<6>(jt->first) != 0) continue;
      ++primaries;
      if (oldest == g_plans.end() || jt->second->last_use < oldest->second->last_use) oldest = jt;
    }
    if (primaries >= kMaxPlans && oldest != g_plans.end()) {
      const PlanKey gone = oldest->first;
      erase_with_sibling_locked(gone);
    }
  }
  if (nnz < 0) {
    if ((h_rowptr = host_view(h_rowptr)) != nullptr) {
      nnz = h_rowptr[m];
    } else if (!hip_ok(hipMemcpyAsync(&nnz, rp + m, sizeof(int), hipMemcpyDeviceToHost, t_stream), "read rowptr[m]") ||
               !hip_ok(hipStreamSynchronize(t_stream), "read rowptr[m]")) {
      // (on the calling thread's library stream: a rowptr produced on that stream just before the call is read after its producer -- a blocking
      // null-stream copy is not ordered behind a non-blocking stream)
      return nullptr;
    }
  }
  if (nnz < 0 || nnz > INT_MAX - (1 << 16)) {
    set_error(kErrTooLarge, "nnz does not leave room for tile arithmetic in int32; shard the matrix");
    return nullptr;
  }
  std::shared_ptr<Plan> p = std::make_shared<Plan>();
  p->device = dev;
  p->key = key;
  p->A.m = m;
  p->A.n = n;
  p->A.nnz = nnz;
  {
    // the view's first non-zero, rowptr[0] (round 5): 0 for a whole matrix; an un-rebased row sub-range (shard.cpp's chunk views) starts later, and
    // `nnz` is then the END offset -- every shape heuristic reads A.count() = nnz - nnz0.  One 4-byte read per plan.
    const int *hv = host_view(h_rowptr);
    int first = 0;
    if (hv) first = hv[0];
    else if (!hip_ok(hipMemcpyAsync(&first, rp, sizeof(int), hipMemcpyDeviceToHost, t_stream), "read rowptr[0]") ||
             !hip_ok(hipStreamSynchronize(t_stream), "read rowptr[0]"))
      return nullptr; // (ordered on the library stream like the read of rowptr[m] above)
    p->A.nnz0 = (first > 0 && first <= nnz) ? first : 0;
  }
  p->A.rp = rp;
  p->A.ci = ci;
  p->A.v = v;
  p->A.aligned16 = (reinterpret_cast<uintptr_t>(ci) % 16 == 0) && (reinterpret_cast<uintptr_t>(v) % 16 == 0) &&
                   nnz >= 8;
  p->last_use = ++g_use_clock;
  p->guard_slot = guard_acquire(dev, &p->A.guard, &p->A.stale);
  if (p->guard_slot >= 0) {
    launch_guard_fill(t_stream, rp, m, const_cast<int *>(p->A.guard));
    if (!hip_ok(hipStreamSynchronize(t_stream), "record the plan guard")) return nullptr; // (a later call may use another stream)
    if (tune_cache_enabled() && !tun(kT_deterministic)) {
      int samples[kGuardSamples], col_samples[kGuardSamples] = {};
      bool ok = hipMemcpy(samples, p->A.guard, sizeof(samples), hipMemcpyDeviceToHost) == hipSuccess;
      // colindex[rowptr[0] + k * (extent - 1) / 63]: 64 four-byte reads, once per plan, only with the tune cache on
      const long long first = samples[0], extent = static_cast<long long>(samples[kGuardSamples - 1]) - first;
      for (int k = 0; ok && ci && extent > 0 && k < kGuardSamples; ++k)
        ok = hipMemcpy(&col_samples[k], ci + first + k * (extent - 1) / (kGuardSamples - 1), sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
      if (ok) {
        p->tune_key = tune_key_of(dev, m, n, nnz, samples, col_samples);
        tune_adopt(*p);
      }
      (void)hipGetLastError();
    }
  }
  g_plans[key] = p;
  return p;
}

const char *const kStaleText =
    "the matrix behind a cached plan changed (same pointers and shape, different rowptr) without spmv_acc_release_plans: the "
    "results of the calls made on it since the change -- including the calling thread's most recent SpMV on these pointers -- "
    "are invalid; the plan has been dropped and the next call on the matrix rebuilds it";

// O(1): the plan the calling thread used last, nothing else
bool report_stale_last_plan() {
  const std::shared_ptr<Plan> p = t_last_plan.lock();
  if (!p || !p->is_stale()) return false;
  set_error(kErrBadArgument, kStaleText);
  t_last_plan.reset();
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find(p->key);
  if (it != g_plans.end() && it->second == p) erase_with_sibling_locked(p->key);
  return true;
}
} // namespace detail

// Every cached plan (any thread's): drops the stale ones, returns how many there were.  For callers that edit matrices in place
// from several threads and want one check after a device-wide synchronisation; not on any hot path.
int check_plans() {
  std::lock_guard<std::mutex> lk(g_mu);
  int dropped = 0; // (matrices: a plan and its rule twin count once)
  std::vector<PlanKey> gone;
  for (auto &kv : g_plans) {
    if (!kv.second->is_stale()) continue;
    PlanKey k = kv.first;
    std::get<6>(k) = 0;
    if (std::find(gone.begin(), gone.end(), k) == gone.end()) gone.push_back(k);
  }
  for (const PlanKey &k : gone) {
    erase_with_sibling_locked(k);
    ++dropped;
  }
  if (dropped) set_error(kErrBadArgument, kStaleText);
  return dropped;
}


// m_only >= 0: only the plans of that row count (a shard's chunk 0 is a view `rowptr + 0` of fewer rows: dropping it must not take the
// caller's whole-shard plan on the same pointer with it)
void release_plans(const int *d_rowptr, int m_only) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto it = g_plans.begin(); it != g_plans.end();) {
    if (!d_rowptr || (std::get<1>(it->first) == d_rowptr && (m_only < 0 || std::get<4>(it->first) == m_only))) {
      it = g_plans.erase(it);
    } else {
      ++it;
    }
  }
  drain_deferred_locked(); // the plans of matrices derived from the ones just dropped
}

bool query_plan(const int *d_rowptr, int m, PlanInfo *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &kv : g_plans) {
    if (std::get<1>(kv.first) == d_rowptr && std::get<4>(kv.first) == m && std::get<6>(kv.first) == 0) { // (a rule twin is not what callers ask about)
      const Plan &p = *kv.second;
      out->nnz = p.A.nnz;
      out->adaptive_branch = p.have_samples ? adaptive_branch(m, p.samples) : 0;
      int rb_vec = 1, rb_rows = kThreads;
      pick_rowblock_shape(m, p.A.count(), rowblock_target_for(p), &rb_vec, &rb_rows);
      out->vec = rb_vec; // lanes per row of the row-block family for this matrix
      out->flat_tiles = p.flat_tiles;
      out->plus_blocks = p.plus_blocks;
      out->aligned16 = p.A.aligned16 ? 1 : 0;
      // the policy of the family that runs this matrix: adaptive's choice if it was timed, else the first family tuned
      const int afam = p.adaptive_family[1] >= 0 ? p.adaptive_family[1] : p.adaptive_family[0]; // (beta != 0 first: the reference's protocol)
      int fam = afam;
      for (int f = 0; fam < 0 && f < kFamilyCount; ++f)
        if (p.stream_policy[f][1] >= 0 || p.stream_policy[f][0] >= 0) fam = f;
      out->stream_policy = fam < 0 ? -1 : (p.stream_policy[fam][1] >= 0 ? p.stream_policy[fam][1] : p.stream_policy[fam][0]);
      out->flat_fixup = p.flat_tiles > 0 ? (p.flat.needs_fixup ? 1 : 0) : -1;
      out->adaptive_family = afam;
      out->adaptive_family_beta0 = p.adaptive_family[0];
      out->settled = p.calls > 0 && !p.tuning_open ? 1 : 0;
      out->last_kernel = p.last_kernel;
      out->col16 = p.last_c16;
      out->slab_passes = p.seg_state == 1 && (tun(kT_slab_segments) >= 1 || (tun(kT_slab_segments) < 0 && p.seg_choice == 1)) ? p.seg_slabs - (p.seg_rest_below > 0 ? 1 : 0) : 0; // (column slabs: the whole-row pass of the two-class form is not counted)
      return true;
    }
  }
  return false;
}

int cached_plan_count() { // (the rule twins of unsettled plans are not counted: they come and go with their plans)
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (auto &kv : g_plans) n += std::get<6>(kv.first) == 0 ? 1 : 0;
  return n;
}

namespace detail {
bool plan_settled_for(const int *rp, const int *ci, const double *v, int m, int n, int strategy, int cls) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return true; // (the call itself will report it)
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find(PlanKey(dev, rp, ci, v, m, n, 0));
  if (it == g_plans.end()) return false;
  const Plan &p = *it->second;
  return strategy >= 0 && strategy < 32 && ((p.settled_for[cls & 1] >> strategy) & 1u);
}

bool rule_twin_exists(const int *rp, const int *ci, const double *v, int m, int n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  return g_plans.find(PlanKey(dev, rp, ci, v, m, n, 1)) != g_plans.end();
}

void drop_rule_twin(const int *rp, const int *ci, const double *v, int m, int n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find(PlanKey(dev, rp, ci, v, m, n, 1));
  // (a twin that a stream capture recorded stays: the graph's kernels read its tables until the caller releases the matrix' plans)
  if (it != g_plans.end() && !it->second->captured) g_plans.erase(it);
}
} // namespace detail

} // namespace spmv_acc
