// solver.hip -- C-ABI (include/mi_osqp.h) over host_core + kernels.
//
// setup(): host analysis + Ruiz scaling + KKT + LDL' per QP (rows E1-E5), upload
// in tile-interleaved, schedule-ordered layout.  solve(): the whole ADMM loop
// runs inside admm_kernel; the host only reacts to "rho changed" requests
// (row E13: numeric refactor on the host, first version) and reads statuses.
// There is NO CPU solve path: without a gfx950 device setup returns
// MI_OSQP_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/mi_osqp.h"
#include "device_types.h"
#include "host_core.hpp"

using namespace miosqp;

static thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    if (getenv("MI_OSQP_DEBUG_HIP")) {                                                   \
      hipError_t _s = hipGetLastError();                                                 \
      if (_s != hipSuccess) fprintf(stderr, "[mi_osqp] stale HIP error before %s (%s:%d): %s\n", #expr, __FILE__, __LINE__, hipGetErrorString(_s)); \
    }                                                                                    \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
      return MI_OSQP_ERR_DEVICE;                                                         \
    }                                                                                    \
  } while (0)

// Every entry point that touches the device runs with the handle's device current and restores the caller's afterwards:
// a handle may be used from any host thread, whatever device that thread had selected.
struct DevGuard {
  int prev = -1; bool switched = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
  DevGuard(const DevGuard &) = delete;
  DevGuard &operator=(const DevGuard &) = delete;
};

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// MI_OSQP_DEBUG_TIMING: wall time of the C-ABI calls that sit in the GOMP drivers' loops, summed per entry point and
// printed when the process exits
struct CallTimer {
  struct Slot { const char *name; double s; long calls; };
  static Slot *slots() { static Slot t[16] = {}; return t; }
  static bool on() { static const bool v = getenv("MI_OSQP_DEBUG_TIMING") != nullptr; return v; }
  static void report() {
    for (Slot *t = slots(); t->name; t++) fprintf(stderr, "[mi_osqp] %-28s %6ld calls %9.3f ms\n", t->name, t->calls, 1e3 * t->s);
  }
  const char *name; double t0;
  explicit CallTimer(const char *n) : name(n), t0(on() ? now_s() : 0.0) {}
  ~CallTimer() {
    if (!on()) return;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    Slot *t = slots();
    static bool registered = false;
    if (!registered) { atexit(report); registered = true; }
    while (t->name && t->name != name && t < slots() + 14) t++;
    t->name = name; t->s += now_s() - t0; t->calls++;
  }
};

// cores this process may use: min(hardware threads, cgroup CPU quota); the GPU
// boxes expose 256 logical CPUs but grant a quota of ~16
static int host_threads() {
  static int cached = 0;
  if (cached) return cached;
  const char *e = getenv("MI_OSQP_HOST_THREADS");
  int t = (int)std::thread::hardware_concurrency();
  if (e) t = atoi(e);
  else {
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64]; long long period = 0;
      if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
        t = std::min<long long>(t, std::max<long long>(1, (atoll(q) + period / 2) / period));
      fclose(f);
    }
    long long q1 = -1, p1 = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &q1) != 1) q1 = -1; fclose(f); }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &p1) != 1) p1 = 0; fclose(f); }
    if (q1 > 0 && p1 > 0) t = std::min<long long>(t, std::max<long long>(1, (q1 + p1 / 2) / p1));
  }
  cached = std::max(1, std::min(t, 128));
  return cached;
}

template <class F>
static void parallel_for(int count, F &&fn) {
  int nt = std::min(host_threads(), count);
  if (nt <= 1) { for (int i = 0; i < count; i++) fn(i, 0); return; }
  std::atomic<int> next{0};
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&, t]() { for (int i; (i = next.fetch_add(1)) < count;) fn(i, t); });
  for (auto &x : th) x.join();
}

// Device memory of freed handles is kept for the next one (the GOMP drivers build a new solver per horizon segment:
// hipMalloc / hipFree of ~60 buffers cost ~11 ms per segment otherwise).  A block is reused for requests between half
// its size and its size; at most 8 GiB are kept, mi_osqp_release_device_cache() returns them to the runtime.
namespace devpool {
static std::mutex mu;
static std::multimap<std::pair<int, size_t>, void *> blocks;      // (device, capacity in bytes) -> pointer
static size_t kept = 0;
constexpr size_t kMaxKept = (size_t)8 << 30;
static void *take(int dev, size_t bytes, size_t &cap) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = blocks.lower_bound({dev, bytes});
  if (it == blocks.end() || it->first.first != dev || it->first.second > 2 * bytes + 4096) return nullptr;
  void *p = it->second; cap = it->first.second; kept -= cap;
  blocks.erase(it);
  return p;
}
static bool give(int dev, void *p, size_t cap) {
  std::lock_guard<std::mutex> lk(mu);
  if (kept + cap > kMaxKept) return false;
  blocks.emplace(std::make_pair(dev, cap), p); kept += cap;
  return true;
}
static void release_all() {
  std::lock_guard<std::mutex> lk(mu);
  for (auto &b : blocks) (void)hipFree(b.second);
  blocks.clear(); kept = 0;
}
}  // namespace devpool

// The same for pinned host memory and for streams + events: hipHostMalloc costs ~0.5 ms per call and a handle needs four
// of them, a stream and five events - 2-3 ms of a 4 ms setup for the small QPs of a sequential GOMP run.
namespace hostpool {
static std::mutex mu;
static std::multimap<size_t, void *> blocks;          // capacity in bytes -> pinned pointer
static size_t kept = 0;
constexpr size_t kMaxKept = (size_t)1 << 30;
static void *take(size_t bytes, size_t &cap) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = blocks.lower_bound(bytes);
  if (it == blocks.end() || it->first > 4 * bytes + 65536) return nullptr;
  void *p = it->second; cap = it->first; kept -= cap;
  blocks.erase(it);
  return p;
}
static void give(void *p, size_t cap) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (kept + cap <= kMaxKept) { blocks.emplace(cap, p); kept += cap; return; }
  }
  (void)hipHostFree(p);
}
static void release_all() {
  std::lock_guard<std::mutex> lk(mu);
  for (auto &b : blocks) (void)hipHostFree(b.second);
  blocks.clear(); kept = 0;
}
// pinned allocation of at least `bytes` (rounded up to 4 KiB); *cap = what to hand back to give()
static hipError_t alloc(void **p, size_t bytes, size_t *cap) {
  bytes = (bytes + 4095) & ~(size_t)4095;
  if (void *q = take(bytes, *cap)) { *p = q; return hipSuccess; }
  *cap = bytes;
  return hipHostMalloc(p, bytes, hipHostMallocPortable);      // (a kept block may serve a handle on another device)
}
}  // namespace hostpool
namespace streampool {
struct Bundle { int dev; hipStream_t stream; hipEvent_t ev[5]; };
static std::mutex mu;
static std::vector<Bundle> idle;
static bool take(int dev, Bundle &out) {
  std::lock_guard<std::mutex> lk(mu);
  for (size_t i = 0; i < idle.size(); i++)
    if (idle[i].dev == dev) { out = idle[i]; idle.erase(idle.begin() + i); return true; }
  return false;
}
static void give(const Bundle &b) {           // (the stream has been synchronised)
  {
    std::lock_guard<std::mutex> lk(mu);
    if (idle.size() < 64) { idle.push_back(b); return; }
  }
  for (hipEvent_t e : b.ev) if (e) (void)hipEventDestroy(e);
  if (b.stream) (void)hipStreamDestroy(b.stream);
}
static void release_all() {
  std::lock_guard<std::mutex> lk(mu);
  for (Bundle &b : idle) { for (hipEvent_t e : b.ev) if (e) (void)hipEventDestroy(e); if (b.stream) (void)hipStreamDestroy(b.stream); }
  idle.clear();
}
}  // namespace streampool

template <class T>
struct DevBuf {
  T *p = nullptr; size_t n = 0;
  size_t cap = 0;            // bytes of the underlying block
  int dev = 0;               // the device it lives on
  int alloc(size_t count) {
    free();
    n = count;
    if (!count) return 0;
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (void *q = devpool::take(dev, bytes, cap)) { p = (T *)q; return 0; }
    hipError_t e = hipMalloc((void **)&p, bytes);
    if (e != hipSuccess) {          // the kept blocks may be what is in the way
      devpool::release_all();
      e = hipMalloc((void **)&p, bytes);
    }
    if (e != hipSuccess) { p = nullptr; g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e); return MI_OSQP_ERR_ALLOC; }
    cap = bytes;
    return 0;
  }
  int zero(hipStream_t s) { if (n && hipMemsetAsync(p, 0, n * sizeof(T), s) != hipSuccess) return MI_OSQP_ERR_DEVICE; return 0; }
  int upload(const std::vector<T> &v) {
    int rc = alloc(v.size());
    if (rc) return rc;
    if (v.size() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return MI_OSQP_ERR_DEVICE;
    return 0;
  }
  void free() { if (p && !devpool::give(dev, p, cap)) (void)hipFree(p); p = nullptr; n = 0; cap = 0; }
  ~DevBuf() { free(); }
};

struct SchedBufs {
  DevBuf<uint32_t> step, idxw, lvl_pos, tail_bar;
  DevBuf<uint64_t> idxw64;
  DevBuf<int32_t> src;
  int upload(const Schedule &s) {
    int rc;
    if ((rc = step.upload(s.step)) || (rc = idxw.upload(s.idxw)) || (rc = idxw64.upload(s.idxw64)) || (rc = src.upload(s.src)) ||
        (rc = lvl_pos.upload(s.lvl_pos)) || (rc = tail_bar.upload(s.tail_bar))) return rc;
    return 0;
  }
  SchedDev view(const Schedule &s) const {
    SchedDev d; d.step = step.p; d.idxw = idxw.p; d.idxw64 = idxw64.p; d.lvl_pos = lvl_pos.p; d.tail_bar = tail_bar.p;
    d.n_phases = s.n_phases; d.nw = s.nw; d.n_levels = s.n_levels;
    d.n_steps = s.n_steps; d.n_slots = s.n_slots;
    return d;
  }
};

struct mi_osqp_batch {
  Settings st;
  std::shared_ptr<const Analysis> anp;        // pattern analysis: shared between the handles of one pattern (analysis cache below)
  int B = 0, BT = 1, ntiles = 0, threads = 512, device = 0, n_cus = 256;
  size_t lds = 0;
  std::vector<QPNumeric> qp;
  bool host_bounds_stale = false;
  hipStream_t stream = nullptr;
  SchedBufs fwd, bwd, chk;
  DevBuf<uint32_t> pinv, xloc;
  DevBuf<double> fwd_val, bwd_val, chk_val, dinv, x, z, y, q, l, u, rho_vec, rho_inv, Dsc, Dsc_inv, Esc, Esc_inv;
  DevBuf<double> dx, dy, out1, out2, dscal, x_out, y_out, xs_global;
  bool global_xs = false;
  int mw_groups = 0, mw_threads = 0;    // > 0: dataflow form of the solves (Analysis::df): mw_groups workgroups of mw_threads threads share the ONE QP of the handle
  DevBuf<uint32_t> mw_bar;              // their grid barrier: arrival count, generation, error word
  DevBuf<unsigned char> rflag;
  DevBuf<double> mw_scratch;            // partial norms / sums of the grid-wide check_kernel
  DevBuf<double> fwd_val0, bwd_val0, dinv0, rho_vec0, rho_inv0, dscal0;   // setup snapshot (reset)
  DevBuf<int> use_work;                       // per slot: the current factor is the working copy (KernelArgs::use_work)
  DevBuf<int> iscal, qp_of_slot, flag, npos;
  DevBuf<int2> pairs;
  // device refactorisation (BlockFactor tables + scratch)
  DevBuf<uint32_t> bf_blk, bf_lvl, bf_utask, bf_tri, bf_dtask, bf_ttask, bf_asm_dst, bf_asm_src, bf_ubig;
  DevBuf<int32_t> fwd_srcblk, bwd_srcblk;
  DevBuf<double> pa_val, Lblk, Dl, dinv_scratch;
  // dense tail (host_core.hpp DenseTail): task tables, the per-QP stream of S^-1 (+ setup snapshot), dense scratch
  DevBuf<uint32_t> dt_task, dt_wave_task, dt_wave_step, dt_tail_bar;
  DevBuf<uint32_t> dt_lt_pos, dt_ltcol_col, dt_tile_tab, dt_wave_tiles, dt_diag_tile, dt_task_step;     // tail_kernel tables
  DevBuf<uint64_t> dt_asm_q64;
  DevBuf<int32_t> dt_src_tile;
  int dt_nh = 0; uint32_t dt_cs_doubles = 0; size_t dt_lds = 0, dt_lds_asm = 0;
  DevBuf<int32_t> dt_src;
  DevBuf<double> dt_val, dt_val0, dt_Sd;
  DevBuf<uint32_t> sp_ptr, sp_ent;      // fused SpMV op (spmv_fused_kernel); empty when not eligible
  DevBuf<uint32_t> sp_ell, sp_rowid;    // its prefetching form (rows sorted by length, <= 4 passes of 512 rows, <= 24 entries per row)
  int sp_npass = 0;
  uint32_t sp_ell_off[4] = {0, 0, 0, 0}, sp_ell_k[4] = {0, 0, 0, 0};
  bool host_rho_stale = false;
  bool host_scaling_stale = false;      // the device has re-equilibrated (ruiz_kernel): P, A, q, D, E, c of the host mirrors lag behind
  DevBuf<int32_t> rz_prow, rz_pcol, rz_arow, rz_acol;      // per entry of triu(P) / A: row, column (row E2 on the device)
  bool clear_rho_updates = true;          // the next solve starts counting rho updates from 0 (setup / update_* / reset happened)
  int *h_npos = nullptr;
  DevBuf<double> stage; DevBuf<int> ids, work;
  int *h_iscal = nullptr;     // pinned (hostpool; *_cap = capacity to hand back)
  double *h_dscal = nullptr;  // pinned
  size_t h_iscal_cap = 0, h_dscal_cap = 0, h_npos_cap = 0, pin_cap = 0;
  double *pin = nullptr;      // pinned staging of the host update paths (a pageable hipMemcpy runs at ~1 GB/s here, and unevenly)
  size_t pin_n = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evf0 = nullptr, evf1 = nullptr, evf2 = nullptr;
  double factor_ms_sum = 0.0, dense_ms_sum = 0.0;
  int64_t refactor_launches = 0, refactor_qps = 0, peak_qps = 0;
  double peak_factor_ms = 0.0, peak_tail_ms = 0.0;
  mi_osqp_stats stats{};
  // last-solve accounting
  int64_t last_total_iters = 0, last_launches = 0, last_refactors = 0;
  double last_device_s = 0.0, last_refactor_s = 0.0, last_compact_s = 0.0, kernel_ms_sum = 0.0;
  int64_t kernel_launches = 0, kernel_qp_iters = 0;
  bool solved_once = false;
  // per-QP failure isolation: QPs whose KKT factor lost its inertia (at setup, in an update or in a rho update).  They
  // report kNonConvex with a NaN solution on every solve until a later refactorisation of theirs succeeds; the rest
  // of the batch is unaffected ([REF] src/osqp-wrapper.h:51-54: solve() never throws, one exit code per solver).
  std::vector<char> failed;
  DevBuf<int> fail_list;
  // raw triu(P) and q as setup received them ([QP][nnzP], [QP][n]): what mi_osqp_batch_reinit_some equilibrates from
  DevBuf<double> rawP, rawq;
  // ---- continuous batching (the per-QP entry points + advance / poll; section "continuous" below)
  struct Cont {
    bool on = false;
    int L = 25;                              // iterations per segment (gcd of the check / rho / max_iter periods)
    unsigned launch_seq = 0;                 // sequence number of the last advance launch (the stop word's currency)
    DevBuf<unsigned> stop;                   // device word: launch in which a QP last finished (advance_kernel)
    int64_t adv_seq = 0, polled_seq = 0;     // segments enqueued / segments whose flags the host has read
    std::vector<char> running, clear_rho;    // per QP: a solve is in flight; its next solve counts rho updates from 0
    std::vector<mi_osqp_info> info;          // per QP: result of its last finished solve
    int n_running = 0;
    // flag copies of the last two advances (pinned): iscal / dscal images + the event behind them
    int *h_is[2] = {nullptr, nullptr}; double *h_ds[2] = {nullptr, nullptr}; size_t h_is_cap[2] = {0, 0}, h_ds_cap[2] = {0, 0};
    hipEvent_t ev[2] = {nullptr, nullptr}; int64_t seq_of[2] = {0, 0};
    // solutions of finished QPs land in pinned host memory straight from check_kernel ([B][n], [B][m])
    double *xh = nullptr, *yh = nullptr; size_t xh_cap = 0, yh_cap = 0;
    // staging ring of the per-QP calls: regions are handed out once per call and recycled when the ring wraps (after a
    // synchronisation), so a call never waits for an earlier call's copy
    char *ring_h = nullptr; size_t ring_cap = 0, ring_h_cap = 0, ring_head = 0;
    DevBuf<char> ring_d;
    DevBuf<int> work;                        // device-built refactorisation work list of an advance
    // The per-QP calls (new data, equilibration, refactorisation, warm start, begin) and the refactorisations after rho
    // updates run on a second stream next to the advance launches: a QP that iterates through hundreds of segments must not
    // wait for the other QPs' updates.  Nothing orders the two streams: a begun solve carries a pending mark that the next
    // advance launch to see it takes up (IS_PENDING), a QP whose rho changed pauses until its refactorisation has run, and
    // the host tells a finished solve from the slot's previous one by its epoch (IS_EPOCH).
    hipStream_t ustream = nullptr;
    bool own_ustream = false;
    hipEvent_t ev_adv = nullptr;             // behind the last advance launch (the refactorisation kernels wait for it)
    std::vector<int> epoch;                  // per QP: solves begun so far (what IS_EPOCH reads once the begin has run)
    DevBuf<double> rz_scratch;               // equilibration scratch of that stream (not the check kernels': they may be running)
    double *keepA = nullptr, *keepl = nullptr, *keepu = nullptr;   // a mi_gomp_scene's QP-major copy of the raw rows: reinit / update keep it current
    DevBuf<unsigned> counter;                // tiles that have left the advance launch in flight
    unsigned *h_done = nullptr; size_t h_done_cap = 0;      // pinned: [0] sequence number of the last advance launch that is over, [1] tiles that iterated in it
    unsigned last_active = 0;                // tiles that iterated in the last launch polled
    hipEvent_t ev_u = nullptr;               // end of the second stream's queue (a launch after an idle one waits for it)
  } cont;
  ~mi_osqp_batch() {
    DevGuard guard(device);
    if (stream) (void)hipStreamSynchronize(stream);      // (the buffers go back to their pools right after: DevBuf remembers its device)
    for (int k = 0; k < 2; k++) { hostpool::give(cont.h_is[k], cont.h_is_cap[k]); hostpool::give(cont.h_ds[k], cont.h_ds_cap[k]); if (cont.ev[k]) (void)hipEventDestroy(cont.ev[k]); }
    hostpool::give(cont.xh, cont.xh_cap); hostpool::give(cont.yh, cont.yh_cap); hostpool::give(cont.ring_h, cont.ring_h_cap);
    if (cont.ustream && cont.own_ustream) { (void)hipStreamSynchronize(cont.ustream); (void)hipStreamDestroy(cont.ustream); }
    if (cont.ev_adv) (void)hipEventDestroy(cont.ev_adv);
    if (cont.ev_u) (void)hipEventDestroy(cont.ev_u);
    hostpool::give(cont.h_done, cont.h_done_cap);
    hostpool::give(h_iscal, h_iscal_cap); hostpool::give(h_dscal, h_dscal_cap); hostpool::give(pin, pin_cap); hostpool::give(h_npos, h_npos_cap);
    if (stream && ev0 && ev1 && evf0 && evf1 && evf2) streampool::give({device, stream, {ev0, ev1, evf0, evf1, evf2}});
    else {
      for (hipEvent_t e : {ev0, ev1, evf0, evf1, evf2}) if (e) (void)hipEventDestroy(e);
      if (stream) (void)hipStreamDestroy(stream);
    }
  }
};

struct mi_osqp_solver { mi_osqp_batch *b = nullptr; };

// ------------------------------------------------------------------ helpers

static Settings to_settings(const mi_osqp_settings *s) {
  Settings t;
  if (!s) return t;
  t.rho = s->rho; t.sigma = s->sigma; t.scaling = s->scaling; t.adaptive_rho = s->adaptive_rho;
  t.adaptive_rho_interval = s->adaptive_rho_interval; t.adaptive_rho_tolerance = s->adaptive_rho_tolerance;
  t.max_iter = s->max_iter; t.eps_abs = s->eps_abs; t.eps_rel = s->eps_rel; t.eps_prim_inf = s->eps_prim_inf;
  t.eps_dual_inf = s->eps_dual_inf; t.alpha = s->alpha; t.scaled_termination = s->scaled_termination;
  t.check_termination = s->check_termination; t.warm_start = s->warm_start; t.verbose = s->verbose;
  return t;
}

// threads per tile of the refactorisation kernel (experiments: MI_OSQP_FACTOR_THREADS)
static int factor_threads() {
  const char *e = getenv("MI_OSQP_FACTOR_THREADS");
  return e ? std::max(64, std::min(1024, atoi(e) / 64 * 64)) : 1024;
}

static size_t lds_bytes(int N, int BT, int threads) {
  int nw = threads / 64;
  return ((size_t)N * BT + (size_t)nw * 14 * BT + 14 * BT) * sizeof(double);
}

static KernelArgs make_args(mi_osqp_batch *h) {
  KernelArgs a{};
  a.n = (*h->anp).n; a.m = (*h->anp).m; a.N = (*h->anp).N; a.B = h->B;
  a.fwd = h->fwd.view((*h->anp).fwd); a.bwd = h->bwd.view((*h->anp).bwd); a.chk = h->chk.view((*h->anp).chk);
  a.pinv = h->pinv.p; a.xloc = h->xloc.p;
  a.fwd_val = h->fwd_val.p; a.bwd_val = h->bwd_val.p; a.chk_val = h->chk_val.p; a.dinv = h->dinv.p;
  a.fwd_val0 = h->fwd_val0.p; a.bwd_val0 = h->bwd_val0.p; a.dt_val0 = h->dt_val0.p;
  a.use_work = (h->fwd_val0.p && h->bwd_val0.p && (!(*h->anp).dt.k || h->dt_val0.p)) ? h->use_work.p : nullptr;     // (before the first snapshot: working copy only)
  a.x = h->x.p; a.z = h->z.p; a.y = h->y.p; a.q = h->q.p; a.l = h->l.p; a.u = h->u.p;
  a.rho_vec = h->rho_vec.p; a.rho_inv = h->rho_inv.p; a.Dsc = h->Dsc.p; a.Dsc_inv = h->Dsc_inv.p;
  a.Esc = h->Esc.p; a.Esc_inv = h->Esc_inv.p; a.dx = h->dx.p; a.dy = h->dy.p; a.out1 = h->out1.p; a.out2 = h->out2.p;
  a.dscal = h->dscal.p; a.iscal = h->iscal.p; a.qp_of_slot = h->qp_of_slot.p;
  a.x_out = h->x_out.p; a.y_out = h->y_out.p;
  a.xs_global = h->global_xs ? h->xs_global.p : nullptr; a.xs_len = (*h->anp).xs_total; a.wide = (*h->anp).wide ? 1 : 0;
  a.mw_groups = h->mw_groups; a.mw_bar = h->mw_bar.p; a.mw_scratch = h->mw_scratch.p;
  a.df = (*h->anp).df ? 1 : 0; a.df_shadow = (unsigned)(*h->anp).Next; a.rflag = h->rflag.p;
  {
    const DenseTail &dt = (*h->anp).dt;
    a.dt.s = dt.s; a.dt.k = dt.k; a.dt.n_phases = dt.n_phases; a.dt.n_steps = dt.n_steps;
    a.dt.task = h->dt_task.p; a.dt.wave_task = h->dt_wave_task.p; a.dt.wave_step = h->dt_wave_step.p; a.dt.tail_bar = h->dt_tail_bar.p;
    a.dt_val = h->dt_val.p;
  }
  const Settings &s = h->st;
  a.sigma = s.sigma; a.alpha = s.alpha; a.eps_abs = s.eps_abs; a.eps_rel = s.eps_rel;
  a.eps_prim_inf = s.eps_prim_inf; a.eps_dual_inf = s.eps_dual_inf; a.rho_tolerance = s.adaptive_rho_tolerance;
  a.check_termination = (int)s.check_termination; a.rho_interval = (int)s.adaptive_rho_interval;
  a.max_iter = (int)s.max_iter; a.scaled_termination = (int)s.scaled_termination; a.scaling = s.scaling ? 1 : 0;
  a.adaptive_rho = (int)s.adaptive_rho; a.iter_begin = 0; a.iter_end = 0; a.info_at_end = 1;
  return a;
}

static FactorArgs make_factor_args(mi_osqp_batch *h, int force_all) {
  FactorArgs a{};
  const Analysis &an = (*h->anp);
  a.n = an.n; a.m = an.m; a.N = an.N; a.B = h->B; a.nnzP = an.Pp[an.n]; a.nnzK = an.nnzK();
  a.pa_len = an.Pp[an.n] + an.Ap[an.n]; a.n_levels = an.bf.n_levels; a.force_all = force_all;
  a.storage = an.bf.storage; a.fwd = h->fwd.view(an.fwd); a.bwd = h->bwd.view(an.bwd);
  a.blk = h->bf_blk.p; a.lvl = h->bf_lvl.p; a.utask = h->bf_utask.p; a.tri4 = h->bf_tri.p; a.dtask = h->bf_dtask.p; a.ubig = h->bf_ubig.p;
  a.ttask = h->bf_ttask.p; a.asm_dst = h->bf_asm_dst.p; a.asm_src = h->bf_asm_src.p;
  a.fwd_srcblk = h->fwd_srcblk.p; a.bwd_srcblk = h->bwd_srcblk.p;
  a.pa_val = h->pa_val.p; a.l = h->l.p; a.u = h->u.p; a.dscal = h->dscal.p;
  a.rho_vec = h->rho_vec.p; a.rho_inv = h->rho_inv.p; a.Lblk = h->Lblk.p; a.Dl = h->Dl.p; a.dinv_scratch = h->dinv_scratch.p;
  a.fwd_val = h->fwd_val.p; a.bwd_val = h->bwd_val.p; a.dinv = h->dinv.p; a.iscal = h->iscal.p; a.npos = h->npos.p; a.use_work = h->use_work.p;
  a.sigma = h->st.sigma; a.home_bt = h->BT; a.dt_k = an.dt.k;
  a.mw_groups = 0; a.mw_bar = h->mw_bar.p;      // (device_refactor_slots decides how many workgroups share a QP)
#ifdef MI_OSQP_DEBUG_BUILD
  { const char *e = getenv("MI_OSQP_FACTOR_SKIP"); a.debug_skip = e ? atoi(e) : 0; }      // timing experiments, diagnostic build only
#endif
  return a;
}

static int ensure_stage(mi_osqp_batch *h, size_t doubles, size_t ints) {
  int rc;
  if (h->stage.n < doubles && (rc = h->stage.alloc(doubles))) return rc;
  if (h->ids.n < ints && (rc = h->ids.alloc(ints))) return rc;
  return 0;
}

static int ensure_pin(mi_osqp_batch *h, size_t doubles) {
  if (h->pin_n >= doubles) return 0;
  if (h->pin) { hostpool::give(h->pin, h->pin_cap); h->pin = nullptr; h->pin_n = 0; }
  HIPCHK(hostpool::alloc((void **)&h->pin, doubles * sizeof(double), &h->pin_cap));
  h->pin_n = h->pin_cap / sizeof(double);
  return 0;
}
// memcpy on the host threads (user arrays of several MB per call in the GOMP drivers' loops)
static void par_copy(double *dst, const double *src, size_t n) {
  const size_t chunk = (size_t)1 << 17;       // 1 MiB
  const int parts = (int)((n + chunk - 1) / chunk);
  if (parts <= 2) { memcpy(dst, src, n * sizeof(double)); return; }
  parallel_for(parts, [&](int i, int) { const size_t b = (size_t)i * chunk; memcpy(dst + b, src + b, std::min(chunk, n - b) * sizeof(double)); });
}

// upload QP-major host rows [nq][len] into a tile-interleaved device array
static int upload_rows(mi_osqp_batch *h, const std::vector<double> &rows, const std::vector<int> *ids, int nq,
                       int len, double *dst) {
  if (!nq || !len) return 0;
  int rc = ensure_stage(h, rows.size(), ids ? ids->size() : 0);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h->stage.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (ids) HIPCHK(hipMemcpyAsync(h->ids.p, ids->data(), ids->size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(launch_interleave(h->stage.p, dst, ids ? h->ids.p : nullptr, nq, len, h->BT, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

template <class G>
static std::vector<double> gather_rows(const std::vector<int> &ids, int len, G &&get) {
  std::vector<double> rows((size_t)ids.size() * len);
  for (size_t j = 0; j < ids.size(); j++) { const std::vector<double> &v = get(ids[j]); std::copy(v.begin(), v.begin() + len, rows.begin() + j * len); }
  return rows;
}

static int upload_rho(mi_osqp_batch *h, const std::vector<int> &ids) {
  int m = (*h->anp).m, nq = (int)ids.size(), rc;
  if (!nq || !m) return 0;
  std::vector<double> r1 = gather_rows(ids, m, [&](int q) -> const std::vector<double> & { return h->qp[q].rho_vec; });
  if ((rc = upload_rows(h, r1, &ids, nq, m, h->rho_vec.p))) return rc;
  std::vector<double> r2 = gather_rows(ids, m, [&](int q) -> const std::vector<double> & { return h->qp[q].rho_inv; });
  return upload_rows(h, r2, &ids, nq, m, h->rho_inv.p);
}

// scaled problem data, scalings, scalars of the listed QPs
static int upload_problem(mi_osqp_batch *h, const std::vector<int> &ids, bool with_matrices) {
  const Analysis &an = (*h->anp);
  int n = an.n, m = an.m, nq = (int)ids.size(), rc;
  if (!nq) return 0;
  auto up = [&](int len, double *dst, auto get) -> int {
    std::vector<double> rows = gather_rows(ids, len, get);
    return upload_rows(h, rows, &ids, nq, len, dst);
  };
  if (with_matrices) {
    int nnzP = an.Pp[n], nnzA = an.Ap[n];
    std::vector<double> pa((size_t)nq * (nnzP + nnzA));
    for (int j = 0; j < nq; j++) {
      const QPNumeric &q = h->qp[ids[j]];
      std::copy(q.Pv.begin(), q.Pv.end(), pa.begin() + (size_t)j * (nnzP + nnzA));
      std::copy(q.Av.begin(), q.Av.end(), pa.begin() + (size_t)j * (nnzP + nnzA) + nnzP);
    }
    if ((rc = ensure_stage(h, std::max<size_t>(pa.size(), 1), ids.size()))) return rc;
    HIPCHK(hipMemcpyAsync(h->ids.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (!pa.empty()) {
      HIPCHK(hipMemcpyAsync(h->stage.p, pa.data(), pa.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIPCHK(launch_scatter(h->stage.p, h->chk_val.p, h->chk.src.p, h->ids.p, nq, nnzP + nnzA, h->chk.view(an.chk), h->BT, h->stream));
      HIPCHK(launch_interleave(h->stage.p, h->pa_val.p, h->ids.p, nq, nnzP + nnzA, h->BT, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    }
    if ((rc = up(n, h->q.p, [&](int q) -> const std::vector<double> & { return h->qp[q].q; }))) return rc;
    if ((rc = up(n, h->Dsc.p, [&](int q) -> const std::vector<double> & { return h->qp[q].D; }))) return rc;
    if ((rc = up(n, h->Dsc_inv.p, [&](int q) -> const std::vector<double> & { return h->qp[q].Dinv; }))) return rc;
    if ((rc = up(m, h->Esc.p, [&](int q) -> const std::vector<double> & { return h->qp[q].E; }))) return rc;
    if ((rc = up(m, h->Esc_inv.p, [&](int q) -> const std::vector<double> & { return h->qp[q].Einv; }))) return rc;
  }
  if ((rc = up(m, h->l.p, [&](int q) -> const std::vector<double> & { return h->qp[q].l; }))) return rc;
  if ((rc = up(m, h->u.p, [&](int q) -> const std::vector<double> & { return h->qp[q].u; }))) return rc;
  return 0;
}

// write c, cinv, rho of the listed QPs into dscal (read-modify-write via host mirror)
static int sync_scalars_to_device(mi_osqp_batch *h, const std::vector<int> &ids, bool with_c) {
  size_t cnt = (size_t)h->ntiles * DS_COUNT * h->BT;
  HIPCHK(hipMemcpy(h->h_dscal, h->dscal.p, cnt * sizeof(double), hipMemcpyDeviceToHost));
  for (int q : ids) {
    double *t = h->h_dscal + (size_t)(q / h->BT) * DS_COUNT * h->BT;
    int b = q % h->BT;
    if (with_c) { t[DS_C * h->BT + b] = h->qp[q].c; t[DS_CINV * h->BT + b] = h->qp[q].cinv; }
    t[DS_RHO * h->BT + b] = h->qp[q].rho;
    t[DS_RHO_EST * h->BT + b] = h->qp[q].rho;
  }
  HIPCHK(hipMemcpy(h->dscal.p, h->h_dscal, cnt * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

static int snapshot(mi_osqp_batch *h) {
  auto cp = [&](DevBuf<double> &dst, DevBuf<double> &src) -> int {
    if (dst.n != src.n) { int rc = dst.alloc(src.n); if (rc) return rc; }
    if (src.n) HIPCHK(hipMemcpyAsync(dst.p, src.p, src.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return 0;
  };
  // the factor streams: only the QPs whose current factor IS the working copy (the others' snapshot is current, and their
  // working copy may be a leftover of an earlier solve: a reset clears flags, it does not copy)
  const Analysis &an = (*h->anp);
  const int nslots = h->ntiles * h->BT;
  auto cps = [&](DevBuf<double> &dst, DevBuf<double> &src, size_t per) -> int {
    const bool fresh = dst.n != src.n;
    if (fresh) { int rc = dst.alloc(src.n); if (rc) return rc; }
    if (src.n) HIPCHK(launch_copy_flagged_streams(dst.p, src.p, fresh ? nullptr : h->use_work.p, 1, nslots, per, h->stream));
    return 0;
  };
  int rc;
  if ((rc = cps(h->fwd_val0, h->fwd_val, (size_t)an.fwd.phys_steps() * 64)) || (rc = cps(h->bwd_val0, h->bwd_val, (size_t)an.bwd.phys_steps() * 64)) ||
      (an.dt.k && (rc = cps(h->dt_val0, h->dt_val, (size_t)an.dt.n_steps * 64))) ||
      (rc = cp(h->dinv0, h->dinv)) || (rc = cp(h->rho_vec0, h->rho_vec)) || (rc = cp(h->rho_inv0, h->rho_inv)) || (rc = cp(h->dscal0, h->dscal))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// download scaled l,u from the device into the host mirrors (after device-side bound updates)
static int sync_bounds_to_host(mi_osqp_batch *h) {
  if (!h->host_bounds_stale) return 0;
  int m = (*h->anp).m, B = h->B, rc;
  if ((rc = ensure_stage(h, (size_t)B * m + 1, 0))) return rc;
  std::vector<double> tmp((size_t)B * m);
  for (int which = 0; which < 2; which++) {
    HIPCHK(launch_deinterleave(which ? h->u.p : h->l.p, h->stage.p, B, m, h->BT, h->stream));
    HIPCHK(hipMemcpyAsync(tmp.data(), h->stage.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int q = 0; q < B; q++) {
      std::vector<double> &dst = which ? h->qp[q].u : h->qp[q].l;
      std::copy(tmp.begin() + (size_t)q * m, tmp.begin() + (size_t)(q + 1) * m, dst.begin());
    }
  }
  h->host_bounds_stale = false;
  return 0;
}

// after device-side rho updates the host mirrors (rho, rho_vec) lag behind
static int sync_rho_to_host(mi_osqp_batch *h) {
  if (!h->host_rho_stale) return 0;
  int rc = sync_bounds_to_host(h);
  if (rc) return rc;
  size_t dcnt = (size_t)h->ntiles * DS_COUNT * h->BT;
  HIPCHK(hipMemcpy(h->h_dscal, h->dscal.p, dcnt * sizeof(double), hipMemcpyDeviceToHost));
  for (int q = 0; q < h->B; q++) {
    double r = h->h_dscal[(size_t)(q / h->BT) * DS_COUNT * h->BT + DS_RHO * h->BT + q % h->BT];
    if (r != h->qp[q].rho) apply_rho((*h->anp), h->qp[q], r);
  }
  h->host_rho_stale = false;
  return 0;
}

static int reset_solve_state(mi_osqp_batch *h, bool cold) {
  int rc;
  // statuses: UNSOLVED, not done (padding lanes of the last tile stay done)
  // (the count of rho updates survives a plain re-solve, as upstream's info->rho_updates does: only setup and the
  //  update_* calls reset it)
  size_t icnt = (size_t)h->ntiles * IS_COUNT * h->BT;
  for (int t = 0; t < h->ntiles; t++)
    for (int b = 0; b < h->BT; b++) {
      int *p = h->h_iscal + (size_t)t * IS_COUNT * h->BT;
      const int keep = h->clear_rho_updates ? 0 : p[IS_RHO_UPDATES * h->BT + b];
      for (int k = 0; k < IS_COUNT; k++) p[k * h->BT + b] = 0;
      p[IS_STATUS * h->BT + b] = -10;
      p[IS_DONE * h->BT + b] = (t * h->BT + b >= h->B) ? 1 : 0;
      p[IS_RHO_UPDATES * h->BT + b] = keep;
    }
  h->clear_rho_updates = false;
  HIPCHK(hipMemcpyAsync(h->iscal.p, h->h_iscal, icnt * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (cold) { if ((rc = h->x.zero(h->stream)) || (rc = h->z.zero(h->stream)) || (rc = h->y.zero(h->stream))) return rc; }
  return 0;
}

// ------------------------------------------------------------------- setup

// Pattern analysis (ordering, symbolic factor, step streams, block-factor and dense-tail tables) depends only on the
// sparsity pattern and the launch shape.  The GOMP drivers build one solver per horizon segment and the same ten
// patterns come back on every run(): the last analyses are kept (process-wide, keyed by a hash of the pattern, verified
// by a full comparison) and shared read-only between handles.  MI_OSQP_ANALYSIS_CACHE=0 switches the cache off.
namespace ancache {
struct Entry {
  uint64_t hash; int64_t n, m; int nw, bt, max_extra, dt_max, tri_waves, n_tiles; std::string env;
  std::vector<int64_t> Pp, Pi, Ap, Ai;
  std::shared_ptr<const Analysis> an;           // null while a thread is still computing it (mi_osqp_prefetch_analysis): others wait
};
static std::mutex mu;
static std::condition_variable cv;
static std::vector<Entry> entries;            // most recently used last
constexpr size_t kMaxEntries = 24;
static uint64_t fnv(uint64_t h, const void *p, size_t bytes) {
  const unsigned char *c = (const unsigned char *)p;
  for (size_t i = 0; i < bytes; i++) { h ^= c[i]; h *= 1099511628211ull; }
  return h;
}
}  // namespace ancache

static int cached_analysis(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const int64_t *Ap, const int64_t *Ai, int nw,
                           int bt, int max_extra, int dt_max, int tri_waves, int n_tiles, std::shared_ptr<const Analysis> &out) {
  n_tiles = n_tiles <= 1 ? 1 : (n_tiles <= 32 ? 32 : (n_tiles <= 128 ? 128 : (n_tiles <= 256 ? 256 : 512)));      // (buckets: the analysis is shared between handles)
  const char *off = getenv("MI_OSQP_ANALYSIS_CACHE");
  const bool use = !(off && atoi(off) == 0) && n > 0 && m >= 0 && Pp && Ap && Pp[0] == 0 && Ap[0] == 0 && Pp[n] >= 0 && Ap[n] >= 0 &&
                   Pp[n] < ((int64_t)1 << 30) && Ap[n] < ((int64_t)1 << 30);
  std::string env;                              // the knobs analyze() reads
  for (const char *k : {"MI_OSQP_DENSE_TAIL", "MI_OSQP_ORDERING", "MI_OSQP_ND_LEAF", "MI_OSQP_RELAX"}) { const char *v = getenv(k); env += v ? v : "-"; env += ';'; }
  uint64_t hsh = 1469598103934665603ull;
  auto same = [&](const ancache::Entry &e) {
    if (e.hash != hsh || e.n != n || e.m != m || e.nw != nw || e.bt != bt || e.max_extra != max_extra || e.dt_max != dt_max || e.tri_waves != tri_waves || e.n_tiles != n_tiles || e.env != env) return false;
    return (int64_t)e.Pi.size() == Pp[n] && (int64_t)e.Ai.size() == Ap[n] && !memcmp(e.Pp.data(), Pp, (size_t)(n + 1) * 8) &&
           !memcmp(e.Pi.data(), Pi, (size_t)Pp[n] * 8) && !memcmp(e.Ap.data(), Ap, (size_t)(n + 1) * 8) && !memcmp(e.Ai.data(), Ai, (size_t)Ap[n] * 8);
  };
  if (use) {
    hsh = ancache::fnv(hsh, Pp, (size_t)(n + 1) * 8); hsh = ancache::fnv(hsh, Pi, (size_t)Pp[n] * 8);
    hsh = ancache::fnv(hsh, Ap, (size_t)(n + 1) * 8); hsh = ancache::fnv(hsh, Ai, (size_t)Ap[n] * 8);
    std::unique_lock<std::mutex> lk(ancache::mu);
    for (;;) {
      size_t i = 0;
      while (i < ancache::entries.size() && !same(ancache::entries[i])) i++;
      if (i == ancache::entries.size()) break;
      if (!ancache::entries[i].an) { ancache::cv.wait(lk); continue; }      // being computed by another thread: wait, look again
      out = ancache::entries[i].an;
      std::rotate(ancache::entries.begin() + i, ancache::entries.begin() + i + 1, ancache::entries.end());
      return MI_OSQP_OK;
    }
    // ours to compute: a placeholder tells the others
    if (ancache::entries.size() >= ancache::kMaxEntries) {
      for (size_t i = 0; i < ancache::entries.size(); i++) if (ancache::entries[i].an) { ancache::entries.erase(ancache::entries.begin() + i); break; }
    }
    ancache::entries.push_back(ancache::Entry{hsh, n, m, nw, bt, max_extra, dt_max, tri_waves, n_tiles, env, {Pp, Pp + n + 1}, {Pi, Pi + Pp[n]}, {Ap, Ap + n + 1}, {Ai, Ai + Ap[n]}, nullptr});
  }
  auto an = std::make_shared<Analysis>();
  const int rc = analyze(n, m, Pp, Pi, Ap, Ai, *an, nw, bt, max_extra, dt_max, tri_waves, n_tiles);
  if (!rc) out = an;
  if (use) {
    std::lock_guard<std::mutex> lk(ancache::mu);
    for (size_t i = 0; i < ancache::entries.size(); i++)
      if (!ancache::entries[i].an && same(ancache::entries[i])) {
        if (rc) ancache::entries.erase(ancache::entries.begin() + i); else ancache::entries[i].an = out;
        break;
      }
    ancache::cv.notify_all();
  }
  return rc;
}

static int refactor_qps(mi_osqp_batch *h, std::vector<int> qps);
static bool host_ruiz(const mi_osqp_batch *h);
// Kernels that spin on their own grid (the dataflow sweeps and grid barriers of a large single QP, the grouped
// refactorisation) need every workgroup of the grid resident.  The grids are clamped to what the device keeps resident
// (max_coresident_groups); two such grids on one device could still starve each other, so their launches - from any
// handle and host thread of the process - take turns per device, from the launch to its completion.
static std::mutex &spin_mutex(int device) {
  static std::mutex mu[64];
  return mu[(unsigned)device % 64u];
}
static int restore_snapshot(mi_osqp_batch *h);
static int materialise_working(mi_osqp_batch *h);
static int cont_leave(mi_osqp_batch *h);      // (a blocking call ends the continuous mode of a handle: section "continuous")

// multi-workgroup mode: a grid barrier that gave up waiting (a workgroup of the grid was not resident) leaves its error
// word set; the results of that launch are garbage
static int mw_barrier_ok(mi_osqp_batch *h) {
  if (h->mw_groups <= 0) return MI_OSQP_OK;
  uint32_t w[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpy(w, h->mw_bar.p, sizeof(w), hipMemcpyDeviceToHost));
  if (w[2]) { g_last_error = "dataflow solve: a wait for a vector entry or a grid barrier timed out"; return MI_OSQP_ERR_DEVICE; }
  return MI_OSQP_OK;
}

// The launch shape of a handle (threads per workgroup, QPs per tile, where the solve vector lives, the grid of a large single
// QP): what the pattern analysis is built for.  Shared by setup and by mi_osqp_prefetch_analysis.
static void derive_shape(mi_osqp_batch *h, int64_t B, int64_t n, int64_t m, int64_t device, int &BT_out, int &max_extra_out) {
  const char *eth = getenv("MI_OSQP_THREADS");
  h->threads = eth ? std::max(64, std::min(1024, atoi(eth) / 64 * 64)) : 512;
  // ---- tile shape (needed by the schedule layout)
  // 2 QPs per tile: iterate_kernel<2> needs 112 VGPRs, so two 512-thread workgroups share a CU and
  // cover each other's barrier stalls; measured best on the 1024-QP headline batch (4 and 1 are slower)
  int BT = B >= 384 ? 2 : 1;
  {
    const char *et = getenv("MI_OSQP_TILE");
    if (et && (atoi(et) == 1 || atoi(et) == 2 || atoi(et) == 4)) BT = atoi(et);
  }
  // vectors of 65 535 entries and more: 32-bit index words (twice the index bytes), the solve vector in global
  // memory, one QP per tile, 512 threads (the only instantiation of the wide kernels)
  const bool wide = n + m >= 65535 || 2 * n + m >= 65535;
  if (wide) { BT = 1; h->threads = std::min(h->threads, 512); }
  const size_t lds_cap = 160 * 1024 - 1024;      // (the kernels carry up to 272 B of static LDS of their own: a vector that fills
                                                 //  the 160 KB to the last byte - extra rows are handed out until it does - cannot launch)
  while (BT > 1 && lds_bytes((int)(n + m), BT, h->threads) > lds_cap) BT /= 2;
  // 16 waves per tile where a tile has its CU to itself anyway - no more tiles than CUs, or a vector of more than half the
  // LDS: 1 024 against 512 threads measured -10 % on 256 GOMP QPs (7 DOF x 100 waypoints), -18 % on 1 024 of them (2 per
  // tile, 100 KB), -4..7 % on 128-QP shards of the headline batch; the headline batch itself (512 tiles of 40 KB: two
  // 8-wave workgroups per CU cover each other's barrier stalls) stays at 8 waves
  if (!eth && !wide && BT <= 2 && !getenv("MI_OSQP_GLOBAL_XS") && lds_bytes((int)(n + m), BT, 1024) <= lds_cap) {     // (16-wave kernels: 1 or 2 QPs per tile)
    int dev = (int)device, cus = 256;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    (void)hipGetLastError();
    const int64_t tiles = (B + BT - 1) / BT;
    if (tiles <= cus || lds_bytes((int)(n + m), BT, 1024) > 80 * 1024) h->threads = 1024;
  }
  // too large for LDS even at one QP per tile: the solve vector goes to a per-tile global buffer
  h->global_xs = wide || lds_bytes((int)(n + m), BT, h->threads) > lds_cap || getenv("MI_OSQP_GLOBAL_XS") != nullptr;
  // rows that may get a second vector position (phase B of the solves): what still fits LDS (analyze() also
  // respects the 16-bit index range of the narrow index words)
  int max_extra = -1;
  if (!h->global_xs) {
    const size_t cap_rows = (lds_cap - lds_bytes(0, BT, h->threads)) / (sizeof(double) * BT);
    max_extra = (int)(cap_rows - (size_t)(n + m));
  }
  // dense tail (inverted Schur complement of the trailing rows): needs the LDS vector and <= 512 rows (one row per
  // thread of dense_inverse_kernel; k^3 flops per refactorisation)
  // one QP whose vector lives in global memory: the dataflow form of the solves, shared by several workgroups (one CU
  // cannot issue the scattered 8-byte gathers / read-modify-writes of a 10^5-row factor fast enough, and a barrier per
  // level costs more than the level: DESIGN.md 7.4).  MI_OSQP_GROUPS = 0: the barrier form, one workgroup.
  h->mw_groups = 0; h->mw_threads = 0;
  if (h->global_xs && BT == 1 && B == 1 && h->threads <= 512) {
    const char *eg = getenv("MI_OSQP_GROUPS"), *ew = getenv("MI_OSQP_GROUP_THREADS");
    // measured on the 316 x 316 grid of config 5 (scripts/mw_probe.py, ms per iteration): 16 x 512 threads 1.56, 32 x 512 1.23,
    // 64 x 256 1.06, 128 x 128 1.01, 256 x 128 0.97 (the barrier form in one workgroup: 6.5); 150 x 150 grid: 128 x 128 0.40
    // (round 3, same probe: 316 x 316 grid 128 x 128 threads 0.868, 256 x 128 0.818, 192 x 128 0.823, 128 x 256 0.824, 256 x 64 0.899;
    //  150 x 150 grid: 0.371 / 0.401 / 0.387 / 0.396 / 0.394 - the larger grid for the larger QP)
    //  and smaller grids for the mid-size ones (`scripts/groups_probe.py`, ms per 25-iteration solve: the 802-waypoint trajectory QP,
    //  N = 43 284: 128 x 128 3.46, 64 x 128 3.25, 256 x 128 4.38; 402 waypoints, N = 21 684: 3.34 / 2.96, 32 x 128 2.93)
    const int64_t Nrows = n + m;
    h->mw_groups = eg ? std::max(0, std::min(256, atoi(eg))) : (Nrows >= 250000 ? 256 : Nrows >= 60000 ? 128 : Nrows >= 30000 ? 64 : 32);
    h->mw_threads = ew ? std::max(64, std::min(512, atoi(ew) / 64 * 64)) : 128;
    if (h->mw_groups * (h->mw_threads / 64) > 2048) h->mw_groups = 2048 / (h->mw_threads / 64);
    // never more workgroups than the device keeps resident at once (their waits are for each other): the schedules are
    // built for the clamped grid; a device that cannot hold two of them gets the barrier form in one workgroup
    int dev = (int)device, cus = 0;
    if ((dev >= 0 || hipGetDevice(&dev) == hipSuccess) && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
      DevGuard guard(dev);
      { const char *ec = getenv("MI_OSQP_ASSUME_CUS"); if (ec && atoi(ec) > 0) cus = std::min(cus, atoi(ec)); }      // (tests: the clamp on a device with fewer CUs)
      const int cap = max_coresident_groups(h->mw_threads, lds_bytes(0, 1, h->threads), cus);
      if (cap > 0 && h->mw_groups > cap) h->mw_groups = cap;
      if (h->mw_groups == 1) h->mw_groups = 0;
    }
    (void)hipGetLastError();
  }
  BT_out = BT; max_extra_out = max_extra;
}

// Shape and analysis of a handle.  The shape rules above were measured on trajectory QPs (sparse chain-like factors: the
// iteration is a chain of phases, two QPs per tile share them, and with more tiles than CUs 8-wave workgroups win: 2 048 QPs of
// 6 DOF x 50 waypoints 3.6 ms per batch solve against 5.4 at one QP per tile in 16 waves).  A pattern whose analysis comes back
// with a DENSE TAIL is the other kind - its iteration streams the inverted Schur complement, i.e. bandwidth - and runs best at
// ONE QP per tile in 16-wave workgroups whatever the batch size (the headline batch, end of round 3: two per tile / 8 waves 37.6
// ms per step, one per tile / 8 waves 36.6, one per tile / 16 waves 32.0; the refactorisation packs four small tasks per wave
// at one QP per tile against two).  So such a pattern is analysed a second time for that shape (both analyses are cached).
static int shape_and_analysis(mi_osqp_batch *h, int64_t B, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const int64_t *Ap,
                              const int64_t *Ai, int64_t device, int &BT, int &max_extra, std::shared_ptr<const Analysis> &out) {
  derive_shape(h, B, n, m, device, BT, max_extra);
  int rc = cached_analysis(n, m, Pp, Pi, Ap, Ai, h->threads / 64, BT, max_extra, h->global_xs ? 0 : 512, h->mw_groups * (h->mw_threads / 64), (int)((B + BT - 1) / BT), out);
  if (rc) return rc;
  const size_t lds_cap = 160 * 1024 - 1024;
  if (out->dt.k > 0 && (BT != 1 || h->threads != 1024) && !h->global_xs && !getenv("MI_OSQP_TILE") && !getenv("MI_OSQP_THREADS") &&
      !getenv("MI_OSQP_NO_DENSE_SHAPE") && lds_bytes((int)(n + m), 1, 1024) + (size_t)2 * 512 * sizeof(double) <= lds_cap) {
    const int BT2 = 1, thr2 = 1024;
    const size_t cap_rows = (lds_cap - lds_bytes(0, BT2, thr2)) / (sizeof(double) * BT2);
    const int max_extra2 = (int)(cap_rows - (size_t)(n + m));
    std::shared_ptr<const Analysis> an2;
    rc = cached_analysis(n, m, Pp, Pi, Ap, Ai, thr2 / 64, BT2, max_extra2, 512, 0, (int)B, an2);
    if (!rc && an2->dt.k > 0) { BT = BT2; h->threads = thr2; max_extra = max_extra2; out = an2; }
  }
  return MI_OSQP_OK;
}

static int batch_setup_impl(mi_osqp_batch *h, int64_t B, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi,
                            const double *Pv, const double *q, const int64_t *Ap, const int64_t *Ai,
                            const double *Av, const double *l, const double *u, int64_t device) {
  double t0 = now_s();
  if (B <= 0 || !Pp || !Ap || (m > 0 && (!l || !u)) || (Pp[n] > 0 && !Pv) || (Ap[n] > 0 && !Av)) return MI_OSQP_ERR_INVALID_DATA;
  if (validate_settings(h->st)) return MI_OSQP_ERR_INVALID_SETTINGS;
  if (h->st.adaptive_rho && !h->st.adaptive_rho_interval)   // deterministic "auto" (upstream non-PROFILING rule)
    h->st.adaptive_rho_interval = h->st.check_termination ? 4 * h->st.check_termination : 100;
  for (int64_t k = 0; k < B * m; k++) if (l[k] > u[k]) return MI_OSQP_ERR_INVALID_DATA;
  int BT = 1, max_extra = -1;
  const size_t lds_cap = 160 * 1024 - 1024;
  const double ta0 = now_s();
  int rc = shape_and_analysis(h, B, n, m, Pp, Pi, Ap, Ai, device, BT, max_extra, h->anp);
  const double t_analysis = now_s() - ta0;
  if (rc) return rc;
  const Analysis &an = (*h->anp);
  h->B = (int)B;
  h->failed.assign((size_t)B, 0);
  // ---- device
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_last_error = "no HIP device visible"; return MI_OSQP_ERR_DEVICE; }
  if (device >= ndev) { g_last_error = "no such HIP device"; return MI_OSQP_ERR_DEVICE; }
  if (device >= 0) h->device = (int)device; else HIPCHK(hipGetDevice(&h->device));
  DevGuard guard(h->device);            // (the caller's current device is restored when setup returns)
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, h->device));
  h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { g_last_error = std::string("device is not gfx950: ") + prop.gcnArchName; return MI_OSQP_ERR_DEVICE; }
  {
    streampool::Bundle sb;
    if (streampool::take(h->device, sb)) { h->stream = sb.stream; h->ev0 = sb.ev[0]; h->ev1 = sb.ev[1]; h->evf0 = sb.ev[2]; h->evf1 = sb.ev[3]; h->evf2 = sb.ev[4]; }
    else {
      HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreate(&h->ev0)); HIPCHK(hipEventCreate(&h->ev1));
      HIPCHK(hipEventCreate(&h->evf0)); HIPCHK(hipEventCreate(&h->evf1)); HIPCHK(hipEventCreate(&h->evf2));
    }
  }
  h->BT = BT; h->ntiles = (int)((B + BT - 1) / BT);
  h->lds = h->global_xs ? lds_bytes(0, BT, h->threads) : lds_bytes(an.Next + 2 * an.dt.k, BT, h->threads);
  if (h->lds > lds_cap) { g_last_error = "internal: LDS budget exceeded"; return MI_OSQP_ERR_ALLOC; }
  // ---- device arrays
  size_t T = (size_t)h->ntiles * BT;
  if ((rc = h->fwd.upload(an.fwd)) || (rc = h->bwd.upload(an.bwd)) || (rc = h->chk.upload(an.chk))) return rc;
  { std::vector<uint32_t> pv(an.pinv.begin(), an.pinv.end()); if ((rc = h->pinv.upload(pv))) return rc; }
  { std::vector<uint32_t> xv(an.xloc.begin(), an.xloc.end()); if ((rc = h->xloc.upload(xv))) return rc; }
  {
    std::vector<int32_t> pr(an.Pi.begin(), an.Pi.begin() + an.Pp[n]), pc(an.Pp[n]), ar(an.Ai.begin(), an.Ai.begin() + an.Ap[n]), ac(an.Ap[n]);
    for (int j = 0; j < n; j++) { for (int k = an.Pp[j]; k < an.Pp[j + 1]; k++) pc[k] = j; for (int k = an.Ap[j]; k < an.Ap[j + 1]; k++) ac[k] = j; }
    if ((rc = h->rz_prow.upload(pr)) || (rc = h->rz_pcol.upload(pc)) || (rc = h->rz_arow.upload(ar)) || (rc = h->rz_acol.upload(ac))) return rc;
  }
  if ((rc = h->use_work.alloc(T)) || (rc = h->use_work.zero(h->stream))) return rc;
#define ALLOC(buf, len) if ((rc = h->buf.alloc((size_t)(len) * T)) || (rc = h->buf.zero(h->stream))) return rc
  ALLOC(fwd_val, (size_t)an.fwd.phys_steps() * 64); ALLOC(bwd_val, (size_t)an.bwd.phys_steps() * 64); ALLOC(chk_val, (size_t)an.chk.phys_steps() * 64); ALLOC(dinv, an.N);
  ALLOC(x, n); ALLOC(z, m); ALLOC(y, m); ALLOC(q, n); ALLOC(l, m); ALLOC(u, m); ALLOC(rho_vec, m); ALLOC(rho_inv, m);
  ALLOC(Dsc, n); ALLOC(Dsc_inv, n); ALLOC(Esc, m); ALLOC(Esc_inv, m); ALLOC(dx, n); ALLOC(dy, m);
  ALLOC(out1, 2 * n + m); ALLOC(out2, 2 * n + m); ALLOC(dscal, DS_COUNT);
  if (h->global_xs) { ALLOC(xs_global, an.xs_total); }
  if ((rc = h->mw_bar.alloc(4 * (size_t)std::max(h->n_cus, 1))) || (rc = h->mw_bar.zero(h->stream))) return rc;      // (one set of barrier words per work tile)
  if (an.df && ((rc = h->rflag.upload(an.rflag)) || (rc = h->mw_scratch.alloc((size_t)8 * 256 * 16)) || (rc = h->mw_scratch.zero(h->stream)))) return rc;
  if (an.dt.k) {
    const DenseTail &dt = an.dt;
    ALLOC(dt_val, (size_t)dt.n_steps * 64);
    if ((rc = h->dt_task.upload(dt.task)) || (rc = h->dt_wave_task.upload(dt.wave_task)) || (rc = h->dt_wave_step.upload(dt.wave_step)) ||
        (rc = h->dt_tail_bar.upload(dt.tail_bar)) || (rc = h->dt_src.upload(dt.src)) ||
        (rc = h->dt_lt_pos.upload(dt.lt_pos)) || (rc = h->dt_ltcol_col.upload(dt.ltcol_col)) || (rc = h->dt_tile_tab.upload(dt.tile_tab)) ||
        (rc = h->dt_wave_tiles.upload(dt.wave_tiles)) || (rc = h->dt_asm_q64.upload(dt.asm_q64)) ||
        (rc = h->dt_src_tile.upload(dt.src_tile)) || (rc = h->dt_diag_tile.upload(dt.diag_tile)) || (rc = h->dt_task_step.upload(dt.task_step)) ||
        (rc = h->dt_Sd.alloc((size_t)dt.k * dt.k * (T + 4)))) return rc;
    // LDS plan of tail_kernel: the staged half of the panel (<= 14 row tiles of 8 KiB; at least the 64 x 65 image of a pivot
    // block) + Pn in operand order (32 KiB), or the compact factor entries of the assembly phase, whichever is larger
    const int nrt = dt.k / 16 - 4;
    h->dt_nh = nrt <= 14 ? std::max(nrt, 1) : (nrt + 1) / 2;
    h->dt_cs_doubles = (uint32_t)std::max(h->dt_nh * 1024, 64 * 65 + 63) / 64 * 64;
    h->dt_lds = std::max<size_t>((size_t)h->dt_cs_doubles + 4096, 2 * 64 * 65) * sizeof(double);       // (the stream write keeps two 64 x 65 images)
    h->dt_lds_asm = (dt.asm_lds_bytes() + 255) & ~(size_t)255;
    if (h->dt_lds > lds_cap || h->dt_lds_asm > lds_cap) { g_last_error = "internal: LDS budget of tail_kernel exceeded"; return MI_OSQP_ERR_ALLOC; }
  }
#undef ALLOC
  if ((rc = h->iscal.alloc((size_t)IS_COUNT * T)) || (rc = h->qp_of_slot.alloc((size_t)h->ntiles * BT)) || (rc = h->flag.alloc(4))) return rc;
  {
    const BlockFactor &bf = an.bf;
    if ((rc = h->bf_blk.upload(bf.blk)) || (rc = h->bf_lvl.upload(bf.lvl)) || (rc = h->bf_ubig.upload(bf.ubig)) || (rc = h->bf_utask.upload(bf.utask4)) ||
        (rc = h->bf_tri.upload(bf.tri4)) || (rc = h->bf_dtask.upload(bf.dtask4)) || (rc = h->bf_ttask.upload(bf.ttask4)) ||
        (rc = h->bf_asm_dst.upload(bf.asm_dst)) || (rc = h->bf_asm_src.upload(bf.asm_src)) ||
        (rc = h->fwd_srcblk.upload(an.fwd_srcblk)) || (rc = h->bwd_srcblk.upload(an.bwd_srcblk))) return rc;
    if ((rc = h->pa_val.alloc((size_t)(an.Pp[n] + an.Ap[n]) * T)) || (rc = h->pa_val.zero(h->stream)) ||
        // (+4 QPs: a refactorisation may pack its work list with up to 4 QPs per workgroup, rounded up)
        (rc = h->Lblk.alloc((size_t)bf.storage * (T + 4))) || (rc = h->Dl.alloc((size_t)an.N * (T + 4))) || (rc = h->Dl.zero(h->stream)) ||
        (rc = h->dinv_scratch.alloc((size_t)an.N * (T + 4))) || (rc = h->npos.alloc(T))) return rc;
    HIPCHK(hostpool::alloc((void **)&h->h_npos, T * sizeof(int), &h->h_npos_cap));
  }
  // tables of the fused SpMV op: rows of [P x ; A'y ; A x] as (value position, vector index) pairs, 16 bits each; used
  // when the compact values of a tile and [x ; y] fit LDS together
  if (!h->global_xs && an.Pp[n] + an.Ap[n] < 65536 && n + m < 65536 &&
      spmv_fused_lds_bytes((int)n, (int)m, an.Pp[n] + an.Ap[n], BT) <= 150 * 1024) {
    const int nnzP = an.Pp[n];
    std::vector<std::vector<uint32_t>> rows((size_t)2 * n + m);
    for (int c = 0; c < (int)n; c++)
      for (int k = an.Pp[c]; k < an.Pp[c + 1]; k++) {
        const int r = an.Pi[k];
        rows[r].push_back((uint32_t)k | ((uint32_t)c << 16));
        if (r != c) rows[c].push_back((uint32_t)k | ((uint32_t)r << 16));
      }
    for (int c = 0; c < (int)n; c++)
      for (int k = an.Ap[c]; k < an.Ap[c + 1]; k++) {
        const int r = an.Ai[k];
        rows[(size_t)n + c].push_back((uint32_t)(nnzP + k) | ((uint32_t)(n + r) << 16));
        rows[(size_t)2 * n + r].push_back((uint32_t)(nnzP + k) | ((uint32_t)c << 16));
      }
    std::vector<uint32_t> ptr{0u}, ent;
    for (auto &rw : rows) { ent.insert(ent.end(), rw.begin(), rw.end()); ptr.push_back((uint32_t)ent.size()); }
    if ((rc = h->sp_ptr.upload(ptr)) || (rc = h->sp_ent.upload(ent))) return rc;
    // prefetching form
    const size_t nrows = rows.size();
    size_t maxlen = 0;
    for (auto &rw : rows) maxlen = std::max(maxlen, rw.size());
    if (nrows <= 4 * 512 && maxlen <= 24) {
      std::vector<uint32_t> order(nrows);
      for (size_t r = 0; r < nrows; r++) order[r] = (uint32_t)r;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return rows[x].size() > rows[y].size(); });
      const int npass = (int)((nrows + 511) / 512);
      std::vector<uint32_t> ell, rowid((size_t)npass * 512, 0xFFFFFFFFu);
      const uint32_t pad = (uint32_t)(an.Pp[n] + an.Ap[n]);          // value position of the zero, vector index 0
      for (int p = 0; p < npass; p++) {
        const size_t r0 = (size_t)p * 512, r1 = std::min(nrows, r0 + 512);
        const uint32_t K = (uint32_t)rows[order[r0]].size();
        h->sp_ell_off[p] = (uint32_t)ell.size(); h->sp_ell_k[p] = K;
        ell.resize(ell.size() + (size_t)K * 512, pad);
        for (size_t r = r0; r < r1; r++) {
          rowid[r] = order[r];
          const auto &rw = rows[order[r]];
          for (size_t k = 0; k < rw.size(); k++) ell[h->sp_ell_off[p] + k * 512 + (r - r0)] = rw[k];
        }
      }
      h->sp_npass = npass;
      if ((rc = h->sp_ell.upload(ell)) || (rc = h->sp_rowid.upload(rowid))) return rc;
    }
  }
  if ((rc = h->x_out.alloc((size_t)B * n)) || (rc = h->y_out.alloc((size_t)B * std::max<int64_t>(m, 1)))) return rc;
  if ((rc = h->rawP.alloc((size_t)B * std::max(an.Pp[n], 1))) || (rc = h->rawq.alloc((size_t)B * n))) return rc;
  if ((rc = h->x_out.zero(h->stream)) || (rc = h->y_out.zero(h->stream))) return rc;
  HIPCHK(hostpool::alloc((void **)&h->h_iscal, (size_t)IS_COUNT * T * sizeof(int), &h->h_iscal_cap));
  HIPCHK(hostpool::alloc((void **)&h->h_dscal, (size_t)DS_COUNT * T * sizeof(double), &h->h_dscal_cap));
  if ((rc = ensure_pin(h, std::max((size_t)2 * B * m, (size_t)B * n)))) return rc;      // (host update paths: bounds, warm starts)
  HIPCHK(hipStreamSynchronize(h->stream));
  double t1 = now_s();
  // ---- per-QP numeric (host threads), uploaded in chunks to bound host memory
  h->qp.resize(B);
  int nnzPin = (int)Pp[n], nnzA = (int)Ap[n];
  const int CH = 128;
  double t_factor = 0.0, t_upload = 0.0;
  // Row E2 (Ruiz equilibration) of a batch runs on the device (ruiz_kernel from the raw data: bit for bit what scale_qp
  // computes), like every later update of the handle; a handful of large QPs keeps the host threads (host_ruiz()).  The
  // host mirrors of the scaled problem are then fetched only if a host path ever needs them (ensure_mirrors).
  const bool dev_ruiz = !host_ruiz(h);
  for (int c0 = 0; c0 < B; c0 += CH) {
    int c1 = (int)std::min<int64_t>(B, c0 + CH);
    double ta = now_s();
    // QPs whose P, A, q equal those of the chunk's first QP (GOMP: all of them, only bounds differ) reuse its
    // equilibration.  The numeric factorisation itself happens on the device, below.
    std::vector<char> dup(c1 - c0, 0);
    std::vector<double> raw_pq((size_t)(c1 - c0) * (an.Pp[n] + n));
    for (int k = 1; k < c1 - c0 && !dev_ruiz; k++) {
      const int qi = c0 + k;
      dup[k] = !memcmp(Pv + (size_t)qi * nnzPin, Pv + (size_t)c0 * nnzPin, sizeof(double) * nnzPin) &&
               !memcmp(Av + (size_t)qi * nnzA, Av + (size_t)c0 * nnzA, sizeof(double) * nnzA) &&
               (!q || !memcmp(q + (size_t)qi * n, q + (size_t)c0 * n, sizeof(double) * n));
    }
    auto numeric = [&](int k, bool second_pass) {
      if ((bool)dup[k] != second_pass) return;
      int qi = c0 + k;
      QPNumeric &Q = h->qp[qi];
      load_qp(an, h->st, Pv + (size_t)qi * nnzPin, q ? q + (size_t)qi * n : nullptr, Av + (size_t)qi * nnzA,
              l + (size_t)qi * m, u + (size_t)qi * m, Q);
      std::copy(Q.Pv.begin(), Q.Pv.end(), raw_pq.begin() + (size_t)k * (an.Pp[n] + n));          // (before the equilibration)
      std::copy(Q.q.begin(), Q.q.end(), raw_pq.begin() + (size_t)k * (an.Pp[n] + n) + an.Pp[n]);
      if (!dev_ruiz && h->st.scaling) { if (second_pass) scale_like(an, h->qp[c0], Q); else scale_qp(an, h->st, Q); }
      set_rho_vec(an, h->st, Q);          // (device equilibration: sized here, refreshed with the mirrors)
    };
    parallel_for(c1 - c0, [&](int k, int) { numeric(k, false); });
    parallel_for(c1 - c0, [&](int k, int) { numeric(k, true); });
    double tb = now_s();
    t_factor += tb - ta;
    std::vector<int> ids(c1 - c0);
    for (int k = 0; k < c1 - c0; k++) ids[k] = c0 + k;
    if (!dev_ruiz && ((rc = upload_rho(h, ids)) || (rc = upload_problem(h, ids, true)))) return rc;
    {     // raw triu(P) and q of the chunk, QP-major (mi_osqp_batch_reinit_some; the device equilibration below)
      const size_t per = (size_t)an.Pp[n] + n;
      if ((rc = ensure_stage(h, raw_pq.size() + 1, 0))) return rc;
      HIPCHK(hipMemcpyAsync(h->stage.p, raw_pq.data(), raw_pq.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
      if (an.Pp[n]) HIPCHK(hipMemcpy2DAsync(h->rawP.p + (size_t)c0 * an.Pp[n], (size_t)an.Pp[n] * 8, h->stage.p, per * 8, (size_t)an.Pp[n] * 8, (size_t)(c1 - c0), hipMemcpyDeviceToDevice, h->stream));
      HIPCHK(hipMemcpy2DAsync(h->rawq.p + (size_t)c0 * n, (size_t)n * 8, h->stage.p + an.Pp[n], per * 8, (size_t)n * 8, (size_t)(c1 - c0), hipMemcpyDeviceToDevice, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    }
    t_upload += now_s() - tb;
  }
  {
    std::vector<int> all(B);
    for (int i = 0; i < B; i++) all[i] = i;
    size_t cnt = (size_t)DS_COUNT * T;
    for (size_t k = 0; k < cnt; k++) h->h_dscal[k] = 0.0;
    HIPCHK(hipMemcpy(h->dscal.p, h->h_dscal, cnt * sizeof(double), hipMemcpyHostToDevice));
    if ((rc = sync_scalars_to_device(h, all, !dev_ruiz))) return rc;          // (c and 1/c belong to the equilibration)
  }
  if (dev_ruiz) {
    double tb = now_s();
    // raw A and bounds as the caller gave them (QP-major) -> ruiz_kernel in its "fresh" form -> scaled P, A, q, bounds, D, E, c
    // in the handle's layout + the scaled values once more QP-major for the check streams
    const size_t cA = (size_t)B * nnzA, cb = (size_t)B * m, cpa = (size_t)B * (an.Pp[n] + nnzA);
    if ((rc = ensure_pin(h, cA + 2 * cb + 1)) || (rc = ensure_stage(h, cA + 2 * cb + cpa + 1, (size_t)B))) return rc;
    par_copy(h->pin, Av, cA);
    if (m) { par_copy(h->pin + cA, l, cb); par_copy(h->pin + cA + cb, u, cb); }
    HIPCHK(hipMemcpyAsync(h->stage.p, h->pin, (cA + 2 * cb) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RuizArgs r{};
    r.n = (int)n; r.m = (int)m; r.nnzP = an.Pp[n]; r.nnzA = nnzA; r.B = (int)B; r.BT = BT; r.iters = (int)h->st.scaling;
    r.ids = nullptr; r.fresh = 1; r.rawP = h->rawP.p; r.rawq = h->rawq.p;
    r.Prow = h->rz_prow.p; r.Pcol = h->rz_pcol.p; r.Arow = h->rz_arow.p; r.Acol = h->rz_acol.p;
    r.rawA = h->stage.p; r.rawl = h->stage.p + cA; r.rawu = h->stage.p + cA + cb;
    r.pa_val = h->pa_val.p; r.q = h->q.p; r.Dsc = h->Dsc.p; r.Dsc_inv = h->Dsc_inv.p; r.Esc = h->Esc.p; r.Esc_inv = h->Esc_inv.p;
    r.l = h->l.p; r.u = h->u.p; r.dscal = h->dscal.p;
    r.dn = h->out1.p; r.en = h->out1.p + (size_t)B * n;
    r.pa_out = h->stage.p + cA + 2 * cb;
    HIPCHK(launch_ruiz(r, h->stream));
    std::vector<int> ids(B);
    for (int i = 0; i < (int)B; i++) ids[i] = i;
    HIPCHK(hipMemcpyAsync(h->ids.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(launch_scatter(r.pa_out, h->chk_val.p, h->chk.src.p, h->ids.p, (int)B, an.Pp[n] + nnzA, h->chk.view(an.chk), BT, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->host_scaling_stale = true; h->host_bounds_stale = true; h->host_rho_stale = true;
    t_factor += now_s() - tb;
  }
  // E5 numeric on the device: KKT assembly + block LDL' + inverted diagonal blocks + scatter into the solve
  // streams of every QP (the kernel every later rho / A update uses); a wrong inertia comes back as an error
  {
    double tb = now_s();
    std::vector<int> all(B);
    for (int i = 0; i < (int)B; i++) all[i] = i;
    if ((rc = refactor_qps(h, std::move(all)))) return rc;
    t_factor += now_s() - tb;
  }
  if ((rc = reset_solve_state(h, true))) return rc;
  if ((rc = snapshot(h))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  mi_osqp_stats &s = h->stats;
  s.n = n; s.m = m; s.N = an.N; s.batch = B; s.tile = BT; s.n_tiles = h->ntiles;
  s.nnz_P_triu = an.Pp[n]; s.nnz_A = nnzA; s.nnz_KKT = an.nnzK(); s.nnz_L = an.nnzL();
  s.n_supernodes = (int64_t)an.sn_start.size() - 1; s.n_blocks = (int64_t)an.chunk_start.size() - 1;
  s.fwd_levels = an.fwd.n_phases; s.bwd_levels = an.bwd.n_phases;
  s.fwd_slots = (int64_t)an.fwd.phys_steps() * 64; s.bwd_slots = (int64_t)an.bwd.phys_steps() * 64; s.chk_slots = (int64_t)an.chk.phys_steps() * 64;
  s.lds_bytes = (int64_t)h->lds; s.threads_per_block = h->threads;
  s.dense_tail_rows = an.dt.k; s.dense_tail_slots = (int64_t)an.dt.n_steps * 64;
  s.setup_seconds_host = t1 - t0; s.setup_seconds_factor = t_factor; s.setup_seconds_upload = t_upload;
  s.nnz_L_before_tail = an.dt.k ? an.Lp[an.dt.s] : an.nnzL();
  s.solve_groups = h->mw_groups; s.solve_group_threads = h->mw_groups > 0 ? h->mw_threads : 0;
  if (getenv("MI_OSQP_DEBUG_TIMING"))
    fprintf(stderr, "[mi_osqp] setup B=%d N=%d: analysis+alloc %.1f ms (analysis %.1f), numeric %.1f ms, upload %.1f ms, rest %.1f ms\n", (int)B, an.N,
            1e3 * (t1 - t0), 1e3 * t_analysis, 1e3 * t_factor, 1e3 * t_upload, 1e3 * (now_s() - t1 - t_factor - t_upload));
  return MI_OSQP_OK;
}

// ------------------------------------------------------------------- solve

// Per-QP failure isolation: the slots of `bad` (slot = tile * BT + b; outside a solve slot == QP) become kNonConvex on
// the device (fail_slots_kernel) and are remembered in h->failed.  qp_of_slot: the slot -> QP table of a solve in flight.
static int fail_slots(mi_osqp_batch *h, KernelArgs a, const std::vector<int> &bad, const std::vector<int> *qp_of_slot, int iter) {
  if (bad.empty()) return 0;
  int rc;
  if (h->fail_list.n < bad.size() && (rc = h->fail_list.alloc(std::max<size_t>(bad.size(), (size_t)h->ntiles * h->BT)))) return rc;
  if (!qp_of_slot) a.qp_of_slot = nullptr;                   // identity layout
  HIPCHK(hipMemcpyAsync(h->fail_list.p, bad.data(), bad.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(launch_fail_slots(a, h->fail_list.p, (int)bad.size(), h->BT, iter, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int s : bad) { const int q = qp_of_slot ? (*qp_of_slot)[s] : s; if (q >= 0 && q < h->B) h->failed[q] = 1; }
  return 0;
}

// Row E13 on the device for a list of slots (tile * BT + b): rho vector from the current bounds and rho, KKT
// assembly, block LDL', scatter into the solve streams.  The work list packs the slots kbt per workgroup.
// Slots whose new factor has the wrong inertia are appended to *bad (the caller isolates them).
static int device_refactor_slots(mi_osqp_batch *h, std::vector<int> work, std::vector<int> *bad) {
  if (work.empty()) return 0;
  const int BT = h->BT;
  int rc;
  FactorArgs fa = make_factor_args(h, 0);
  // QPs per workgroup of this refactorisation: one while every QP can have a CU of its own (a lone tile is
  // latency-bound: 3.3 ms with one QP, 4.6 ms with two), the solve tiling otherwise (measured: 605 QPs take
  // 10.9 ms whether packed 1, 2 or 4 per workgroup - the memory system, not the tiling, is the limit there)
  const int nq = (int)work.size();
  int kbt = nq <= h->n_cus ? 1 : BT;
  { const char *e = getenv("MI_OSQP_FACTOR_BT"); if (e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4)) kbt = atoi(e); }
  const int wtiles = (nq + kbt - 1) / kbt;
  work.resize((size_t)wtiles * kbt, -1);
  if (h->work.n < work.size() && (rc = h->work.alloc((size_t)h->ntiles * BT + 4))) return rc;
  HIPCHK(hipMemcpyAsync(h->work.p, work.data(), work.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  fa.work = h->work.p;
  // Short work lists (the rho updates of the last few QPs of a batch, a strong-scaling shard, a handle with one QP): the
  // block tasks of a QP are shared by several workgroups with barriers of the group between the phases of a level
  // (3 x levels x ~6 us): one QP of config 5 (ms): 1: 275, 8: 45, 16: 28, 32: 19, 64: 15, 128: 14
  // (measured in round 3: the LDS-resident single-workgroup form of factor_kernel does NOT beat the group of workgroups on a
  //  lone QP - 0.88 against 0.43 ms at config 2: a lone QP's levels hold hundreds of block tasks, the kernel is short of
  //  waves, not of memory latency - so short lists keep their groups and the LDS form serves one-workgroup-per-QP launches)
  if (kbt == 1 && h->mw_bar.p && (*h->anp).N >= 1000) {
    const char *eg = getenv("MI_OSQP_FACTOR_GROUPS");
    const int cap = h->B == 1 ? std::max(4, std::min(64, (*h->anp).N / 600)) : 8;
    int G = eg ? std::max(1, std::min(256, atoi(eg))) : cap;
    G = std::min(G, std::max(1, h->n_cus / wtiles));
    static const int resident = max_coresident_factor_groups(factor_threads(), 1);      // workgroups of factor_kernel per CU
    if (resident > 0) G = std::min(G, std::max(1, resident * h->n_cus / wtiles));
    fa.mw_groups = G > 1 ? G : 0;
  }
  std::unique_lock<std::mutex> spin_lock(spin_mutex(h->device), std::defer_lock);
  if (fa.mw_groups > 1) spin_lock.lock();
  HIPCHK(hipEventRecord(h->evf0, h->stream));
  if (fa.mw_groups > 1) HIPCHK(hipMemsetAsync(h->mw_bar.p, 0, 4 * sizeof(uint32_t) * (size_t)wtiles, h->stream));
  HIPCHK(launch_factor(fa, kbt, wtiles, factor_threads(), h->stream));
  HIPCHK(hipEventRecord(h->evf1, h->stream));
  if ((*h->anp).dt.k) {      // the tail blocks now hold the Schur complement: invert it into the stream of the symmetric product
    const DenseTail &dt = (*h->anp).dt;
    TailArgs da{};
    da.n = (*h->anp).n; da.N = (*h->anp).N; da.s = dt.s; da.k = dt.k; da.kbt = kbt; da.home_bt = BT;
    da.storage = (*h->anp).bf.storage; da.n_slots = dt.n_steps * 64u; da.n_lt = dt.n_lt; da.n_ltcol = dt.n_ltcol; da.n_quads = (uint32_t)(dt.asm_q64.size() / 64);
    da.nh = h->dt_nh; da.cs_doubles = h->dt_cs_doubles; da.work = h->work.p;
    da.lt_pos = h->dt_lt_pos.p; da.ltcol_col = h->dt_ltcol_col.p; da.tile_tab = h->dt_tile_tab.p; da.wave_tiles = h->dt_wave_tiles.p;
    da.dt_task = h->dt_task.p; da.dt_task_step = h->dt_task_step.p; da.n_tasks = (uint32_t)(dt.task.size() / 4);
    da.asm_q64 = h->dt_asm_q64.p; da.diag_tile = h->dt_diag_tile.p; da.src_tile = h->dt_src_tile.p;
    da.Lblk = h->Lblk.p; da.Dl = h->Dl.p; da.Sd = h->dt_Sd.p; da.dt_val = h->dt_val.p; da.dinv = h->dinv.p; da.npos = h->npos.p; da.iscal = h->iscal.p;
    unsigned long long *d_trace = nullptr;
    const bool tracing = getenv("MI_OSQP_TAIL_TRACE") != nullptr;          // timing stamps only; results are unaffected
    if (tracing) { HIPCHK(hipMalloc((void **)&d_trace, (size_t)wtiles * kbt * 8 * sizeof(unsigned long long))); HIPCHK(hipMemsetAsync(d_trace, 0, (size_t)wtiles * kbt * 64, h->stream)); }
    da.trace = d_trace;
    HIPCHK(launch_tail(da, wtiles * kbt, h->dt_lds_asm, h->dt_lds, h->stream));
    if (tracing) {
      std::vector<unsigned long long> tr((size_t)wtiles * kbt * 8);
      HIPCHK(hipStreamSynchronize(h->stream));
      HIPCHK(hipMemcpy(tr.data(), d_trace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      (void)hipFree(d_trace);
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t i = 0; i < tr.size(); i++) sum[i % 8] += (double)tr[i] / (wtiles * kbt);
      fprintf(stderr, "[mi_osqp] tail_kernel, %d QPs, mean shader clocks of wave 0 per QP: assembly %.0f, pivot blocks %.0f, panel stores %.0f, stream write %.0f; "
              "staging %.0f, Gn %.0f, trailing %.0f, barrier waits of the panel phase %.0f\n", wtiles * kbt, sum[0], sum[1], sum[2], sum[3], sum[4], sum[5], sum[6], sum[7]);
    }
  }
  HIPCHK(hipEventRecord(h->evf2, h->stream));
  HIPCHK(hipMemcpyAsync(h->h_iscal, h->iscal.p, (size_t)h->ntiles * IS_COUNT * BT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (spin_lock.owns_lock()) spin_lock.unlock();
  if (fa.mw_groups > 1) {
    std::vector<uint32_t> w(4 * (size_t)wtiles, 0u);
    HIPCHK(hipMemcpy(w.data(), h->mw_bar.p, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int t = 0; t < wtiles; t++)
      if (w[4 * (size_t)t + 2]) {
        // a workgroup of a group never showed up: the streams of the listed QPs hold a half-written factor.  Every listed
        // QP is without a valid factor (kNonConvex on every solve) until a later refactorisation of it succeeds.
        for (int sl : work) if (sl >= 0 && sl < h->B) h->failed[(size_t)sl] = 1;
        g_last_error = "refactorisation on several workgroups: a barrier of the group timed out"; return MI_OSQP_ERR_DEVICE;
      }
  }
  {
    float f = 0.f, d = 0.f;
    HIPCHK(hipEventElapsedTime(&f, h->evf0, h->evf1)); HIPCHK(hipEventElapsedTime(&d, h->evf1, h->evf2));
    h->factor_ms_sum += f; h->dense_ms_sum += (*h->anp).dt.k ? d : 0.0; h->refactor_launches++; h->refactor_qps += nq;
    if (nq >= h->peak_qps) { h->peak_qps = nq; h->peak_factor_ms = f; h->peak_tail_ms = (*h->anp).dt.k ? d : 0.0; }
  }
  for (int s : work)            // (only the listed slots: at setup the flags of padding slots are not initialised yet)
    if (s >= 0 && h->h_iscal[(size_t)(s / BT) * IS_COUNT * BT + IS_NEED_REFACTOR * BT + s % BT] < 0) {
      if (getenv("MI_OSQP_DEBUG_TIMING")) {
        (void)hipMemcpy(h->h_npos, h->npos.p, (size_t)h->ntiles * BT * sizeof(int), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mi_osqp] slot %d: %d positive pivots, expected %d\n", s, h->h_npos[s], (*h->anp).n);
      }
      if (bad) bad->push_back(s);
    }
  return 0;
}

// Refactorisation outside a solve (setup, update_A, constraint-type changes; slot == QP): QPs whose factor loses its
// inertia are isolated (kNonConvex on every solve until a later refactorisation of theirs succeeds); the call fails
// only when that happens to EVERY QP of the handle (a single QP: OsqpSolver::Init / the update reports the error).
static int refactor_qps(mi_osqp_batch *h, std::vector<int> qps) {
  if (qps.empty()) return 0;
  std::vector<int> bad;
  const std::vector<int> listed = qps;
  int rc = device_refactor_slots(h, std::move(qps), &bad);
  if (rc) return rc;
  for (int q : listed) h->failed[q] = 0;
  if ((rc = fail_slots(h, make_args(h), bad, nullptr, 0))) return rc;
  if ((int)bad.size() == h->B) { g_last_error = "the KKT factor lost its inertia"; return MI_OSQP_ERR_NONCONVEX; }
  return 0;
}

// exchange the complete device state of slot pairs (slot = tile*BT + b)
static int apply_swaps(mi_osqp_batch *h, const std::vector<int2> &pairs) {
  if (pairs.empty()) return 0;
  const Analysis &an = (*h->anp);
  int np = (int)pairs.size(), BT = h->BT, n = an.n, m = an.m, rc;
  if (h->pairs.n < pairs.size() && (rc = h->pairs.alloc(std::max<size_t>(pairs.size(), (size_t)h->ntiles * BT)))) return rc;
  HIPCHK(hipMemcpyAsync(h->pairs.p, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
  hipStream_t st = h->stream;
  HIPCHK(launch_swap_sched(h->fwd_val.p, h->pairs.p, np, h->fwd.view(an.fwd), BT, st));
  HIPCHK(launch_swap_sched(h->bwd_val.p, h->pairs.p, np, h->bwd.view(an.bwd), BT, st));
  HIPCHK(launch_swap_sched(h->chk_val.p, h->pairs.p, np, h->chk.view(an.chk), BT, st));
  struct PL { double *p; int len; };
  const PL plain[] = {{h->dinv.p, an.N}, {h->x.p, n}, {h->z.p, m}, {h->y.p, m}, {h->q.p, n}, {h->l.p, m}, {h->u.p, m},
                      {h->rho_vec.p, m}, {h->rho_inv.p, m}, {h->Dsc.p, n}, {h->Dsc_inv.p, n}, {h->Esc.p, m}, {h->Esc_inv.p, m},
                      {h->pa_val.p, an.Pp[n] + an.Ap[n]}, {h->dscal.p, DS_COUNT}};
  for (const PL &a : plain) HIPCHK(launch_swap_plain(a.p, h->pairs.p, np, a.len, BT, st));
  HIPCHK(launch_swap_int(h->iscal.p, h->pairs.p, np, IS_COUNT, BT, st));
  return 0;
}

// The ADMM loop runs in segments that end at every termination-check / rho-update
// point: iterate_kernel (E6-E10) -> check_kernel (E11-E14) -> host reads the flags,
// (optionally) compacts the QPs still iterating into the leading tiles (slot swaps on the device),
// runs the device refactorisation for the QPs whose rho changed, and continues.
// The swaps are undone at the end - also when the loop ends with an error - so outside a solve
// every array is in the identity layout.
static int gcd_i(int a, int b);
static int segment_length(const Settings &S);
static int ensure_advance_buffers(mi_osqp_batch *h);
static int wait_launch_over(mi_osqp_batch *h, unsigned seq);

// The latency regime - no more tiles than CUs, the solve vector in LDS (a lone trajectory QP, a strong-scaling shard, the
// stragglers' world): the host round trip per segment (two launches, a copy of the flags, a stream synchronisation) is a
// tenth to a third of such a solve.  Here ONE advance_kernel launch carries every QP to its end or to its next rho update
// (the QP pauses), the host waits on a word in pinned memory, refactors the paused QPs with the grouped factor_kernel and
// launches again: host round trips = rho-update rounds + 1.  Same arithmetic, same per-QP iteration counts.
// Measured (round 3): NOT a win as it stands - advance_kernel's fused bodies iterate ~10 % slower than iterate_kernel
// (register allocation shared with the check), which eats the saved round trips (config 2: 0.73 against 0.69 ms).  Hence
// opt-in (MI_OSQP_ADVANCE_SOLVE=1; tests/test_gpu_continuous.py runs it for parity).
static bool small_batch_path(const mi_osqp_batch *h) {
  if (!getenv("MI_OSQP_ADVANCE_SOLVE")) return false;
  if (h->global_xs || h->mw_groups > 0 || h->ntiles > h->n_cus || h->BT != 1) return false;      // (tiles of 2 QPs at 16 waves spill in the fused kernel)
  if (getenv("MI_OSQP_COMPACT") && !(*h->anp).dt.k) return false;
  return segment_length(h->st) >= 5;
}
static int solve_small(mi_osqp_batch *h, KernelArgs a) {
  mi_osqp_batch::Cont &c = h->cont;
  const int BT = h->BT, nslots = h->ntiles * BT, L = segment_length(h->st);
  int rc;
  if ((rc = ensure_advance_buffers(h))) return rc;
  HIPCHK(hipMemsetAsync(c.counter.p, 0, 2 * sizeof(unsigned), h->stream));
  a.info_at_end = 1;
  const int max_segments = (int)((h->st.max_iter + L - 1) / L) + 1;
  std::vector<int> paused;
  for (int round = 0; round < 100000; round++) {
    const unsigned seq = ++c.launch_seq ? c.launch_seq : ++c.launch_seq;       // (never 0: the word's idle value)
    c.h_done[0] = 0;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(launch_advance(a, BT, h->ntiles, h->threads, h->lds, h->stream, max_segments, L, c.h_is[0], c.h_ds[0], nullptr, seq, c.counter.p, c.h_done));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    if ((rc = wait_launch_over(h, seq))) return rc;
    h->last_launches++; h->kernel_launches++;
    paused.clear();
    bool any_active = false;
    for (int sl = 0; sl < nslots && sl < h->B; sl++) {
      const int *t = c.h_is[0] + (size_t)(sl / BT) * IS_COUNT * BT;
      const int b = sl % BT;
      if (t[IS_NEED_REFACTOR * BT + b] == 1) paused.push_back(sl);      // paused - or finished at max_iter on a rho-update iteration
      else if (!t[IS_DONE * BT + b]) any_active = true;       // (cannot happen: a tile leaves only done or paused)
    }
    bool any_paused = false;
    for (int sl : paused) any_paused = any_paused || c.h_is[0][(size_t)(sl / BT) * IS_COUNT * BT + IS_PENDING * BT + sl % BT] == 2;
    if (paused.empty()) { if (any_active) continue; break; }
    // ---- row E13 for the paused QPs: the grouped refactorisation of short lists, then resume (or isolate: kNonConvex)
    const double tr = now_s();
    std::vector<int> bad;
    if ((rc = device_refactor_slots(h, paused, &bad))) return rc;
    HIPCHK(launch_resume_flagged(a, nslots, BT, h->stream));
    for (int sl : bad) h->failed[(size_t)sl] = 1;
    h->host_rho_stale = true;
    h->last_refactors += (int64_t)paused.size();
    h->last_refactor_s += now_s() - tr;
    if (!any_paused && !any_active) break;        // (only finished QPs were refactored: nothing is left to iterate)
  }
  {     // device time of the launches of this solve (events around the last launch; earlier ones through the stream order)
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) { h->last_device_s += ms * 1e-3; h->kernel_ms_sum += ms; }
  }
  // the final flags are in the pinned image already
  memcpy(h->h_iscal, c.h_is[0], (size_t)h->ntiles * IS_COUNT * BT * sizeof(int));
  for (int qi = 0; qi < h->B; qi++)
    h->last_total_iters += h->h_iscal[(size_t)(qi / BT) * IS_COUNT * BT + IS_ITER * BT + qi % BT];
  h->solved_once = true;
  return MI_OSQP_OK;
}

static int solve_impl(mi_osqp_batch *h, double *d_x_out, hipStream_t user_stream) {
  hipStream_t keep = h->stream;
  struct Restore { mi_osqp_batch *h; hipStream_t s; ~Restore() { h->stream = s; } } restore{h, keep};
  if (user_stream) h->stream = user_stream;
  int rc;
  if ((rc = reset_solve_state(h, !h->st.warm_start))) return rc;
  KernelArgs a = make_args(h);
  if (d_x_out) a.x_out = d_x_out;
  h->last_total_iters = h->last_launches = h->last_refactors = 0;
  h->last_device_s = h->last_refactor_s = h->last_compact_s = 0.0;
  const int BT = h->BT, nslots = h->ntiles * BT;
  const Settings &S = h->st;
  std::vector<int> qp_of_slot(nslots);
  for (int s = 0; s < nslots; s++) qp_of_slot[s] = s < h->B ? s : -1;
  HIPCHK(hipMemcpyAsync(h->qp_of_slot.p, qp_of_slot.data(), nslots * sizeof(int), hipMemcpyHostToDevice, h->stream));
  {     // QPs without a valid factor (isolated earlier) are kNonConvex from the start
    std::vector<int> bad;
    for (int q = 0; q < h->B; q++) if (h->failed[q]) bad.push_back(q);
    if ((rc = fail_slots(h, a, bad, &qp_of_slot, 0))) return rc;
  }
  if (small_batch_path(h)) return solve_small(h, a);
  std::vector<std::vector<int2>> rounds;
  // compaction (re-pairing the QPs still iterating into fewer tiles) is implemented and tested but OFF by default:
  // since the value streams are per QP, a finished QP costs no bytes anyway, and moving data only breaks even
  const bool no_compact = getenv("MI_OSQP_COMPACT") == nullptr || (*h->anp).dt.k != 0;      // (and not combined with the dense tail)
  if (!no_compact) { const int rc_m = materialise_working(h); if (rc_m) return rc_m; }      // (its slot swaps move working streams)
  auto loop = [&]() -> int {
    int iter = 0, ntl = h->ntiles;     // tiles [0, ntl) hold every QP that is still iterating
    while (true) {
      int seg_end = (int)S.max_iter;
      if (S.check_termination > 0) seg_end = std::min<int64_t>(seg_end, (iter / S.check_termination + 1) * S.check_termination);
      if (S.adaptive_rho && S.adaptive_rho_interval > 0)
        seg_end = std::min<int64_t>(seg_end, (iter / S.adaptive_rho_interval + 1) * S.adaptive_rho_interval);
      a.iter_begin = iter; a.iter_end = seg_end; a.info_at_end = 1;
      {
        std::unique_lock<std::mutex> spin_lock(spin_mutex(h->device), std::defer_lock);
        if (h->mw_groups > 0) spin_lock.lock();
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        if (h->mw_groups > 0) HIPCHK(hipMemsetAsync(h->mw_bar.p, 0, 4 * sizeof(uint32_t), h->stream));
        HIPCHK(launch_iterate(a, BT, ntl, h->mw_groups > 0 ? h->mw_threads : h->threads, h->lds, h->stream));
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        {
          // The check of more 16-wave tiles than the CUs hold at once (two each) runs in 8-wave workgroups - four per CU, one round
          // instead of two; the check schedule is walked stream by stream by however many waves there are, row by row in the
          // same order (headline batch: 0.19 -> 0.11 ms per check, 31.8 -> 31.5 ms per step).  MI_OSQP_CHECK_THREADS forces.
          int chk_threads = h->mw_groups > 0 ? h->mw_threads : h->threads;
          if (h->mw_groups <= 0 && h->threads == 1024 && ntl > 2 * h->n_cus) chk_threads = 512;
          if (getenv("MI_OSQP_CHECK_THREADS") && h->mw_groups <= 0) chk_threads = std::max(64, std::min(h->threads, atoi(getenv("MI_OSQP_CHECK_THREADS")) / 64 * 64));
          HIPCHK(launch_check(a, BT, ntl, chk_threads, h->lds, h->stream));
        }
        HIPCHK(hipMemcpyAsync(h->h_iscal, h->iscal.p, (size_t)ntl * IS_COUNT * BT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
      }
      if ((rc = mw_barrier_ok(h))) {
        // waves that gave up waiting have consumed "not yet" patterns: x, y, z (and with them every later warm start) are
        // garbage.  Back to the last good state: the factor / rho of the snapshot, cold iterates.
        const std::string why = g_last_error;
        (void)restore_snapshot(h);
        g_last_error = why + " (the handle is back in the state of its last setup / update, cold-started)";
        return rc;
      }
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
      h->last_device_s += ms * 1e-3; h->kernel_ms_sum += ms; h->kernel_launches++; h->last_launches++;
      iter = seg_end;
      // slots still iterating / asking for a refactorisation.  A QP that runs into max_iter at a rho-update iteration
      // has finished AND asks for its refactorisation: upstream adapts rho (and refactors) before it leaves the loop,
      // and the next Solve() of a warm-started solver continues from that factor.
      std::vector<int> active, work;
      for (int s = 0; s < ntl * BT; s++) {
        const int *t = h->h_iscal + (size_t)(s / BT) * IS_COUNT * BT;
        if (qp_of_slot[s] < 0) continue;
        if (!t[IS_DONE * BT + s % BT]) active.push_back(s);
        if (t[IS_NEED_REFACTOR * BT + s % BT]) work.push_back(s);
      }
      const int n_ref = (int)work.size();
      if (active.empty() && !n_ref) break;
      // ---- compaction
      int target = ((int)active.size() + BT - 1) / BT;
      bool compacted_now = false;
      const int ntl_before = ntl;
      std::vector<int> bad;             // slots whose refactorisation lost the inertia: isolated below
      if (!no_compact && target < ntl && !active.empty()) {
        compacted_now = true;
        double tc = now_s();
        std::vector<char> is_active(ntl * BT, 0);
        for (int s : active) is_active[s] = 1;
        std::vector<int2> pairs;
        int hole = 0;
        for (int k = (int)active.size() - 1; k >= 0 && active[k] >= target * BT; k--) {
          while (hole < target * BT && is_active[hole]) hole++;
          pairs.push_back(int2{hole, active[k]});
          std::swap(qp_of_slot[hole], qp_of_slot[active[k]]);
          is_active[hole] = 1;
        }
        int rc2 = apply_swaps(h, pairs);
        rounds.push_back(std::move(pairs));              // (recorded first: the caller undoes whatever part was applied)
        if (rc2) return rc2;
        HIPCHK(hipMemcpyAsync(h->qp_of_slot.p, qp_of_slot.data(), nslots * sizeof(int), hipMemcpyHostToDevice, h->stream));
        ntl = target;
        h->last_compact_s += now_s() - tc;
      }
      // ---- row E13 on the device: rho vector, KKT assembly, block LDL', scatter into the schedules
      if (n_ref) {
        double tr = now_s();
        int rc2;
        if (!compacted_now) {
          // work list: the flagged slots of the whole batch (fewer, fuller tiles = fewer rounds over the CUs)
          if ((rc2 = device_refactor_slots(h, std::move(work), &bad))) return rc2;
        } else {
          // after a compaction of this segment the host copy of the flags is stale: flag-driven sweep over the tiles
          FactorArgs fa = make_factor_args(h, 0);
          fa.mw_groups = 0;
          HIPCHK(launch_factor(fa, BT, ntl_before, factor_threads(), h->stream));
          HIPCHK(hipMemcpyAsync(h->h_iscal, h->iscal.p, (size_t)ntl_before * IS_COUNT * BT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
          HIPCHK(hipStreamSynchronize(h->stream));
          for (int s = 0; s < ntl_before * BT; s++)
            if (qp_of_slot[s] >= 0 && h->h_iscal[(size_t)(s / BT) * IS_COUNT * BT + IS_NEED_REFACTOR * BT + s % BT] < 0) bad.push_back(s);
        }
        // a rho update that makes the factor lose its inertia ends THAT QP as kNonConvex ([EXT] osqp_solve: adapt_rho
        // fails -> OSQP_NON_CVX, break); every other QP of the batch goes on
        if ((rc2 = fail_slots(h, a, bad, &qp_of_slot, iter))) return rc2;
        h->host_rho_stale = true;
        h->last_refactors += n_ref;
        h->last_refactor_s += now_s() - tr;
      }
      if (!bad.empty()) {
        // (with a compaction in this segment `active` holds pre-swap slot numbers: rebuild it from the table)
        std::vector<char> isbad(nslots, 0);
        for (int s : bad) isbad[s] = 1;
        if (compacted_now) { active.clear(); for (int s = 0; s < ntl * BT; s++) if (qp_of_slot[s] >= 0 && !isbad[s]) active.push_back(s); }
        else active.erase(std::remove_if(active.begin(), active.end(), [&](int s) { return isbad[s] != 0; }), active.end());
      }
      if (active.empty()) break;
    }
    return MI_OSQP_OK;
  };
  rc = loop();
  // ---- undo the compaction (reverse order; swaps are involutions) - also after an error, so that the handle stays usable
  {
    double tc = now_s();
    for (int r = (int)rounds.size() - 1; r >= 0; r--) { int rc2 = apply_swaps(h, rounds[r]); if (rc2 && !rc) rc = rc2; }
    if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
    HIPCHK(hipMemcpyAsync(h->h_iscal, h->iscal.p, (size_t)nslots * IS_COUNT * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->last_compact_s += now_s() - tc;
  }
  for (int qi = 0; qi < h->B; qi++)
    h->last_total_iters += h->h_iscal[(size_t)(qi / BT) * IS_COUNT * BT + IS_ITER * BT + qi % BT];
  h->solved_once = true;
  return MI_OSQP_OK;
}

// ---------------------------------------------------------------- C entry points

extern "C" {

void mi_osqp_default_settings(mi_osqp_settings *s) {
  if (!s) return;
  s->rho = 0.1; s->sigma = 1e-6; s->scaling = 10; s->adaptive_rho = 1; s->adaptive_rho_interval = 0;
  s->adaptive_rho_tolerance = 5.0; s->max_iter = 4000; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->alpha = 1.6; s->scaled_termination = 0;
  s->check_termination = 25; s->warm_start = 1; s->verbose = 0;
}

const char *mi_osqp_exit_code_name(int64_t c) {
  static const char *names[] = {"kOptimal", "kPrimalInfeasible", "kDualInfeasible", "kOptimalInaccurate",
                                "kPrimalInfeasibleInaccurate", "kDualInfeasibleInaccurate", "kMaxIterations",
                                "kInterrupted", "kTimeLimitReached", "kNonConvex", "kUnknown"};
  return (c >= 0 && c <= 10) ? names[c] : "kUnknown";
}
const char *mi_osqp_error_name(int64_t e) {
  static const char *names[] = {"ok", "invalid data", "invalid settings", "sparsity pattern changed",
                                "non-convex problem / KKT inertia", "device error", "null argument", "allocation / size"};
  return (e >= 0 && e <= 7) ? names[e] : "unknown";
}
const char *mi_osqp_version(void) { return "mi-osqp 0.1 (gfx950)"; }
const char *mi_osqp_last_error(void) { return g_last_error.c_str(); }

static int64_t exit_code_of(int status) {
  switch (status) {
    case 1: return MI_OSQP_EXIT_OPTIMAL;
    case 2: return MI_OSQP_EXIT_OPTIMAL_INACCURATE;
    case -3: return MI_OSQP_EXIT_PRIMAL_INFEASIBLE;
    case 3: return MI_OSQP_EXIT_PRIMAL_INFEASIBLE_INACCURATE;
    case -4: return MI_OSQP_EXIT_DUAL_INFEASIBLE;
    case 4: return MI_OSQP_EXIT_DUAL_INFEASIBLE_INACCURATE;
    case -2: return MI_OSQP_EXIT_MAX_ITERATIONS;
    case -5: return MI_OSQP_EXIT_INTERRUPTED;
    case -6: return MI_OSQP_EXIT_TIME_LIMIT_REACHED;
    case -7: return MI_OSQP_EXIT_NON_CONVEX;
    default: return MI_OSQP_EXIT_UNKNOWN;
  }
}

int mi_osqp_batch_setup(mi_osqp_batch **out, int64_t B, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi,
                        const double *Pv, const double *q, const int64_t *Ap, const int64_t *Ai, const double *Av,
                        const double *l, const double *u, const mi_osqp_settings *settings, int64_t device) {
  CallTimer timer_("batch_setup");
  if (!out) return MI_OSQP_ERR_NULL;
  *out = nullptr;
  if (n <= 0 || m < 0) return MI_OSQP_ERR_INVALID_DATA;
  mi_osqp_batch *h = new (std::nothrow) mi_osqp_batch();
  if (!h) return MI_OSQP_ERR_ALLOC;
  h->st = to_settings(settings);
  int rc = batch_setup_impl(h, B, n, m, Pp, Pi, Pv, q, Ap, Ai, Av, l, u, device);
  if (rc) { delete h; return rc; }
  *out = h;
  return MI_OSQP_OK;
}

// The pattern analysis a later setup of B QPs with this pattern on this device will ask for, computed now into the
// process-wide cache (a planner that knows its coming patterns - the ten horizons of a GOMP run - calls this from spare
// host threads; a setup that arrives while it is still running waits for it instead of computing it twice).
int mi_osqp_prefetch_analysis(int64_t B, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const int64_t *Ap, const int64_t *Ai, int64_t device) {
  CallTimer timer_("prefetch_analysis");
  if (B <= 0 || n <= 0 || m < 0 || !Pp || !Ap || (Pp[n] > 0 && !Pi) || (Ap[n] > 0 && !Ai)) return MI_OSQP_ERR_INVALID_DATA;
  mi_osqp_batch tmp;
  int BT = 1, max_extra = -1;
  std::shared_ptr<const Analysis> an;
  return shape_and_analysis(&tmp, B, n, m, Pp, Pi, Ap, Ai, device, BT, max_extra, an);
}

void mi_osqp_batch_free(mi_osqp_batch *h) { delete h; }
void mi_osqp_release_device_cache(void) { devpool::release_all(); hostpool::release_all(); streampool::release_all(); }

int mi_osqp_batch_solve(mi_osqp_batch *h) {
  CallTimer timer_("batch_solve");
  if (!h) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  const int rc = solve_impl(h, nullptr, nullptr);
  if (CallTimer::on() && getenv("MI_OSQP_DEBUG_SOLVES"))
    fprintf(stderr, "[mi_osqp] solve B=%d N=%d: %ld segments of <= 25 iterations, %.0f QP-iterations in total, iterate %.2f ms, refactor %.2f ms (%ld)\n", h->B,
            (*h->anp).N, (long)h->last_launches, (double)h->last_total_iters, 1e3 * h->last_device_s, 1e3 * h->last_refactor_s, (long)h->last_refactors);
  return rc;
}

int mi_osqp_batch_solve_device(mi_osqp_batch *h, double *d_x_out, int32_t *d_status, int32_t *d_iters, void *stream) {
  if (!h) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  int rc = solve_impl(h, d_x_out, (hipStream_t)stream);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  if (d_x_out) {   // keep the internal copy coherent for get_primal()
    HIPCHK(hipMemcpyAsync(h->x_out.p, d_x_out, (size_t)h->B * (*h->anp).n * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  if (d_status || d_iters) HIPCHK(launch_gather_status(h->iscal.p, d_status, d_iters, h->B, h->BT, s));
  HIPCHK(hipStreamSynchronize(s));
  return MI_OSQP_OK;
}

int mi_osqp_batch_get_primal(mi_osqp_batch *h, double *x) {
  CallTimer timer_("batch_get_primal");
  if (!h || !x) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  HIPCHK(hipMemcpy(x, h->x_out.p, (size_t)h->B * (*h->anp).n * sizeof(double), hipMemcpyDeviceToHost));
  return MI_OSQP_OK;
}
int mi_osqp_batch_get_dual(mi_osqp_batch *h, double *y) {
  if (!h || !y) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  if ((*h->anp).m) HIPCHK(hipMemcpy(y, h->y_out.p, (size_t)h->B * (*h->anp).m * sizeof(double), hipMemcpyDeviceToHost));
  return MI_OSQP_OK;
}

int mi_osqp_batch_get_info(mi_osqp_batch *h, mi_osqp_info *info) {
  CallTimer timer_("batch_get_info");
  if (!h || !info) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  size_t icnt = (size_t)h->ntiles * IS_COUNT * h->BT, dcnt = (size_t)h->ntiles * DS_COUNT * h->BT;
  HIPCHK(hipMemcpy(h->h_iscal, h->iscal.p, icnt * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(h->h_dscal, h->dscal.p, dcnt * sizeof(double), hipMemcpyDeviceToHost));
  const int BT = h->BT;
  for (int q = 0; q < h->B; q++) {
    const int *ti = h->h_iscal + (size_t)(q / BT) * IS_COUNT * BT;
    const double *td = h->h_dscal + (size_t)(q / BT) * DS_COUNT * BT;
    int b = q % BT;
    mi_osqp_info &I = info[q];
    I.iter = ti[IS_ITER * BT + b]; I.status_val = ti[IS_STATUS * BT + b]; I.exit_code = exit_code_of((int)I.status_val);
    I.obj_val = td[DS_OBJ * BT + b]; I.pri_res = td[DS_PRI_RES * BT + b]; I.dua_res = td[DS_DUA_RES * BT + b];
    I.rho_updates = ti[IS_RHO_UPDATES * BT + b]; I.rho_estimate = td[DS_RHO_EST * BT + b]; I.rho = td[DS_RHO * BT + b];
  }
  return MI_OSQP_OK;
}

int mi_osqp_batch_get_stats(mi_osqp_batch *h, mi_osqp_stats *st) {
  if (!h || !st) return MI_OSQP_ERR_NULL;
  *st = h->stats;
  return MI_OSQP_OK;
}

int mi_osqp_batch_get_ordering(mi_osqp_batch *h, int64_t *kkt_perm) {
  if (!h || !kkt_perm) return MI_OSQP_ERR_NULL;
  const Analysis &an = (*h->anp);
  for (int k = 0; k < an.N; k++) kkt_perm[k] = an.perm[(size_t)k];
  return MI_OSQP_OK;
}

int mi_osqp_batch_last_solve_stats(mi_osqp_batch *h, int64_t *total_iters, int64_t *kernel_launches, double *device_seconds,
                                   double *refactor_seconds, int64_t *refactor_count, double *compact_seconds) {
  if (h && compact_seconds) *compact_seconds = h->last_compact_s;
  if (!h) return MI_OSQP_ERR_NULL;
  if (total_iters) *total_iters = h->last_total_iters;
  if (kernel_launches) *kernel_launches = h->last_launches;
  if (device_seconds) *device_seconds = h->last_device_s;
  if (refactor_seconds) *refactor_seconds = h->last_refactor_s;
  if (refactor_count) *refactor_count = h->last_refactors;
  return MI_OSQP_OK;
}

int mi_osqp_batch_kernel_time(mi_osqp_batch *h, double *avg_ms, int64_t *launches) {
  if (!h) return MI_OSQP_ERR_NULL;
  if (avg_ms) *avg_ms = h->kernel_launches ? h->kernel_ms_sum / (double)h->kernel_launches : 0.0;
  if (launches) *launches = h->kernel_launches;
  h->kernel_ms_sum = 0.0; h->kernel_launches = 0;
  return MI_OSQP_OK;
}

int mi_osqp_batch_refactor_time(mi_osqp_batch *h, double *factor_ms, double *dense_ms, int64_t *launches, int64_t *qps) {
  if (!h) return MI_OSQP_ERR_NULL;
  if (factor_ms) *factor_ms = h->factor_ms_sum;
  if (dense_ms) *dense_ms = h->dense_ms_sum;
  if (launches) *launches = h->refactor_launches;
  if (qps) *qps = h->refactor_qps;
  h->factor_ms_sum = h->dense_ms_sum = 0.0; h->refactor_launches = h->refactor_qps = 0;
  h->peak_qps = 0; h->peak_factor_ms = h->peak_tail_ms = 0.0;
  return MI_OSQP_OK;
}

int mi_osqp_batch_refactor_peak(mi_osqp_batch *h, int64_t *qps, double *factor_ms, double *tail_ms) {
  if (!h) return MI_OSQP_ERR_NULL;
  if (qps) *qps = h->peak_qps;
  if (factor_ms) *factor_ms = h->peak_factor_ms;
  if (tail_ms) *tail_ms = h->peak_tail_ms;
  return MI_OSQP_OK;
}

int mi_osqp_batch_reset(mi_osqp_batch *h) {
  CallTimer timer_("batch_reset");
  if (!h) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  return restore_snapshot(h);
}
}  // extern "C"
// the state right after setup / the last update that refactored: factor, rho vectors and scalars of the snapshot, cold iterates
// every QP's current factor into the working copy (what the slot swaps of the compaction path move around)
static int materialise_working(mi_osqp_batch *h) {
  const Analysis &an = (*h->anp);
  const int nslots = h->ntiles * h->BT;
  if (!h->fwd_val0.p || !h->use_work.p) return MI_OSQP_OK;
  HIPCHK(launch_copy_flagged_streams(h->fwd_val.p, h->fwd_val0.p, h->use_work.p, 0, nslots, (size_t)an.fwd.phys_steps() * 64, h->stream));
  HIPCHK(launch_copy_flagged_streams(h->bwd_val.p, h->bwd_val0.p, h->use_work.p, 0, nslots, (size_t)an.bwd.phys_steps() * 64, h->stream));
  if (an.dt.k && h->dt_val0.p) HIPCHK(launch_copy_flagged_streams(h->dt_val.p, h->dt_val0.p, h->use_work.p, 0, nslots, (size_t)an.dt.n_steps * 64, h->stream));
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->use_work.p, 1, (size_t)nslots, h->stream));
  return MI_OSQP_OK;
}
static int restore_snapshot(mi_osqp_batch *h) {
  h->clear_rho_updates = true;
  auto cp = [&](DevBuf<double> &dst, DevBuf<double> &src) -> int {
    if (src.n) HIPCHK(hipMemcpyAsync(dst.p, src.p, src.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return 0;
  };
  int rc;
  const size_t nflags = (size_t)h->ntiles * h->BT;
  if (getenv("MI_OSQP_RESET_COPIES")) {
    // (experiments: everything back into the working copy, as before round 3)
    if ((rc = cp(h->fwd_val, h->fwd_val0)) || (rc = cp(h->bwd_val, h->bwd_val0)) || (rc = cp(h->dt_val, h->dt_val0))) return rc;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->use_work.p, 1, nflags, h->stream));
  } else {
    // every QP's factor is its snapshot again: one word per QP instead of 1 GB of stream copies at the headline batch
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->use_work.p, 0, nflags, h->stream));
  }
  if ((rc = cp(h->dinv, h->dinv0)) || (rc = cp(h->rho_vec, h->rho_vec0)) || (rc = cp(h->rho_inv, h->rho_inv0)) || (rc = cp(h->dscal, h->dscal0))) return rc;
  if ((rc = reset_solve_state(h, true))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  // the host mirrors of rho follow the snapshot lazily (sync_rho_to_host, only the host update paths need them)
  h->host_rho_stale = true;
  return MI_OSQP_OK;
}
extern "C" {

int mi_osqp_batch_warm_start_x(mi_osqp_batch *h, const double *x) {
  CallTimer timer_("batch_warm_start_x");
  if (!h || !x) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  h->st.warm_start = 1;
  size_t cnt = (size_t)h->B * (*h->anp).n;
  int rc;
  if ((rc = ensure_stage(h, cnt, 0)) || (rc = ensure_pin(h, cnt))) return rc;
  par_copy(h->pin, x, cnt);
  HIPCHK(hipMemcpyAsync(h->stage.p, h->pin, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  KernelArgs a = make_args(h);
  HIPCHK(launch_warm_start(a, h->BT, h->ntiles, h->threads, h->lds, h->stream, h->stage.p));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MI_OSQP_OK;
}

static int update_bounds_on_host(mi_osqp_batch *h, const double *l, const double *u);
static int ensure_mirrors(mi_osqp_batch *h);
static int update_bounds_on_device(mi_osqp_batch *h, const double *d_l, const double *d_u, hipStream_t s, const double *h_l, const double *h_u);

int mi_osqp_batch_update_bounds(mi_osqp_batch *h, const double *l, const double *u) {
  CallTimer timer_("batch_update_bounds");
  if (!h || !l || !u) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  // through pinned memory to the device, where the rows are scaled and checked (bounds_kernel); only when a row changes
  // its type (equality / inequality / free: new rho vector, new factor) the host mirrors take over
  const size_t cnt = (size_t)h->B * (*h->anp).m;
  if (!cnt) return MI_OSQP_OK;
  int rc;
  if ((rc = ensure_stage(h, 2 * cnt, 0)) || (rc = ensure_pin(h, 2 * cnt))) return rc;
  {
    // in slices of 4 MB: the transfer of one runs while the host threads copy the next (as in device_update)
    const size_t slice = (size_t)1 << 19;
    for (int part = 0; part < 2; part++) {
      const double *src = part ? u : l;
      for (size_t o = 0; o < cnt; o += slice) {
        const size_t len = std::min(slice, cnt - o), off = (size_t)part * cnt + o;
        par_copy(h->pin + off, src + o, len);
        HIPCHK(hipMemcpyAsync(h->stage.p + off, h->pin + off, len * sizeof(double), hipMemcpyHostToDevice, h->stream));
      }
    }
  }
  return update_bounds_on_device(h, h->stage.p, h->stage.p + cnt, h->stream, l, u);
}

static int update_bounds_on_host(mi_osqp_batch *h, const double *l, const double *u) {
  { int rc0 = ensure_mirrors(h); if (rc0) return rc0; }
  h->clear_rho_updates = true;
  const Analysis &an = (*h->anp);
  int m = an.m, B = h->B, rc;
  for (size_t k = 0; k < (size_t)B * m; k++) if (l[k] > u[k]) return MI_OSQP_ERR_INVALID_DATA;
  if ((rc = sync_rho_to_host(h))) return rc;
  h->host_bounds_stale = false;   // host copy becomes authoritative
  std::vector<int> changed;
  for (int q = 0; q < B; q++) {
    QPNumeric &Q = h->qp[q];
    for (int i = 0; i < m; i++) {
      Q.l[i] = std::max(l[(size_t)q * m + i], -kInfty); Q.u[i] = std::min(u[(size_t)q * m + i], kInfty);
      if (h->st.scaling) { Q.l[i] *= Q.E[i]; Q.u[i] *= Q.E[i]; }
    }
    if (refresh_rho_types(an, Q)) changed.push_back(q);
  }
  std::vector<int> all(B);
  for (int i = 0; i < B; i++) all[i] = i;
  if ((rc = upload_problem(h, all, false))) return rc;
  if ((rc = refactor_qps(h, changed))) return rc;      // constraint types changed: new rho vector + factor on the device
  if (!changed.empty() && (rc = snapshot(h))) return rc;
  return MI_OSQP_OK;
}

int mi_osqp_batch_update_bounds_device(mi_osqp_batch *h, const double *d_l, const double *d_u, void *stream) {
  if (!h || !d_l || !d_u) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  return update_bounds_on_device(h, d_l, d_u, stream ? (hipStream_t)stream : h->stream, nullptr, nullptr);
}

// (h_l / h_u: the same bounds on the host when the caller has them, else they are fetched if the host path is needed)
static int update_bounds_on_device(mi_osqp_batch *h, const double *d_l, const double *d_u, hipStream_t s, const double *h_l, const double *h_u) {
  h->clear_rho_updates = true;
  int m = (*h->anp).m, B = h->B;
  if (!m) return MI_OSQP_OK;
  // pass 1: validate + detect constraint-type changes without writing
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), s));
  HIPCHK(launch_bounds(d_l, d_u, h->out1.p /*scratch*/, h->out2.p /*scratch*/, h->Esc.p, h->rho_vec.p, h->dscal.p, h->flag.p, B,
                       m, h->BT, h->st.scaling ? 1 : 0, s));
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (flag & 2) return MI_OSQP_ERR_INVALID_DATA;
  if (flag & 1) {   // a row changed type -> needs the refactor path: go through the host
    if (h_l && h_u) return update_bounds_on_host(h, h_l, h_u);
    std::vector<double> hl((size_t)B * m), hu((size_t)B * m);
    HIPCHK(hipMemcpy(hl.data(), d_l, hl.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hu.data(), d_u, hu.size() * sizeof(double), hipMemcpyDeviceToHost));
    return update_bounds_on_host(h, hl.data(), hu.data());
  }
  HIPCHK(launch_bounds(d_l, d_u, h->l.p, h->u.p, h->Esc.p, h->rho_vec.p, h->dscal.p, h->flag.p, B, m, h->BT,
                       h->st.scaling ? 1 : 0, s));
  HIPCHK(hipStreamSynchronize(s));
  h->host_bounds_stale = true;
  return MI_OSQP_OK;
}

static int update_A_values(mi_osqp_batch *h, const int64_t *Ap, const int64_t *Ai, const double *Av);

// The host mirrors of the scaled problem after the device has re-equilibrated: fetched when a host path needs them
// (a bounds update that changes row types, MI_OSQP_HOST_RUIZ).
static int ensure_mirrors(mi_osqp_batch *h) {
  if (!h->host_scaling_stale) return 0;
  const Analysis &an = (*h->anp);
  const int n = an.n, m = an.m, B = h->B, nnzP = an.Pp[n], nnzA = an.Ap[n], pa_len = nnzP + nnzA;
  int rc;
  if ((rc = ensure_stage(h, (size_t)B * std::max({pa_len, n, m, 1}) + 1, 0))) return rc;
  std::vector<double> tmp;
  auto down = [&](const double *src, int len) -> int {
    tmp.resize((size_t)B * len);
    if (!len) return 0;
    HIPCHK(launch_deinterleave(src, h->stage.p, B, len, h->BT, h->stream));
    HIPCHK(hipMemcpyAsync(tmp.data(), h->stage.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  };
  if ((rc = down(h->pa_val.p, pa_len))) return rc;
  for (int q = 0; q < B; q++) {
    std::copy(tmp.begin() + (size_t)q * pa_len, tmp.begin() + (size_t)q * pa_len + nnzP, h->qp[q].Pv.begin());
    std::copy(tmp.begin() + (size_t)q * pa_len + nnzP, tmp.begin() + (size_t)(q + 1) * pa_len, h->qp[q].Av.begin());
  }
  if ((rc = down(h->q.p, n))) return rc;
  for (int q = 0; q < B; q++) std::copy(tmp.begin() + (size_t)q * n, tmp.begin() + (size_t)(q + 1) * n, h->qp[q].q.begin());
  if ((rc = down(h->Dsc.p, n))) return rc;
  for (int q = 0; q < B; q++) for (int j = 0; j < n; j++) { h->qp[q].D[j] = tmp[(size_t)q * n + j]; h->qp[q].Dinv[j] = 1.0 / h->qp[q].D[j]; }
  if ((rc = down(h->Esc.p, m))) return rc;
  for (int q = 0; q < B; q++) for (int i = 0; i < m; i++) { h->qp[q].E[i] = tmp[(size_t)q * m + i]; h->qp[q].Einv[i] = 1.0 / h->qp[q].E[i]; }
  const size_t dcnt = (size_t)h->ntiles * DS_COUNT * h->BT;
  HIPCHK(hipMemcpy(h->h_dscal, h->dscal.p, dcnt * sizeof(double), hipMemcpyDeviceToHost));
  for (int q = 0; q < B; q++) {
    h->qp[q].c = h->h_dscal[(size_t)(q / h->BT) * DS_COUNT * h->BT + DS_C * h->BT + q % h->BT];
    h->qp[q].cinv = 1.0 / h->qp[q].c;
  }
  h->host_scaling_stale = false;
  h->host_bounds_stale = true;
  if ((rc = sync_bounds_to_host(h))) return rc;
  for (int q = 0; q < B; q++) (void)refresh_rho_types(an, h->qp[q]);
  h->host_rho_stale = true;
  return sync_rho_to_host(h);
}

// Row E13's A / bounds update with row E2 on the device: raw values through pinned memory, ruiz_kernel (unscale, new A,
// equilibrate, scale the bounds), the check streams re-scattered, ONE refactorisation of every QP.  l / u null: bounds kept.
static int device_update(mi_osqp_batch *h, const double *Av, const double *l, const double *u,
                         const double *dA = nullptr, const double *dl = nullptr, const double *du = nullptr) {      // (dA / dl / du: the same data already in HBM)
  const Analysis &an = (*h->anp);
  const int n = an.n, m = an.m, B = h->B, nnzP = an.Pp[n], nnzA = an.Ap[n], pa_len = nnzP + nnzA;
  h->clear_rho_updates = true;
  const size_t cA = (size_t)B * nnzA, cb = (l || dl) ? (size_t)B * m : 0, cpa = (size_t)B * pa_len;
  int rc;
  if ((rc = ensure_pin(h, cA + 2 * cb + 1)) || (rc = ensure_stage(h, cA + 2 * cb + cpa + 1, (size_t)B))) return rc;
  if (!dA) {
    // through pinned memory in slices of 4 MB: the transfer of a slice runs while the host threads copy the next one
    // (256 GOMP QPs of config 4: 30 MB, ~2 ms when copied whole and then sent)
    const size_t slice = (size_t)1 << 19;
    struct Part { const double *src; size_t off, len; };
    std::vector<Part> parts{{Av, 0, cA}};
    if (l) { parts.push_back({l, cA, cb}); parts.push_back({u, cA + cb, cb}); }
    for (const Part &pt : parts)
      for (size_t o = 0; o < pt.len; o += slice) {
        const size_t len = std::min(slice, pt.len - o);
        par_copy(h->pin + pt.off + o, pt.src + o, len);
        HIPCHK(hipMemcpyAsync(h->stage.p + pt.off + o, h->pin + pt.off + o, len * sizeof(double), hipMemcpyHostToDevice, h->stream));
      }
  }
  RuizArgs r{};
  r.n = n; r.m = m; r.nnzP = nnzP; r.nnzA = nnzA; r.B = B; r.BT = h->BT; r.iters = (int)h->st.scaling;
  r.Prow = h->rz_prow.p; r.Pcol = h->rz_pcol.p; r.Arow = h->rz_arow.p; r.Acol = h->rz_acol.p;
  if (dA) { r.rawA = dA; r.rawl = dl; r.rawu = du; }
  else { r.rawA = h->stage.p; r.rawl = l ? h->stage.p + cA : nullptr; r.rawu = l ? h->stage.p + cA + cb : nullptr; }
  r.pa_val = h->pa_val.p; r.q = h->q.p; r.Dsc = h->Dsc.p; r.Dsc_inv = h->Dsc_inv.p; r.Esc = h->Esc.p; r.Esc_inv = h->Esc_inv.p;
  r.l = h->l.p; r.u = h->u.p; r.dscal = h->dscal.p;
  r.dn = h->out1.p; r.en = h->out1.p + (size_t)B * n;              // (scratch of check_kernel: (2n + m) doubles per QP)
  r.pa_out = h->stage.p + cA + 2 * cb;
  HIPCHK(launch_ruiz(r, h->stream));
  std::vector<int> ids(B);
  for (int i = 0; i < B; i++) ids[i] = i;
  HIPCHK(hipMemcpyAsync(h->ids.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(launch_scatter(r.pa_out, h->chk_val.p, h->chk.src.p, h->ids.p, B, pa_len, h->chk.view(an.chk), h->BT, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->host_scaling_stale = true; h->host_bounds_stale = true; h->host_rho_stale = true;
  if ((rc = refactor_qps(h, std::move(ids)))) return rc;          // (factor_kernel derives the rho vectors from the bounds in force)
  return snapshot(h);
}
// Device or host equilibration: ruiz_kernel gives every QP one workgroup - right for a batch (256 GOMP QPs: 22 -> 7 ms per
// update), wrong for a handful of large QPs (one QP of 48 k entries: 5 ms on a host thread, 14 ms in one workgroup).
// MI_OSQP_HOST_RUIZ=1 / MI_OSQP_DEVICE_RUIZ=1 force one or the other (same bits either way).
static bool host_ruiz(const mi_osqp_batch *h) {
  if (getenv("MI_OSQP_HOST_RUIZ")) return true;
  if (getenv("MI_OSQP_DEVICE_RUIZ")) return false;
  const Analysis &an = (*h->anp);
  const long per_qp = (long)an.Pp[an.n] + an.Ap[an.n] + an.n + an.m;
  return h->B < 16 || per_qp > 65536;
}

int mi_osqp_batch_update_A(mi_osqp_batch *h, const int64_t *Ap, const int64_t *Ai, const double *Av) {
  CallTimer timer_("batch_update_A");
  if (!h || !Ap || !Ai || !Av) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  const Analysis &an = (*h->anp);
  for (int j = 0; j <= an.n; j++) if (Ap[j] != an.Ap[j]) return MI_OSQP_ERR_PATTERN_CHANGED;
  for (int k = 0; k < an.Ap[an.n]; k++) if (Ai[k] != an.Ai[k]) return MI_OSQP_ERR_PATTERN_CHANGED;
  if (!host_ruiz(h)) return device_update(h, Av, nullptr, nullptr);
  int rc = update_A_values(h, Ap, Ai, Av);
  if (rc || (rc = mi_osqp_batch_refactor_device(h))) return rc;
  return snapshot(h);
}

// QPSolver::update with the new values already in HBM (QP-major [B][nnzA], [B][m], natural CSC order of the pattern given at
// setup - a planner that assembles its rows on the device, or mi_gomp_scene): nothing crosses PCIe but a validation flag.
int mi_osqp_batch_update_A_bounds_device(mi_osqp_batch *h, const double *d_Av, const double *d_l, const double *d_u, void *stream) {
  CallTimer timer_("batch_update_A_bounds_device");
  if (!h || !d_Av || !d_l || !d_u) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  const Analysis &an = (*h->anp);
  const int m = an.m, B = h->B, nnzA = an.Ap[an.n];
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  if (host_ruiz(h)) {            // a handful of large QPs: equilibrated on host threads, so the values come down first
    std::vector<double> hA((size_t)B * nnzA), hl((size_t)B * m), hu((size_t)B * m);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(hA.data(), d_Av, hA.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hl.data(), d_l, hl.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hu.data(), d_u, hu.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<int64_t> Ap(an.Ap.begin(), an.Ap.end()), Ai(an.Ai.begin(), an.Ai.end());
    return mi_osqp_batch_update_A_bounds(h, Ap.data(), Ai.data(), hA.data(), hl.data(), hu.data());
  }
  if (m) {                       // l <= u, checked without writing (pass 1 of the bounds update)
    HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), s));
    HIPCHK(launch_bounds(d_l, d_u, h->out1.p, h->out2.p, h->Esc.p, h->rho_vec.p, h->dscal.p, h->flag.p, B, m, h->BT, h->st.scaling ? 1 : 0, s));
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flag & 2) return MI_OSQP_ERR_INVALID_DATA;
  } else HIPCHK(hipStreamSynchronize(s));
  return device_update(h, nullptr, nullptr, nullptr, d_Av, d_l, d_u);
}

// new A values and new bounds, ONE refactorisation (QPSolver::update: [REF] src/osqp-wrapper.h:33-43)
int mi_osqp_batch_update_A_bounds(mi_osqp_batch *h, const int64_t *Ap, const int64_t *Ai, const double *Av, const double *l, const double *u) {
  CallTimer timer_("batch_update_A_bounds");
  if (!h || !Ap || !Ai || !Av || !l || !u) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  const Analysis &an = (*h->anp);
  const int m = an.m, B = h->B;
  {
    const size_t tot = (size_t)B * m, chunk = (size_t)1 << 17;
    std::atomic<int> bad{0};
    const int parts = (int)((tot + chunk - 1) / chunk);
    auto scan = [&](int i, int) { const size_t e = std::min(tot, ((size_t)i + 1) * chunk); for (size_t k = (size_t)i * chunk; k < e; k++) if (l[k] > u[k]) { bad.store(1); return; } };
    if (parts <= 2) { for (int i = 0; i < parts; i++) scan(i, 0); } else parallel_for(parts, scan);
    if (bad.load()) return MI_OSQP_ERR_INVALID_DATA;
  }
  for (int j = 0; j <= an.n; j++) if (Ap[j] != an.Ap[j]) return MI_OSQP_ERR_PATTERN_CHANGED;
  for (int k = 0; k < an.Ap[an.n]; k++) if (Ai[k] != an.Ai[k]) return MI_OSQP_ERR_PATTERN_CHANGED;
  if (!host_ruiz(h)) return device_update(h, Av, l, u);
  int rc = update_A_values(h, Ap, Ai, Av);        // (the host mirrors are authoritative from here on)
  if (rc) return rc;
  for (int q = 0; q < B; q++) {
    QPNumeric &Q = h->qp[q];
    for (int i = 0; i < m; i++) {
      Q.l[i] = std::max(l[(size_t)q * m + i], -kInfty); Q.u[i] = std::min(u[(size_t)q * m + i], kInfty);
      if (h->st.scaling) { Q.l[i] *= Q.E[i]; Q.u[i] *= Q.E[i]; }
    }
    (void)refresh_rho_types(an, Q);
  }
  std::vector<int> all(B);
  for (int i = 0; i < B; i++) all[i] = i;
  if ((rc = upload_problem(h, all, false)) || (rc = refactor_qps(h, all))) return rc;       // (factor_kernel derives the rho vectors from the new bounds)
  return snapshot(h);
}

static int update_A_values(mi_osqp_batch *h, const int64_t *Ap, const int64_t *Ai, const double *Av) {
  { int rc0 = ensure_mirrors(h); if (rc0) return rc0; }
  h->clear_rho_updates = true;
  const Analysis &an = (*h->anp);
  int n = an.n, B = h->B, nnzA = an.Ap[n], rc;
  for (int j = 0; j <= n; j++) if (Ap[j] != an.Ap[j]) return MI_OSQP_ERR_PATTERN_CHANGED;
  for (int k = 0; k < nnzA; k++) if (Ai[k] != an.Ai[k]) return MI_OSQP_ERR_PATTERN_CHANGED;
  if ((rc = sync_bounds_to_host(h)) || (rc = sync_rho_to_host(h))) return rc;
  // host: unscale, new values, Ruiz rescale (O(nnz) per QP); device: the numeric refactorisation of every QP
  // (factor_kernel on the new scaled values and the current rho vectors) - the same code path a rho update takes
  const int CH = 128;
  for (int c0 = 0; c0 < B; c0 += CH) {
    int c1 = std::min(B, c0 + CH);
    parallel_for(c1 - c0, [&](int k, int) {
      int qi = c0 + k;
      QPNumeric &Q = h->qp[qi];
      if (h->st.scaling) unscale_qp(an, Q);
      std::copy(Av + (size_t)qi * nnzA, Av + (size_t)(qi + 1) * nnzA, Q.Av.begin());
      if (h->st.scaling) scale_qp(an, h->st, Q);
    });
    std::vector<int> ids(c1 - c0);
    for (int k = 0; k < c1 - c0; k++) ids[k] = c0 + k;
    if ((rc = upload_problem(h, ids, true)) || (rc = sync_scalars_to_device(h, ids, true))) return rc;
  }
  return MI_OSQP_OK;
}

int mi_osqp_batch_refactor_device(mi_osqp_batch *h) {
  if (!h) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  { const int rc_ = cont_leave(h); if (rc_) return rc_; }
  std::vector<int> all(h->B);
  for (int i = 0; i < h->B; i++) all[i] = i;
  return refactor_qps(h, std::move(all));
}

int mi_osqp_batch_spmv(mi_osqp_batch *h, const double *d_x, const double *d_y, double *d_Px, double *d_Aty, double *d_Ax,
                       void *stream) {
  if (!h) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  KernelArgs a = make_args(h);
  if (h->sp_ptr.n && !getenv("MI_OSQP_SPMV_STREAM")) {      // one read of P and A for all three products
    SpmvFused t{};
    t.ptr = h->sp_ptr.p; t.ent = h->sp_ent.p; t.pa_val = h->pa_val.p; t.pa_len = (*h->anp).Pp[(*h->anp).n] + (*h->anp).Ap[(*h->anp).n];
    if (h->sp_npass && !getenv("MI_OSQP_SPMV_NO_PREFETCH")) {
      t.ell = h->sp_ell.p; t.rowid = h->sp_rowid.p; t.n_pass = h->sp_npass;
      for (int p = 0; p < 4; p++) { t.ell_off[p] = h->sp_ell_off[p]; t.ell_k[p] = h->sp_ell_k[p]; }
    }
    HIPCHK(launch_spmv_fused(a, t, h->BT, h->ntiles, h->n_cus, s, d_x, d_y, d_Px, d_Aty, d_Ax));
    HIPCHK(hipStreamSynchronize(s));
    return MI_OSQP_OK;
  }
  size_t lds = (size_t)((*h->anp).Next + 2 * (*h->anp).n + (*h->anp).m) * h->BT * sizeof(double);
  a.op_out_lds = !h->global_xs && lds <= 160 * 1024 - 1024;
  if (!a.op_out_lds) lds = h->lds;
  HIPCHK(launch_spmv(a, h->BT, h->ntiles, h->threads, lds, s, d_x, d_y, d_Px, d_Aty, d_Ax));
  HIPCHK(hipStreamSynchronize(s));
  return MI_OSQP_OK;
}

int mi_osqp_batch_kkt_solve(mi_osqp_batch *h, const double *d_rhs, double *d_sol, void *stream) {
  if (!h || !d_rhs || !d_sol) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  KernelArgs a = make_args(h);
  {
    std::unique_lock<std::mutex> spin_lock(spin_mutex(h->device), std::defer_lock);
    if (h->mw_groups > 0) spin_lock.lock();
    if (h->mw_groups > 0) HIPCHK(hipMemsetAsync(h->mw_bar.p, 0, 4 * sizeof(uint32_t), s));
    HIPCHK(launch_kkt_solve(a, h->BT, h->ntiles, h->mw_groups > 0 ? h->mw_threads : h->threads, h->lds, s, d_rhs, d_sol));
    HIPCHK(hipStreamSynchronize(s));
  }
  return mw_barrier_ok(h);
}

int mi_osqp_debug_trace_kkt_solve(mi_osqp_batch *h, int32_t which, const double *d_rhs, double *d_sol, uint32_t *out,
                                  int64_t cap, int64_t *dims) {
  if (!h || !dims) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  const int nw = h->threads / 64;
  const int64_t fp = (*h->anp).fwd.n_phases, bp = (*h->anp).bwd.n_phases, words = 8 + 4 * nw + (fp + bp) * nw * 2;
  dims[0] = fp; dims[1] = bp; dims[2] = nw; dims[3] = words;
  if (!out) return MI_OSQP_OK;
  if (which == 1 || which == 2) {
    const Schedule &sc = which == 1 ? (*h->anp).fwd : (*h->anp).bwd;
    if ((int64_t)sc.phase.size() > cap) { g_last_error = "trace: output too small"; return MI_OSQP_ERR_INVALID_DATA; }
    std::copy(sc.phase.begin(), sc.phase.end(), out);
    return MI_OSQP_OK;
  }
  if (!d_rhs || !d_sol) return MI_OSQP_ERR_NULL;
  if (2 * words > cap) { g_last_error = "trace: output too small"; return MI_OSQP_ERR_INVALID_DATA; }
  uint32_t *d_tr = nullptr;
  HIPCHK(hipMalloc(&d_tr, (size_t)2 * words * 4));
  KernelArgs a = make_args(h);
  hipError_t e = launch_kkt_trace(a, h->BT, h->ntiles, h->threads, h->lds, h->stream, d_rhs, d_sol, d_tr, (uint32_t)words);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(out, d_tr, (size_t)2 * words * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d_tr);
  HIPCHK(e);
  return MI_OSQP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ continuous
// Per-QP entry points + a non-blocking advance: the reference's SQP loop is per trajectory - solve, check, re-linearise,
// update, solve again ([REF] src/gomp-solver.h:70-88) - so the QPs of a batch do not finish together and must not wait
// for each other.  A QP's solve is begun with solve_begin_some, advance() enqueues one segment (L iterations + check +
// the refactorisations the check asks for) for every QP that is iterating, poll() reports the QPs that finished, and the
// caller updates / warm-starts / begins them again while the rest keeps iterating.  Every QP counts its iterations from
// its own begin (IS_CUR), so it takes exactly the iterations, rho updates and checks of a blocking solve of its own:
// results are those of mi_osqp_batch_solve bit for bit.  Nothing in these calls waits for the device except poll().

static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// iterations per segment of advance_kernel: the gcd of the periods at which a solve looks at itself
static int segment_length(const Settings &S) {
  int L = (int)S.max_iter;
  if (S.check_termination > 0) L = gcd_i(L, (int)S.check_termination);
  if (S.adaptive_rho && S.adaptive_rho_interval > 0) L = gcd_i(L, (int)S.adaptive_rho_interval);
  return std::max(1, L);
}
// the pinned images advance_kernel publishes flags / scalars in, its tile counter and completion word
static int ensure_advance_buffers(mi_osqp_batch *h) {
  mi_osqp_batch::Cont &c = h->cont;
  const size_t icnt = (size_t)h->ntiles * IS_COUNT * h->BT, dcnt = (size_t)h->ntiles * DS_COUNT * h->BT;
  for (int k = 0; k < 2; k++) {
    if (!c.h_is[k]) HIPCHK(hostpool::alloc((void **)&c.h_is[k], icnt * sizeof(int), &c.h_is_cap[k]));
    if (!c.h_ds[k]) HIPCHK(hostpool::alloc((void **)&c.h_ds[k], dcnt * sizeof(double), &c.h_ds_cap[k]));
  }
  int rc;
  if (!c.counter.p) { if ((rc = c.counter.alloc(4)) || (rc = c.counter.zero(h->stream))) return rc; }
  if (!c.h_done) { HIPCHK(hostpool::alloc((void **)&c.h_done, 64, &c.h_done_cap)); c.h_done[0] = c.h_done[1] = 0; }
  return MI_OSQP_OK;
}
// wait for the completion word of launch `seq` (no runtime call in the way); fallback: the stream
static int wait_launch_over(mi_osqp_batch *h, unsigned seq) {
  mi_osqp_batch::Cont &c = h->cont;
  auto over = [&]() { return __atomic_load_n(c.h_done, __ATOMIC_ACQUIRE) == seq; };
  const double t0 = now_s();
  for (int spin = 0; !over(); spin++) {
    if (spin < 20000) { __builtin_ia32_pause(); continue; }
    struct timespec ts{0, 20000};
    nanosleep(&ts, nullptr);
    if ((spin & 1023) == 0 && now_s() - t0 > 5.0) { HIPCHK(hipStreamSynchronize(h->stream)); if (!over()) { g_last_error = "advance launch ended without reporting"; return MI_OSQP_ERR_DEVICE; } }
  }
  return MI_OSQP_OK;
}

static int cont_enter(mi_osqp_batch *h) {
  mi_osqp_batch::Cont &c = h->cont;
  if (c.on) return MI_OSQP_OK;
  if (h->global_xs || h->mw_groups > 0) { g_last_error = "continuous batching serves LDS-resident QPs (a large single QP has no batch to be continuous in)"; return MI_OSQP_ERR_INVALID_DATA; }
  const Analysis &an = (*h->anp);
  const int B = h->B, n = an.n, m = an.m, BT = h->BT, nslots = h->ntiles * BT;
  c.L = segment_length(h->st);
  c.running.assign((size_t)B, 0); c.clear_rho.assign((size_t)B, 0); c.epoch.assign((size_t)B, 0);
  c.info.assign((size_t)B, mi_osqp_info{});
  c.n_running = 0; c.adv_seq = c.polled_seq = 0; c.launch_seq = 0;
  const size_t icnt = (size_t)h->ntiles * IS_COUNT * BT;
  { const int rc0 = ensure_advance_buffers(h); if (rc0) return rc0; }
  for (int k = 0; k < 2; k++) {
    if (!c.ev[k]) HIPCHK(hipEventCreateWithFlags(&c.ev[k], hipEventDisableTiming));
    c.seq_of[k] = 0;
  }
  if (!c.xh) HIPCHK(hostpool::alloc((void **)&c.xh, (size_t)B * n * sizeof(double), &c.xh_cap));
  if (!c.yh) HIPCHK(hostpool::alloc((void **)&c.yh, (size_t)B * std::max(m, 1) * sizeof(double), &c.yh_cap));
  if (!c.ring_h) {
    // a few rounds of per-QP calls: (A values + bounds + a warm start) of every QP, twice
    const size_t per_qp = ((size_t)an.Ap[n] + (size_t)an.Pp[n] + 2 * (size_t)m + (size_t)n + 16) * sizeof(double);      // (a whole call - ids, rows in, scaled values out - fits one lap)
    size_t want = std::max<size_t>((size_t)4 << 20, std::min<size_t>((size_t)256 << 20, 2 * per_qp * (size_t)B));
    // (tests: MI_OSQP_CONT_RING_KB = a ring barely larger than one whole-batch call, so that every few calls wrap)
    if (const char *er = getenv("MI_OSQP_CONT_RING_KB")) want = std::max<size_t>((size_t)atol(er) << 10, per_qp * (size_t)B + ((size_t)64 << 10));
    HIPCHK(hostpool::alloc((void **)&c.ring_h, want, &c.ring_h_cap));
    c.ring_cap = c.ring_h_cap;
    int rc = c.ring_d.alloc(c.ring_cap);
    if (rc) return rc;
    c.ring_head = 0;
  }
  int rc;
  if (c.work.n < (size_t)nslots + 4 && (rc = c.work.alloc((size_t)nslots + 4))) return rc;
  if (!c.ustream) {
    // By default the preparation and the refactorisations share the handle's stream with the advance launches (in order):
    // measured on the GOMP obstacle scene (ten handles driven by ten host threads) a second stream per handle LOSES - 20
    // streams on the runtime's 4-10 hardware queues wait for each other's kernels (900 against 620 trajectories/s,
    // profiles/r03).  MI_OSQP_CONT_STREAMS=2 gives every handle its second stream; the protocol is the same either way.
    { const char *e = getenv("MI_OSQP_CONT_STREAMS"); c.own_ustream = e && atoi(e) == 2; }
    if (c.own_ustream) HIPCHK(hipStreamCreateWithFlags(&c.ustream, hipStreamNonBlocking));
    else c.ustream = h->stream;
    HIPCHK(hipEventCreateWithFlags(&c.ev_adv, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c.ev_u, hipEventDisableTiming));
    if ((rc = c.rz_scratch.alloc((size_t)B * (n + m) + 1))) return rc;
  }
  if ((rc = c.counter.zero(h->stream))) return rc;
  c.h_done[0] = c.h_done[1] = 0; c.last_active = 0;
  // the state of the last blocking solve, if any, stays valid; every slot is idle until its solve is begun
  HIPCHK(hipMemcpy(h->h_iscal, h->iscal.p, icnt * sizeof(int), hipMemcpyDeviceToHost));
  for (int t = 0; t < h->ntiles; t++)
    for (int b = 0; b < BT; b++) {
      int *p = h->h_iscal + (size_t)t * IS_COUNT * BT;
      p[IS_DONE * BT + b] = 1; p[IS_CUR * BT + b] = 0; p[IS_PENDING * BT + b] = 0; p[IS_EPOCH * BT + b] = 0;
      if (h->clear_rho_updates) p[IS_RHO_UPDATES * BT + b] = 0;
      if (t * BT + b < B && h->failed[(size_t)t * BT + b]) p[IS_NEED_REFACTOR * BT + b] = -1;
    }
  h->clear_rho_updates = false;
  HIPCHK(hipMemcpyAsync(h->iscal.p, h->h_iscal, icnt * sizeof(int), hipMemcpyHostToDevice, h->stream));
  memcpy(c.h_is[0], h->h_iscal, icnt * sizeof(int)); memcpy(c.h_is[1], h->h_iscal, icnt * sizeof(int));      // the host images advance_kernel writes
  if (!c.stop.p && (rc = c.stop.alloc(4))) return rc;
  if ((rc = c.stop.zero(h->stream))) return rc;
  std::vector<int> ident((size_t)nslots);
  for (int sl = 0; sl < nslots; sl++) ident[sl] = sl < B ? sl : -1;
  HIPCHK(hipMemcpyAsync(h->qp_of_slot.p, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  c.on = true;
  return MI_OSQP_OK;
}

// leave the continuous mode (a blocking call follows): wait for what is enqueued, forget the solves in flight
static int cont_leave(mi_osqp_batch *h) {
  mi_osqp_batch::Cont &c = h->cont;
  if (!c.on) return MI_OSQP_OK;
  if (c.ustream) HIPCHK(hipStreamSynchronize(c.ustream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t icnt = (size_t)h->ntiles * IS_COUNT * h->BT;
  HIPCHK(hipMemcpy(h->h_iscal, h->iscal.p, icnt * sizeof(int), hipMemcpyDeviceToHost));
  for (int q = 0; q < h->B; q++) {
    h->failed[q] = h->h_iscal[(size_t)(q / h->BT) * IS_COUNT * h->BT + IS_NEED_REFACTOR * h->BT + q % h->BT] < 0;
    if (c.clear_rho[q]) h->clear_rho_updates = true;      // (coarser than per QP, like the whole-batch update calls)
  }
  // the solutions of the finished QPs, for the whole-batch getters
  HIPCHK(hipMemcpy(h->x_out.p, c.xh, (size_t)h->B * (*h->anp).n * sizeof(double), hipMemcpyHostToDevice));
  if ((*h->anp).m) HIPCHK(hipMemcpy(h->y_out.p, c.yh, (size_t)h->B * (*h->anp).m * sizeof(double), hipMemcpyHostToDevice));
  c.on = false; c.n_running = 0;
  h->host_bounds_stale = h->host_rho_stale = true;
  return MI_OSQP_OK;
}

// a region of the staging ring: host pointer + the device address of the same offset
struct RingSpan { char *host; char *dev; };
static int ring_take(mi_osqp_batch *h, size_t bytes, RingSpan &out) {
  mi_osqp_batch::Cont &c = h->cont;
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > c.ring_cap) { g_last_error = "per-QP call larger than the staging ring"; return MI_OSQP_ERR_ALLOC; }
  if (c.ring_head + bytes > c.ring_cap) { HIPCHK(hipStreamSynchronize(c.ustream)); HIPCHK(hipStreamSynchronize(h->stream)); c.ring_head = 0; }      // everything handed out so far has been consumed
  out.host = c.ring_h + c.ring_head; out.dev = c.ring_d.p + c.ring_head;
  c.ring_head += bytes;
  return MI_OSQP_OK;
}
// Everything ONE call hands out must come from one lap of the ring.  A wrap between two spans of a call frees - at the wrap's
// synchronisation - nothing of the call's own earlier spans (they are filled and enqueued after it), and the head runs into
// them, unsynchronised, once it has advanced that far again: the staging of a QP's new rows could be overwritten while its
// transfer or its equilibration was still reading it (seen as a sporadic memory fault on the fourth run of 128 UR5e
// trajectories, whose updates take 140 KB of ring per QP).  So a call reserves its total first; the takes that follow cannot wrap.
static int ring_reserve(mi_osqp_batch *h, size_t bytes, int spans) {
  mi_osqp_batch::Cont &c = h->cont;
  bytes += (size_t)256 * (size_t)spans;
  if (bytes > c.ring_cap) { g_last_error = "per-QP call larger than the staging ring"; return MI_OSQP_ERR_ALLOC; }
  if (c.ring_head + bytes > c.ring_cap) { HIPCHK(hipStreamSynchronize(c.ustream)); HIPCHK(hipStreamSynchronize(h->stream)); c.ring_head = 0; }
  return MI_OSQP_OK;
}
static int ring_upload(mi_osqp_batch *h, const RingSpan &sp, size_t bytes) {
  if (bytes) HIPCHK(hipMemcpyAsync(sp.dev, sp.host, bytes, hipMemcpyHostToDevice, h->cont.ustream));
  return MI_OSQP_OK;
}

static int cont_check_ids(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, bool must_be_idle) {
  if (n_ids < 0 || (n_ids > 0 && !ids)) return MI_OSQP_ERR_NULL;
  if (n_ids > h->B) return MI_OSQP_ERR_INVALID_DATA;
  for (int64_t j = 0; j < n_ids; j++) {
    if (ids[j] < 0 || ids[j] >= h->B) return MI_OSQP_ERR_INVALID_DATA;
    if (must_be_idle && h->cont.running[(size_t)ids[j]]) { g_last_error = "QP " + std::to_string(ids[j]) + " is still iterating"; return MI_OSQP_ERR_INVALID_DATA; }
  }
  return MI_OSQP_OK;
}

// the id list (as int) and, on request, the selection map of the tile kernels (slot -> position in the list + 1)
static int cont_stage_ids(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, int **d_ids, int **d_sel) {
  const int nslots = h->ntiles * h->BT;
  RingSpan sp;
  int rc = ring_take(h, ((size_t)n_ids + (d_sel ? (size_t)nslots : 0)) * sizeof(int), sp);
  if (rc) return rc;
  int *hi = (int *)sp.host;
  for (int64_t j = 0; j < n_ids; j++) hi[j] = (int)ids[j];
  if (d_sel) {
    int *hs = hi + n_ids;
    for (int sl = 0; sl < nslots; sl++) hs[sl] = 0;
    for (int64_t j = 0; j < n_ids; j++) hs[ids[j]] = (int)j + 1;        // (a QP listed twice: its last row wins)
    *d_sel = (int *)sp.dev + n_ids;
  }
  *d_ids = (int *)sp.dev;
  return ring_upload(h, sp, ((size_t)n_ids + (d_sel ? (size_t)nslots : 0)) * sizeof(int));
}

// the listed slots' share of the setup snapshot (mi_osqp_batch_reset restores it)
static int snapshot_some(mi_osqp_batch *h, const int *d_ids, int nq, hipStream_t st) {
  const Analysis &an = (*h->anp);
  HIPCHK(launch_copy_slot_streams(h->fwd_val0.p, h->fwd_val.p, d_ids, nq, (size_t)an.fwd.phys_steps() * 64, st));
  HIPCHK(launch_copy_slot_streams(h->bwd_val0.p, h->bwd_val.p, d_ids, nq, (size_t)an.bwd.phys_steps() * 64, st));
  if (an.dt.k) HIPCHK(launch_copy_slot_streams(h->dt_val0.p, h->dt_val.p, d_ids, nq, (size_t)an.dt.n_steps * 64, st));
  HIPCHK(launch_copy_slot_rows(h->dinv0.p, h->dinv.p, d_ids, nq, an.N, h->BT, st));
  HIPCHK(launch_copy_slot_rows(h->rho_vec0.p, h->rho_vec.p, d_ids, nq, an.m, h->BT, st));
  HIPCHK(launch_copy_slot_rows(h->rho_inv0.p, h->rho_inv.p, d_ids, nq, an.m, h->BT, st));
  HIPCHK(launch_copy_slot_rows(h->dscal0.p, h->dscal.p, d_ids, nq, DS_COUNT, h->BT, st));
  return MI_OSQP_OK;
}

// the refactorisation kernels for a device-resident work list of `count` entries (slots, -1 = none): one QP per workgroup,
// no group sharing (other handles' kernels may hold the CUs: nothing here may spin on a co-resident partner)
static int enqueue_refactor_list(mi_osqp_batch *h, const int *d_work, int count) {
  if (count <= 0) return MI_OSQP_OK;
  const Analysis &an = (*h->anp);
  hipStream_t st = h->cont.ustream;         // (every refactorisation of the continuous mode: one stream, one scratch)
  FactorArgs fa = make_factor_args(h, 0);
  fa.work = d_work; fa.mw_groups = 0;
  HIPCHK(launch_factor(fa, 1, count, factor_threads(), st));
  if (an.dt.k) {
    const DenseTail &dt = an.dt;
    TailArgs da{};
    da.n = an.n; da.N = an.N; da.s = dt.s; da.k = dt.k; da.kbt = 1; da.home_bt = h->BT;
    da.storage = an.bf.storage; da.n_slots = dt.n_steps * 64u; da.n_lt = dt.n_lt; da.n_ltcol = dt.n_ltcol; da.n_quads = (uint32_t)(dt.asm_q64.size() / 64);
    da.nh = h->dt_nh; da.cs_doubles = h->dt_cs_doubles; da.work = d_work;
    da.lt_pos = h->dt_lt_pos.p; da.ltcol_col = h->dt_ltcol_col.p; da.tile_tab = h->dt_tile_tab.p; da.wave_tiles = h->dt_wave_tiles.p;
    da.dt_task = h->dt_task.p; da.dt_task_step = h->dt_task_step.p; da.n_tasks = (uint32_t)(dt.task.size() / 4);
    da.asm_q64 = h->dt_asm_q64.p; da.diag_tile = h->dt_diag_tile.p; da.src_tile = h->dt_src_tile.p;
    da.Lblk = fa.Lblk; da.Dl = fa.Dl; da.Sd = h->dt_Sd.p; da.dt_val = h->dt_val.p; da.dinv = h->dinv.p; da.npos = h->npos.p; da.iscal = h->iscal.p;
    da.trace = nullptr;
    HIPCHK(launch_tail(da, count, h->dt_lds_asm, h->dt_lds, st));
  }
  return MI_OSQP_OK;
}

// new A values + bounds of the listed QPs: equilibration on the device (fresh: from the raw P and q of setup; else unscale
// with the scaling in force first, as QPSolver::update does), check streams, refactorisation, snapshot.  All enqueued.
// (dA / dl / du non-null: the new data is on the device already, [B][nnzA] / [B][m] by QP number - the GOMP scene's kept rows)
static int cont_new_data(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, const double *Av, const double *l, const double *u, bool fresh,
                         const double *dA = nullptr, const double *dl = nullptr, const double *du = nullptr) {
  const Analysis &an = (*h->anp);
  const int n = an.n, m = an.m, nnzP = an.Pp[n], nnzA = an.Ap[n], pa_len = nnzP + nnzA, nq = (int)n_ids;
  int rc;
  if (!dA) for (size_t k = 0; k < (size_t)nq * m; k++) if (l[k] > u[k]) return MI_OSQP_ERR_INVALID_DATA;
  int *d_ids = nullptr;
  const size_t cA = (size_t)nq * nnzA, cb = (size_t)nq * m, cpa = (size_t)nq * pa_len;
  if ((rc = ring_reserve(h, (size_t)nq * sizeof(int) + (dA ? 0 : (cA + 2 * cb) * sizeof(double)) + std::max<size_t>(cpa, 1) * sizeof(double), 3))) return rc;
  if ((rc = cont_stage_ids(h, n_ids, ids, &d_ids, nullptr))) return rc;
  RingSpan in{nullptr, nullptr}, out;
  if (!dA && (rc = ring_take(h, (cA + 2 * cb) * sizeof(double), in))) return rc;
  if ((rc = ring_take(h, std::max<size_t>(cpa, 1) * sizeof(double), out))) return rc;
  double *hin = (double *)in.host, *din = (double *)in.dev;
  if (!dA) {
    memcpy(hin, Av, cA * sizeof(double)); memcpy(hin + cA, l, cb * sizeof(double)); memcpy(hin + cA + cb, u, cb * sizeof(double));
    if ((rc = ring_upload(h, in, (cA + 2 * cb) * sizeof(double)))) return rc;
  }
  hipStream_t us = h->cont.ustream;
  if (!dA && h->cont.keepA) {
    HIPCHK(launch_keep_rows(h->cont.keepA, din, d_ids, nq, nnzA, us));
    HIPCHK(launch_keep_rows(h->cont.keepl, din + cA, d_ids, nq, m, us));
    HIPCHK(launch_keep_rows(h->cont.keepu, din + cA + cb, d_ids, nq, m, us));
  }
  KernelArgs ka = make_args(h);
  if (fresh) HIPCHK(launch_fresh_slots(ka, d_ids, nq, h->BT, h->st.rho, us));
  RuizArgs r{};
  r.n = n; r.m = m; r.nnzP = nnzP; r.nnzA = nnzA; r.B = nq; r.BT = h->BT; r.iters = (int)h->st.scaling;
  r.ids = d_ids; r.fresh = fresh ? 1 : 0; r.rawP = h->rawP.p; r.rawq = h->rawq.p;
  r.Prow = h->rz_prow.p; r.Pcol = h->rz_pcol.p; r.Arow = h->rz_arow.p; r.Acol = h->rz_acol.p;
  if (dA) { r.rawA = dA; r.rawl = dl; r.rawu = du; r.raw_by_qp = 1; }
  else { r.rawA = din; r.rawl = din + cA; r.rawu = din + cA + cb; }
  r.pa_val = h->pa_val.p; r.q = h->q.p; r.Dsc = h->Dsc.p; r.Dsc_inv = h->Dsc_inv.p; r.Esc = h->Esc.p; r.Esc_inv = h->Esc_inv.p;
  r.l = h->l.p; r.u = h->u.p; r.dscal = h->dscal.p;
  r.dn = h->cont.rz_scratch.p; r.en = h->cont.rz_scratch.p + (size_t)h->B * n;        // (not the check kernels' scratch: they may be running)
  r.pa_out = (double *)out.dev;
  HIPCHK(launch_ruiz(r, us));
  HIPCHK(launch_scatter(r.pa_out, h->chk_val.p, h->chk.src.p, d_ids, nq, pa_len, h->chk.view(an.chk), h->BT, us));
  if ((rc = enqueue_refactor_list(h, d_ids, nq))) return rc;
  if ((rc = snapshot_some(h, d_ids, nq, us))) return rc;
  for (int64_t j = 0; j < n_ids; j++) { h->cont.clear_rho[(size_t)ids[j]] = 1; h->failed[(size_t)ids[j]] = 0; }
  h->host_scaling_stale = true; h->host_bounds_stale = true; h->host_rho_stale = true;
  return MI_OSQP_OK;
}

extern "C" {

int mi_osqp_batch_reinit_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, const double *Av, const double *l, const double *u) {
  CallTimer timer_("batch_reinit_some");
  if (!h || (n_ids > 0 && (!Av || !l || !u))) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  int rc;
  if ((rc = cont_enter(h)) || (rc = cont_check_ids(h, n_ids, ids, true))) return rc;
  if (!n_ids) return MI_OSQP_OK;
  return cont_new_data(h, n_ids, ids, Av, l, u, true);
}

int mi_osqp_batch_update_A_bounds_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, const double *Av, const double *l, const double *u) {
  CallTimer timer_("batch_update_A_bounds_some");
  if (!h || (n_ids > 0 && (!Av || !l || !u))) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  int rc;
  if ((rc = cont_enter(h)) || (rc = cont_check_ids(h, n_ids, ids, true))) return rc;
  if (!n_ids) return MI_OSQP_OK;
  return cont_new_data(h, n_ids, ids, Av, l, u, false);
}

int mi_osqp_batch_warm_start_x_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, const double *x) {
  CallTimer timer_("batch_warm_start_x_some");
  if (!h || (n_ids > 0 && !x)) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  int rc;
  if ((rc = cont_enter(h)) || (rc = cont_check_ids(h, n_ids, ids, true))) return rc;
  if (!n_ids) return MI_OSQP_OK;
  h->st.warm_start = 1;
  const size_t cnt = (size_t)n_ids * (*h->anp).n;
  int *d_ids = nullptr, *d_sel = nullptr;
  RingSpan sp;
  if ((rc = ring_reserve(h, ((size_t)n_ids + (size_t)h->ntiles * h->BT) * sizeof(int) + cnt * sizeof(double), 2))) return rc;
  if ((rc = cont_stage_ids(h, n_ids, ids, &d_ids, &d_sel)) || (rc = ring_take(h, cnt * sizeof(double), sp))) return rc;
  memcpy(sp.host, x, cnt * sizeof(double));
  if ((rc = ring_upload(h, sp, cnt * sizeof(double)))) return rc;
  KernelArgs a = make_args(h);
  a.sel = d_sel;
  HIPCHK(launch_warm_start(a, h->BT, h->ntiles, h->threads, h->lds, h->cont.ustream, (const double *)sp.dev));
  return MI_OSQP_OK;
}

int mi_osqp_batch_solve_begin_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids) {
  CallTimer timer_("batch_solve_begin_some");
  if (!h) return MI_OSQP_ERR_NULL;
  DevGuard guard(h->device);
  int rc;
  if ((rc = cont_enter(h)) || (rc = cont_check_ids(h, n_ids, ids, true))) return rc;
  if (!n_ids) return MI_OSQP_OK;
  mi_osqp_batch::Cont &c = h->cont;
  RingSpan sp;
  if ((rc = ring_take(h, 2 * (size_t)n_ids * sizeof(int), sp))) return rc;
  int *hi = (int *)sp.host;
  for (int64_t j = 0; j < n_ids; j++) {
    const size_t q = (size_t)ids[j];
    hi[j] = (int)q; hi[n_ids + j] = c.clear_rho[q] ? 1 : 0;
    c.clear_rho[q] = 0;
  }
  if ((rc = ring_upload(h, sp, 2 * (size_t)n_ids * sizeof(int)))) return rc;
  KernelArgs a = make_args(h);
  a.x_out = c.xh; a.y_out = c.yh;
  // (a QP whose last refactorisation lost the inertia carries flag -1 on the device: start_slots_kernel ends it as kNonConvex)
  HIPCHK(launch_start_slots(a, (const int *)sp.dev, (const int *)sp.dev + n_ids, (int)n_ids, h->BT, h->st.warm_start ? 0 : 1, c.ustream));
  for (int64_t j = 0; j < n_ids; j++) {
    const size_t q = (size_t)ids[j];
    if (!c.running[q]) { c.running[q] = 1; c.n_running++; }
    c.epoch[q]++;
  }
  return MI_OSQP_OK;
}

int mi_osqp_batch_advance(mi_osqp_batch *h, int64_t n_segments) {
  CallTimer timer_("batch_advance");
  if (!h) return MI_OSQP_ERR_NULL;
  if (n_segments <= 0) return MI_OSQP_ERR_INVALID_DATA;
  DevGuard guard(h->device);
  int rc;
  if ((rc = cont_enter(h))) return rc;
  mi_osqp_batch::Cont &c = h->cont;
  if (c.adv_seq - c.polled_seq >= 2) { g_last_error = "advance: two advances are waiting for poll()"; return MI_OSQP_ERR_INVALID_DATA; }
  const int BT = h->BT, nslots = h->ntiles * BT;
  KernelArgs a = make_args(h);
  a.x_out = c.xh; a.y_out = c.yh;
  // ONE launch: every tile runs up to n_segments segments (L iterations + check each) and publishes its flags and the
  // solutions of finished QPs in pinned host memory; with several segments the launch ends early for everybody once a QP
  // has finished (the caller wants to react to it), and for a tile whose QP asks for a refactorisation
  a.info_at_end = 1;
  // Nothing orders the two streams - unless the last launch found nothing to iterate: then whatever is running waits for
  // its preparation / refactorisation on the second stream, and launching again at once would only spin.  Such a launch
  // waits for the second stream's queue.
  if (c.last_active == 0) {
    HIPCHK(hipEventRecord(c.ev_u, c.ustream));
    HIPCHK(hipStreamWaitEvent(h->stream, c.ev_u, 0));
  }
  c.launch_seq++;
  const int par = (int)(c.adv_seq & 1);
  HIPCHK(launch_advance(a, BT, h->ntiles, h->threads, h->lds, h->stream, (int)std::min<int64_t>(n_segments, 1 << 20), c.L, c.h_is[par], c.h_ds[par],
                        n_segments > 1 ? c.stop.p : nullptr, c.launch_seq, c.counter.p, c.h_done));
  HIPCHK(hipEventRecord(c.ev[par], h->stream));
  if (h->st.adaptive_rho) {
    // row E13 without the host, next to the following launches: the QPs this launch paused (their rho changed) are listed on
    // the device, refactored and marked to resume - or ended as kNonConvex when the new factor lost its inertia
    HIPCHK(hipStreamWaitEvent(c.ustream, c.ev[par], 0));
    HIPCHK(launch_worklist(h->iscal.p, c.work.p, nslots, BT, c.ustream));
    if ((rc = enqueue_refactor_list(h, c.work.p, nslots))) return rc;
    HIPCHK(launch_resume_flagged(a, nslots, BT, c.ustream));
  }
  c.adv_seq++;
  c.seq_of[par] = c.adv_seq;
  h->host_rho_stale = true;
  return MI_OSQP_OK;
}

int mi_osqp_batch_poll(mi_osqp_batch *h, int64_t wait, int64_t *n_finished, int64_t *ids_out, int64_t capacity) {
  CallTimer timer_("batch_poll");
  if (!h || !n_finished) return MI_OSQP_ERR_NULL;
  *n_finished = 0;
  mi_osqp_batch::Cont &c = h->cont;
  if (!c.on || c.polled_seq >= c.adv_seq) return MI_OSQP_OK;      // nothing enqueued
  DevGuard guard(h->device);
  const int64_t seq = c.polled_seq + 1;
  const int par = (int)((seq - 1) & 1);
  // the launch is over when its last tile has written the sequence number into pinned memory (no runtime call in the way:
  // waking up from hipEventSynchronize costs up to milliseconds when several host threads wait on several streams)
  auto over = [&]() { return __atomic_load_n(c.h_done, __ATOMIC_ACQUIRE) >= (unsigned)seq; };
  if (!over()) {
    if (!wait) { *n_finished = -1; return MI_OSQP_OK; }
    const double t0 = now_s();
    for (int spin = 0; !over(); spin++) {
      if (spin < 2000) { __builtin_ia32_pause(); continue; }
      struct timespec ts{0, 20000};                      // 20 us
      nanosleep(&ts, nullptr);
      if ((spin & 1023) == 0 && now_s() - t0 > 5.0) { HIPCHK(hipEventSynchronize(c.ev[par])); if (!over()) { g_last_error = "advance launch ended without reporting"; return MI_OSQP_ERR_DEVICE; } }
    }
  }
  c.last_active = c.h_done[1];
  const int BT = h->BT;
  int64_t nf = 0;
  auto finished = [&](int q) {
    if (!c.running[q]) return false;
    const int *ti = c.h_is[par] + (size_t)(q / BT) * IS_COUNT * BT;
    const int b = q % BT;
    return ti[IS_DONE * BT + b] != 0 && ti[IS_PENDING * BT + b] == 0 && ti[IS_EPOCH * BT + b] == c.epoch[(size_t)q];
  };
  // (the caller's buffer must take every finished QP of this advance: with less room nothing is consumed)
  int64_t would = 0;
  for (int q = 0; q < h->B; q++) would += finished(q) ? 1 : 0;
  if (would > capacity || (would > 0 && !ids_out)) { *n_finished = would; g_last_error = "poll: ids_out too small"; return MI_OSQP_ERR_INVALID_DATA; }
  for (int q = 0; q < h->B; q++) {
    if (!finished(q)) continue;
    const int *ti = c.h_is[par] + (size_t)(q / BT) * IS_COUNT * BT;
    const int b = q % BT;
    const double *td = c.h_ds[par] + (size_t)(q / BT) * DS_COUNT * BT;
    mi_osqp_info &I = c.info[(size_t)q];
    I.iter = ti[IS_ITER * BT + b]; I.status_val = ti[IS_STATUS * BT + b]; I.exit_code = exit_code_of((int)I.status_val);
    I.obj_val = td[DS_OBJ * BT + b]; I.pri_res = td[DS_PRI_RES * BT + b]; I.dua_res = td[DS_DUA_RES * BT + b];
    I.rho_updates = ti[IS_RHO_UPDATES * BT + b]; I.rho_estimate = td[DS_RHO_EST * BT + b]; I.rho = td[DS_RHO * BT + b];
    if (ti[IS_NEED_REFACTOR * BT + b] < 0) h->failed[(size_t)q] = 1;
    c.running[q] = 0; c.n_running--;
    ids_out[nf++] = q;
  }
  c.polled_seq = seq;
  *n_finished = nf;
  return MI_OSQP_OK;
}

int mi_osqp_batch_get_primal_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, double *x_out) {
  if (!h || (n_ids > 0 && (!ids || !x_out))) return MI_OSQP_ERR_NULL;
  if (!h->cont.on) return MI_OSQP_ERR_INVALID_DATA;
  const size_t n = (size_t)(*h->anp).n;
  for (int64_t j = 0; j < n_ids; j++) {
    if (ids[j] < 0 || ids[j] >= h->B) return MI_OSQP_ERR_INVALID_DATA;
    memcpy(x_out + (size_t)j * n, h->cont.xh + (size_t)ids[j] * n, n * sizeof(double));
  }
  return MI_OSQP_OK;
}
int mi_osqp_batch_get_dual_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, double *y_out) {
  if (!h || (n_ids > 0 && (!ids || !y_out))) return MI_OSQP_ERR_NULL;
  if (!h->cont.on) return MI_OSQP_ERR_INVALID_DATA;
  const size_t m = (size_t)(*h->anp).m;
  for (int64_t j = 0; j < n_ids; j++) {
    if (ids[j] < 0 || ids[j] >= h->B) return MI_OSQP_ERR_INVALID_DATA;
    memcpy(y_out + (size_t)j * m, h->cont.yh + (size_t)ids[j] * m, m * sizeof(double));
  }
  return MI_OSQP_OK;
}
int mi_osqp_batch_get_info_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, mi_osqp_info *info) {
  if (!h || (n_ids > 0 && (!ids || !info))) return MI_OSQP_ERR_NULL;
  if (!h->cont.on) return MI_OSQP_ERR_INVALID_DATA;
  for (int64_t j = 0; j < n_ids; j++) {
    if (ids[j] < 0 || ids[j] >= h->B) return MI_OSQP_ERR_INVALID_DATA;
    info[j] = h->cont.info[(size_t)ids[j]];
  }
  return MI_OSQP_OK;
}
int64_t mi_osqp_batch_running(mi_osqp_batch *h) { return h && h->cont.on ? h->cont.n_running : 0; }

}  // extern "C"

// ------------------------------------------------------------------ gomp scene
// SURVEY 8(f) rank 2 on the device: the re-linearised 3-D / obstacle rows of the GOMP constraint matrix
// ([REF] src/constraints/constraint-builder.h:90-136) and the feasibility check that decides the SQP loop
// ([REF] src/gomp-solver.h:141-199), for balls whose kinematics are built-in models.  The scene keeps the raw constraint
// data of every QP of a batch handle on the device; re-linearising a QP rewrites its 3-D rows there (gomp_relinearise_kernel)
// and runs QPSolver::update from them - no A values cross PCIe in an SQP step, only the trajectory (n doubles) and one flag.
struct mi_gomp_scene {
  mi_osqp_batch *h = nullptr;
  int dims = 0, W = 0, n_balls = 0, n_lines = 0, row0 = 0, n_rows3d = 0;
  double con_lo[3] = {-1e30, -1e30, -1e30}, con_hi[3] = {1e30, 1e30, 1e30};
  DevBuf<GompBallDev> balls;
  DevBuf<GompLineDev> lines;
  DevBuf<int> aidx;
  DevBuf<double> A, l, u;                     // QP-major raw rows as ConstraintBuilder::build() lays them out
  int *h_ok = nullptr; size_t h_ok_cap = 0;   // the kernel's verdicts (pinned host memory)
  hipEvent_t ev = nullptr;
  ~mi_gomp_scene() { hostpool::give(h_ok, h_ok_cap); if (ev) (void)hipEventDestroy(ev); }
};

static int gomp_launch(mi_gomp_scene *sc, int64_t n_ids, const int64_t *ids, const double *x, int write_rows /* 0 never, 1 always, 2 unless accepted */, int32_t *ok_out) {
  mi_osqp_batch *h = sc->h;
  const Analysis &an = (*h->anp);
  const int n = an.n, m = an.m, nq = (int)n_ids;
  int rc;
  // trajectories and ids through the handle's staging ring (pinned, asynchronous), the verdicts straight into pinned host memory
  int *d_ids = nullptr;
  RingSpan sp;
  if ((rc = ring_reserve(h, (size_t)nq * sizeof(int) + (size_t)nq * n * sizeof(double), 2))) return rc;
  if ((rc = cont_stage_ids(h, n_ids, ids, &d_ids, nullptr)) || (rc = ring_take(h, (size_t)nq * n * sizeof(double), sp))) return rc;
  memcpy(sp.host, x, (size_t)nq * n * sizeof(double));
  if ((rc = ring_upload(h, sp, (size_t)nq * n * sizeof(double)))) return rc;
  hipStream_t st = h->cont.ustream;
  GompArgs g{};
  g.dims = sc->dims; g.W = sc->W; g.n_balls = sc->n_balls; g.n_lines = sc->n_lines; g.n = n; g.m = m; g.nnzA = an.Ap[n]; g.n_ids = nq;
  g.row0 = sc->row0; g.write_rows = write_rows;
  g.ids = d_ids; g.balls = sc->balls.p; g.lines = sc->lines.p; g.aidx = sc->aidx.p; g.traj = (const double *)sp.dev;
  for (int k = 0; k < 3; k++) { g.con_lo[k] = sc->con_lo[k]; g.con_hi[k] = sc->con_hi[k]; }
  g.A = sc->A.p; g.l = sc->l.p; g.u = sc->u.p; g.ok = sc->h_ok;
  HIPCHK(launch_gomp_relinearise(g, st));
  HIPCHK(hipEventRecord(sc->ev, st));
  HIPCHK(hipEventSynchronize(sc->ev));
  for (int j = 0; j < nq; j++) ok_out[j] = sc->h_ok[j];
  return MI_OSQP_OK;
}

extern "C" {

int mi_gomp_scene_create(mi_gomp_scene **out, mi_osqp_batch *h, int64_t dims, int64_t waypoints, int64_t n_balls, const mi_gomp_ball *balls,
                         int64_t n_lines, const mi_gomp_line *lines, const double *con_lo, const double *con_hi) {
  if (!out) return MI_OSQP_ERR_NULL;
  *out = nullptr;
  if (!h || (n_balls > 0 && !balls) || (n_lines > 0 && !lines)) return MI_OSQP_ERR_NULL;
  const Analysis &an = (*h->anp);
  const int D = (int)dims, W = (int)waypoints;
  if (D < 1 || D > 8 || W < 2 || n_balls < 0 || n_lines < 0 || an.n != 2 * D * W) return MI_OSQP_ERR_INVALID_DATA;
  // rows of the reference's layout ([REF] constraint-builder.h:34-44): links, position / velocity / acceleration boxes, then
  // D W (3 + |lines| |balls|) rows for the 3-D part, of which the populated ones come first
  const int row0 = (W - 1) * D + D * (W + W - 1 + W - 2);
  int rows3d = 0;
  for (int b = 0; b < (int)n_balls; b++) {
    if (balls[b].model < MI_GM_UR5E_FLANGE || balls[b].model > MI_GM_TABLE) return MI_OSQP_ERR_INVALID_DATA;
    if ((balls[b].model <= MI_GM_UR5E_ELBOW && D != 6) || (balls[b].model >= MI_GM_YAW_2LINK && D != 3)) return MI_OSQP_ERR_INVALID_DATA;
    rows3d += W * ((balls[b].is_gripper ? 3 : 0) + (int)n_lines);
  }
  if (row0 + rows3d > an.m) return MI_OSQP_ERR_INVALID_DATA;
  DevGuard guard(h->device);
  mi_gomp_scene *sc = new (std::nothrow) mi_gomp_scene();
  if (!sc) return MI_OSQP_ERR_ALLOC;
  std::unique_ptr<mi_gomp_scene> own(sc);
  sc->h = h; sc->dims = D; sc->W = W; sc->n_balls = (int)n_balls; sc->n_lines = (int)n_lines; sc->row0 = row0; sc->n_rows3d = rows3d;
  for (int k = 0; k < 3; k++) { sc->con_lo[k] = con_lo ? con_lo[k] : -1e30; sc->con_hi[k] = con_hi ? con_hi[k] : 1e30; }
  // where the entries of the 3-D rows sit in A's value array: row r of waypoint w holds D entries in the columns of q_w
  std::vector<int> aidx((size_t)rows3d * D, -1);
  {
    int r = row0;
    for (int b = 0; b < (int)n_balls; b++)
      for (int w = 0; w < W; w++)
        for (int k = 0; k < (balls[b].is_gripper ? 3 : 0) + (int)n_lines; k++, r++)
          for (int j = 0; j < D; j++) {
            const int col = w * D + j;
            const int *lo = an.Ai.data() + an.Ap[col], *hi = an.Ai.data() + an.Ap[col + 1];
            const int *it = std::lower_bound(lo, hi, r);
            if (it == hi || *it != r) { g_last_error = "the constraint matrix does not hold the 3-D rows of this scene"; return MI_OSQP_ERR_INVALID_DATA; }
            aidx[(size_t)(r - row0) * D + j] = (int)(it - an.Ai.data());
          }
  }
  std::vector<GompBallDev> hb((size_t)n_balls);
  for (int b = 0; b < (int)n_balls; b++) { hb[b].model = balls[b].model; hb[b].is_gripper = balls[b].is_gripper; hb[b].radius = balls[b].radius; for (int k = 0; k < 12; k++) hb[b].param[k] = balls[b].param[k]; }
  std::vector<GompLineDev> hl((size_t)n_lines);
  for (int li = 0; li < (int)n_lines; li++) {
    const double nrm = std::hypot(lines[li].dir[0], lines[li].dir[1]);
    if (!(nrm > 0.0)) return MI_OSQP_ERR_INVALID_DATA;
    hl[li].D[0] = lines[li].dir[0] / nrm; hl[li].D[1] = lines[li].dir[1] / nrm; hl[li].D[2] = 0.0;
    for (int k = 0; k < 3; k++) hl[li].A[k] = lines[li].point[k];
    hl[li].below = lines[li].below; hl[li].pad = 0;
  }
  int rc;
  if ((rc = sc->balls.upload(hb)) || (rc = sc->lines.upload(hl)) || (rc = sc->aidx.upload(aidx)) ||
      (rc = sc->A.alloc((size_t)h->B * std::max(an.Ap[an.n], 1))) || (rc = sc->l.alloc((size_t)h->B * std::max(an.m, 1))) || (rc = sc->u.alloc((size_t)h->B * std::max(an.m, 1)))) return rc;
  HIPCHK(hostpool::alloc((void **)&sc->h_ok, (size_t)h->B * sizeof(int), &sc->h_ok_cap));
  HIPCHK(hipEventCreateWithFlags(&sc->ev, hipEventDisableTiming));
  if (h->cont.keepA) { g_last_error = "the solver already has a scene"; return MI_OSQP_ERR_INVALID_DATA; }
  h->cont.keepA = sc->A.p; h->cont.keepl = sc->l.p; h->cont.keepu = sc->u.p;
  *out = own.release();
  return MI_OSQP_OK;
}

void mi_gomp_scene_free(mi_gomp_scene *sc) {
  if (!sc) return;
  DevGuard guard(sc->h->device);
  (void)hipStreamSynchronize(sc->h->stream);
  if (sc->h->cont.ustream) (void)hipStreamSynchronize(sc->h->cont.ustream);
  if (sc->h->cont.keepA == sc->A.p) sc->h->cont.keepA = sc->h->cont.keepl = sc->h->cont.keepu = nullptr;
  delete sc;
}

// the raw constraint data of the listed QPs as the host built them (ConstraintBuilder::build): kept on the device
int mi_gomp_scene_set_rows(mi_gomp_scene *sc, int64_t n_ids, const int64_t *ids, const double *Av, const double *l, const double *u) {
  if (!sc || (n_ids > 0 && (!ids || !Av || !l || !u))) return MI_OSQP_ERR_NULL;
  mi_osqp_batch *h = sc->h;
  DevGuard guard(h->device);
  const Analysis &an = (*h->anp);
  const size_t nnzA = (size_t)an.Ap[an.n], m = (size_t)an.m;
  hipStream_t st = h->cont.on && h->cont.ustream ? h->cont.ustream : h->stream;
  for (int64_t j = 0; j < n_ids; j++) {
    if (ids[j] < 0 || ids[j] >= h->B) return MI_OSQP_ERR_INVALID_DATA;
    HIPCHK(hipMemcpyAsync(sc->A.p + (size_t)ids[j] * nnzA, Av + (size_t)j * nnzA, nnzA * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sc->l.p + (size_t)ids[j] * m, l + (size_t)j * m, m * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sc->u.p + (size_t)ids[j] * m, u + (size_t)j * m, m * sizeof(double), hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipStreamSynchronize(st));          // (the caller's arrays are pageable)
  return MI_OSQP_OK;
}
int mi_gomp_scene_get_rows(mi_gomp_scene *sc, int64_t id, double *Av, double *l, double *u) {
  if (!sc) return MI_OSQP_ERR_NULL;
  mi_osqp_batch *h = sc->h;
  if (id < 0 || id >= h->B) return MI_OSQP_ERR_INVALID_DATA;
  DevGuard guard(h->device);
  const Analysis &an = (*h->anp);
  const size_t nnzA = (size_t)an.Ap[an.n], m = (size_t)an.m;
  if (Av) HIPCHK(hipMemcpy(Av, sc->A.p + (size_t)id * nnzA, nnzA * sizeof(double), hipMemcpyDeviceToHost));
  if (l) HIPCHK(hipMemcpy(l, sc->l.p + (size_t)id * m, m * sizeof(double), hipMemcpyDeviceToHost));
  if (u) HIPCHK(hipMemcpy(u, sc->u.p + (size_t)id * m, m * sizeof(double), hipMemcpyDeviceToHost));
  return MI_OSQP_OK;
}

// withObstacles(con_3d, x) for the listed QPs' kept rows, without touching the solver (tests; the first linearisation of a segment)
int mi_gomp_assemble_some(mi_gomp_scene *sc, int64_t n_ids, const int64_t *ids, const double *x, int32_t *ok_out) {
  if (!sc || (n_ids > 0 && (!ids || !x || !ok_out))) return MI_OSQP_ERR_NULL;
  if (!n_ids) return MI_OSQP_OK;
  DevGuard guard(sc->h->device);
  int rc;
  if ((rc = cont_enter(sc->h)) || (rc = cont_check_ids(sc->h, n_ids, ids, false))) return rc;
  return gomp_launch(sc, n_ids, ids, x, 1, ok_out);
}

// One SQP step of the listed (finished) QPs on the device: isSolutionOK of their solutions x; the QPs whose trajectory is not
// acceptable are re-linearised around it (their kept 3-D rows rewritten) and updated (= builder.withObstacles(con_3d, x) +
// QPSolver::update, [REF] src/gomp-solver.h:79-87) - they are ready for solve_begin_some.  ok_out[j] = 1: trajectory accepted.
int mi_gomp_relinearise_some(mi_gomp_scene *sc, int64_t n_ids, const int64_t *ids, const double *x, int32_t *ok_out) {
  CallTimer timer_("gomp_relinearise_some");
  if (!sc || (n_ids > 0 && (!ids || !x || !ok_out))) return MI_OSQP_ERR_NULL;
  if (!n_ids) return MI_OSQP_OK;
  mi_osqp_batch *h = sc->h;
  DevGuard guard(h->device);
  int rc;
  if ((rc = cont_enter(h)) || (rc = cont_check_ids(h, n_ids, ids, true))) return rc;
  // the acceptance test; the trajectories that fail it get new 3-D rows in the same launch (an accepted trajectory's rows
  // stay as they are: its solver is not updated) ...
  if ((rc = gomp_launch(sc, n_ids, ids, x, 2, ok_out))) return rc;
  std::vector<int64_t> bad;
  for (int64_t j = 0; j < n_ids; j++) if (!ok_out[j]) bad.push_back(ids[j]);
  if (bad.empty()) return MI_OSQP_OK;
  // ... and the update from the device-resident data
  return cont_new_data(h, (int64_t)bad.size(), bad.data(), nullptr, nullptr, nullptr, false, sc->A.p, sc->l.p, sc->u.p);
}

}  // extern "C"

extern "C" {

// ------------------------------------------------------------------ single QP

int mi_osqp_setup(mi_osqp_solver **out, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Pv,
                  const double *q, const int64_t *Ap, const int64_t *Ai, const double *Av, const double *l,
                  const double *u, const mi_osqp_settings *settings) {
  if (!out) return MI_OSQP_ERR_NULL;
  *out = nullptr;
  mi_osqp_batch *b = nullptr;
  int rc = mi_osqp_batch_setup(&b, 1, n, m, Pp, Pi, Pv, q, Ap, Ai, Av, l, u, settings, -1);
  if (rc) return rc;
  mi_osqp_solver *s = new (std::nothrow) mi_osqp_solver();
  if (!s) { delete b; return MI_OSQP_ERR_ALLOC; }
  s->b = b; *out = s;
  return MI_OSQP_OK;
}
void mi_osqp_free(mi_osqp_solver *h) { if (h) { delete h->b; delete h; } }
int mi_osqp_update_A_bounds(mi_osqp_solver *h, const int64_t *Ap, const int64_t *Ai, const double *Av, const double *l, const double *u) {
  return h ? mi_osqp_batch_update_A_bounds(h->b, Ap, Ai, Av, l, u) : MI_OSQP_ERR_NULL;
}
int mi_osqp_update_A(mi_osqp_solver *h, const int64_t *Ap, const int64_t *Ai, const double *Av) {
  return h ? mi_osqp_batch_update_A(h->b, Ap, Ai, Av) : MI_OSQP_ERR_NULL;
}
int mi_osqp_update_bounds(mi_osqp_solver *h, const double *l, const double *u) {
  return h ? mi_osqp_batch_update_bounds(h->b, l, u) : MI_OSQP_ERR_NULL;
}
int mi_osqp_warm_start_x(mi_osqp_solver *h, const double *x) { return h ? mi_osqp_batch_warm_start_x(h->b, x) : MI_OSQP_ERR_NULL; }
int mi_osqp_solve(mi_osqp_solver *h, mi_osqp_info *info) {
  if (!h) return MI_OSQP_ERR_NULL;
  int rc = mi_osqp_batch_solve(h->b);
  if (rc) return rc;
  if (info) return mi_osqp_batch_get_info(h->b, info);
  return MI_OSQP_OK;
}
int mi_osqp_get_primal(mi_osqp_solver *h, double *x) { return h ? mi_osqp_batch_get_primal(h->b, x) : MI_OSQP_ERR_NULL; }
int mi_osqp_get_dual(mi_osqp_solver *h, double *y) { return h ? mi_osqp_batch_get_dual(h->b, y) : MI_OSQP_ERR_NULL; }
int mi_osqp_get_stats(mi_osqp_solver *h, mi_osqp_stats *st) { return h ? mi_osqp_batch_get_stats(h->b, st) : MI_OSQP_ERR_NULL; }

// ------------------------------------------------------------ multi-GPU batch
// block partition of the batch over the devices, one host thread per shard, no data-path collective.

}  // extern "C"

// One long-lived worker thread per shard (a planner calls update / warm start / solve / get_* several times per SQP step:
// creating and joining a thread per shard and call cost ~0.1 ms of host time each).  A job is posted to every worker; the
// caller waits for all of them - or, for solve_async, comes back later (wait).
struct ShardWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, busy = false, quit = false;
  int rc = 0;
  std::string err;
  void start() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<int()> f = std::move(job);
        has_job = false;
        lk.unlock();
        const int r = f();
        const std::string e = r ? g_last_error : std::string();
        lk.lock();
        rc = r; err = e; busy = false;
        cv.notify_all();
      }
    });
  }
  void post(std::function<int()> f) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return !busy; });
    job = std::move(f); has_job = true; busy = true; rc = 0; err.clear();
    cv.notify_all();
  }
  int wait(std::string &e) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return !busy; });
    e = err;
    return rc;
  }
  ~ShardWorker() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

struct mi_osqp_multi {
  int64_t B = 0, n = 0, m = 0;
  std::vector<int64_t> dev, begin;                 // begin has one entry more than there are shards
  std::vector<mi_osqp_batch *> shard;
  std::vector<std::unique_ptr<ShardWorker>> worker;
  bool async_pending = false;                      // a solve_async whose wait() has not been called
  ~mi_osqp_multi() { worker.clear(); for (mi_osqp_batch *b : shard) delete b; }      // (the workers finish their jobs first)
  size_t count() const { return shard.size(); }
};

// wait for the jobs posted to the workers; first error wins, its text becomes this thread's last error
static int multi_join(mi_osqp_multi *h) {
  int rc = MI_OSQP_OK;
  for (size_t k = 0; k < h->worker.size(); k++) {
    std::string e;
    const int r = h->worker[k]->wait(e);
    if (r && !rc) { rc = r; g_last_error = "shard " + std::to_string(k) + ": " + e; }
  }
  h->async_pending = false;
  return rc;
}
// run fn(shard index) on the shard's worker thread and wait for all of them
template <class F>
static int multi_fan_out(mi_osqp_multi *h, F &&fn) {
  const size_t ns = h->count();
  if (h->async_pending) { const int rc0 = multi_join(h); if (rc0) return rc0; }
  if (ns == 1) { const int rc = fn(0); if (rc) g_last_error = "shard 0: " + g_last_error; return rc; }
  if (h->worker.size() != ns) {
    h->worker.clear();
    for (size_t k = 0; k < ns; k++) { h->worker.emplace_back(new ShardWorker()); h->worker.back()->start(); }
  }
  for (size_t k = 0; k < ns; k++) h->worker[k]->post([&fn, k]() -> int { return fn(k); });
  return multi_join(h);
}

extern "C" {

int mi_osqp_multi_batch_setup(mi_osqp_multi **out, int64_t n_devices, const int64_t *devices, int64_t B, int64_t n, int64_t m,
                              const int64_t *Pp, const int64_t *Pi, const double *Pv, const double *q, const int64_t *Ap,
                              const int64_t *Ai, const double *Av, const double *l, const double *u,
                              const mi_osqp_settings *settings) {
  if (!out) return MI_OSQP_ERR_NULL;
  *out = nullptr;
  if (n_devices <= 0 || B <= 0 || n <= 0 || m < 0 || !Pp || !Ap) return MI_OSQP_ERR_INVALID_DATA;
  if (n_devices > B) n_devices = B;                 // never an empty shard
  mi_osqp_multi *h = new (std::nothrow) mi_osqp_multi();
  if (!h) return MI_OSQP_ERR_ALLOC;
  h->B = B; h->n = n; h->m = m;
  const int64_t base = B / n_devices, rem = B % n_devices;
  h->begin.push_back(0);
  for (int64_t k = 0; k < n_devices; k++) {
    h->dev.push_back(devices ? devices[k] : k);
    h->begin.push_back(h->begin.back() + base + (k < rem ? 1 : 0));
  }
  h->shard.assign((size_t)n_devices, nullptr);
  const int64_t nnzP = Pp[n], nnzA = Ap[n];
  int rc = multi_fan_out(h, [&](size_t k) -> int {
    const int64_t b0 = h->begin[k], nb = h->begin[k + 1] - b0;
    return mi_osqp_batch_setup(&h->shard[k], nb, n, m, Pp, Pi, Pv ? Pv + b0 * nnzP : nullptr, q ? q + b0 * n : nullptr, Ap, Ai,
                               Av ? Av + b0 * nnzA : nullptr, l ? l + b0 * m : nullptr, u ? u + b0 * m : nullptr, settings, h->dev[k]);
  });
  if (rc) { delete h; return rc; }
  *out = h;
  return MI_OSQP_OK;
}

void mi_osqp_multi_batch_free(mi_osqp_multi *h) { delete h; }
int64_t mi_osqp_multi_batch_shards(mi_osqp_multi *h) { return h ? (int64_t)h->count() : 0; }
int mi_osqp_multi_batch_shard(mi_osqp_multi *h, int64_t k, int64_t *device, int64_t *begin, int64_t *end, mi_osqp_batch **handle) {
  if (!h) return MI_OSQP_ERR_NULL;
  if (k < 0 || k >= (int64_t)h->count()) return MI_OSQP_ERR_INVALID_DATA;
  if (device) *device = h->shard[k]->device;
  if (begin) *begin = h->begin[k];
  if (end) *end = h->begin[k + 1];
  if (handle) *handle = h->shard[k];
  return MI_OSQP_OK;
}
int mi_osqp_multi_batch_update_A(mi_osqp_multi *h, const int64_t *Ap, const int64_t *Ai, const double *Av) {
  if (!h || !Ap || !Ai || !Av) return MI_OSQP_ERR_NULL;
  const int64_t nnzA = Ap[h->n];
  return multi_fan_out(h, [&](size_t k) { return mi_osqp_batch_update_A(h->shard[k], Ap, Ai, Av + h->begin[k] * nnzA); });
}
int mi_osqp_multi_batch_update_bounds(mi_osqp_multi *h, const double *l, const double *u) {
  if (!h || !l || !u) return MI_OSQP_ERR_NULL;
  return multi_fan_out(h, [&](size_t k) { return mi_osqp_batch_update_bounds(h->shard[k], l + h->begin[k] * h->m, u + h->begin[k] * h->m); });
}
int mi_osqp_multi_batch_update_A_bounds(mi_osqp_multi *h, const int64_t *Ap, const int64_t *Ai, const double *Av, const double *l, const double *u) {
  if (!h || !Ap || !Ai || !Av || !l || !u) return MI_OSQP_ERR_NULL;
  const int64_t nnzA = Ap[h->n];
  return multi_fan_out(h, [&](size_t k) {
    return mi_osqp_batch_update_A_bounds(h->shard[k], Ap, Ai, Av + h->begin[k] * nnzA, l + h->begin[k] * h->m, u + h->begin[k] * h->m);
  });
}
int mi_osqp_multi_batch_warm_start_x(mi_osqp_multi *h, const double *x) {
  if (!h || !x) return MI_OSQP_ERR_NULL;
  return multi_fan_out(h, [&](size_t k) { return mi_osqp_batch_warm_start_x(h->shard[k], x + h->begin[k] * h->n); });
}
int mi_osqp_multi_batch_solve(mi_osqp_multi *h) {
  if (!h) return MI_OSQP_ERR_NULL;
  return multi_fan_out(h, [&](size_t k) { return mi_osqp_batch_solve(h->shard[k]); });
}
// solve on every shard without waiting: the call returns once the jobs are with the shard workers; wait() joins them and
// reports the first error.  Any other multi-batch call joins a pending solve first.
int mi_osqp_multi_batch_solve_async(mi_osqp_multi *h) {
  if (!h) return MI_OSQP_ERR_NULL;
  if (h->async_pending) { const int rc0 = multi_join(h); if (rc0) return rc0; }
  const size_t ns = h->count();
  if (h->worker.size() != ns) {
    h->worker.clear();
    for (size_t k = 0; k < ns; k++) { h->worker.emplace_back(new ShardWorker()); h->worker.back()->start(); }
  }
  for (size_t k = 0; k < ns; k++) { mi_osqp_batch *b = h->shard[k]; h->worker[k]->post([b]() -> int { return mi_osqp_batch_solve(b); }); }
  h->async_pending = true;
  return MI_OSQP_OK;
}
int mi_osqp_multi_batch_wait(mi_osqp_multi *h) {
  if (!h) return MI_OSQP_ERR_NULL;
  return h->async_pending ? multi_join(h) : (int)MI_OSQP_OK;
}
int mi_osqp_multi_batch_get_primal(mi_osqp_multi *h, double *x) {
  if (!h || !x) return MI_OSQP_ERR_NULL;
  return multi_fan_out(h, [&](size_t k) { return mi_osqp_batch_get_primal(h->shard[k], x + h->begin[k] * h->n); });
}
int mi_osqp_multi_batch_get_dual(mi_osqp_multi *h, double *y) {
  if (!h || !y) return MI_OSQP_ERR_NULL;
  return multi_fan_out(h, [&](size_t k) { return mi_osqp_batch_get_dual(h->shard[k], y + h->begin[k] * h->m); });
}
int mi_osqp_multi_batch_get_info(mi_osqp_multi *h, mi_osqp_info *info) {
  if (!h || !info) return MI_OSQP_ERR_NULL;
  return multi_fan_out(h, [&](size_t k) { return mi_osqp_batch_get_info(h->shard[k], info + h->begin[k]); });
}

// ------------------------------------------------------ host-only diagnostics

int mi_osqp_debug_host_kkt_solve(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Pv,
                                 const int64_t *Ap, const int64_t *Ai, const double *Av, const double *l,
                                 const double *u, const mi_osqp_settings *settings, int64_t tile, const double *rhs,
                                 double *sol_schedule, double *sol_direct, mi_osqp_stats *st) {
  Settings s = to_settings(settings);
  if (validate_settings(s)) return MI_OSQP_ERR_INVALID_SETTINGS;
  Analysis an;
  // tile > 1: the dataflow form for tile - 1 workgroups per QP (global solve vector, no dense tail)
  int rc = tile > 1 ? analyze(n, m, Pp, Pi, Ap, Ai, an, 8, 1, -1, 0, (int)tile - 1) : analyze(n, m, Pp, Pi, Ap, Ai, an);
  if (rc) return rc;
  QPNumeric Q;
  load_qp(an, s, Pv, nullptr, Av, l, u, Q);
  if (s.scaling) scale_qp(an, s, Q);
  set_rho_vec(an, s, Q);
  std::vector<double> w;
  if ((rc = factor_qp(an, s, Q, w))) return rc;
  if (sol_direct) direct_kkt_solve(an, Q, rhs, sol_direct);
  if (sol_schedule && !replay_kkt_solve(an, Q, rhs, sol_schedule)) {
    g_last_error = "schedule streams are not race-free / deadlock-free"; return MI_OSQP_ERR_INVALID_DATA;
  }
  if (st) {
    memset(st, 0, sizeof(*st));
    st->n = n; st->m = m; st->N = an.N; st->batch = 1; st->tile = 1; st->n_tiles = 1;
    st->nnz_P_triu = an.Pp[n]; st->nnz_A = an.Ap[n]; st->nnz_KKT = an.nnzK(); st->nnz_L = an.nnzL();
    st->n_supernodes = (int64_t)an.sn_start.size() - 1; st->n_blocks = (int64_t)an.chunk_start.size() - 1;
    st->fwd_levels = an.fwd.n_phases; st->bwd_levels = an.bwd.n_phases;
    st->fwd_slots = (int64_t)an.fwd.phys_steps() * 64; st->bwd_slots = (int64_t)an.bwd.phys_steps() * 64; st->chk_slots = (int64_t)an.chk.phys_steps() * 64;
    st->dense_tail_rows = an.dt.k; st->dense_tail_slots = (int64_t)an.dt.n_steps * 64;
    st->nnz_L_before_tail = an.dt.k ? an.Lp[an.dt.s] : an.nnzL();
  }
  return MI_OSQP_OK;
}

int mi_osqp_debug_host_block_factor(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Pv,
                                    const int64_t *Ap, const int64_t *Ai, const double *Av, const double *l,
                                    const double *u, const mi_osqp_settings *settings, double *dL, double *dD,
                                    int64_t *counts) {
  Settings s = to_settings(settings);
  if (validate_settings(s)) return MI_OSQP_ERR_INVALID_SETTINGS;
  Analysis an;
  int rc = analyze(n, m, Pp, Pi, Ap, Ai, an);
  if (rc) return rc;
  QPNumeric Q, R;
  load_qp(an, s, Pv, nullptr, Av, l, u, Q);
  if (s.scaling) scale_qp(an, s, Q);
  set_rho_vec(an, s, Q);
  std::vector<double> w;
  if ((rc = factor_qp(an, s, Q, w))) return rc;
  if ((rc = replay_block_factor(an, s, Q, R))) return rc;
  double mL = 0.0, sL = 1e-300, mD = 0.0;
  // with a dense tail the device factor holds L only for the columns before it (and S^-1 instead of the rest)
  const int ts = an.dt.k ? an.dt.s : an.N;
  for (int k = 0; k < an.Lp[ts]; k++) { mL = std::max(mL, std::fabs(Q.Lx[k] - R.Lx[k])); sL = std::max(sL, std::fabs(Q.Lx[k])); }
  for (size_t k = (size_t)an.nnzL(); k < Q.Lx.size(); k++) { mL = std::max(mL, std::fabs(Q.Lx[k] - R.Lx[k])); sL = std::max(sL, std::fabs(Q.Lx[k])); }
  for (int k = 0; k < ts; k++) mD = std::max(mD, std::fabs(Q.Dlinv[k] - R.Dlinv[k]) / std::fabs(Q.Dlinv[k]));
  if (an.dt.k) {
    double mM = 0.0, sM = 1e-300;
    for (size_t k = 0; k < Q.Minv.size(); k++) { mM = std::max(mM, std::fabs(Q.Minv[k] - R.Minv[k])); sM = std::max(sM, std::fabs(Q.Minv[k])); }
    mL = std::max(mL, mM / sM * sL);
  }
  if (dL) *dL = mL / sL;
  if (dD) *dD = mD;
  if (counts) {
    counts[0] = (int64_t)an.bf.n_blocks(); counts[1] = (int64_t)an.bf.tri.size() / 2;
    counts[2] = an.bf.storage; counts[3] = an.bf.n_levels;
  }
  return MI_OSQP_OK;
}

}  // extern "C"
