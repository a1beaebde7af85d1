// kernels.hip -- hand-written gfx950 kernels of the OSQP ADMM iterate
// (SURVEY.md section 8(a) rows E6-E14).  fp64 throughout, no MFMA: the path is
// sparse and HBM-bound.
//
// Execution model (MI355X-first, not a translation of anything):
//   * B QPs share ONE sparsity pattern; they are cut into tiles of BT QPs whose
//     values are interleaved innermost ([entry][BT]) so that one wave64 load
//     instruction streams 64 consecutive entries x BT doubles (512 B .. 2 KiB).
//   * ONE workgroup owns ONE tile for the WHOLE solve: QPs are independent, so
//     no grid-wide synchronisation ever exists; the KKT solve vector lives in
//     LDS (N*BT doubles), the factor streams from HBM in schedule order.
//   * The triangular solves are "pull" schedules built on the host
//     (host_core.cpp): every wave walks one linear stream of wave-steps per
//     sweep; per elimination level, phase A = steps that gather from the LDS
//     vector and reduce groups of T lanes into one target row, phase B = the
//     in-chunk triangle applied as a product with the inverted diagonal block
//     (same kind of step, "store" flavour); phase boundaries are barrier counts
//     in the step descriptors (sched_format.h).
//   * The same row-task machinery evaluates P x, A' y and A x for the residuals.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "device_types.h"
#include "sched_format.h"

namespace miosqp {

static hipError_t ensure_dynamic_lds(const void *kern, size_t lds);      // (launchers, bottom of the file)

#define MI_INFTY 1e30
#define MI_MIN_SCALING 1e-4
#define MI_DIV_TOL 1e-30
#define MI_RHO_MIN 1e-6
#define MI_RHO_MAX 1e6
#define MI_NOROW 0xFFFFFFFFu
// Timing experiments that skip parts of the refactorisation (wrong results by design) exist only in a diagnostic build
// (MI_OSQP_CXXFLAGS=-DMI_OSQP_DEBUG_BUILD, scripts/profile_factor.py); the product binary has no such switch.
#ifdef MI_OSQP_DEBUG_BUILD
#define MI_DBG_SKIP(a) ((a).debug_skip)
// fault injection (diagnostic build): MI_OSQP_DEBUG_DROP_GROUP names the launch that loses a workgroup of its grid
// (iterate / check / kkt / factor; "1" = iterate)
static bool debug_drop_group(const char *which) {
  const char *e = getenv("MI_OSQP_DEBUG_DROP_GROUP");
  return e && (!strcmp(e, which) || (!strcmp(e, "1") && !strcmp(which, "iterate")));
}
#else
#define MI_DBG_SKIP(a) 0
#endif

template <int BT>
struct VecBT;
template <>
struct VecBT<1> { double v[1]; };
template <>
struct VecBT<2> { double v[2]; };
template <>
struct VecBT<4> { double v[4]; };

template <int BT>
__device__ __forceinline__ void load_bt(const double *__restrict__ p, double (&o)[BT]) {
  if constexpr (BT == 1) {
    o[0] = p[0];
  } else if constexpr (BT == 2) {
    double2 t = *reinterpret_cast<const double2 *>(p);
    o[0] = t.x; o[1] = t.y;
  } else {
    double2 t0 = reinterpret_cast<const double2 *>(p)[0];
    double2 t1 = reinterpret_cast<const double2 *>(p)[1];
    o[0] = t0.x; o[1] = t0.y; o[2] = t1.x; o[3] = t1.y;
  }
}

// buffer (SRSRC) loads: wave-uniform base in SGPRs + a 32-bit per-lane offset + a
// scalar step offset, so the deep prefetch below costs no 64-bit address VGPRs
typedef unsigned int mi_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mi_u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t mi_rsrc;
__device__ __forceinline__ mi_rsrc make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double shfl_down_d(double v, int d) { return __shfl_down(v, d, 64); }

// DPP move of a double (both halves); invalid source lanes read 0
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#define DPP_XOR1 0xB1       /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E       /* quad_perm [2,3,0,1] */
#define DPP_SHL(n) (0x100 + (n))   /* row_shl:n -- lane i reads lane i+n of its 16-lane row */

// Schedule tables are read with wave-uniform addresses.  Going through the constant
// address space makes them SMEM loads (lgkmcnt), so waiting for a table entry never
// drains the vector-memory queue that holds the value prefetches (vmcnt is in-order).
typedef const uint32_t __attribute__((address_space(4))) *mi_cptr;
__device__ __forceinline__ mi_cptr as_const(const uint32_t *p) { return (mi_cptr)(uintptr_t)p; }

// LDS-only workgroup barrier: waits for this wave's LDS traffic but NOT for its
// outstanding global loads, so register prefetches stay in flight across it
// (a __syncthreads() would drain vmcnt to 0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// value of a slot from its source code (host_core.hpp Schedule::src composed with a position map)
__device__ __forceinline__ double slot_value(int32_t mp, const double *src, size_t stride, int b) {
  return mp >= 0 ? src[(size_t)mp * stride + b] : (mp == MI_SRC_ONE ? 1.0 : 0.0);
}

// ----------------------------------------------------------------- row steps
// Reduce acc[] over aligned groups of T = 2^lt lanes and apply it to the group's
// target row (`row`: every lane of a group carries it, 0xFFFF = none).  The BT
// per-QP partial sums are "transposed" across lanes on the way (after log2(BT)
// steps every lane carries ONE QP's partial sum), then summed toward the first
// lanes of the group with DPP row shifts; the two cross-row steps (16, 32) use the
// gfx950 v_permlane16_swap / v_permlane32_swap -- no LDS round trip anywhere.
// sub = true : base[row] = old - sum (old = the row's previous value, fetched by
// flush_prefetch before the reduction) ; false: base[row] = sum   (wave-uniform)

// value of lane + 16 for rows 0 and 2 / of lane + 32 for lanes 0..31 (other lanes: don't care)
__device__ __forceinline__ double down16_d(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ double down32_d(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)rh[1], (int)rl[1]);
}

// which QP component a lane ends up carrying after the transposition, and how many writer lanes a group has
template <int BT>
__device__ __forceinline__ int flush_q(int lane) {
  if constexpr (BT == 4) return ((lane & 1) ? 2 : 0) + ((lane & 2) ? 1 : 0);
  else if constexpr (BT == 2) return lane & 1;
  else return 0;
}
template <int BT, uint32_t NONE = 0xFFFFu>
__device__ __forceinline__ void flush_prefetch(uint32_t lt, uint32_t row, const double *base, int lane, double (&oldv)[BT]) {
#pragma unroll
  for (int b = 0; b < BT; b++) oldv[b] = 0.0;
  if (row == NONE) return;
  const double *dst = base + (size_t)row * BT;
  if (lt == 0) { load_bt<BT>(dst, oldv); return; }
  if constexpr (BT == 4) {
    if (lt == 1) { const double2 t = *reinterpret_cast<const double2 *>(dst + ((lane & 1) ? 2 : 0)); oldv[0] = t.x; oldv[1] = t.y; return; }
  }
  if (((uint32_t)lane & ((1u << lt) - 1u)) < (uint32_t)BT) oldv[0] = dst[flush_q<BT>(lane)];
}

template <int BT, uint32_t NONE = 0xFFFFu>
__device__ __forceinline__ void reduce_write(double (&acc)[BT], uint32_t lt, uint32_t row, double *base, int lane, bool sub,
                                             const double (&oldv)[BT]) {
  const bool has_row = row != NONE;
  if (lt == 0) {
    if (has_row) {
      double *dst = base + (size_t)row * BT;
#pragma unroll
      for (int b = 0; b < BT; b++) dst[b] = sub ? oldv[b] - acc[b] : acc[b];
    }
    return;
  }
  const uint32_t T = 1u << lt;
  double kp;
  if constexpr (BT == 4) {
    const bool o0 = lane & 1;
    const double s0 = o0 ? acc[0] : acc[2], s1 = o0 ? acc[1] : acc[3];
    double k0 = o0 ? acc[2] : acc[0], k1 = o0 ? acc[3] : acc[1];
    k0 += dpp_d<DPP_XOR1>(s0); k1 += dpp_d<DPP_XOR1>(s1);
    if (lt == 1) {                       // groups of 2 lanes: even lane owns QPs 0,1 ; odd lane QPs 2,3
      if (has_row) {
        double *dst = base + (size_t)row * BT + (o0 ? 2 : 0);
        dst[0] = sub ? oldv[0] - k0 : k0; dst[1] = sub ? oldv[1] - k1 : k1;
      }
      return;
    }
    const bool o1 = lane & 2;
    const double sd = o1 ? k0 : k1;
    kp = o1 ? k1 : k0;
    kp += dpp_d<DPP_XOR2>(sd);
    if (lt > 2) kp += dpp_d<DPP_SHL(4)>(kp);
    if (lt > 3) kp += dpp_d<DPP_SHL(8)>(kp);
  } else if constexpr (BT == 2) {
    const bool o0 = lane & 1;
    const double sd = o0 ? acc[0] : acc[1];
    kp = o0 ? acc[1] : acc[0];
    kp += dpp_d<DPP_XOR1>(sd);
    if (lt > 1) kp += dpp_d<DPP_SHL(2)>(kp);
    if (lt > 2) kp += dpp_d<DPP_SHL(4)>(kp);
    if (lt > 3) kp += dpp_d<DPP_SHL(8)>(kp);
  } else {
    kp = acc[0];
    kp += dpp_d<DPP_SHL(1)>(kp);
    if (lt > 1) kp += dpp_d<DPP_SHL(2)>(kp);
    if (lt > 2) kp += dpp_d<DPP_SHL(4)>(kp);
    if (lt > 3) kp += dpp_d<DPP_SHL(8)>(kp);
  }
  if (lt > 4) kp += down16_d(kp);
  if (lt > 5) kp += down32_d(kp);
  if (has_row && ((uint32_t)lane & (T - 1)) < (uint32_t)BT)
    base[(size_t)row * BT + flush_q<BT>(lane)] = sub ? oldv[0] - kp : kp;
}

// ---- per-wave step streams with a register ring -------------------------------------
// Every wave walks ONE linear stream of wave-steps per schedule (host_core.hpp Schedule,
// sched_format.h).  The next PF steps of the stream live in a register ring: right after
// step q has been consumed, its registers are refilled with step q + PF.  So PF steps
// (PF KiB at BT = 2) per wave are in flight at all times, across phase boundaries -- a
// phase boundary is just a number of LDS-only barriers the descriptor of the next step
// asks for.  No vector load other than the stream itself is issued on this path (VMEM
// returns in order: waiting for anything else would drain the ring); descriptors ride
// in one VGPR per ring revolution (lane st = step st) and are read with v_readlane,
// target rows ride in the high half of the index word.  The refill is one unconditional
// straight-line site: reads past the end of a stream hit the next wave's steps or the
// buffer bound (which returns 0) and are never consumed, so there is no branch around a
// load and the compiler keeps ONE copy of the ring.
// Ring depth of the sweeps (steps in flight per wave; a multiple of MI_D_LOOKAHEAD + 1).  Round 2 sweep on the 1024-QP
// headline batch (scripts/tune_rings.sh, ms per step): 15 / 16 (sweeps / dense-tail product): 38.8; 9 / 8: 37.2; 6 / 8: 37.3;
// 9 / 16: 37.8 - a many-stream read kernel reaches the full 6.1-6.4 TB/s with 64 KB in flight per CU
// (scripts/probes/stream_probe.hip), the shallower rings cost 88 instead of 119 VGPRs and start up faster after a barrier.
#ifndef MI_PFV
#define MI_PFV 9
#endif
// (a ring twice as deep for the global-vector mode of the large single QPs was measured in round 2: no change - 8.97 ms
// per iteration at config 5 either way; that mode is bound by the global gathers / read-modify-writes of the solve vector
// and the full drains at its phase barriers, not by the value stream)
#define MI_PFV_OF(BT, GX) MI_PFV
// 16 waves per tile = a tile that has its CU to itself (no more tiles than CUs: lone QPs, GOMP batches of <= 256, the QPs
// still iterating in a continuous batch): the factor streams come from L2 after the first iteration and every sweep is a
// chain of phases with about one step per wave, so what counts is how fast a ring starts up after it was emptied - measured
// in round 3 (scripts/tail_latency_probe.py, us per iteration, config 2 / 256 x 7 DOF x 100 waypoints): ring of 18: 30.1 /
// 47.3, 15: 27.8 / 45.0, 9: 23.9 / 41.4, 6: 22.3 / 39.8, 3: 21.7 / 42.6.
#ifndef MI_PFV_LAT
#define MI_PFV_LAT 6
#endif
#define MI_PFV_NT(NT, GX) (((NT) > 512 && !(GX)) ? MI_PFV_LAT : MI_PFV)
template <int BT, int PF>
struct Ring { double v[PF][BT]; uint32_t gi[PF]; uint32_t gr[PF]; uint32_t desc; };     // gr: target rows of the wide index words (unused otherwise)
// The value streams of a tile: ONE stream per QP ([slot][step][64] doubles, 8 B per lane and load) plus the shared
// index words / descriptors.  A QP that has finished (or a padding slot) gets a null descriptor: its loads
// return 0 without touching memory, so a half-done tile streams half the bytes.
template <int BT>
struct ValSrc { mi_rsrc vals[BT], idx, step; };

template <int BT, bool WIDE>
__device__ __forceinline__ void load_step(const ValSrc<BT> &vs, uint32_t stepno, int lane, double (&v)[BT], uint32_t &gi, uint32_t &gr) {
  if constexpr (WIDE) {
    const mi_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(vs.idx, (uint32_t)lane * 8u, stepno * 512u, 0);
    gi = t.x; gr = t.y;
  } else {
    gi = __builtin_amdgcn_raw_buffer_load_b32(vs.idx, (uint32_t)lane * 4u, stepno * 256u, 0);
  }
#pragma unroll
  for (int b = 0; b < BT; b++) {
    const mi_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(vs.vals[b], (uint32_t)lane * 8u, stepno * 512u, 0);
    v[b] = __hiloint2double((int)t.y, (int)t.x);
  }
}
// lane st gets the descriptor of step pos + st, a no-op at and past `end`
template <int BT>
__device__ __forceinline__ uint32_t load_desc(const ValSrc<BT> &vs, uint32_t pos, uint32_t end, int lane) {
  const uint32_t d = __builtin_amdgcn_raw_buffer_load_b32(vs.step, (uint32_t)lane * 4u, pos * 4u, 0);
  return pos + (uint32_t)lane < end ? d : MI_D_NOOP;
}

// Walk the stream [begin, end) of this wave, then pass `tail` more barriers.
//   SUB  = true : triangular solves (xs[row] -= sum, or xs[row] = sum on store steps); false: out[row] = sum (SpMV)
//   BAR  = the schedule has barriers (GX: full __syncthreads, the vector is in global memory)
//   TR   = debug instantiations: 1: lane 0 logs the shader clock before / after every barrier into
//          tr[(ordinal of the barrier * nw + wave) * 2 + {0, 1}]; 2: also the time spent waiting for ring slots
template <int BT, int PF, bool SUB, bool BAR, bool GX, int TR = 0, bool WIDE = false>
__device__ __forceinline__ void run_stream(const ValSrc<BT> &vs, uint32_t begin, uint32_t end, uint32_t tail, double *xs,
                                           double *out, int lane, uint32_t *tr = nullptr, int wave = 0, int nw = 0,
                                           uint32_t *tw = nullptr) {
  double *base = SUB ? xs : out;
  // The ring starts empty (all no-ops) and the first revolution only fills it: there is no separate prologue
  // whose load order the compiler could permute -- with one, the wait counts of the loop (merged over both
  // entries) collapse to nearly vmcnt(0) and the ring drains at every step.
  Ring<BT, PF> r;
  r.desc = MI_D_NOOP;
#pragma unroll
  for (int st = 0; st < PF; st++) {
    r.gi[st] = 0u; r.gr[st] = 0u;
#pragma unroll
    for (int b = 0; b < BT; b++) r.v[st][b] = 0.0;
  }
  constexpr uint32_t NONE = WIDE ? 0xFFFFFFFFu : 0xFFFFu;
  static_assert(PF % (MI_D_LOOKAHEAD + 1) == 0, "the gather slots must line up across ring revolutions");
  constexpr int GS = MI_D_LOOKAHEAD + 1;      // gather slots: step q uses slot q % GS
  double acc[BT], xq[GS][BT];
#pragma unroll
  for (int b = 0; b < BT; b++) {
    acc[b] = 0.0;
#pragma unroll
    for (int g = 0; g < GS; g++) xq[g][b] = 0.0;
  }
  uint32_t nbar_seen = 0, wait_cycles = 0, real_steps = 0;
  auto barrier = [&]() {
    if constexpr (TR) { if (lane == 0) tr[((size_t)nbar_seen * nw + wave) * 2] = (uint32_t)__builtin_amdgcn_s_memtime(); }
    if constexpr (GX) __syncthreads(); else lds_barrier();
    if constexpr (TR) { if (lane == 0) tr[((size_t)nbar_seen * nw + wave) * 2 + 1] = (uint32_t)__builtin_amdgcn_s_memtime(); nbar_seen++; }
  };
  for (uint32_t npos = begin; npos < end + PF; npos += PF) {   // this revolution runs steps npos - PF + st, loads npos + st
    const uint32_t dnext = load_desc(vs, npos, end, lane);     // older than every refill below
#pragma unroll
    for (int st = 0; st < PF; st++) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)r.desc, st);
      if constexpr (BAR) { for (uint32_t nb = MI_D_NBAR(d); nb; nb--) barrier(); }
      const uint32_t type = MI_D_TYPE(d);
      if constexpr (TR >= 2) {       // time spent waiting for the ring slot of this step (exact count: 2 loads per younger slot + dnext)
        if (type != 3u) {
          const uint32_t t0 = (uint32_t)__builtin_amdgcn_s_memtime();
          asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
          wait_cycles += (uint32_t)__builtin_amdgcn_s_memtime() - t0;
          real_steps++;
        }
      }
      if (type == MI_D_TYPE_ROW) {
        const uint32_t w = r.gi[st];
        // Vector gathers run MI_D_LOOKAHEAD steps ahead of their fma (flags set by the host: no barrier in
        // between).  Inside one phase no step reads what another one writes (pull schedule; checked by the
        // host replay), so a gather may pass the flushes of the steps before it.
        if (!(d & MI_D_PRE)) load_bt<BT>(xs + (size_t)(WIDE ? w : (w & 0xFFFFu)) * BT, xq[st % GS]);
        const bool sub = SUB && !(d & MI_D_STORE);
        const bool flush = d & MI_D_FLUSH;
        // the old value of the target row (read-modify-write) is fetched before the reduction starts
        double oldv[BT];
        const uint32_t lt = MI_D_LT(d), row = WIDE ? r.gr[st] : (w >> 16);
        if (flush && sub) flush_prefetch<BT, NONE>(lt, row, base, lane, oldv);
        if (d & MI_D_AHEAD) {
          const int s2 = (st + MI_D_LOOKAHEAD) % PF;         // past the end of this revolution: the slot was refilled already
          load_bt<BT>(xs + (size_t)(WIDE ? r.gi[s2] : (r.gi[s2] & 0xFFFFu)) * BT, xq[(st + MI_D_LOOKAHEAD) % GS]);
        }
#pragma unroll
        for (int b = 0; b < BT; b++) acc[b] = fma(r.v[st][b], xq[st % GS][b], acc[b]);
        if (flush) {
          reduce_write<BT, NONE>(acc, lt, row, base, lane, sub, oldv);
#pragma unroll
          for (int b = 0; b < BT; b++) acc[b] = 0.0;
        }
      }
      // ring refill: this slot now carries step npos + st
      load_step<BT, WIDE>(vs, npos + (uint32_t)st, lane, r.v[st], r.gi[st], r.gr[st]);
    }
    r.desc = dnext;
  }
  if constexpr (BAR) { for (uint32_t nb = tail; nb; nb--) barrier(); }
  if constexpr (TR >= 2) { if (lane == 0) { tw[2 * wave] = wait_cycles; tw[2 * wave + 1] = real_steps; } }
}

// Triangular solve with the solve vector in GLOBAL memory (GX: KKT systems that do not fit LDS - the large single QPs).
// Same streams, same arithmetic and order as run_stream; what differs is when the vector is touched.  There, gathers run
// 2 steps ahead and a flush step reads the old value of its target row right before the read-modify-write: with LDS
// that costs nothing, with global memory every gather exposes most of an L2 round trip (VMEM returns in order, behind
// the value ring) and every flush step - half of the steps - drains the whole queue (`vmcnt(0)`): 0.55 us per step at
// config 5.  Here BOTH the gathered values and the old values of the flush targets are requested LA = 8 steps ahead
// (index words and descriptors of those steps are in the ring already: PF = 18 > LA), and whatever was requested across
// a phase barrier is requested again right after it in one burst (another wave may have written those entries before
// the barrier; within a phase no step reads what another one writes - host replay).
#define MI_GX_PF 18
#define MI_GX_LA 8
// Barrier of the multi-workgroup mode: G workgroups (one per CU, all resident: G <= 64 on 256 CUs) share one QP.  The
// recipe of MI355X_MICROARCH.md (inter-workgroup visibility): every storing wave has drained its stores (__syncthreads
// waits vmcnt(0)), lane 0 releases at agent scope (writes the XCD's dirty L2 lines back), arrives on a monotonic counter,
// polls the generation word with relaxed loads + s_sleep, acquires at agent scope (invalidates this CU's L1) and lets its
// workgroup go.  A wait of more than ~2 s raises the error word instead of hanging the GPU (every workgroup then runs out
// the same way; the host reports a device error).
struct Mw { unsigned *bar; unsigned G; };
__device__ __forceinline__ void grid_barrier(const Mw &mw) {
  __syncthreads();
  if (threadIdx.x == 0 && !__hip_atomic_load(&mw.bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {      // (after a time-out nobody waits again)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned gen = __hip_atomic_load(&mw.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned arrived = __hip_atomic_fetch_add(&mw.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == mw.G) {
      __hip_atomic_store(&mw.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&mw.bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
      while (__hip_atomic_load(&mw.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { __hip_atomic_store(&mw.bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        if (__hip_atomic_load(&mw.bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}
__device__ __forceinline__ void wg_or_grid_barrier(const Mw &mw) { if (mw.G > 1) grid_barrier(mw); else __syncthreads(); }
template <int BT, int PF, int LA, bool WIDE>
__device__ __forceinline__ void run_stream_gx(const ValSrc<BT> &vs, uint32_t begin, uint32_t end, uint32_t tail, double *xs, int lane) {
  constexpr int G = LA + 1;                   // slots of requested values: step q uses slot q % G (G divides PF)
  static_assert(PF % G == 0 && PF > LA + 4, "slots must line up across ring revolutions; the ring must hold the steps looked ahead at");
  constexpr uint32_t NONE = WIDE ? 0xFFFFFFFFu : 0xFFFFu;
  Ring<BT, PF> r;
  r.desc = MI_D_NOOP;
#pragma unroll
  for (int st = 0; st < PF; st++) {
    r.gi[st] = 0u; r.gr[st] = 0u;
#pragma unroll
    for (int b = 0; b < BT; b++) r.v[st][b] = 0.0;
  }
  double acc[BT], xq[G][BT], oq[G][BT];
#pragma unroll
  for (int b = 0; b < BT; b++) {
    acc[b] = 0.0;
#pragma unroll
    for (int g = 0; g < G; g++) { xq[g][b] = 0.0; oq[g][b] = 0.0; }
  }
  for (uint32_t npos = begin; npos < end + PF; npos += PF) {   // this revolution runs steps npos - PF + st, loads npos + st
    const uint32_t dnext = load_desc(vs, npos, end, lane);
    // request the vector entries step (st + j) of this revolution needs: its gather operand and, for a subtracting flush
    // step, the old value of its target row (no-op steps request entry 0: no branch around a load)
    auto request = [&](int sj, uint32_t dj) {                   // sj: position in the ring (static), dj: its descriptor
      const bool row_step = MI_D_TYPE(dj) == MI_D_TYPE_ROW;
      const uint32_t w = r.gi[sj % PF];
      const uint32_t gidx = row_step ? (WIDE ? w : (w & 0xFFFFu)) : 0u;
      load_bt<BT>(xs + (size_t)gidx * BT, xq[sj % G]);
      const bool rmw = row_step && (dj & MI_D_FLUSH) && !(dj & MI_D_STORE);
      const uint32_t row = WIDE ? r.gr[sj % PF] : (w >> 16);
      flush_prefetch<BT, NONE>(rmw ? MI_D_LT(dj) : 0u, rmw ? row : 0u, xs, lane, oq[sj % G]);      // (not a flush: entry 0, unused)
    };
#pragma unroll
    for (int st = 0; st < PF; st++) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)r.desc, st);
      const uint32_t nb0 = MI_D_NBAR(d);
      if (nb0) {
        for (uint32_t nb = nb0; nb; nb--) __syncthreads();
        // everything requested before the barrier may be stale: request the next LA steps again, in one burst
#pragma unroll
        for (int j = 0; j < LA; j++) {
          const uint32_t dj = st + j < PF ? (uint32_t)__builtin_amdgcn_readlane((int)r.desc, (st + j) % PF)
                                          : (uint32_t)__builtin_amdgcn_readlane((int)dnext, (st + j) % PF);
          request(st + j, dj);
        }
      }
      // look-ahead: step st + LA (its ring slot and descriptor arrived long ago); its slot (st + LA) % G is the one step
      // st - 1 used
      {
        const uint32_t dl = st + LA < PF ? (uint32_t)__builtin_amdgcn_readlane((int)r.desc, (st + LA) % PF)
                                         : (uint32_t)__builtin_amdgcn_readlane((int)dnext, (st + LA) % PF);
        request(st + LA, dl);
        if (MI_D_TYPE(d) == MI_D_TYPE_ROW) {
#pragma unroll
          for (int b = 0; b < BT; b++) acc[b] = fma(r.v[st][b], xq[st % G][b], acc[b]);
          if (d & MI_D_FLUSH) {
            const uint32_t w = r.gi[st];
            const uint32_t lt = MI_D_LT(d), row = WIDE ? r.gr[st] : (w >> 16);
            reduce_write<BT, NONE>(acc, lt, row, xs, lane, !(d & MI_D_STORE), oq[st % G]);
#pragma unroll
            for (int b = 0; b < BT; b++) acc[b] = 0.0;
          }
        }
      }
      load_step<BT, WIDE>(vs, npos + (uint32_t)st, lane, r.v[st], r.gi[st], r.gr[st]);
    }
    r.desc = dnext;
  }
  for (uint32_t nb = tail; nb; nb--) __syncthreads();
}

// ---- dataflow form of a sweep (host_core.hpp Analysis::df): ONE QP, the vector in global memory, any number of workgroups.
// No barriers between the levels: every entry a sweep produces is written exactly once (a subtracting flush reads the old
// value at `row` and writes at `row + shadow`) and holds the bit pattern MI_DF_NOTYET until then; a step whose gather meets
// that pattern loads again until the value is there.  All accesses of the vector are agent-scope relaxed atomics, i.e.
// sc1 loads / stores (past the CU's L1, written through: the 8-byte "granule" hand-off of MI355X_MICROARCH.md, one hop
// ~1 us instead of a grid barrier of ~6 us per level).  Gathers and old values are requested LA steps ahead like in
// run_stream_gx; a request that came back too early just costs the re-load.  Every wave of the grid is resident (<= 64
// workgroups of 8 waves) and the streams are ordered by level, so the waits cannot form a cycle; a wait of more than ~2 s
// (a lost store would be a bug) raises the error word instead of hanging the GPU, and every later wait gives up at once.
#define MI_DF_NOTYET 0x7FF8D0F1A5B4C3D2ll          // a quiet NaN with a payload no arithmetic produces
__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double df_notyet() { return __longlong_as_double(MI_DF_NOTYET); }
template <int PF, int LA>
__device__ __forceinline__ void run_stream_df(const ValSrc<1> &vs, uint32_t begin, uint32_t end, double *xs, uint32_t shadow, int lane,
                                              unsigned *err) {
  constexpr int G = LA + 1;
  static_assert(PF % G == 0 && PF > LA + 4, "slots must line up across ring revolutions; the ring must hold the steps looked ahead at");
  Ring<1, PF> r;
  r.desc = MI_D_NOOP;
#pragma unroll
  for (int st = 0; st < PF; st++) { r.gi[st] = 0u; r.gr[st] = 0u; r.v[st][0] = 0.0; }
  double acc = 0.0, xq[G], oq[G];
#pragma unroll
  for (int g = 0; g < G; g++) { xq[g] = 0.0; oq[g] = 0.0; }
  bool dead = false;                                                       // a wait has timed out: results are garbage, do not wait again
  for (uint32_t npos = begin; npos < end + PF; npos += PF) {
    const uint32_t dnext = load_desc(vs, npos, end, lane);
    auto request = [&](int sj, uint32_t dj) {
      const bool row_step = MI_D_TYPE(dj) == MI_D_TYPE_ROW;
      xq[sj % G] = ld_sc1(xs + (row_step ? r.gi[sj % PF] : 0u));
      const bool rmw = row_step && (dj & MI_D_FLUSH) && !(dj & MI_D_STORE);
      const uint32_t row = r.gr[sj % PF];
      const bool mine = rmw && row != 0xFFFFFFFFu && ((uint32_t)lane & ((1u << MI_D_LT(dj)) - 1u)) == 0u;
      oq[sj % G] = ld_sc1(xs + (mine ? row : 0u));                         // (not needed: entry 0, unused - no branch around a load)
    };
#pragma unroll
    for (int st = 0; st < PF; st++) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)r.desc, st);
      const uint32_t dl = st + LA < PF ? (uint32_t)__builtin_amdgcn_readlane((int)r.desc, (st + LA) % PF)
                                       : (uint32_t)__builtin_amdgcn_readlane((int)dnext, (st + LA) % PF);
      request(st + LA, dl);
      if (MI_D_TYPE(d) == MI_D_TYPE_ROW) {
        double xv = xq[st % G];
        if (__builtin_amdgcn_ballot_w64(__double_as_longlong(xv) == MI_DF_NOTYET) && !dead) {
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
          uint32_t spins = 0;
          do {
            xv = ld_sc1(xs + r.gi[st]);
            if ((++spins & 63u) == 0u &&
                (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
              __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              dead = true;
              break;
            }
          } while (__builtin_amdgcn_ballot_w64(__double_as_longlong(xv) == MI_DF_NOTYET));
        }
        acc = fma(r.v[st][0], xv, acc);
        if (d & MI_D_FLUSH) {
          const uint32_t lt = MI_D_LT(d), row = r.gr[st];
          const bool sub = !(d & MI_D_STORE);
          double kp = acc;
          if (lt > 0) kp += dpp_d<DPP_SHL(1)>(kp);
          if (lt > 1) kp += dpp_d<DPP_SHL(2)>(kp);
          if (lt > 2) kp += dpp_d<DPP_SHL(4)>(kp);
          if (lt > 3) kp += dpp_d<DPP_SHL(8)>(kp);
          if (lt > 4) kp += down16_d(kp);
          if (lt > 5) kp += down32_d(kp);
          if (row != 0xFFFFFFFFu && ((uint32_t)lane & ((1u << lt) - 1u)) == 0u)
            st_sc1(xs + (size_t)row + (sub ? shadow : 0u), sub ? oq[st % G] - kp : kp);
          acc = 0.0;
        }
      }
      load_step<1, true>(vs, npos + (uint32_t)st, lane, r.v[st], r.gi[st], r.gr[st]);
    }
    r.desc = dnext;
  }
}

// One triangular solve: this wave's whole stream of the schedule.
template <int BT, int PF, bool GX, int TR = 0, bool WIDE = false>
__device__ __forceinline__ void run_tri(const SchedDev &s, const ValSrc<BT> &vals, double *xs, int wave, int lane,
                                        uint32_t *tr = nullptr, uint32_t *tw = nullptr) {
  mi_cptr lp = as_const(s.lvl_pos);
  const uint32_t begin = lp[wave], end = lp[(size_t)s.n_levels * s.nw + wave], tail = as_const(s.tail_bar)[wave];
  if constexpr (GX && TR == 0 && BT <= 2) run_stream_gx<BT, MI_GX_PF, MI_GX_LA, WIDE>(vals, begin, end, tail, xs, lane);     // (BT = 4: registers)
  else run_stream<BT, PF, true, true, GX, TR, WIDE>(vals, begin, end, tail, xs, nullptr, lane, tr, wave, s.nw, tw);
}

// SpMV with the same streams: levels [l0, l1) of the check schedule (independent rows, no barriers)
template <int BT, int PF, bool WIDE = false>
__device__ __forceinline__ void run_spmv(const SchedDev &s, const ValSrc<BT> &vals, double *xs, double *out, int wave,
                                         int lane, int l0, int l1, int nwaves_here = 0) {
  // the rows are independent (no barriers): a schedule laid out for more waves than are here (the single large QPs,
  // whose check schedule is built for the whole grid) is walked stream by stream; normally one trip
  mi_cptr lp = as_const(s.lvl_pos);
  const int step = nwaves_here > 0 ? nwaves_here : (int)s.nw;
  for (int vw = wave; vw < (int)s.nw; vw += step) {
    const uint32_t begin = lp[(size_t)l0 * s.nw + vw], end = lp[(size_t)l1 * s.nw + vw];
    run_stream<BT, PF, false, false, false, 0, WIDE>(vals, begin, end, 0u, xs, out, lane);
  }
}

// --------------------------------------------------------- block reductions
// K values per thread, reduced over all threads with the same (tid % BT);
// result broadcast to every thread of that class.  red: nw*K*BT + K*BT doubles.
template <int BT, int K, bool IS_MAX>
__device__ __forceinline__ void block_reduce(double (&v)[K], double *red, int tid, int wave, int nw, int lane) {
#pragma unroll
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int off = BT; off < 64; off <<= 1) {
      const double o = shfl_xor_d(v[k], off);
      v[k] = IS_MAX ? fmax(v[k], o) : v[k] + o;
    }
  }
  if (lane < BT) {
#pragma unroll
    for (int k = 0; k < K; k++) red[(wave * K + k) * BT + lane] = v[k];
  }
  __syncthreads();
  double *res = red + nw * K * BT;
  if (tid < K * BT) {
    const int k = tid / BT, b = tid % BT;
    double r = red[k * BT + b];
    for (int w = 1; w < nw; w++) {
      const double o = red[(w * K + k) * BT + b];
      r = IS_MAX ? fmax(r, o) : r + o;
    }
    res[k * BT + b] = r;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = res[k * BT + (tid % BT)];
  __syncthreads();
}

// One QP shared by mw.G workgroups (check_kernel of the single large QPs): every workgroup publishes its block result,
// all meet at a grid barrier, and every workgroup combines the G partial results in the same order - all of them hold the
// same numbers afterwards, so the control flow that follows stays uniform over the grid.  (max of non-negative numbers / sums)
template <int K, bool IS_MAX>
__device__ __forceinline__ void grid_reduce(double (&v)[K], double *red, const Mw &mw, double *scratch, int ltid, int lwave, int lnw, int lane) {
  if (mw.G <= 1) return;
  if (ltid == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) __hip_atomic_store(scratch + (size_t)blockIdx.x * 16 + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  grid_barrier(mw);
  double w[K];
#pragma unroll
  for (int k = 0; k < K; k++) w[k] = 0.0;
  for (unsigned g = (unsigned)ltid; g < mw.G; g += blockDim.x) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      const double o = __hip_atomic_load(scratch + (size_t)g * 16 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      w[k] = IS_MAX ? fmax(w[k], o) : w[k] + o;
    }
  }
  block_reduce<1, K, IS_MAX>(w, red, ltid, lwave, lnw, lane);
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = w[k];
}

// ------------------------------------------------------------ the ADMM kernel

template <int BT>
struct TilePtrs {
  const double *dinv;
  double *x, *z, *y;
  const double *q, *l, *u, *rho_vec, *rho_inv, *Dsc, *Dsc_inv, *Esc, *Esc_inv;
  double *dx, *dy, *out1, *out2, *dscal;
  int *iscal;
  ValSrc<BT> vfwd, vbwd, vchk;
  mi_rsrc vdt[BT];                 // dense tail: the stream of the inverted Schur complement, one per QP
  int act;                         // bit b: QP b of the tile streams (is not finished / skipped); wave-uniform
};

template <int BT>
__device__ __forceinline__ TilePtrs<BT> tile_ptrs(const KernelArgs &a, int tile, bool skip_done = false) {
  TilePtrs<BT> p;
  const size_t n = a.n, m = a.m, N = a.N, t = tile;
  p.dinv = a.dinv + t * N * BT;
  p.x = a.x + t * n * BT; p.z = a.z + t * m * BT; p.y = a.y + t * m * BT;
  p.q = a.q + t * n * BT; p.l = a.l + t * m * BT; p.u = a.u + t * m * BT;
  p.rho_vec = a.rho_vec + t * m * BT; p.rho_inv = a.rho_inv + t * m * BT;
  p.Dsc = a.Dsc + t * n * BT; p.Dsc_inv = a.Dsc_inv + t * n * BT;
  p.Esc = a.Esc + t * m * BT; p.Esc_inv = a.Esc_inv + t * m * BT;
  p.dx = a.dx + t * n * BT; p.dy = a.dy + t * m * BT;
  p.out1 = a.out1 + t * (2 * n + m) * BT; p.out2 = a.out2 + t * (2 * n + m) * BT;
  p.dscal = a.dscal + t * DS_COUNT * BT;
  p.iscal = a.iscal + t * IS_COUNT * BT;
  auto idx_rsrc = [&](const SchedDev &sd) { return a.wide ? make_rsrc(sd.idxw64, sd.n_steps * 512u) : make_rsrc(sd.idxw, sd.n_steps * 256u); };
  p.vfwd.idx = idx_rsrc(a.fwd); p.vfwd.step = make_rsrc(a.fwd.step, a.fwd.n_steps * 4u);
  p.vbwd.idx = idx_rsrc(a.bwd); p.vbwd.step = make_rsrc(a.bwd.step, a.bwd.n_steps * 4u);
  p.vchk.idx = idx_rsrc(a.chk); p.vchk.step = make_rsrc(a.chk.step, a.chk.n_steps * 4u);
  p.act = 0;
#pragma unroll
  for (int bb = 0; bb < BT; bb++) {
    const size_t slot = t * BT + bb;
    const bool off = skip_done && p.iscal[IS_DONE * BT + bb] != 0;     // wave-uniform: inside a solve, finished QPs (and padding slots) stream nothing
    if (!off) p.act |= 1 << bb;
    const bool wk = !a.use_work || a.use_work[slot] != 0;                // (wave-uniform) working copy or snapshot of the factor
    const double *fv = wk ? a.fwd_val : a.fwd_val0, *bv = wk ? a.bwd_val : a.bwd_val0, *dv = wk ? a.dt_val : a.dt_val0;
    p.vfwd.vals[bb] = off ? make_rsrc(nullptr, 0u) : make_rsrc(fv + slot * a.fwd.n_steps * 64, a.fwd.n_steps * 512u);
    p.vbwd.vals[bb] = off ? make_rsrc(nullptr, 0u) : make_rsrc(bv + slot * a.bwd.n_steps * 64, a.bwd.n_steps * 512u);
    p.vchk.vals[bb] = off ? make_rsrc(nullptr, 0u) : make_rsrc(a.chk_val + slot * a.chk.n_steps * 64, a.chk.n_steps * 512u);
    p.vdt[bb] = (off || !a.dt.k) ? make_rsrc(nullptr, 0u) : make_rsrc(dv + slot * a.dt.n_steps * 64, a.dt.n_steps * 512u);
  }
  return p;
}

// The solve vector normally lives in LDS.  For KKT systems that do not fit (N*BT*8 B > ~150 KiB, e.g.
// the reference's own 802-waypoint example, N = 43 284) it lives in a per-tile global buffer instead:
// same code, generic pointer, full barriers (global stores must complete before other waves read).
// GX is a compile-time flag on purpose: a runtime-selected generic pointer would turn every access of the
// LDS variant into FLAT instructions (which also count on vmcnt and would drain the value stream).
template <int BT, bool GX>
__device__ __forceinline__ double *solve_vector(const KernelArgs &a, double *smem, int tile, double *&scratch) {
  if constexpr (GX) { scratch = smem; return a.xs_global + (size_t)tile * (a.xs_len) * BT; }
  else { scratch = smem + (size_t)a.xs_len * BT; return smem; }
}


// ---- dense tail: x_tail = S^-1 t_tail as one symmetric product (host_core.hpp DenseTail) ------------------
// lane i reads lane (i + 1) % 64
__device__ __forceinline__ double rol1_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x134 /* wave_rol:1 */, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x134, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
#ifndef MI_DT_PF
#define MI_DT_PF 8
#endif
// Every wave walks its tasks (64 x 64 blocks of M, 64 or 32 steps each) through a PF-step register ring of the
// value stream; per step and QP: two fmas (column sum stays in the lane, row sum travels) and two rotations.
// xt = the tail of the solve vector (t, read only), yr / yc = the two accumulation vectors (LDS, [k][BT]).
// NQ = QP streams served: all BT of the tile, or ONE (slot b0 of the tile: the other QPs of the tile have finished, or
// BT = 1) with a ring twice as deep - a lone stream is bound by the loads it keeps in flight (one memory latency per
// ring revolution), and the registers the finished QP's ring held are free.
template <int BT, int NQ, int PF>
__device__ __forceinline__ void dense_tail_apply(const DenseTailDev &dt, const mi_rsrc (&vals)[NQ], int b0, const double *xt,
                                                 double *yr, double *yc, int wave, int lane) {
  static_assert(NQ == BT || NQ == 1, "all QPs of the tile or one");
  static_assert(32 % PF == 0, "task lengths (32 / 64 steps) must be multiples of the ring depth");
  auto ldv = [&](const double *p, double (&o)[NQ]) {
    if constexpr (NQ == BT) load_bt<BT>(p, o); else o[0] = p[b0];
  };
  mi_cptr tk = as_const(dt.task);
  uint32_t t = as_const(dt.wave_task)[wave];
  const uint32_t begin = as_const(dt.wave_step)[wave], end = as_const(dt.wave_step)[wave + 1];
  double rv[PF][NQ], tj[NQ], ti[NQ], accr[NQ], accc[NQ];
#pragma unroll
  for (int b = 0; b < NQ; b++) {
    tj[b] = 0.0; ti[b] = 0.0; accr[b] = 0.0; accc[b] = 0.0;
#pragma unroll
    for (int st = 0; st < PF; st++) rv[st][b] = 0.0;
  }
  uint32_t left = 0, I0 = 0, J0 = 0, flags = 0, nst = 0;
  for (uint32_t pos = begin; pos < end + PF; pos += PF) {       // this revolution consumes steps [pos - PF, pos) and loads [pos, pos + PF)
    if (pos > begin && left == 0) {                             // the next task starts (task lengths are multiples of PF)
      I0 = tk[4 * t]; J0 = tk[4 * t + 1]; nst = tk[4 * t + 3];
      const uint32_t fl = tk[4 * t + 2];
      for (uint32_t nb = fl >> 8; nb; nb--) lds_barrier();
      flags = fl & 255u; left = nst;
      const uint32_t s0 = flags & 1u;
      ldv(xt + (size_t)(J0 + (uint32_t)lane) * BT, tj);
      ldv(xt + (size_t)(I0 + (((uint32_t)lane + s0) & 63u)) * BT, ti);
#pragma unroll
      for (int b = 0; b < NQ; b++) { accr[b] = 0.0; accc[b] = 0.0; }
    }
#pragma unroll
    for (int st = 0; st < PF; st++) {
#pragma unroll
      for (int b = 0; b < NQ; b++) {
        accc[b] = fma(rv[st][b], ti[b], accc[b]);
        accr[b] = fma(rv[st][b], tj[b], accr[b]);
      }
#pragma unroll
      for (int b = 0; b < NQ; b++) { ti[b] = rol1_d(ti[b]); accr[b] = rol1_d(accr[b]); }
#pragma unroll
      for (int b = 0; b < NQ; b++) {
        const mi_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(vals[b], (uint32_t)lane * 8u, (pos + (uint32_t)st) * 512u, 0);
        rv[st][b] = __hiloint2double((int)w.y, (int)w.x);
      }
    }
    if (pos > begin) {
      left -= PF;
      if (left == 0) {                                          // the task is complete: add its two 64-vectors of sums
        const uint32_t il = ((uint32_t)lane + (flags & 1u) + nst) & 63u;      // the row this lane's travelling sum belongs to
        double *rvec = (flags & 2u) ? yc : yr, *cvec = (flags & 2u) ? yr : yc;
#pragma unroll
        for (int b = 0; b < NQ; b++) {
          const int bb = NQ == BT ? b : b0;
          rvec[(size_t)(I0 + il) * BT + bb] += accr[b];
          cvec[(size_t)(J0 + (uint32_t)lane) * BT + bb] += accc[b];
        }
        t++;
      }
    }
  }
  for (uint32_t nb = as_const(dt.tail_bar)[wave]; nb; nb--) lds_barrier();
}

// what happens between the forward and the backward sweep: D^-1 on the rows before the dense tail, S^-1 on the tail
template <int BT, bool GX>
__device__ __forceinline__ void kkt_middle(const KernelArgs &a, const double *dinv, const mi_rsrc (&vdt)[BT], int act, double *xs,
                                           int tid, int nthr, int wave, int lane) {
  if (!a.dt.k) {
    constexpr int U = GX ? 8 : 1;               // (global vector: U dependent xloc -> xs gathers in flight per thread)
    const int tot = a.N * BT;
    for (int e0 = tid; e0 < tot; e0 += nthr * U) {
      size_t pos[U]; double v[U], dv[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const int e = e0 + u * nthr < tot ? e0 + u * nthr : e0; pos[u] = (size_t)a.xloc[e / BT] * BT + e % BT; dv[u] = dinv[e]; }
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = xs[pos[u]];
#pragma unroll
      for (int u = 0; u < U; u++) if (e0 + u * nthr < tot) xs[pos[u]] = v[u] * dv[u];
    }
    __syncthreads();
    return;
  }
  if constexpr (!GX) {
    const int s = a.dt.s, k = a.dt.k;
    double *yr = xs + (size_t)a.xs_len * BT, *yc = yr + (size_t)k * BT;      // overlays the reduction scratch of check_kernel
    for (int e = tid; e < a.N * BT; e += nthr) {
      const int i = e / BT;
      if (i < s) xs[(size_t)a.xloc[i] * BT + e % BT] *= dinv[e];
      else { yr[e - s * BT] = dinv[e] * xs[e]; yc[e - s * BT] = 0.0; }          // dinv of a tail row = diagonal of S^-1
    }
    __syncthreads();
    if constexpr (BT == 1) {
      dense_tail_apply<1, 1, 2 * MI_DT_PF>(a.dt, vdt, 0, xs + (size_t)s * BT, yr, yc, wave, lane);
    } else {
      // exactly one QP of the tile still streams (the others have finished): serve it alone, with the deeper ring
      const int act_u = __builtin_amdgcn_readfirstlane(act);
      if (BT == 2 && (act_u == 1 || act_u == 2)) {
        const mi_rsrc one[1] = {act_u == 2 ? vdt[BT - 1] : vdt[0]};
        dense_tail_apply<BT, 1, 2 * MI_DT_PF>(a.dt, one, act_u == 2 ? 1 : 0, xs + (size_t)s * BT, yr, yc, wave, lane);
      } else {
        dense_tail_apply<BT, BT, MI_DT_PF>(a.dt, vdt, 0, xs + (size_t)s * BT, yr, yc, wave, lane);
      }
    }
    for (int e = tid; e < k * BT; e += nthr) xs[(size_t)s * BT + e] = yr[e] + yc[e];
    __syncthreads();
  }
}

// K solve on the LDS vector: fwd levels, D^-1, bwd levels (row E7)
// Dataflow form: where the sweeps leave the final value of permuted row e (host_core.hpp Analysis::df_floc / df_bloc)
__device__ __forceinline__ uint32_t df_floc(uint32_t e, uint32_t f, uint32_t xl, uint32_t sh) { return (f & 4u) ? xl : ((f & 1u) ? e + sh : e); }
__device__ __forceinline__ uint32_t df_bloc(uint32_t e, uint32_t f, uint32_t sh) { return (f & 4u) ? e : ((f & 2u) ? e + sh : e); }
// entries the forward sweep is going to write: "not yet" (by whoever stores the right-hand side entry of row e)
__device__ __forceinline__ void df_arm_fwd(double *xs, uint32_t e, uint32_t f, uint32_t xl, uint32_t sh) {
  if (f & 1u) st_sc1(xs + e + sh, df_notyet());
  if (f & 4u) st_sc1(xs + xl, df_notyet());
}

// (tid / nthr / wave: of the workgroup, or - one QP shared by mw.G workgroups - of the whole grid)
template <int BT, int PF, bool GX, bool WIDE = false>
__device__ __forceinline__ void kkt_solve_lds(const KernelArgs &a, const TilePtrs<BT> &p, double *xs,
                                              int tid, int nthr, int wave, int nw, int lane, const Mw &mw = Mw{nullptr, 1u}) {
  if constexpr (GX && BT == 1 && WIDE) {
    if (a.df) {
      // the caller has stored the right-hand side, armed the forward entries and passed a barrier
      mi_cptr lpf = as_const(a.fwd.lvl_pos), lpb = as_const(a.bwd.lvl_pos);
      run_stream_df<MI_GX_PF, MI_GX_LA>(p.vfwd, lpf[wave], lpf[(size_t)a.fwd.n_levels * a.fwd.nw + wave], xs, a.df_shadow, lane, mw.bar + 2);
      wg_or_grid_barrier(mw);
      // D^-1, results to xloc; arm the entries the backward sweep writes
      const uint32_t sh = a.df_shadow;
      constexpr int U = 8;
      for (int e0 = tid; e0 < a.N; e0 += nthr * U) {
        uint32_t xl[U], f[U]; double v[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u * nthr < a.N ? e0 + u * nthr : e0; xl[u] = a.xloc[e]; f[u] = a.rflag[e]; dv[u] = p.dinv[e]; }
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u * nthr < a.N ? e0 + u * nthr : e0; v[u] = ld_sc1(xs + df_floc((uint32_t)e, f[u], xl[u], sh)); }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int e = e0 + u * nthr;
          if (e < a.N) {
            st_sc1(xs + xl[u], v[u] * dv[u]);
            if (f[u] & 2u) st_sc1(xs + xl[u] + sh, df_notyet());
            if (f[u] & 4u) st_sc1(xs + e, df_notyet());
          }
        }
      }
      wg_or_grid_barrier(mw);
      run_stream_df<MI_GX_PF, MI_GX_LA>(p.vbwd, lpb[wave], lpb[(size_t)a.bwd.n_levels * a.bwd.nw + wave], xs, a.df_shadow, lane, mw.bar + 2);
      wg_or_grid_barrier(mw);
      return;
    }
  }
  run_tri<BT, PF, GX, 0, WIDE>(a.fwd, p.vfwd, xs, wave, lane);
  kkt_middle<BT, GX>(a, p.dinv, p.vdt, p.act, xs, tid, nthr, wave, lane);
  run_tri<BT, PF, GX, 0, WIDE>(a.bwd, p.vbwd, xs, wave, lane);
}

// E6-E10 for iterations (iter_begin, iter_end] of one tile.  Lean on purpose: the
// residual / termination / rho logic lives in check_kernel, so this kernel needs
// little beyond the register ring of the step streams.
// (n_iter iterations; a device function so that advance_kernel can run it segment after segment)
template <int BT, int NT, bool GX, bool WIDE = false>
__device__ __forceinline__ void iterate_body(const KernelArgs &a, double *smem, int n_iter) {
  // multi-workgroup mode (global vector, one QP): the grid is ONE tile; thread / wave numbers run over the grid and the
  // barriers between phases are grid barriers
  const Mw mw{a.mw_bar, GX && BT == 1 && a.mw_groups > 1 ? (unsigned)a.mw_groups : 1u};
  const bool multi = GX && BT == 1 && mw.G > 1;
  const int tile = multi ? 0 : blockIdx.x;
  const int tid = multi ? blockIdx.x * blockDim.x + threadIdx.x : threadIdx.x, nthr = multi ? blockDim.x * mw.G : blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int b = tid % BT;
  const int n = a.n, m = a.m, N = a.N;
  double *unused_scratch;
  double *xs = solve_vector<BT, GX>(a, smem, tile, unused_scratch);
  const TilePtrs<BT> p = tile_ptrs<BT>(a, tile, true);
  const int done = p.iscal[IS_DONE * BT + b];
  if (__syncthreads_and(done)) return;
  const double alpha = a.alpha, sigma = a.sigma;
  bool df = false;
  if constexpr (GX && BT == 1 && WIDE) df = a.df != 0;
  const uint32_t sh = a.df_shadow;
  // ---- E6 of the first iteration: rhs into the permuted solve vector
  for (int e = tid; e < N * BT; e += nthr) {
    const int i = e / BT;
    double v;
    if (i < n) v = sigma * p.x[e] - p.q[e];
    else { const int ez = e - n * BT; v = p.z[ez] - p.rho_inv[ez] * p.y[ez]; }
    const uint32_t pe = a.pinv[i];
    if (df) { st_sc1(xs + pe, v); df_arm_fwd(xs, pe, a.rflag[pe], a.xloc[pe], sh); }
    else xs[(size_t)pe * BT + b] = v;
  }
  if (df && tid == 0) st_sc1(xs + 2 * (size_t)sh, 0.0);            // the entry padding slots gather
  if constexpr (GX) wg_or_grid_barrier(mw); else __syncthreads();
  for (int iter = 1; iter <= n_iter; iter++) {
    const bool do_info = a.info_at_end && iter == n_iter;     // delta_x / delta_y are only needed by check_kernel
    // ---- E7
    kkt_solve_lds<BT, MI_PFV_NT(NT, GX), GX, WIDE>(a, p, xs, tid, nthr, wave, nw, lane, mw);
    // ---- E8-E10 fused with E6 of the next iteration (run_tri ends with a barrier): every thread replaces the
    // solution entry it has just consumed by the next right-hand side entry - same position, no other reader
    // (global-vector mode: one workgroup walks 4 x 10^5 entries, each a dependent pinv -> xs gather: U entries per thread
    //  are loaded before any of them is stored, else every trip costs a full memory round trip - 1.5 ms per iteration at
    //  config 5; U = 1 with the vector in LDS: same code as before)
    constexpr int U = GX ? 4 : 1;
    // (dataflow form: the solution of row pe sits at df_bloc(pe); the new right-hand side goes to pe and the entries the
    //  forward sweep writes are armed - same thread, loads before stores, so an entry that is both is read first)
    uint32_t fl[U], xl[U];
    for (int e0 = tid; e0 < n * BT; e0 += nthr * U) {
      size_t pos[U]; double xt[U], xp[U], qv[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const int e = e0 + u * nthr < n * BT ? e0 + u * nthr : e0; pos[u] = (size_t)a.pinv[e / BT] * BT + b; }
      if (df) {
#pragma unroll
        for (int u = 0; u < U; u++) { fl[u] = a.rflag[pos[u]]; xl[u] = a.xloc[pos[u]]; }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * nthr < n * BT ? e0 + u * nthr : e0;
        xt[u] = df ? ld_sc1(xs + df_bloc((uint32_t)pos[u], fl[u], sh)) : xs[pos[u]]; xp[u] = p.x[e]; qv[u] = p.q[e];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * nthr;
        if (e < n * BT) {
          double xn = alpha * xt[u] + (1.0 - alpha) * xp[u];
          if (!done) { p.x[e] = xn; if (do_info) p.dx[e] = xn - xp[u]; } else xn = xp[u];
          if (df) { st_sc1(xs + pos[u], sigma * xn - qv[u]); df_arm_fwd(xs, (uint32_t)pos[u], fl[u], xl[u], sh); }
          else xs[pos[u]] = sigma * xn - qv[u];
        }
      }
    }
    for (int e0 = tid; e0 < m * BT; e0 += nthr * U) {
      size_t pos[U]; double nu[U], zp[U], yv[U], ri[U], rv[U], lo[U], up[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const int e = e0 + u * nthr < m * BT ? e0 + u * nthr : e0; pos[u] = (size_t)a.pinv[n + e / BT] * BT + b; }
      if (df) {
#pragma unroll
        for (int u = 0; u < U; u++) { fl[u] = a.rflag[pos[u]]; xl[u] = a.xloc[pos[u]]; }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * nthr < m * BT ? e0 + u * nthr : e0;
        nu[u] = df ? ld_sc1(xs + df_bloc((uint32_t)pos[u], fl[u], sh)) : xs[pos[u]]; zp[u] = p.z[e]; yv[u] = p.y[e]; ri[u] = p.rho_inv[e]; rv[u] = p.rho_vec[e]; lo[u] = p.l[e]; up[u] = p.u[e];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * nthr;
        if (e < m * BT) {
          double zt = zp[u] - ri[u] * yv[u];
          zt += ri[u] * nu[u];
          const double zr = alpha * zt + (1.0 - alpha) * zp[u];
          double zn = fmin(fmax(zr + ri[u] * yv[u], lo[u]), up[u]);
          const double dyv = rv[u] * (zr - zn);
          double yn = yv[u] + dyv;
          if (!done) { p.z[e] = zn; p.y[e] = yn; if (do_info) p.dy[e] = dyv; } else { zn = zp[u]; yn = yv[u]; }
          if (df) { st_sc1(xs + pos[u], zn - ri[u] * yn); df_arm_fwd(xs, (uint32_t)pos[u], fl[u], xl[u], sh); }
          else xs[pos[u]] = zn - ri[u] * yn;
        }
      }
    }
    if constexpr (GX) wg_or_grid_barrier(mw); else __syncthreads();
  }
}
template <int BT, int NT, bool GX, bool WIDE = false>
__global__ __launch_bounds__(NT) void iterate_kernel(KernelArgs a) {
  extern __shared__ double smem[];
  iterate_body<BT, NT, GX, WIDE>(a, smem, a.iter_end - a.iter_begin);
}

// E11-E14 after a segment of n_iter iterations: residuals, termination and infeasibility tests, rho estimate / update
// request, solution store.  Every QP counts its iterations itself (IS_CUR: iterations of its current solve so far): QPs
// of a continuous batch begin their solves at different launches.  Returns (to every thread of the workgroup) bit 0: a
// QP of the tile finished in this check, bit 1: a QP of the tile asks for its refactorisation.
template <int BT, int NT, bool GX, bool WIDE = false>
__device__ __forceinline__ int check_body(const KernelArgs &a, double *smem, int n_iter) {
  // one QP shared by the grid (single large QPs, see iterate_kernel): tid / nthr / wave run over the grid, ltid / lwave /
  // lnw over the workgroup (block reductions); barriers between phases that exchange data are grid barriers and every
  // block reduction is followed by a reduction over the workgroups
  bool dfm = false;
  if constexpr (GX && BT == 1 && WIDE) dfm = a.df != 0 && a.mw_groups > 1;
  const Mw mw{a.mw_bar, dfm ? (unsigned)a.mw_groups : 1u};
  const int tile = dfm ? 0 : blockIdx.x;
  const int ltid = threadIdx.x, lwave = __builtin_amdgcn_readfirstlane(ltid >> 6), lnw = blockDim.x >> 6;
  const int tid = dfm ? blockIdx.x * blockDim.x + threadIdx.x : threadIdx.x, nthr = dfm ? blockDim.x * mw.G : blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int b = tid % BT;
  auto sync = [&]() { if constexpr (GX && BT == 1 && WIDE) wg_or_grid_barrier(mw); else __syncthreads(); };
  auto gscratch = [&](int slot) { return a.mw_scratch + (size_t)slot * 256 * 16; };
  const int n = a.n, m = a.m;
  double *red;
  double *xs = solve_vector<BT, GX>(a, smem, tile, red);
  const TilePtrs<BT> p = tile_ptrs<BT>(a, tile, true);
  // global QP id of this thread's class: during a solve the QPs still iterating are
  // compacted into the leading tiles (solver.hip), so the id comes from a table
  const int qp = a.qp_of_slot[tile * BT + b];
  int done = p.iscal[IS_DONE * BT + b];
  if (__syncthreads_and(done)) return 0;
  const int iter = p.iscal[IS_CUR * BT + b] + n_iter;
  int status = p.iscal[IS_STATUS * BT + b];
  int rho_updates = p.iscal[IS_RHO_UPDATES * BT + b];
  double rho = p.dscal[DS_RHO * BT + b];
  const double c = p.dscal[DS_C * BT + b], cinv = p.dscal[DS_CINV * BT + b];
  const bool unscale = a.scaling && !a.scaled_termination;
  const bool is_last = iter >= a.max_iter;
  const bool is_check = a.check_termination && (iter % a.check_termination == 0);
  const bool is_rho = a.adaptive_rho && a.rho_interval && (iter % a.rho_interval == 0);
  // (the tests below hold barriers: what one QP of the tile needs, every thread of the workgroup walks through)
  bool any_check = is_check || is_last, any_last = is_last;
  if constexpr (BT > 1) { any_check = __syncthreads_or(any_check); any_last = __syncthreads_or(any_last); }
  int need_refactor = 0;
  // ---- E11: [x;y] -> LDS, P x / A' y / A x
  for (int e = tid; e < n * BT; e += nthr) xs[e] = p.x[e];
  for (int e = tid; e < m * BT; e += nthr) xs[(size_t)n * BT + e] = p.y[e];
  sync();
  run_spmv<BT, MI_PFV, WIDE>(a.chk, p.vchk, xs, p.out1, wave, lane, 0, 3, nw);
  sync();
  // residual vectors and the norms termination + rho estimate need
  double mx[14];
#pragma unroll
  for (int k = 0; k < 14; k++) mx[k] = 0.0;
  double sm[1] = {0.0};
  for (int e = tid; e < n * BT; e += nthr) {
    const double px = p.out1[e], aty = p.out1[(size_t)n * BT + e], qv = p.q[e], xv = p.x[e];
    const double di = p.Dsc_inv[e];
    double dres = qv + px;
    dres += aty;
    mx[0] = fmax(mx[0], fabs(dres));      mx[1] = fmax(mx[1], fabs(di * dres));
    mx[2] = fmax(mx[2], fabs(qv));        mx[3] = fmax(mx[3], fabs(aty));   mx[4] = fmax(mx[4], fabs(px));
    mx[5] = fmax(mx[5], fabs(di * qv));   mx[6] = fmax(mx[6], fabs(di * aty)); mx[7] = fmax(mx[7], fabs(di * px));
    sm[0] += 0.5 * xv * px + qv * xv;
  }
  for (int e = tid; e < m * BT; e += nthr) {
    const double ax = p.out1[(size_t)2 * n * BT + e], zv = p.z[e], ei = p.Esc_inv[e];
    const double pres = ax - zv;
    mx[8] = fmax(mx[8], fabs(pres));      mx[9] = fmax(mx[9], fabs(ei * pres));
    mx[10] = fmax(mx[10], fabs(zv));      mx[11] = fmax(mx[11], fabs(ax));
    mx[12] = fmax(mx[12], fabs(ei * zv)); mx[13] = fmax(mx[13], fabs(ei * ax));
  }
  block_reduce<BT, 14, true>(mx, red, ltid, lwave, lnw, lane);
  block_reduce<BT, 1, false>(sm, red, ltid, lwave, lnw, lane);
  if constexpr (GX && BT == 1 && WIDE) {
    grid_reduce<14, true>(mx, red, mw, gscratch(0), ltid, lwave, lnw, lane);
    grid_reduce<1, false>(sm, red, mw, gscratch(1), ltid, lwave, lnw, lane);
  }
  const double pri_res = (m == 0) ? 0.0 : (unscale ? mx[9] : mx[8]);
  const double dua_res = unscale ? cinv * mx[1] : mx[0];
  double obj = sm[0];
  if (a.scaling) obj *= cinv;
  const double pri_nrm = unscale ? fmax(mx[12], mx[13]) : fmax(mx[10], mx[11]);
  const double dua_nrm = unscale ? cinv * fmax(fmax(mx[5], mx[6]), mx[7]) : fmax(fmax(mx[2], mx[3]), mx[4]);

  // ---- E12 ingredients that do not depend on the tolerances
  // infeasibility certificates on delta_y / delta_x
  double mi[4] = {0.0, 0.0, 0.0, 0.0};   // norm_dy, norm_dx, |Dinv A'dy|, |Dinv P dx|
  double si[2] = {0.0, 0.0};             // ineq_lhs, q'dx
  for (int e = tid; e < n * BT; e += nthr) {
    const double d = p.dx[e];
    xs[e] = d;
    mi[1] = fmax(mi[1], unscale ? fabs(p.Dsc[e] * d) : fabs(d));
    si[1] += p.q[e] * d;
  }
  for (int e = tid; e < m * BT; e += nthr) {
    double d = p.dy[e];
    const double lo = p.l[e], up = p.u[e];
    if (up > MI_INFTY * MI_MIN_SCALING) {
      if (lo < -MI_INFTY * MI_MIN_SCALING) d = 0.0; else d = fmin(d, 0.0);
    } else if (lo < -MI_INFTY * MI_MIN_SCALING) d = fmax(d, 0.0);
    xs[(size_t)n * BT + e] = d;
    mi[0] = fmax(mi[0], unscale ? fabs(p.Esc[e] * d) : fabs(d));
    si[0] += up * fmax(d, 0.0) + lo * fmin(d, 0.0);
  }
  sync();
  run_spmv<BT, MI_PFV, WIDE>(a.chk, p.vchk, xs, p.out2, wave, lane, 0, 3, nw);
  sync();
  for (int e = tid; e < n * BT; e += nthr) {
    const double pdx = p.out2[e], atdy = p.out2[(size_t)n * BT + e];
    const double di = unscale ? p.Dsc_inv[e] : 1.0;
    mi[2] = fmax(mi[2], fabs(di * atdy));
    mi[3] = fmax(mi[3], fabs(di * pdx));
  }
  block_reduce<BT, 4, true>(mi, red, ltid, lwave, lnw, lane);
  block_reduce<BT, 2, false>(si, red, ltid, lwave, lnw, lane);
  if constexpr (GX && BT == 1 && WIDE) {
    grid_reduce<4, true>(mi, red, mw, gscratch(2), ltid, lwave, lnw, lane);
    grid_reduce<2, false>(si, red, mw, gscratch(3), ltid, lwave, lnw, lane);
  }
  const double norm_dy = mi[0], norm_dx = mi[1];
  const double cost_scaling = unscale ? c : 1.0;

  // tolerance-dependent decision; approx = 10x tolerances (max_iter path)
  auto decide = [&](bool approx) -> int {
    const double f = approx ? 10.0 : 1.0;
    const double eps_abs = f * a.eps_abs, eps_rel = f * a.eps_rel;
    const double eps_pinf = f * a.eps_prim_inf, eps_dinf = f * a.eps_dual_inf;
    // rows of A dx outside the recession cone (needs its own reduction)
    double viol[1] = {0.0};
    for (int e = tid; e < m * BT; e += nthr) {
      double adx = p.out2[(size_t)2 * n * BT + e];
      if (unscale) adx *= p.Esc_inv[e];
      const double lo = p.l[e], up = p.u[e];
      if ((up < MI_INFTY * MI_MIN_SCALING && adx > eps_dinf * norm_dx) ||
          (lo > -MI_INFTY * MI_MIN_SCALING && adx < -eps_dinf * norm_dx)) viol[0] = 1.0;
    }
    block_reduce<BT, 1, true>(viol, red, ltid, lwave, lnw, lane);
    if constexpr (GX && BT == 1 && WIDE) grid_reduce<1, true>(viol, red, mw, gscratch(approx ? 5 : 4), ltid, lwave, lnw, lane);
    if (pri_res > MI_INFTY || dua_res > MI_INFTY) return -7;   // non-convex / diverged
    int prim_ok = 0, dual_ok = 0, prim_inf = 0, dual_inf = 0;
    if (m == 0) prim_ok = 1;
    else {
      const double eps_prim = eps_abs + eps_rel * pri_nrm;
      if (pri_res < eps_prim) prim_ok = 1;
      else if (norm_dy > MI_DIV_TOL && si[0] < -eps_pinf * norm_dy) prim_inf = mi[2] < eps_pinf * norm_dy;
    }
    const double eps_dual = eps_abs + eps_rel * dua_nrm;
    if (dua_res < eps_dual) dual_ok = 1;
    else if (norm_dx > MI_DIV_TOL && si[1] < -cost_scaling * eps_dinf * norm_dx &&
             mi[3] < cost_scaling * eps_dinf * norm_dx) dual_inf = viol[0] == 0.0;
    if (prim_ok && dual_ok) return approx ? 2 : 1;
    if (prim_inf) return approx ? 3 : -3;
    if (dual_inf) return approx ? 4 : -4;
    return 0;
  };

  int new_status = 0;
  if (any_check) { const int s0 = decide(false); if (is_check || is_last) new_status = s0; }
  double rho_est = p.dscal[DS_RHO_EST * BT + b];
  // ---- E13: rho estimate from the SCALED residual norms
  auto rho_estimate = [&]() -> double {
    const double pr = mx[8] / (fmax(mx[10], mx[11]) + MI_DIV_TOL);
    const double du = mx[0] / (fmax(fmax(mx[2], mx[3]), mx[4]) + MI_DIV_TOL);
    double e = rho * sqrt(pr / du);
    return fmin(fmax(e, MI_RHO_MIN), MI_RHO_MAX);
  };
  // (at max_iter on a rho-update iteration that is not a check iteration upstream adapts rho BEFORE its final
  //  check_termination: the update happens even when that check then reports "solved")
  if (!done && is_rho && (new_status == 0 || (is_last && !is_check))) {
    const double rn = rho_estimate();
    rho_est = rn;
    if (rn > rho * a.rho_tolerance || rn < rho / a.rho_tolerance) {
      need_refactor = 1;
      rho = fmin(fmax(rn, MI_RHO_MIN), MI_RHO_MAX);
      rho_updates++;
    }
  }
  if (any_last) {
    const int s1 = decide(true);            // (every thread: the decision holds barriers)
    if (!done && new_status == 0 && is_last) new_status = s1 ? s1 : -2;      // -2: max iterations reached
  }
  if (!done && new_status != 0) {
    // ---- E14: store_solution
    done = 1; status = new_status;
    rho_est = rho_estimate();            // upstream's closing compute_rho_estimate: with the rho in force now

    if (tid < BT) {
      p.dscal[DS_PRI_RES * BT + b] = pri_res; p.dscal[DS_DUA_RES * BT + b] = dua_res;
      p.dscal[DS_OBJ * BT + b] = (status == -3 || status == 3) ? MI_INFTY
                                 : (status == -4 || status == 4) ? -MI_INFTY
                                 : (status == -7) ? __builtin_nan("") : obj;
      p.iscal[IS_ITER * BT + b] = iter;
    }
  }
  if (tid < BT) p.dscal[DS_RHO_EST * BT + b] = rho_est;
  // outputs for QPs that finished in this pass (done is uniform per class b)
  const int just_done = done && (p.iscal[IS_DONE * BT + b] == 0);
  sync();
  if (just_done && qp >= 0) {
    const bool has_sol = !(status == -3 || status == 3 || status == -4 || status == 4 || status == -7);
    const double nanv = __builtin_nan("");
    for (int e = tid; e < n * BT; e += nthr) {
      const int i = e / BT;
      a.x_out[(size_t)qp * n + i] = has_sol ? (a.scaling ? p.Dsc[e] * p.x[e] : p.x[e]) : nanv;
      if (!has_sol) p.x[e] = 0.0;
    }
    for (int e = tid; e < m * BT; e += nthr) {
      const int j = e / BT;
      a.y_out[(size_t)qp * m + j] = has_sol ? (a.scaling ? p.Esc[e] * p.y[e] * cinv : p.y[e]) : nanv;
      if (!has_sol) { p.y[e] = 0.0; p.z[e] = 0.0; }
    }
  }
  sync();
  const bool was_done = done && !just_done;          // idle before this check: its flags (a failed factor's -1) stay
  if (tid < BT) {
    p.iscal[IS_DONE * BT + b] = done; p.iscal[IS_STATUS * BT + b] = status;
    p.iscal[IS_RHO_UPDATES * BT + b] = rho_updates;
    if (!was_done) { p.iscal[IS_NEED_REFACTOR * BT + b] = need_refactor; p.iscal[IS_CUR * BT + b] = iter; }
    p.dscal[DS_RHO * BT + b] = rho;
  }
  int ev = (just_done ? 1 : 0) | (need_refactor ? 2 : 0);
  if constexpr (BT > 1) ev = (__syncthreads_or(ev & 1) ? 1 : 0) | (__syncthreads_or(ev & 2) ? 2 : 0);
  return ev;
}
template <int BT, int NT, bool GX, bool WIDE = false>
__global__ __launch_bounds__(NT) void check_kernel(KernelArgs a) {
  extern __shared__ double smem[];
  (void)check_body<BT, NT, GX, WIDE>(a, smem, a.iter_end - a.iter_begin);
}

// ---- segments back to back in ONE launch (LDS-resident tiles) -------------------------------------------------------------
// The host round trip per segment (two launches, a copy of the flags, a synchronisation: 0.1-0.25 ms) is what a lone small
// QP and a continuous batch spend most of their time on.  Here a tile runs up to max_segments segments - seg_len iterations +
// the check - by itself and leaves early
//   * when all of its QPs have finished,
//   * when one of its QPs asks for a refactorisation (the refactorisation kernels follow in stream order),
//   * with a stop word: at the first segment boundary after ANY QP of the launch has finished - the caller wants to react
//     to that QP (re-linearise, update, begin again) while the others are not held up for long.
// On its way out a tile publishes its flags - and check_body the solutions of finished QPs - in host memory (pinned,
// zero-copy): no copy follows the launch, the host reads them once the launch is over.
struct AdvanceArgs {
  int max_segments, seg_len;
  int *host_is;            // [tile][IS_COUNT][BT] image of the int scalars in pinned host memory, or null
  double *host_ds;         // [tile][DS_COUNT][BT] image of the double scalars, or null
  unsigned *stop;          // device word holding the sequence number of the last launch in which a QP finished, or null
  unsigned seq;            // this launch's sequence number (> 0)
  unsigned *counter;       // device words: [0] tiles that have left this launch, [1] tiles that iterated in it (the last tile resets both)
  unsigned *host_done;     // pinned words: [0] receives seq when every tile has left and published, [1] the tiles that iterated
};
template <int BT, int NT>
__global__ __launch_bounds__(NT) void advance_kernel(KernelArgs a, AdvanceArgs v) {
  extern __shared__ double smem[];
  const int tile = blockIdx.x, tid = threadIdx.x;
  int *is = a.iscal + (size_t)tile * IS_COUNT * BT;
  // solves begun (or resumed after their refactorisation) on the handle's second stream join the first launch that sees them
  if (tid < BT && __hip_atomic_load(&is[IS_PENDING * BT + tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 1) {
    is[IS_PENDING * BT + tid] = 0; is[IS_DONE * BT + tid] = 0;
  }
  __syncthreads();
  bool iterated = false;
  for (int sgm = 0; sgm < v.max_segments; sgm++) {
    const int done = tid < BT ? is[IS_DONE * BT + tid] : 1;
    if (__syncthreads_and(done)) break;
    iterated = true;
    iterate_body<BT, NT, false>(a, smem, v.seg_len);
    // (inlined next to iterate_body the two bodies share one register allocation - 128 VGPRs + spills at 16 waves against
    //  68 for iterate_kernel alone - and every iteration pays ~10 % for it; check_body out of line, as a real function, was
    //  measured twice as slow: the call's register convention puts scratch traffic into the sweeps)
    const int ev = check_body<BT, NT, false>(a, smem, v.seg_len);
    if ((ev & 1) && v.stop && tid == 0) __hip_atomic_store(v.stop, v.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ev & 2) {
      // a QP whose rho changed pauses (reads "done") until the refactorisation kernels, which follow on the second stream,
      // have given it its new factor: the other QPs of the tile and of the launch go on
      __syncthreads();
      if (tid < BT && is[IS_NEED_REFACTOR * BT + tid] == 1 && !is[IS_DONE * BT + tid]) { is[IS_DONE * BT + tid] = 1; is[IS_PENDING * BT + tid] = 2; }
    }
    if (v.stop && sgm + 1 < v.max_segments) {
      const unsigned sw = __hip_atomic_load(v.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (a tile that misses it runs one more segment)
      if (__syncthreads_or(sw == v.seq)) break;
    }
    __syncthreads();
  }
  if (!v.host_is) return;
  // flags of this tile -> host (after the solutions check_body stored: system-scope fence in between), then the tile counts
  // itself out; the last one tells the host that the launch is over
  __syncthreads();
  if (tid < IS_COUNT * BT) v.host_is[(size_t)tile * IS_COUNT * BT + tid] = is[tid];
  if (v.host_ds && tid < DS_COUNT * BT) v.host_ds[(size_t)tile * DS_COUNT * BT + tid] = a.dscal[(size_t)tile * DS_COUNT * BT + tid];
  __threadfence_system();
  __syncthreads();
  if (tid == 0 && v.counter) {
    if (iterated) __hip_atomic_fetch_add(v.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned left = __hip_atomic_fetch_add(v.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (left == gridDim.x) {
      const unsigned active = __hip_atomic_load(v.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(v.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(v.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v.host_done[1] = active;
      __threadfence_system();
      __hip_atomic_store(v.host_done, v.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------- standalone ops

// Px, A'y, Ax for QP-major x[B][n], y[B][m]  (rows E11 / E14)
template <int BT, int NT, bool GX, bool WIDE = false>
__global__ __launch_bounds__(NT) void spmv_kernel(KernelArgs a, const double *__restrict__ gx,
                                                    const double *__restrict__ gy, double *gPx,
                                                    double *gAty, double *gAx) {
  extern __shared__ double smem[];
  const int tile = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.n, m = a.m;
  double *lds_rest;
  double *xs = solve_vector<BT, GX>(a, smem, tile, lds_rest);
  const TilePtrs<BT> p = tile_ptrs<BT>(a, tile);
  // results go to LDS when the launcher provided room for them (op_out_lds), else to the tile's global scratch
  double *res = a.op_out_lds ? smem + (size_t)(n + m) * BT : p.out1;
  // coalesced QP-major I/O: consecutive threads touch consecutive elements of one QP
  for (int bb = 0; bb < BT; bb++) {
    const int q = tile * BT + bb;
    const bool ok = q < a.B;
    for (int i = tid; i < n; i += nthr) xs[(size_t)i * BT + bb] = (ok && gx) ? gx[(size_t)q * n + i] : 0.0;
    for (int i = tid; i < m; i += nthr) xs[((size_t)n + i) * BT + bb] = (ok && gy) ? gy[(size_t)q * m + i] : 0.0;
  }
  __syncthreads();
  run_spmv<BT, MI_PFV, WIDE>(a.chk, p.vchk, xs, res, wave, lane, 0, 3, (int)(blockDim.x >> 6));
  __syncthreads();
  for (int bb = 0; bb < BT; bb++) {
    const int q = tile * BT + bb;
    if (q >= a.B) continue;
    if (gPx) for (int i = tid; i < n; i += nthr) gPx[(size_t)q * n + i] = res[(size_t)i * BT + bb];
    if (gAty) for (int i = tid; i < n; i += nthr) gAty[(size_t)q * n + i] = res[((size_t)n + i) * BT + bb];
    if (gAx) for (int i = tid; i < m; i += nthr) gAx[(size_t)q * m + i] = res[((size_t)2 * n + i) * BT + bb];
  }
}


// P x, A'y, A x with ONE read of the matrices (rows E11 / E14 as an op).  The three products share their values: P is
// symmetric (upper triangle stored, each entry serves two rows) and A serves A x by rows and A'y by columns, so the
// compact values of a tile (pa_val, 8 (nnz(P) + nnz(A)) bytes per QP: 57 KB at config 3) are staged in LDS once and every
// output row is a short gather-dot-product over LDS in a fixed order (one thread per row: no atomics, deterministic).
// SURVEY 8(d) counts the three SpMVs separately (A twice); this kernel moves 0.65x those bytes.
template <int BT>
__global__ __launch_bounds__(512) void spmv_fused_kernel(KernelArgs a, SpmvFused t, const double *__restrict__ gx,
                                                         const double *__restrict__ gy, double *gPx, double *gAty, double *gAx) {
  extern __shared__ double smem[];
  const int tile = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int n = a.n, m = a.m, len = t.pa_len;
  double *vals = smem, *xs = smem + (size_t)(len + 1) * BT;
  {     // the tile's values: one contiguous block of len * BT doubles
    const double *src = t.pa_val + (size_t)tile * len * BT;
    const int tot = len * BT, pairs = tot / 2;
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(vals);
    for (int e = tid; e < pairs; e += nthr) d2[e] = s2[e];
    if (tid == 0 && (tot & 1)) vals[tot - 1] = src[tot - 1];
  }
  for (int bb = 0; bb < BT; bb++) {           // coalesced QP-major input
    const int q = tile * BT + bb;
    const bool ok = q < a.B;
    for (int i = tid; i < n; i += nthr) xs[(size_t)i * BT + bb] = (ok && gx) ? gx[(size_t)q * n + i] : 0.0;
    for (int i = tid; i < m; i += nthr) xs[((size_t)n + i) * BT + bb] = (ok && gy) ? gy[(size_t)q * m + i] : 0.0;
  }
  auto put = [&](uint32_t r, const double (&acc)[BT]) {
    double *out; size_t stride; int idx;
    if (r < (uint32_t)n) { out = gPx; stride = (size_t)n; idx = (int)r; }
    else if (r < (uint32_t)(2 * n)) { out = gAty; stride = (size_t)n; idx = (int)r - n; }
    else { out = gAx; stride = (size_t)m; idx = (int)r - 2 * n; }
    if (!out) return;
#pragma unroll
    for (int b = 0; b < BT; b++) { const int q = tile * BT + b; if (q < a.B) out[(size_t)q * stride + idx] = acc[b]; }
  };
  if (t.ell) {
    // Prefetching variant: the index words of this thread's (<= 4) rows are loaded BEFORE the staging barrier - coalesced,
    // independent, hidden behind the value loads - so that only LDS work is left after it.
    constexpr int PMAX = 4, KMAX = 24;
    uint32_t w[PMAX][KMAX], rid[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; p++) {
      rid[p] = p < t.n_pass ? t.rowid[p * 512 + tid] : 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < KMAX; k++) w[p][k] = (p < t.n_pass && (uint32_t)k < t.ell_k[p]) ? t.ell[t.ell_off[p] + (uint32_t)k * 512u + (uint32_t)tid] : 0u;
    }
    vals[(size_t)len * BT + (tid % BT)] = 0.0;        // the zero the padding entries point at
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PMAX; p++) {
      if (p >= t.n_pass) break;
      double acc[BT];
#pragma unroll
      for (int b = 0; b < BT; b++) acc[b] = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; k++) {
        if ((uint32_t)k >= t.ell_k[p]) break;
        const uint32_t vp = w[p][k] & 0xFFFFu, vi = w[p][k] >> 16;
#pragma unroll
        for (int b = 0; b < BT; b++) acc[b] = fma(vals[(size_t)vp * BT + b], xs[(size_t)vi * BT + b], acc[b]);
      }
      if (rid[p] != 0xFFFFFFFFu) put(rid[p], acc);
    }
    return;
  }
  __syncthreads();
  for (int r = tid; r < 2 * n + m; r += nthr) {
    double acc[BT];
#pragma unroll
    for (int b = 0; b < BT; b++) acc[b] = 0.0;
    for (uint32_t e = t.ptr[r]; e < t.ptr[r + 1]; e++) {
      const uint32_t w = t.ent[e], vp = w & 0xFFFFu, vi = w >> 16;
#pragma unroll
      for (int b = 0; b < BT; b++) acc[b] = fma(vals[(size_t)vp * BT + b], xs[(size_t)vi * BT + b], acc[b]);
    }
    put((uint32_t)r, acc);
  }
}

// The same product with ONE QP per workgroup (69 KB of LDS at config 3: two workgroups per CU, so that the staging of
// one overlaps the row work and the stores of the other; the tile-wide kernel above needs 138 KB and runs its tiles in
// non-overlapping rounds).  The QP's values are every BT-th double of its tile's block; the BT workgroups of a tile
// get block ids 8 apart, i.e. the same XCD (block -> XCD is round-robin), so the lines one of them fetches serve the other
// from the shared L2.  Tried and slower: persistent workgroups with the index words kept in registers and double-buffered
// staging (one workgroup of 8 waves per CU has too few loads in flight: 47 us instead of 33-35), and one workgroup
// per tile serving its QPs one after the other (47 us).
__global__ __launch_bounds__(512) void spmv_fused_qp_kernel(KernelArgs a, SpmvFused t, int BT, const double *__restrict__ gx,
                                                           const double *__restrict__ gy, double *gPx, double *gAty, double *gAx) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int g = blockIdx.x, grp = g / (8 * BT), within = g % (8 * BT);
  const int tile = grp * 8 + within % 8, b = within / 8, q = tile * BT + b;
  const int n = a.n, m = a.m, len = t.pa_len;
  if (q >= a.B) return;
  double *vals = smem, *xs = smem + (size_t)len + 1;
  const double *src = t.pa_val + (size_t)tile * len * BT + b;
  for (int e = tid; e < len; e += nthr) vals[e] = src[(size_t)e * BT];
  for (int i = tid; i < n; i += nthr) xs[i] = gx ? gx[(size_t)q * n + i] : 0.0;
  for (int i = tid; i < m; i += nthr) xs[n + i] = gy ? gy[(size_t)q * m + i] : 0.0;
  constexpr int PMAX = 4, KMAX = 24;
  uint32_t w[PMAX][KMAX], rid[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; p++) {
    rid[p] = p < t.n_pass ? t.rowid[p * 512 + tid] : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < KMAX; k++) w[p][k] = (p < t.n_pass && (uint32_t)k < t.ell_k[p]) ? t.ell[t.ell_off[p] + (uint32_t)k * 512u + (uint32_t)tid] : 0u;
  }
  if (tid == 0) vals[len] = 0.0;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PMAX; p++) {
    if (p >= t.n_pass) break;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      if ((uint32_t)k >= t.ell_k[p]) break;
      acc = fma(vals[w[p][k] & 0xFFFFu], xs[w[p][k] >> 16], acc);
    }
    const uint32_t r = rid[p];
    if (r == 0xFFFFFFFFu) continue;
    if (r < (uint32_t)n) { if (gPx) gPx[(size_t)q * n + r] = acc; }
    else if (r < (uint32_t)(2 * n)) { if (gAty) gAty[(size_t)q * n + (r - n)] = acc; }
    else if (gAx) gAx[(size_t)q * m + (r - 2 * n)] = acc;
  }
}
size_t spmv_fused_lds_bytes(int n, int m, int pa_len, int BT) { return ((size_t)pa_len + 1 + n + m) * BT * sizeof(double); }
hipError_t launch_spmv_fused(const KernelArgs &a, const SpmvFused &t, int BT, int tiles, int n_cus, hipStream_t st,
                             const double *x, const double *y, double *Px, double *Aty, double *Ax) {
  (void)n_cus;
  if (t.ell && tiles % 8 == 0 && !getenv("MI_OSQP_SPMV_TILE")) {      // one QP per workgroup
    const size_t lq = spmv_fused_lds_bytes(a.n, a.m, t.pa_len, 1);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_fused_qp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(spmv_fused_qp_kernel, dim3(tiles * BT), dim3(512), lq, st, a, t, BT, x, y, Px, Aty, Ax);
    return hipGetLastError();
  }
  const size_t lds = spmv_fused_lds_bytes(a.n, a.m, t.pa_len, BT);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, st, a, t, x, y, Px, Aty, Ax);
    return hipGetLastError();
  };
  if (BT == 1) return go(&spmv_fused_kernel<1>);
  if (BT == 2) return go(&spmv_fused_kernel<2>);
  return go(&spmv_fused_kernel<4>);
}

// sol = K^-1 rhs for QP-major rhs[B][N]  (row E7)
template <int BT, int NT, bool GX, bool WIDE = false>
__global__ __launch_bounds__(NT) void kkt_solve_kernel(KernelArgs a, const double *__restrict__ rhs, double *sol) {
  extern __shared__ double smem[];
  const Mw mw{a.mw_bar, GX && BT == 1 && a.mw_groups > 1 ? (unsigned)a.mw_groups : 1u};      // (multi-workgroup mode: see iterate_kernel)
  const bool multi = GX && BT == 1 && mw.G > 1;
  const int tile = multi ? 0 : blockIdx.x;
  const int tid = multi ? blockIdx.x * blockDim.x + threadIdx.x : threadIdx.x, nthr = multi ? blockDim.x * mw.G : blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int N = a.N;
  double *lds_rest;
  double *xs = solve_vector<BT, GX>(a, smem, tile, lds_rest);
  const TilePtrs<BT> p = tile_ptrs<BT>(a, tile);
  bool df = false;
  if constexpr (GX && BT == 1 && WIDE) df = a.df != 0;
  if (df) {
    if constexpr (GX && BT == 1 && WIDE) {
      const uint32_t sh = a.df_shadow;
      for (int i = tid; i < N; i += nthr) {
        const uint32_t pe = a.pinv[i];
        st_sc1(xs + pe, rhs[i]);
        df_arm_fwd(xs, pe, a.rflag[pe], a.xloc[pe], sh);
      }
      if (tid == 0) st_sc1(xs + 2 * (size_t)sh, 0.0);
      wg_or_grid_barrier(mw);
      kkt_solve_lds<BT, MI_PFV_NT(NT, GX), GX, WIDE>(a, p, xs, tid, nthr, wave, nw, lane, mw);
      for (int i = tid; i < N; i += nthr) { const uint32_t pe = a.pinv[i]; sol[i] = ld_sc1(xs + df_bloc(pe, a.rflag[pe], sh)); }
    }
    return;
  }
  for (int bb = 0; bb < BT; bb++) {           // coalesced QP-major I/O
    const int q = tile * BT + bb;
    for (int i = tid; i < N; i += nthr) xs[(size_t)a.pinv[i] * BT + bb] = q < a.B ? rhs[(size_t)q * N + i] : 0.0;
  }
  __syncthreads();
  kkt_solve_lds<BT, MI_PFV_NT(NT, GX), GX, WIDE>(a, p, xs, tid, nthr, wave, nw, lane, mw);
  for (int bb = 0; bb < BT; bb++) {
    const int q = tile * BT + bb;
    if (q < a.B) for (int i = tid; i < N; i += nthr) sol[(size_t)q * N + i] = xs[(size_t)a.pinv[i] * BT + bb];
  }
}

// Debug twin of kkt_solve_kernel<2, 512, false> (MI_OSQP trace entry point, scripts/trace_phases.py): same
// solve, plus per-phase / per-wave shader-clock stamps of tiles {0, gridDim/2} copied to trace[2][words].
// Layout of one tile's words: [0..3] = memtime / memrealtime at start and end (low words), [4..5] = memtime before / after
// the step between the sweeps (D^-1 scaling, dense tail), [6..7] unused,
// then per sweep and wave (cycles spent waiting for ring slots, real steps), then fwd stamps [n_phases_fwd][nw][2], then bwd stamps.
template <int TRL>
__global__ __launch_bounds__(512) void kkt_trace_kernel(KernelArgs a, const double *__restrict__ rhs, double *sol,
                                                       uint32_t *trace, uint32_t words) {
  constexpr int BT = 2;
  extern __shared__ double smem[];
  const int tile = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
  const int N = a.N;
  double *lds_rest;
  double *xs = solve_vector<BT, false>(a, smem, tile, lds_rest);
  uint32_t *tr = reinterpret_cast<uint32_t *>(xs + ((size_t)a.xs_len + 2 * (size_t)a.dt.k) * BT);    // behind the dense tail's accumulation vectors
  const TilePtrs<BT> p = tile_ptrs<BT>(a, tile);
  for (uint32_t i = tid; i < words; i += nthr) tr[i] = 0;
  for (int bb = 0; bb < BT; bb++) {
    const int q = tile * BT + bb;
    for (int i = tid; i < N; i += nthr) xs[(size_t)a.pinv[i] * BT + bb] = q < a.B ? rhs[(size_t)q * N + i] : 0.0;
  }
  __syncthreads();
  if (tid == 0) { tr[0] = (uint32_t)__builtin_amdgcn_s_memtime(); tr[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); }
  uint32_t *twf = tr + 8, *twb = twf + 2 * nw;          // per wave: cycles waited for ring slots, real steps
  uint32_t *trf = twb + 2 * nw, *trb = trf + (size_t)a.fwd.n_phases * nw * 2;
  run_tri<BT, MI_PFV, false, TRL>(a.fwd, p.vfwd, xs, wave, lane, trf, twf);
  if (tid == 0) tr[4] = (uint32_t)__builtin_amdgcn_s_memtime();
  kkt_middle<BT, false>(a, p.dinv, p.vdt, p.act, xs, tid, nthr, wave, lane);
  if (tid == 0) tr[5] = (uint32_t)__builtin_amdgcn_s_memtime();
  run_tri<BT, MI_PFV, false, TRL>(a.bwd, p.vbwd, xs, wave, lane, trb, twb);
  if (tid == 0) { tr[2] = (uint32_t)__builtin_amdgcn_s_memtime(); tr[3] = (uint32_t)__builtin_amdgcn_s_memrealtime(); }
  for (int bb = 0; bb < BT; bb++) {
    const int q = tile * BT + bb;
    if (q < a.B) for (int i = tid; i < N; i += nthr) sol[(size_t)q * N + i] = xs[(size_t)a.pinv[i] * BT + bb];
  }
  __syncthreads();
  const int sel = tile == 0 ? 0 : (tile == (int)gridDim.x / 2 ? 1 : -1);
  if (sel >= 0) for (uint32_t i = tid; i < words; i += nthr) trace[(size_t)sel * words + i] = tr[i];
}

// warm start (row E14): x <- Dinv .* x0 ; z <- A x   (QP-major x0[B][n])
template <int BT, int NT, bool GX, bool WIDE = false>
__global__ __launch_bounds__(NT) void warm_start_kernel(KernelArgs a, const double *__restrict__ x0) {
  extern __shared__ double smem[];
  const int tile = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), b = tid % BT;
  const int n = a.n, m = a.m;
  const int qp = tile * BT + b;
  // per-QP form (a.sel): only the addressed QPs of the tile take a new x and z; input row = position in the caller's list
  const int row = a.sel ? a.sel[qp] - 1 : (qp < a.B ? qp : -1);
  if (!__syncthreads_or(row >= 0)) return;
  double *lds_rest;
  double *xs = solve_vector<BT, GX>(a, smem, tile, lds_rest);
  const TilePtrs<BT> p = tile_ptrs<BT>(a, tile);
  for (int e = tid; e < n * BT; e += nthr) {
    double v = row >= 0 ? x0[(size_t)row * n + e / BT] : 0.0;
    if (a.scaling) v *= p.Dsc_inv[e];
    xs[e] = v;
    if (row >= 0 || !a.sel) p.x[e] = v;
  }
  for (int e = tid; e < m * BT; e += nthr) xs[(size_t)n * BT + e] = 0.0;
  __syncthreads();
  run_spmv<BT, MI_PFV, WIDE>(a.chk, p.vchk, xs, p.out1, wave, lane, 2, 3, (int)(blockDim.x >> 6));
  __syncthreads();
  for (int e = tid; e < m * BT; e += nthr) if (row >= 0 || !a.sel) p.z[e] = p.out1[(size_t)2 * n * BT + e];
}

// ------------------------------------------- device refactorisation (row E13)
// Block left-looking LDL' on the chunk structure (host_core.hpp BlockFactor).
// One workgroup per tile; lane = (row i of a <=16-row block, QP b).  Per level of
// chunk columns:  U (pull updates, one wave per target block, the scaled source
// block B*D staged in wave-private LDS and read back as broadcasts),
// D (LDL' of the diagonal block inside one wave), T (row-wise triangular solve).
// The result is scattered straight into the forward/backward schedule order.

#define MI_BS(k, b, j) ((((k) * BT + (b)) << 4) + (j))
// order LDS traffic of one wave (lanes exchange data through wave-private LDS)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int BT>
__device__ __forceinline__ void fct_update(const FactorArgs &a, double *Lb, const double *Dl, double *Ss,
                                           const uint4 ud, int lane) {
  // lane = (h, i, b): i = row of the <=16-row target block, b = QP of the tile, and h splits the
  // source work (the k range of a source block / the members of a rank-1 batch) over the 64/(16*BT)
  // lane groups that would otherwise idle; the partial results are summed at the end.
  // (Software-pipelining the operand loads across batches was tried: it spills at 128 VGPRs and is slower.)
  constexpr int H = 64 / (MI_CHUNK * BT);
  mi_cptr tri4 = as_const(a.tri4);        // resolved triples: {offset A, offset B, first column of K, (h_A << 16) | (w_K << 8) | h_B}
  // the task's descriptor (BlockFactor::utask4) arrives in registers: the caller loads the next one while this task runs
  const uint32_t off = ud.x, q0 = ud.y, qm = q0 + (ud.z & 0x3FFFFFu), q1 = qm + ud.w, h = ud.z >> 27, w = (ud.z >> 22) & 31u;
  const int hh = lane / (MI_CHUNK * BT), i = (lane / BT) % MI_CHUNK, b = lane % BT;
  const bool row_ok = (uint32_t)i < h;
  double acc[MI_CHUNK];
#pragma unroll
  for (int j = 0; j < MI_CHUNK; j++) acc[j] = (hh == 0 && row_ok && (uint32_t)j < w) ? Lb[((size_t)off + j * h + i) * BT + b] : 0.0;
  // ---- rank-1 sources (one-column chunks), 8 per batch: all operand loads of a batch are in flight
  // together, the 8 scaled B columns go through wave-private LDS; group hh applies members g = hh mod H
  constexpr int G = 8;                     // rank-1 sources per batch (16 at BT = 1 fits the registers but is not faster)
  for (uint32_t q = q0; q < ((MI_DBG_SKIP(a) & 1) ? q0 : qm); q += G) {
    double av[G / H], bv[G / H];
#pragma unroll
    for (int gg = 0; gg < G / H; gg++) {
      const int g = gg * H + hh;
      av[gg] = 0.0; bv[gg] = 0.0;
      if (q + g < qm) {
        const uint4 t4 = reinterpret_cast<const uint4 *>(a.tri4)[q + g];      // one 16-byte load per member, no dependent chain
        const uint32_t ao = t4.x, bo = t4.y, kc0 = t4.z, ah = t4.w >> 16, bh = t4.w & 255u;
        if ((uint32_t)i < ah) av[gg] = Lb[((size_t)ao + i) * BT + b];
        if ((uint32_t)i < bh) bv[gg] = Lb[((size_t)bo + i) * BT + b] * Dl[(size_t)kc0 * BT + b];
      }
    }
#pragma unroll
    for (int gg = 0; gg < G / H; gg++) Ss[MI_BS(gg * H + hh, b, i)] = bv[gg];
    wave_sync();
#pragma unroll
    for (int gg = 0; gg < G / H; gg++) {
      const double2 *bs = reinterpret_cast<const double2 *>(&Ss[MI_BS(gg * H + hh, b, 0)]);
#pragma unroll
      for (int j2 = 0; j2 < MI_CHUNK / 2; j2++) {
        const double2 bvv = bs[j2];
        acc[2 * j2] = fma(-av[gg], bvv.x, acc[2 * j2]);
        acc[2 * j2 + 1] = fma(-av[gg], bvv.y, acc[2 * j2 + 1]);
      }
    }
    wave_sync();
  }
  // ---- general sources (width > 1): group hh handles the columns k = hh mod H
  for (uint32_t q = qm; q < ((MI_DBG_SKIP(a) & 2) ? qm : q1); q++) {
    const uint32_t ao = tri4[4 * q], bo = tri4[4 * q + 1], kc0 = tri4[4 * q + 2], pk = tri4[4 * q + 3];
    const uint32_t ah = pk >> 16, aw = (pk >> 8) & 255u, bh = pk & 255u;
    // all operand loads of the triple are issued up front (fixed unroll, predicated):
    // lane (hh, i, b) needs A[i, k] and provides (B .* d)[i, k] for its columns k
    const bool brow = (uint32_t)i < bh, arow = (uint32_t)i < ah;
    double avk[MI_CHUNK / H], bvk[MI_CHUNK / H];
#pragma unroll
    for (int kk = 0; kk < MI_CHUNK / H; kk++) {
      const uint32_t k = (uint32_t)(kk * H + hh);
      avk[kk] = (arow && k < aw) ? Lb[((size_t)ao + k * ah + i) * BT + b] : 0.0;
      bvk[kk] = (brow && k < aw) ? Lb[((size_t)bo + k * bh + i) * BT + b] * Dl[((size_t)kc0 + k) * BT + b] : 0.0;
    }
#pragma unroll
    for (int kk = 0; kk < MI_CHUNK / H; kk++) Ss[MI_BS(kk * H + hh, b, i)] = bvk[kk];
    wave_sync();
#pragma unroll
    for (int kk = 0; kk < MI_CHUNK / H; kk++) {
      if ((uint32_t)(kk * H) < aw) {          // uniform over the wave (some groups may run one zero column extra)
        const double2 *bs = reinterpret_cast<const double2 *>(&Ss[MI_BS(kk * H + hh, b, 0)]);
#pragma unroll
        for (int j2 = 0; j2 < MI_CHUNK / 2; j2++) {
          const double2 bvv = bs[j2];
          acc[2 * j2] = fma(-avk[kk], bvv.x, acc[2 * j2]);
          acc[2 * j2 + 1] = fma(-avk[kk], bvv.y, acc[2 * j2 + 1]);
        }
      }
    }
    wave_sync();
  }
  // ---- combine the H partial results
  if constexpr (H > 1) {
#pragma unroll
    for (int j = 0; j < MI_CHUNK; j++) {
      if constexpr (H == 4) acc[j] += shfl_xor_d(acc[j], 16);
      acc[j] += shfl_xor_d(acc[j], 32);
    }
  }
  if (hh == 0 && row_ok) {
#pragma unroll
    for (int j = 0; j < MI_CHUNK; j++) if ((uint32_t)j < w) Lb[((size_t)off + j * h + i) * BT + b] = acc[j];
  }
}

// The same update for SEVERAL small tasks at once (four at one QP per workgroup, two at two): lane group g owns task g entirely - its
// descriptor, its sources (two rank-1 sources per trip; a general source four columns per trip), its accumulators; the groups
// share nothing but the instruction stream and the wave-private staging rows (row c * 4 + g belongs to group g).  The factors
// of the trajectory QPs are thousands of tasks with a handful of sources each: what they cost is the per-task chain of
// dependent loads, four of which now run side by side.
template <int BT>
__device__ __forceinline__ void fct_update_quad(const FactorArgs &a, double *Lb, const double *Dl, double *Ss, const uint4 ud, const bool valid, int lane) {
  constexpr int NG = 64 / (MI_CHUNK * BT);
  const int g = lane / (MI_CHUNK * BT), i = (lane / BT) % MI_CHUNK, b = lane % BT;
  const uint4 *tri4 = reinterpret_cast<const uint4 *>(a.tri4);
  const uint32_t off = ud.x, q0 = ud.y, qm = q0 + (ud.z & 0x3FFFFFu), q1 = qm + ud.w, h = ud.z >> 27, w = (ud.z >> 22) & 31u;
  const bool row_ok = valid && (uint32_t)i < h;
  constexpr int WQ = 8;                    // columns of a small task's target (BlockFactor::ubig)
  double acc[WQ];
#pragma unroll
  for (int j = 0; j < WQ; j++) acc[j] = (row_ok && (uint32_t)j < w) ? Lb[((size_t)off + j * h + i) * BT + b] : 0.0;
  for (uint32_t q = q0; __builtin_amdgcn_ballot_w64(valid && q < qm) != 0ull; q += 2) {
    double av[2], bv[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      av[c] = 0.0; bv[c] = 0.0;
      if (valid && q + c < qm) {
        const uint4 t4 = tri4[q + c];
        const uint32_t ao = t4.x, bo = t4.y, kc0 = t4.z, ah = t4.w >> 16, bh = t4.w & 255u;
        if ((uint32_t)i < ah) av[c] = Lb[((size_t)ao + i) * BT + b];
        if ((uint32_t)i < bh) bv[c] = Lb[((size_t)bo + i) * BT + b] * Dl[(size_t)kc0 * BT + b];
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) Ss[MI_BS(c * NG + g, b, i)] = bv[c];
    wave_sync();
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const double2 *bs = reinterpret_cast<const double2 *>(&Ss[MI_BS(c * NG + g, b, 0)]);
#pragma unroll
      for (int j2 = 0; j2 < WQ / 2; j2++) {
        const double2 bvv = bs[j2];
        acc[2 * j2] = fma(-av[c], bvv.x, acc[2 * j2]);
        acc[2 * j2 + 1] = fma(-av[c], bvv.y, acc[2 * j2 + 1]);
      }
    }
    wave_sync();
  }
  for (uint32_t q = qm; __builtin_amdgcn_ballot_w64(valid && q < q1) != 0ull; q++) {
    uint32_t ao = 0, bo = 0, kc0 = 0, ah = 0, aw = 0, bh = 0;
    if (valid && q < q1) { const uint4 t4 = tri4[q]; ao = t4.x; bo = t4.y; kc0 = t4.z; ah = t4.w >> 16; aw = (t4.w >> 8) & 255u; bh = t4.w & 255u; }
    const bool brow = (uint32_t)i < bh, arow = (uint32_t)i < ah;
    for (uint32_t k0 = 0; __builtin_amdgcn_ballot_w64(k0 < aw) != 0ull; k0 += 2) {
      double avk[2], bvk[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const uint32_t k = k0 + (uint32_t)c;
        avk[c] = (arow && k < aw) ? Lb[((size_t)ao + k * ah + i) * BT + b] : 0.0;
        bvk[c] = (brow && k < aw) ? Lb[((size_t)bo + k * bh + i) * BT + b] * Dl[((size_t)kc0 + k) * BT + b] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < 2; c++) Ss[MI_BS(c * NG + g, b, i)] = bvk[c];
      wave_sync();
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const double2 *bs = reinterpret_cast<const double2 *>(&Ss[MI_BS(c * NG + g, b, 0)]);
#pragma unroll
        for (int j2 = 0; j2 < WQ / 2; j2++) {
          const double2 bvv = bs[j2];
          acc[2 * j2] = fma(-avk[c], bvv.x, acc[2 * j2]);
          acc[2 * j2 + 1] = fma(-avk[c], bvv.y, acc[2 * j2 + 1]);
        }
      }
      wave_sync();
    }
  }
  if (row_ok) {
#pragma unroll
    for (int j = 0; j < WQ; j++) if ((uint32_t)j < w) Lb[((size_t)off + j * h + i) * BT + b] = acc[j];
  }
}

template <int BT>
__device__ __forceinline__ void fct_diag(const FactorArgs &a, double *Lb, double *Dl, double *dinv, double *Ss,
                                         const uint4 tb, int lane, int &npos) {
  const uint32_t off = tb.x, c0 = tb.z, w = tb.w & 255u;
  const int i = lane / BT, b = lane % BT;
  const bool ok = (uint32_t)i < w && lane < MI_CHUNK * BT;
  if (lane < MI_CHUNK * BT) {
#pragma unroll
    for (int k = 0; k < MI_CHUNK; k++) Ss[MI_BS(k, b, i)] = (ok && (uint32_t)k < w) ? Lb[((size_t)off + k * w + i) * BT + b] : 0.0;
  }
  wave_sync();
  for (uint32_t j = 0; j < w; j++) {
    const double d = Ss[MI_BS(j, b, j)];
    const double di = 1.0 / d;
    double lij = 0.0;
    if (ok && (uint32_t)i > j) { lij = Ss[MI_BS(j, b, i)] * di; Ss[MI_BS(j, b, i)] = lij; }
    if (ok && (uint32_t)i == j) {
      Dl[((size_t)c0 + j) * BT + b] = d; dinv[((size_t)c0 + j) * BT + b] = di;
      if (d > 0.0) npos++;
    }
    wave_sync();
    if (ok && (uint32_t)i > j) {
      const double ld = lij * d;
      for (uint32_t k = j + 1; k <= (uint32_t)i; k++) Ss[MI_BS(k, b, i)] -= ld * Ss[MI_BS(j, b, k)];
    }
    wave_sync();
  }
  if (ok) {
    for (uint32_t k = 0; k < (uint32_t)i; k++) Lb[((size_t)off + k * w + i) * BT + b] = Ss[MI_BS(k, b, i)];
  }
  // inv(L_JJ) for the solve schedules (phase B = product with the inverted diagonal block), IN PLACE in the
  // staged copy (L itself is already back in global memory): for j = w-2 .. 0, column j of X = inv(L) follows from
  // X L = I:  X[i,j] = -L[i,j] - sum_{p=j+1}^{i-1} X[i,p] L[p,j]; lane (i, b) owns row i.  Register-light on purpose
  // (a column-per-lane variant with 16 live values spilled).  Stored in the unused upper triangle of the block:
  // inv[i,k] at (row k, col i).
  if (w >= 2) {
    for (int j = (int)w - 2; j >= 0; j--) {
      double x = 0.0;
      if (ok && i > j) {
        x = -Ss[MI_BS(j, b, i)];
        for (int p2 = j + 1; p2 < i; p2++) x = fma(-Ss[MI_BS(p2, b, i)], Ss[MI_BS(j, b, p2)], x);
      }
      wave_sync();
      if (ok && i > j) Ss[MI_BS(j, b, i)] = x;
      wave_sync();
    }
    if (ok) {
      for (uint32_t k = 0; k < (uint32_t)i; k++) Lb[((size_t)off + (size_t)i * w + k) * BT + b] = Ss[MI_BS(k, b, i)];
    }
  }
}

template <int BT>
__device__ __forceinline__ void fct_trsm(const FactorArgs &a, double *Lb, const double *Dl, const double *dinv,
                                         double *Ss, const uint4 td, int lane) {
  const uint32_t off = td.x, h = td.z >> 8, w = td.z & 255u, c0 = td.y;
  const uint32_t db_off = td.w;
  const int i = lane / BT, b = lane % BT;
  // stage the diagonal block: Ss(k, b, j) = L_JJ[j,k] (strictly lower)
  if (lane < MI_CHUNK * BT) {
    for (uint32_t k = 0; k < w; k++)
      Ss[MI_BS(k, b, i)] = ((uint32_t)i < w && (uint32_t)i > k) ? Lb[((size_t)db_off + k * w + i) * BT + b] : 0.0;
  }
  wave_sync();
  if ((uint32_t)i < h && lane < MI_CHUNK * BT) {
    double ld[MI_CHUNK];
#pragma unroll
    for (int j = 0; j < MI_CHUNK; j++) {
      if ((uint32_t)j < w) {
        double v = Lb[((size_t)off + j * h + i) * BT + b];
#pragma unroll
        for (int k = 0; k < j; k++) v = fma(-ld[k], Ss[MI_BS(k, b, j)], v);
        ld[j] = v;                                           // = l_ij * d_j
        Lb[((size_t)off + j * h + i) * BT + b] = v * dinv[((size_t)c0 + j) * BT + b];
      }
    }
  }
  wave_sync();
}

// Several NARROW diagonal blocks (<= 4 columns) per wave: lane group g works in rows 4 g .. 4 g + 3 of the wave's staging area.
// Same operations as fct_diag.
template <int BT>
__device__ __forceinline__ void fct_diag_quad(double *Lb, double *Dl, double *dinv, double *Ss, const uint4 tb, const bool valid, int lane, int &npos) {
  const int g = lane / (MI_CHUNK * BT), i = (lane / BT) % MI_CHUNK, b = lane % BT;
  const uint32_t off = tb.x, c0 = tb.z, w = tb.w & 255u;
  const bool ok = valid && (uint32_t)i < w;
#define MI_QS(k, j) Ss[MI_BS(4 * g + (k), b, (j))]
#pragma unroll
  for (int k = 0; k < 4; k++) MI_QS(k, i) = (ok && (uint32_t)k < w) ? Lb[((size_t)off + k * w + i) * BT + b] : 0.0;
  wave_sync();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const bool on = ok && (uint32_t)j < w;
    const double d = MI_QS(j, j);
    const double di = 1.0 / d;
    double lij = 0.0;
    if (on && i > j) { lij = MI_QS(j, i) * di; MI_QS(j, i) = lij; }
    if (on && i == j) {
      Dl[((size_t)c0 + j) * BT + b] = d; dinv[((size_t)c0 + j) * BT + b] = di;
      if (d > 0.0) npos++;
    }
    wave_sync();
    if (on && i > j) {
      const double ld = lij * d;
      for (int k = j + 1; k <= i; k++) MI_QS(k, i) -= ld * MI_QS(j, k);
    }
    wave_sync();
  }
  if (ok) { for (int k = 0; k < i; k++) Lb[((size_t)off + k * w + i) * BT + b] = MI_QS(k, i); }
#pragma unroll
  for (int j = 2; j >= 0; j--) {
    const bool on = ok && (uint32_t)(j + 2) <= w && i > j;
    double x = 0.0;
    if (on) {
      x = -MI_QS(j, i);
      for (int p2 = j + 1; p2 < i; p2++) x = fma(-MI_QS(p2, i), MI_QS(j, p2), x);
    }
    wave_sync();
    if (on) MI_QS(j, i) = x;
    wave_sync();
  }
  if (ok && w >= 2) { for (int k = 0; k < i; k++) Lb[((size_t)off + (size_t)i * w + k) * BT + b] = MI_QS(k, i); }
}
// Several triangular solves against NARROW diagonal blocks per wave (same operations as fct_trsm).
template <int BT>
__device__ __forceinline__ void fct_trsm_quad(double *Lb, const double *dinv, double *Ss, const uint4 td, const bool valid, int lane) {
  const int g = lane / (MI_CHUNK * BT), i = (lane / BT) % MI_CHUNK, b = lane % BT;
  const uint32_t off = td.x, h = td.z >> 8, w = td.z & 255u, c0 = td.y, db_off = td.w;
#pragma unroll
  for (int k = 0; k < 4; k++) MI_QS(k, i) = (valid && (uint32_t)k < w && (uint32_t)i < w && i > k) ? Lb[((size_t)db_off + k * w + i) * BT + b] : 0.0;
  wave_sync();
  if (valid && (uint32_t)i < h) {
    double ld[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if ((uint32_t)j < w) {
        double v = Lb[((size_t)off + j * h + i) * BT + b];
#pragma unroll
        for (int k = 0; k < j; k++) v = fma(-ld[k], MI_QS(k, j), v);
        ld[j] = v;
        Lb[((size_t)off + j * h + i) * BT + b] = v * dinv[((size_t)c0 + j) * BT + b];
      }
    }
  }
  wave_sync();
#undef MI_QS
}

// LBL: the block storage, D and the new inverted diagonal live in LDS instead of the global scratch (one workgroup per QP,
// no dense tail: nobody else reads them).  Worth 6 % on 256 GOMP QPs of 1 600 rows (0.67 -> 0.63 ms); a lone QP stays
// faster on a group of workgroups with global storage (its levels hold hundreds of block tasks: it is short of waves).
template <int BT, bool LBL = false>
__global__ __launch_bounds__(1024) void factor_kernel(FactorArgs a) {
  extern __shared__ double smem[];
  // (a QP shared by mw_groups consecutive workgroups - short work lists: a lone QP is a chain of latency-bound levels, and
  //  the single large QPs have 10^4 .. 10^5 block tasks: threads / waves are numbered over the group, the barriers between
  //  the phases of a level are barriers of the group, each group with its own counters)
  bool multi = false;
  if constexpr (BT == 1) multi = a.mw_groups > 1;
  const unsigned G = multi ? (unsigned)a.mw_groups : 1u;
  const int tile = (int)(blockIdx.x / G), gidx = (int)(blockIdx.x % G);
  const Mw mw{a.mw_bar + 4 * (size_t)tile, G};
  const int tid = gidx * (int)blockDim.x + (int)threadIdx.x, nthr = (int)(blockDim.x * G);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6, b = tid % BT;
  auto sync = [&]() { if constexpr (BT == 1) wg_or_grid_barrier(mw); else __syncthreads(); };
  const int m = a.m, N = a.N;
  // The QP of lane class b: with a work list the flagged QPs of the whole batch are packed BT per workgroup (a
  // refactorisation is latency-bound per tile, so fewer, fuller tiles = fewer rounds over the CUs); per-QP arrays
  // are addressed through their home slot, the block storage / D scratch through the work tile.
  const int slot = a.work ? a.work[tile * BT + b] : tile * BT + b;
  const size_t hbt = (size_t)a.home_bt;
  const size_t home = slot >= 0 ? (size_t)slot / hbt : 0, hb = slot >= 0 ? (size_t)slot % hbt : 0;
  auto H = [&](size_t len, size_t i) { return (home * len + i) * hbt + hb; };      // element i of a [tile][len][home_bt] array of this QP
  int flag = 0;
  if (slot >= 0) flag = a.work ? 1 : (a.force_all ? (slot < a.B) : a.iscal[H(IS_COUNT, IS_NEED_REFACTOR)]);
  if (!__syncthreads_or(flag)) return;
  double *Ss = smem + (size_t)(threadIdx.x >> 6) * MI_CHUNK * MI_CHUNK * BT;
  double *lds_rest = smem + (size_t)(blockDim.x >> 6) * MI_CHUNK * MI_CHUNK * BT;
  double *Lb, *Dl;
  if constexpr (LBL) { Lb = lds_rest; Dl = lds_rest + (size_t)a.storage * BT; }
  else { Lb = a.Lblk + (size_t)tile * a.storage * BT; Dl = a.Dl + (size_t)tile * N * BT; }
  const double rho = a.dscal[H(DS_COUNT, DS_RHO)];
  // ---- rho vector of the QPs being refactored ([EXT] osqp_update_rho)
  if (flag && !a.force_all) {
    for (int e = tid; e < m * BT; e += nthr) {
      const size_t k = H(m, e / BT);
      const double l = a.l[k], u = a.u[k];
      double rv;
      if (l < -MI_INFTY * MI_MIN_SCALING && u > MI_INFTY * MI_MIN_SCALING) rv = MI_RHO_MIN;
      else if (u - l < 1e-4) rv = 1e3 * rho;
      else rv = rho;
      a.rho_vec[k] = rv; a.rho_inv[k] = 1.0 / rv;
    }
  }
  {    // zero the block storage (16-byte stores; storage * BT is even or the tail is handled singly)
    const size_t tot = (size_t)a.storage * BT, pairs = tot / 2;
    double2 *L2p = reinterpret_cast<double2 *>(Lb);
    for (size_t e = tid; e < pairs; e += nthr) L2p[e] = make_double2(0.0, 0.0);
    if (tid == 0 && (tot & 1)) Lb[tot - 1] = 0.0;
    if (multi && tid == 0 && slot >= 0) a.npos[slot] = 0;
  }
  sync();
  // ---- assemble the permuted KKT into block storage
  {
    // 4 entries per thread and trip: the table reads, then the value reads, then the stores (independent loads in flight)
    constexpr int UA = 2;
    const int tot = a.nnzK * BT;
    for (int e0 = tid; e0 < tot; e0 += nthr * UA) {
      uint32_t src[UA], dst[UA];
      double v[UA];
#pragma unroll
      for (int u = 0; u < UA; u++) {
        const int e = e0 + u * nthr;
        const int k = e < tot ? e / BT : 0;
        src[u] = a.asm_src[k]; dst[u] = a.asm_dst[k];
      }
#pragma unroll
      for (int u = 0; u < UA; u++) {
        const uint32_t kind = src[u] >> 29, idx = src[u] & 0x1FFFFFFFu;
        if (kind == 0) v[u] = a.pa_val[H(a.pa_len, idx)];
        else if (kind == 1) v[u] = a.pa_val[H(a.pa_len, idx)] + a.sigma;
        else if (kind == 2) v[u] = a.sigma;
        else if (kind == 3) v[u] = a.pa_val[H(a.pa_len, (size_t)a.nnzP + idx)];
        else v[u] = -a.rho_inv[H(m, idx)];
      }
#pragma unroll
      for (int u = 0; u < UA; u++) if (e0 + u * nthr < tot) Lb[(size_t)dst[u] * BT + b] = v[u];
    }
  }
  sync();
  int npos = 0, bad_inertia = 0;
  double *dnew;
  if constexpr (LBL) dnew = lds_rest + ((size_t)a.storage + (size_t)N) * BT;
  else dnew = a.dinv_scratch + (size_t)tile * N * BT;
  for (int L = 0; L < a.n_levels; L++) {
    const uint32_t *lv = a.lvl + 6 * L;
    // (task descriptors: one 16-byte load each, requested one task ahead - a lone QP is a chain of dependent memory round
    //  trips, and descriptor -> block table -> operands used to be three of them per task)
    const uint4 *ut4 = reinterpret_cast<const uint4 *>(a.utask), *dt4 = reinterpret_cast<const uint4 *>(a.dtask), *tt4 = reinterpret_cast<const uint4 *>(a.ttask);
    if (lv[1] > lv[0]) {
      uint32_t ub = lv[1];                 // [lv[0], ub): one task per wave; [ub, lv[1]): the small tasks, one per 16-lane group
      if constexpr (BT <= 2) { if (!(MI_DBG_SKIP(a) & 32)) ub = a.ubig[3 * L]; }
      uint32_t t = lv[0] + wave;
      uint4 nxt = t < ub ? ut4[t] : make_uint4(0, 0, 0, 0);
      for (; t < ub; t += nw) {
        const uint4 cur = nxt;
        if (t + nw < ub) nxt = ut4[t + nw];
        fct_update<BT>(a, Lb, Dl, Ss, cur, lane);
      }
      if constexpr (BT <= 2) {
        constexpr uint32_t NG = 64 / (MI_CHUNK * BT);
        const uint32_t g = (uint32_t)lane / (MI_CHUNK * BT);
        uint32_t tq = ub + NG * (uint32_t)wave;
        uint4 nq = tq + g < lv[1] ? ut4[tq + g] : make_uint4(0, 0, 0, 0);
        for (; tq < lv[1]; tq += NG * (uint32_t)nw) {
          const uint4 cur = nq;
          const bool valid = tq + g < lv[1];
          const uint32_t tn = tq + NG * (uint32_t)nw + g;
          if (tn < lv[1]) nq = ut4[tn];
          fct_update_quad<BT>(a, Lb, Dl, Ss, cur, valid, lane);
        }
      }
      sync();
    }
    if (!(MI_DBG_SKIP(a) & 4)) {
      uint32_t db = lv[3];                 // [lv[2], db): one block per wave; [db, lv[3]): the narrow ones, four per wave
      if constexpr (BT <= 2) { if (!(MI_DBG_SKIP(a) & 32)) db = a.ubig[3 * L + 1]; }
      uint32_t t = lv[2] + wave;
      uint4 nxt = t < db ? dt4[t] : make_uint4(0, 0, 0, 0);
      for (; t < db; t += nw) {
        const uint4 cur = nxt;
        if (t + nw < db) nxt = dt4[t + nw];
        fct_diag<BT>(a, Lb, Dl, dnew, Ss, cur, lane, npos);
      }
      if constexpr (BT <= 2) {
        constexpr uint32_t NG = 64 / (MI_CHUNK * BT);
        const uint32_t g = (uint32_t)lane / (MI_CHUNK * BT);
        uint32_t tq = db + NG * (uint32_t)wave;
        uint4 nq = tq + g < lv[3] ? dt4[tq + g] : make_uint4(0, 0, 0, 0);
        for (; tq < lv[3]; tq += NG * (uint32_t)nw) {
          const uint4 cur = nq;
          const bool valid = tq + g < lv[3];
          const uint32_t tn = tq + NG * (uint32_t)nw + g;
          if (tn < lv[3]) nq = dt4[tn];
          fct_diag_quad<BT>(Lb, Dl, dnew, Ss, cur, valid, lane, npos);
        }
      }
    }
    sync();
    if (lv[5] > lv[4] && !(MI_DBG_SKIP(a) & 8)) {
      uint32_t tbg = lv[5];
      if constexpr (BT <= 2) { if (!(MI_DBG_SKIP(a) & 32)) tbg = a.ubig[3 * L + 2]; }
      uint32_t t = lv[4] + wave;
      uint4 nxt = t < tbg ? tt4[t] : make_uint4(0, 0, 0, 0);
      for (; t < tbg; t += nw) {
        const uint4 cur = nxt;
        if (t + nw < tbg) nxt = tt4[t + nw];
        fct_trsm<BT>(a, Lb, Dl, dnew, Ss, cur, lane);
      }
      if constexpr (BT <= 2) {
        constexpr uint32_t NG = 64 / (MI_CHUNK * BT);
        const uint32_t g = (uint32_t)lane / (MI_CHUNK * BT);
        uint32_t tq = tbg + NG * (uint32_t)wave;
        uint4 nq = tq + g < lv[5] ? tt4[tq + g] : make_uint4(0, 0, 0, 0);
        for (; tq < lv[5]; tq += NG * (uint32_t)nw) {
          const uint4 cur = nq;
          const bool valid = tq + g < lv[5];
          const uint32_t tn = tq + NG * (uint32_t)nw + g;
          if (tn < lv[5]) nq = tt4[tn];
          fct_trsm_quad<BT>(Lb, dnew, Ss, cur, valid, lane);
        }
      }
      sync();
    }
  }
  // ---- inertia: positive pivots of QP b summed over the workgroup (each pivot counted by one lane)
  {
    __shared__ int s_npos[4];
    if (threadIdx.x < 4) s_npos[threadIdx.x] = 0;
    __syncthreads();
    if (npos) atomicAdd(&s_npos[b], npos);
    __syncthreads();
    if (multi) {                            // the workgroups add their counts; everybody reads the total behind a grid barrier
      if (threadIdx.x == 0 && slot >= 0 && s_npos[0]) atomicAdd(&a.npos[slot], s_npos[0]);
      sync();
      if (threadIdx.x == 0 && slot >= 0) s_npos[0] = __hip_atomic_load(&a.npos[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    } else if (tid < BT && slot >= 0) a.npos[slot] = s_npos[b];
    if (tid < BT && flag && s_npos[b] != a.n && !MI_DBG_SKIP(a) && !a.dt_k) bad_inertia = 1;   // (dense tail: dense_inverse_kernel adds its pivots and checks)
  }
  // ---- scatter into the solve schedules (only the refactored QPs)
  if (flag && !(MI_DBG_SKIP(a) & 16)) {
    double *fv = a.fwd_val + (size_t)slot * a.fwd.n_steps * 64, *bv = a.bwd_val + (size_t)slot * a.bwd.n_steps * 64;   // this thread's QP stream
    auto scatter = [&](double *dst, const int32_t *map, uint32_t n_slots) {
      constexpr int US = 4;                 // table reads, then value reads, then stores: 4 independent chains per thread
      const uint32_t tot = n_slots * (uint32_t)BT;
      for (uint32_t e0 = tid; e0 < tot; e0 += (uint32_t)nthr * US) {
        int32_t mp[US];
        double v[US];
#pragma unroll
        for (int u = 0; u < US; u++) { const uint32_t e = e0 + (uint32_t)(u * nthr); mp[u] = e < tot ? map[e / BT] : MI_SRC_ZERO; }
#pragma unroll
        for (int u = 0; u < US; u++) v[u] = slot_value(mp[u], Lb, BT, b);
#pragma unroll
        for (int u = 0; u < US; u++) { const uint32_t e = e0 + (uint32_t)(u * nthr); if (e < tot) dst[e / BT] = v[u]; }   // dst = the stream of QP b = e % BT
      }
    };
    scatter(fv, a.fwd_srcblk, a.fwd.n_slots);
    scatter(bv, a.bwd_srcblk, a.bwd.n_slots);
    for (int e = tid; e < N * BT; e += nthr) a.dinv[H(N, e / BT)] = dnew[e];
  }
  if (tid < BT && slot >= 0 && flag && a.use_work) a.use_work[slot] = 1;      // this QP's factor now lives in the working copy
  sync();
  __threadfence();          // (continuous batching: an advance launch on another stream may read the flag)
  if (tid < BT && slot >= 0) a.iscal[H(IS_COUNT, IS_NEED_REFACTOR)] = bad_inertia ? -1 : 0;   // -1: the new factor has the wrong inertia
}

template <int BT, bool LBL = false>
static hipError_t launch_factor_t(const FactorArgs &a, int tiles, int threads, hipStream_t st) {
  size_t lds = factor_lds_bytes(BT, threads);
  if (LBL) lds += ((size_t)a.storage + 2 * (size_t)a.N) * BT * sizeof(double);
  if (a.mw_groups > 1) { if (BT != 1 || !a.mw_bar || LBL) return hipErrorInvalidValue; tiles *= a.mw_groups; }
#ifdef MI_OSQP_DEBUG_BUILD
  if (a.mw_groups > 1 && debug_drop_group("factor")) tiles--;       // fault injection: a workgroup of the last group never shows up
#endif
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&factor_kernel<BT, LBL>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((factor_kernel<BT, LBL>), dim3(tiles), dim3(threads), lds, st, a);
  return hipGetLastError();
}
size_t factor_lds_bytes(int BT, int threads) { return (size_t)(threads / 64) * MI_CHUNK * MI_CHUNK * BT * sizeof(double); }
bool factor_fits_lds(const FactorArgs &a, int threads) {
  return !a.dt_k && factor_lds_bytes(1, threads) + ((size_t)a.storage + 2 * (size_t)a.N) * sizeof(double) <= 160 * 1024 - 1024;
}
hipError_t launch_factor(const FactorArgs &a, int BT, int tiles, int threads, hipStream_t st) {
  // one QP per workgroup, no dense tail (its kernels read the block storage), everything within LDS: the LDS-resident form
  if (BT == 1 && a.mw_groups <= 1 && !getenv("MI_OSQP_FACTOR_GLOBAL") && factor_fits_lds(a, threads)) return launch_factor_t<1, true>(a, tiles, threads, st);
  switch (BT) {
    case 1: return launch_factor_t<1>(a, tiles, threads, st);
    case 2: return launch_factor_t<2>(a, tiles, threads, st);
    default: return launch_factor_t<4>(a, tiles, threads, st);
  }
}


// ---- dense tail: S and M = S^-1 (host_core.hpp DenseTail) -----------------------------------------------------
// One workgroup (8 waves) per refactored QP, after factor_kernel.  Everything dense runs on the matrix cores
// (v_mfma_f64_16x16x4_f64: the fp64 vector rate with 1/16 of the instructions and a quarter of the LDS operand traffic).
//
// Data: the k x k Schur complement as 16 x 16 tiles (lower triangle, I >= J) in a per-QP scratch, each tile stored in the
// register order of an MFMA accumulator ([lane][reg]: row = lane / 16 + 4 reg, col = lane % 16), so accumulator loads and
// stores are contiguous 32-byte pieces per lane.
//
// Phase 0, assembly: S(I, J) = KKT block - sum_c L[I, c] d_c L[J, c]' over the columns c before the tail.  The entries of
// L in the tail rows of those columns (10 k values at config 3) and their D are staged in LDS once; per tile, 4 source
// columns make one MFMA step whose operands are LDS gathers through a per-step index word (host tables, shared by all
// QPs, L2 resident).  factor_kernel used to do this with 16 x 16 rank-1 block updates at ~10 % useful work from global
// memory.
//
// Phases 1 .. k/64, inversion: symmetric sweep operator on 64 x 64 pivot blocks,
//     Pn = -inv(A_pp);  Gn = A_:p Pn;  A_ij += Gn_i A_pj (i, j outside p);  A_:p = -Gn;  A_pp = Pn      ->  A = -inv(S)
// (no pivoting, like the LDL' it replaces; the scalar pivots met inside the pivot blocks ARE the pivots of LDL', which is
// how the inertia check still works).  Per pass: the pivot block is swept inside LDS (its 16 x 16 diagonal tiles scalar,
// by one wave; the rest of its 64 x 64 as MFMA tile products); the panel C = A_:p is staged in LDS in halves, in MFMA
// operand order; every wave owns row tiles (snake-dealt by work), computes their Gn' = Pn C' - whose accumulator
// registers ARE the A-operand fragments of the trailing update, no transposition - and updates its tiles A_ij, j <= i,
// with a 3-deep prefetch of the accumulator tiles.  Traffic per QP: k/64 read-modify-write passes over the triangle
// instead of k/16 (12 MB instead of 45 MB at k = 448); flops k^3 on the matrix cores.
typedef double mi_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mi_v4d mfma_f64(double a, double b, mi_v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ mi_v4d ld_tile(const double *A, uint32_t tix, int lane) {
  const double2 *p = reinterpret_cast<const double2 *>(A + (size_t)tix * 256 + (size_t)lane * 4);
  const double2 v0 = p[0], v1 = p[1];
  return mi_v4d{v0.x, v0.y, v1.x, v1.y};
}
__device__ __forceinline__ void st_tile(double *A, uint32_t tix, int lane, mi_v4d v) {
  double2 *p = reinterpret_cast<double2 *>(A + (size_t)tix * 256 + (size_t)lane * 4);
  p[0] = make_double2(v[0], v[1]); p[1] = make_double2(v[2], v[3]);
}
__device__ __forceinline__ uint32_t tile_ix(int I, int J) { return (uint32_t)(I * (I + 1) / 2 + J); }

#define MI_TAIL_NW 8
#define MI_TAIL_TS 65              // row stride of the pivot block's LDS image

// Phase 0 of the dense tail as a kernel of its own: 16 waves per QP (1024 threads), 93 KB of LDS for the compact factor
// entries at config 3.  What bounds it (timing experiments of round 2, 1024 QPs: 0.89 ms; without the staging gather
// 0.78; with the MFMA replaced by one scalar fma 0.75; with random instead of table indices 1.05): vector-ALU issue -
// ~40 instructions of index unpacking, address arithmetic and operand scaling around every MFMA - and the LDS gathers,
// not the matrix pipe and not memory latency.  Next step if it matters: byte offsets and the column index packed into one
// 64-bit table word read through a buffer descriptor (no address arithmetic), D pre-negated.
__global__ __launch_bounds__(1024) void tail_assemble_kernel(TailArgs a) {
  extern __shared__ double smem[];
  const int g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = a.work[g];
  if (slot < 0) return;
  const int k = a.k, kbt = a.kbt, wt = g / kbt, wb = g % kbt;
  double *A = a.Sd + (size_t)g * k * k;
  const double *Lb = a.Lblk + (size_t)wt * a.storage * kbt + wb;
  const double *Dl = a.Dl + (size_t)wt * a.N * kbt + wb;
  const int l15 = lane & 15, l4 = lane >> 4;
  {
    double *La = smem, *Dc = smem + a.n_lt + 1;
    for (int e = tid; e < a.n_lt; e += nthr) La[e] = Lb[(size_t)a.lt_pos[e] * kbt];
    if (tid == 0) La[a.n_lt] = 0.0;
    for (int c = tid; c < a.n_ltcol; c += nthr) Dc[c] = -Dl[(size_t)a.ltcol_col[c] * kbt];       // (negated: S = KKT block MINUS L d L')
    __syncthreads();
    // The quads of a wave's tiles are one linear stream of 64-bit table words (tiles are laid out wave-major, each padded
    // to whole blocks of QB quads): low word = LDS byte offset of operand A's entry, high word = offset of operand B's
    // entry | offset of the source column's -d << 17.  Read through a buffer descriptor with a scalar offset (no address
    // arithmetic), one block ahead of its use; every load of the loop sits in straight-line code (clamped indices instead
    // of branches: a branch around a load makes the compiler collapse its wait counts).  The kernel is bound by vector-ALU
    // issue around its MFMAs, so the loop body is kept to ~8 vector instructions per step (the first version spent ~40 on
    // index unpacking, 64-bit addresses and operand scaling: 0.63 ms for 605 QPs).  The initial accumulator (the KKT
    // block) of the next tile is fetched one tile ahead.
    constexpr int QB = 8;
    const uint32_t t_begin = a.wave_tiles[wave], t_end = a.wave_tiles[wave + 1];
    if (t_begin < t_end) {
      const uint4 *ttab = reinterpret_cast<const uint4 *>(a.tile_tab);
      auto load_init = [&](const uint4 &tt) {
        const int I = (int)(tt.x >> 16), J = (int)(tt.x & 0xFFFFu);
        mi_v4d v;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int row = l4 + 4 * r, col = l15;
          if (I == J && row < col) { const int t = row; row = col; col = t; }        // diagonal blocks hold their lower triangle
          v[r] = Lb[((size_t)tt.y + (size_t)col * 16 + row) * kbt];
        }
        return v;
      };
      const mi_rsrc tab = make_rsrc(a.asm_q64, a.n_quads * 512u);
      const uint32_t q_last = a.n_quads ? a.n_quads - 1 : 0u;
      mi_u32x2 nxt[QB];
      auto fetch = [&](uint32_t qb) {
#pragma unroll
        for (int u = 0; u < QB; u++) {
          const uint32_t q = qb + u < q_last ? qb + u : q_last;
          nxt[u] = __builtin_amdgcn_raw_buffer_load_b64(tab, (uint32_t)lane * 8u, q * 512u, 0);
        }
      };
#pragma unroll
      for (int u = 0; u < QB; u++) nxt[u] = mi_u32x2{0u, 0u};
      uint32_t qnext = ttab[t_begin].z;                      // first quad not yet requested
      if (a.n_quads) { fetch(qnext); qnext += QB; }
      const char *lds = reinterpret_cast<const char *>(smem);
      const uint32_t dc_off = (uint32_t)(a.n_lt + 1) * 8u;
      mi_v4d acc_next = load_init(ttab[t_begin]);
      for (uint32_t ti = t_begin; ti < t_end; ti++) {
        const uint4 tt = ttab[ti];
        mi_v4d acc = acc_next;
        acc_next = load_init(ttab[ti + 1 < t_end ? ti + 1 : ti]);
        for (uint32_t q = tt.z; q < tt.w; q += QB) {
          mi_u32x2 cur[QB];
#pragma unroll
          for (int u = 0; u < QB; u++) cur[u] = nxt[u];
          fetch(qnext); qnext += QB;
          double av[QB], bv[QB];
#pragma unroll
          for (int u = 0; u < QB; u++) {
            av[u] = *reinterpret_cast<const double *>(lds + cur[u].x);
            bv[u] = *reinterpret_cast<const double *>(lds + (cur[u].y & 0x1FFFFu)) * *reinterpret_cast<const double *>(lds + dc_off + (cur[u].y >> 17));
          }
#pragma unroll
          for (int u = 0; u < QB; u++) acc = mfma_f64(av[u], bv[u], acc);
        }
        st_tile(A, tile_ix((int)(tt.x >> 16), (int)(tt.x & 0xFFFFu)), lane, acc);
      }
    }
  }
}

// MAXSLOT = row tiles a wave may own = ceil((k / 16 - 4) / 8): 4 at k = 512, 3 up to 448, ...; every slot costs 32 VGPRs
// (its Gn fragments), which the smaller instantiations spend on a deeper accumulator ring (PFT tiles in flight per wave)
template <int MI_TAIL_MAXSLOT, int PFT>
__global__ __launch_bounds__(512) void tail_kernel(TailArgs a) {
  extern __shared__ double smem[];
  const int g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = a.work[g];
  if (slot < 0) return;
  const int k = a.k, kbt = a.kbt, wt = g / kbt, wb = g % kbt, nt = k / 16;
  const size_t hbt = (size_t)a.home_bt, home = (size_t)slot / hbt, hb = (size_t)slot % hbt;
  auto H = [&](size_t len, size_t i) { return (home * len + i) * hbt + hb; };
  double *A = a.Sd + (size_t)g * k * k;                     // the tiles tail_assemble_kernel left here
  (void)wt; (void)wb;
  const int l15 = lane & 15, l4 = lane >> 4;
  // timing stamps of wave 0 (MI_OSQP_TAIL_TRACE=1: a.trace != null; results are not affected): assembly, pivot blocks,
  // panel + trailing updates, stream write - shader clocks summed over the passes
  unsigned long long tr_t = a.trace ? __builtin_amdgcn_s_memtime() : 0ull, tr_acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  auto stamp = [&](int which) {
    if (a.trace && wave == 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr_acc[which] += t - tr_t; tr_t = t; }
  };
  stamp(0);
  // ------------------------------------------------------------------ phases 1 .. k/64: blocked sweep
  const int nrt = nt - 4;                                   // row tiles outside a pivot block
  const int nh = a.nh;                                      // row tiles of one staged half of the panel (host: <= 14)
  double *Cs = smem;                                        // [nh][16 steps][64 lanes]: C in MFMA operand order; doubles as the pivot block's image T
  double *Ps = smem + (size_t)a.cs_doubles;                 // [4][16][64]: Pn as A operand of Gn' = Pn C'
  double *Ts = Cs;                                          // [64][MI_TAIL_TS]
  int npos = 0;
  for (int p = 0; p < k / 64; p++) {
    const int P0 = 4 * p;
    auto nonp = [&](int x) { return x < P0 ? x : x + 4; };  // x-th row tile outside the pivot block
    // ---- pivot block -> LDS (full symmetric image)
    for (int t = wave; t < 10; t += MI_TAIL_NW) {
      int ta = 0, tb = t;                                   // t -> (ta >= tb) in the 4 x 4 lower triangle
      while (tb > ta) { tb -= ta + 1; ta++; }
      const mi_v4d v = ld_tile(A, tile_ix(P0 + ta, P0 + tb), lane);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * ta + l4 + 4 * r, col = 16 * tb + l15;
        if (ta != tb || row >= col) {       // (a diagonal tile is symmetric only up to round-off: its lower triangle is the truth)
          Ts[row * MI_TAIL_TS + col] = v[r];
          Ts[col * MI_TAIL_TS + row] = v[r];
        }
      }
    }
    __syncthreads();
    // ---- Ts <- -inv(Ts): four 16-wide sub-sweeps
    for (int q = 0; q < 4; q++) {
      if (wave == 0) {
        // The 16 x 16 diagonal tile, scalar: U <- -inv(U).  Lane (r = lane % 16, g = lane / 16) keeps U[r][4g .. 4g+3] in
        // registers; per pivot only the pivot row travels (through 16 doubles of LDS: U is symmetric, so the row also
        // serves as the pivot column).  The update is u - (ta tb) di: commutative in (ta, tb), i.e. U stays EXACTLY
        // symmetric.  1/d by v_rcp_f64 + two Newton steps (full precision, not correctly rounded).
        double *U = Ts + (16 * q) * MI_TAIL_TS + 16 * q;
        double *rowbuf = Ps;                                // (Ps is rewritten after the block is done)
        const int r = l15, c0 = l4 * 4;
        double u[4];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) u[cc] = U[r * MI_TAIL_TS + c0 + cc];
        for (int kk = 0; kk < 16; kk++) {
          if (r == kk) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++) rowbuf[c0 + cc] = u[cc];
          }
          wave_sync();
          const double d = rowbuf[kk], ta = rowbuf[r];
          double tb[4];
#pragma unroll
          for (int cc = 0; cc < 4; cc++) tb[cc] = rowbuf[c0 + cc];
          wave_sync();
          if (lane == 0 && d > 0.0) npos++;
          double di = __builtin_amdgcn_rcp(d);
          di = fma(fma(-d, di, 1.0), di, di);
          di = fma(fma(-d, di, 1.0), di, di);
#pragma unroll
          for (int cc = 0; cc < 4; cc++) {
            const int c = c0 + cc;
            double v;
            if (r != kk && c != kk) v = fma(-(ta * tb[cc]), di, u[cc]);
            else if (r == kk && c == kk) v = -di;
            else if (c == kk) v = ta * di;
            else v = tb[cc] * di;
            u[cc] = v;
          }
        }
#pragma unroll
        for (int cc = 0; cc < 4; cc++) U[r * MI_TAIL_TS + c0 + cc] = u[cc];
      }
      __syncthreads();
      // the other three tile rows a of the block: gn_a' = Un c_a' (accumulator registers = A-operand fragments of gn_a),
      // then T[a, b] += gn_a c_b' for the three b != q
      int ta = -1;
      mi_v4d gnT = {0.0, 0.0, 0.0, 0.0};
      if (wave < 3) {
        ta = wave < q ? wave : wave + 1;
#pragma unroll
        for (int s = 0; s < 4; s++)
          gnT = mfma_f64(Ts[(16 * q + l15) * MI_TAIL_TS + 16 * q + 4 * s + l4], Ts[(16 * ta + l15) * MI_TAIL_TS + 16 * q + 4 * s + l4], gnT);
        for (int bi = 0; bi < 3; bi++) {
          const int tb = bi < q ? bi : bi + 1;
          mi_v4d acc;
#pragma unroll
          for (int r = 0; r < 4; r++) acc[r] = Ts[(16 * ta + l4 + 4 * r) * MI_TAIL_TS + 16 * tb + l15];
#pragma unroll
          for (int s = 0; s < 4; s++) acc = mfma_f64(gnT[s], Ts[(16 * tb + l15) * MI_TAIL_TS + 16 * q + 4 * s + l4], acc);
#pragma unroll
          for (int r = 0; r < 4; r++) Ts[(16 * ta + l4 + 4 * r) * MI_TAIL_TS + 16 * tb + l15] = acc[r];
        }
      }
      __syncthreads();
      if (wave < 3) {                                       // the panel column takes g_a = -gn_a (and its mirror image)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int arow = 16 * ta + l15, kcol = 16 * q + l4 + 4 * r;
          Ts[arow * MI_TAIL_TS + kcol] = -gnT[r];
          Ts[kcol * MI_TAIL_TS + arow] = -gnT[r];
        }
      }
      __syncthreads();
    }
    // ---- Pn: to the scratch (A_pp = Pn) and, in A-operand order, to Ps
    for (int t = wave; t < 10; t += MI_TAIL_NW) {
      int ta = 0, tb = t;
      while (tb > ta) { tb -= ta + 1; ta++; }
      mi_v4d v;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = Ts[(16 * ta + l4 + 4 * r) * MI_TAIL_TS + 16 * tb + l15];
      st_tile(A, tile_ix(P0 + ta, P0 + tb), lane, v);
    }
    for (int e = tid; e < 4 * 16 * 64; e += nthr) {
      const int kt = e >> 10, sp = (e >> 6) & 15, l = e & 63;
      Ps[e] = Ts[(16 * kt + (l & 15)) * MI_TAIL_TS + 4 * sp + (l >> 4)];
    }
    __syncthreads();
    stamp(1);
    if (nrt == 0) continue;
    // ---- panel staging: positions [x0, x1) of the non-pivot row tiles -> Cs (16 steps x 64 lanes per row tile)
    auto stage = [&](int x0, int x1) {
      constexpr int MAXE = 7;            // (x1 - x0) * 4 <= 56 pieces over 8 waves
      mi_v4d v[MAXE];
      const int ne = (x1 - x0) * 4;
#pragma unroll
      for (int i = 0; i < MAXE; i++) {          // (loads past the end re-read the last piece: no branch around a load)
        const int e0 = wave + i * MI_TAIL_NW, e = e0 < ne ? e0 : ne - 1;
        const int t = nonp(x0 + (e >> 2)), qq = e & 3;
        v[i] = ld_tile(A, t < P0 ? tile_ix(P0 + qq, t) : tile_ix(t, P0 + qq), lane);
      }
#pragma unroll
      for (int i = 0; i < MAXE; i++) {
        const int e = wave + i * MI_TAIL_NW;
        if (e < ne) {
          const int x = x0 + (e >> 2), qq = e & 3, t = nonp(x);
          double *dst = Cs + (size_t)(x - x0) * 1024;
          if (t < P0) {               // stored tile (P0 + qq, t): rows = pivot columns, cols = the rows of C: already operand order
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(4 * qq + r) * 64 + lane] = v[i][r];
          } else {                    // stored tile (t, P0 + qq): rows = the rows of C, cols = pivot columns: transpose on the way
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(4 * qq + (l15 >> 2)) * 64 + ((l15 & 3) << 4) + l4 + 4 * r] = v[i][r];
          }
        }
      }
    };
    // row tile of this wave's slot sl: the (sl * 8 + (sl even ? wave : 7 - wave))-th largest, i.e. position nrt - 1 - that
    auto slot_pos = [&](int sl) { const int d = sl * MI_TAIL_NW + ((sl & 1) ? MI_TAIL_NW - 1 - wave : wave); return d < nrt ? nrt - 1 - d : -1; };
    double Gn[MI_TAIL_MAXSLOT][16];
    auto compute_g = [&](int x0, int x1) {                  // Gn' = Pn C' for the owned row tiles staged in [x0, x1)
#pragma unroll
      for (int sl = 0; sl < MI_TAIL_MAXSLOT; sl++) {
        const int x = slot_pos(sl);
        if (x < x0 || x >= x1) continue;
        const double *cs = Cs + (size_t)(x - x0) * 1024 + lane;
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
          mi_v4d gacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int sp = 0; sp < 16; sp++) gacc = mfma_f64(Ps[(kt * 16 + sp) * 64 + lane], cs[sp * 64], gacc);
#pragma unroll
          for (int r = 0; r < 4; r++) Gn[sl][4 * kt + r] = gacc[r];
        }
      }
    };
    // A(I, J) += Gn_I C_J' for the owned row tiles at positions [rx0, rx1) and the column positions [y0, y1) staged in Cs
    auto trailing = [&](int rx0, int rx1, int y0, int y1) {
#pragma unroll
      for (int sl = 0; sl < MI_TAIL_MAXSLOT; sl++) {
        const int x = slot_pos(sl);
        if (x < rx0 || x >= rx1) continue;
        const int I = nonp(x), yend = x < y1 - 1 ? x : y1 - 1;           // columns y0 .. yend (y <= x)
        if (yend < y0) continue;
        auto upd = [&](mi_v4d c, int y) {
          const double *cs = Cs + (size_t)(y - y0) * 1024 + lane;
#pragma unroll
          for (int s = 0; s < 16; s++) c = mfma_f64(Gn[sl][s], cs[s * 64], c);
          return c;
        };
        auto tix = [&](int y) { return tile_ix(I, nonp(y)); };
        // PFT accumulator tiles in flight (an HBM round trip under load is several tile updates long).  Straight-line
        // ring: every trip updates PFT tiles; positions past yend work on the wave's dummy tile behind the triangle
        // (loaded, updated with the last column's operand, stored - never read by anyone), so that no load or store of
        // the loop sits behind a branch and the wait counts stay exact.
        const uint32_t dummy = (uint32_t)(nt * (nt + 1) / 2) + (uint32_t)wave;
        auto tix_or_dummy = [&](int y) { return y <= yend ? tix(y) : dummy; };
        mi_v4d c[PFT];
#pragma unroll
        for (int u = 0; u < PFT; u++) c[u] = ld_tile(A, tix_or_dummy(y0 + u), lane);
        for (int y = y0; y <= yend; y += PFT) {
#pragma unroll
          for (int u = 0; u < PFT; u++) {
            c[u] = upd(c[u], y + u <= yend ? y + u : yend);
            st_tile(A, tix_or_dummy(y + u), lane, c[u]);
            c[u] = ld_tile(A, tix_or_dummy(y + u + PFT), lane);
          }
        }
      }
    };
    // steps of a pass: (stage [sx0, sx1)) -> (Gn of the owned rows staged there, when gflag) -> (trailing update of the
    // owned rows [rx0, rx1) against the staged columns).  One staging when the whole panel fits; else the halves
    // [0, h0) and [h0, nrt), h0 <= nh:  H1 (Gn only) -> H0 (Gn; every row against the columns of H0) -> H1 (its rows against itself)
    const int h0 = nrt <= nh ? 0 : nrt - nh, nsteps = nrt <= nh ? 1 : 3;
    for (int stp = 0; stp < nsteps; stp++) {
      int sx0, sx1, rx0, rx1, gflag;
      if (nsteps == 1) { sx0 = 0; sx1 = nrt; rx0 = 0; rx1 = nrt; gflag = 1; }
      else if (stp == 0) { sx0 = h0; sx1 = nrt; rx0 = 0; rx1 = 0; gflag = 1; }
      else if (stp == 1) { sx0 = 0; sx1 = h0; rx0 = 0; rx1 = nrt; gflag = 1; }
      else { sx0 = h0; sx1 = nrt; rx0 = h0; rx1 = nrt; gflag = 0; }
      if (stp) __syncthreads();         // everybody is done with the previous staging
      stamp(7);
      stage(sx0, sx1);
      stamp(4);
      __syncthreads();
      stamp(7);
      if (gflag) compute_g(sx0, sx1);
      stamp(5);
      trailing(rx0, rx1, sx0, sx1);
      stamp(6);
    }
    // ---- the panel takes G = -Gn (nobody reads the old panel from the scratch any more: every staging is behind us)
#pragma unroll
    for (int sl = 0; sl < MI_TAIL_MAXSLOT; sl++) {
      const int x = slot_pos(sl);
      if (x < 0) continue;
      const int t = nonp(x);
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        if (t < P0) {
          st_tile(A, tile_ix(P0 + kt, t), lane, mi_v4d{-Gn[sl][4 * kt], -Gn[sl][4 * kt + 1], -Gn[sl][4 * kt + 2], -Gn[sl][4 * kt + 3]});
        } else {                      // element (row i = lane % 16, col kk = lane / 16 + 4 r) of tile (t, P0 + kt)
          double *dst = A + (size_t)tile_ix(t, P0 + kt) * 256;
#pragma unroll
          for (int r = 0; r < 4; r++) dst[(((l15 & 3) << 4) + l4 + 4 * r) * 4 + (l15 >> 2)] = -Gn[sl][4 * kt + r];
        }
      }
    }
    stamp(2);
    __syncthreads();
    stamp(7);
  }
  // ---- M = -A into the QP's stream of the symmetric product, its diagonal into dinv
  {
    // The stream is the sequence of the product's tasks: per task one 64 x 64 block of M in "circulant" order (step q, lane
    // l: M[I0 + (l + s0 + q) % 64, J0 + l]).  The block's 16 tiles are read coalesced into an LDS image (two images: the
    // tiles of the next block load while this one is written) and written out step by step as contiguous 512-byte rows -
    // a transposition through LDS instead of 100 k scattered 8-byte reads of the scratch (0.27 M clocks per QP).
    double *dv = a.dt_val + (size_t)slot * a.n_slots;
    for (uint32_t t = 0; t < a.n_tasks; t++) {
      double *Tb = smem + (size_t)(t & 1u) * (64 * MI_TAIL_TS);
      const uint32_t I0 = a.dt_task[4 * t], J0 = a.dt_task[4 * t + 1], fl = a.dt_task[4 * t + 2] & 255u, nsteps = a.dt_task[4 * t + 3];
      const bool diag = fl & 1u;
      for (int idx = wave; idx < 16; idx += MI_TAIL_NW) {
        const int ta = idx >> 2, tb = idx & 3;
        if (diag && ta < tb) continue;
        const mi_v4d v = ld_tile(A, tile_ix((int)I0 / 16 + ta, (int)J0 / 16 + tb), lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * ta + l4 + 4 * r, col = 16 * tb + l15;
          if (!diag) Tb[row * MI_TAIL_TS + col] = v[r];
          else if (ta != tb || row >= col) { Tb[row * MI_TAIL_TS + col] = v[r]; Tb[col * MI_TAIL_TS + row] = v[r]; }
        }
      }
      __syncthreads();
      const uint32_t s0 = diag ? 1u : 0u, step0 = a.dt_task_step[t];
      for (uint32_t q = wave; q < nsteps; q += MI_TAIL_NW) {
        const uint32_t sh = s0 + q, i = ((uint32_t)lane + sh) & 63u;
        double val = -Tb[i * MI_TAIL_TS + lane];
        if (diag && sh == 32u && lane >= 32) val = 0.0;          // the pairs at distance 32 appear twice in a diagonal block
        dv[(size_t)(step0 + q) * 64 + lane] = val;
      }
    }
    for (int i = tid; i < k; i += nthr) a.dinv[H(a.N, (size_t)a.s + i)] = -A[a.diag_tile[i]];
  }
  if (tid == 0) {
    const int total = a.npos[slot] + npos;
    a.npos[slot] = total;
    if (total != a.n) a.iscal[H(IS_COUNT, IS_NEED_REFACTOR)] = -1;
  }
  if (a.trace) {
    __syncthreads();
    stamp(3);
    if (tid == 0) for (int i = 0; i < 8; i++) a.trace[(size_t)g * 8 + i] = tr_acc[i];
  }
}
hipError_t launch_tail(const TailArgs &a, int nwork, size_t lds_asm, size_t lds, hipStream_t st) {
  if (a.k > 512 || (a.k & 63) || a.k < 64) return hipErrorInvalidValue;
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tail_assemble_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_asm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tail_assemble_kernel, dim3(nwork), dim3(1024), lds_asm, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(512), lds, st, a);
    return hipGetLastError();
  };
  const int slots = (a.k / 16 - 4 + MI_TAIL_NW - 1) / MI_TAIL_NW;
  if (slots <= 1) return go(&tail_kernel<1, 4>);
  if (slots == 2) return go(&tail_kernel<2, 4>);
  if (slots == 3) return go(&tail_kernel<3, 4>);
  return go(&tail_kernel<4, 4>);
}

// ------------------------------------------------- layout / upload kernels

// dst[tile][i][b] = src[q][i]   (q = ids ? ids[j] : j)
__global__ void interleave_kernel(const double *__restrict__ src, double *dst, const int *ids, int nq,
                                  int len, int BT) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nq * len) return;
  const int j = (int)(g / len), i = (int)(g % len);
  const int q = ids ? ids[j] : j;
  dst[((size_t)(q / BT) * len + i) * BT + (q % BT)] = src[g];
}
// dst[tile][phys(slot, b)] = map[slot] >= 0 ? src[q][map[slot]] : 0   (logical slots -> physical layout)
__global__ void scatter_kernel(const double *__restrict__ src, double *dst, const int *__restrict__ map,
                               const int *ids, int nq, int srclen, SchedDev sd, int BT) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t slots = sd.n_slots;
  if (g >= (size_t)nq * slots) return;
  const int j = (int)(g / slots);
  const uint32_t s = (uint32_t)(g % slots);
  const int q = ids ? ids[j] : j;
  dst[(size_t)q * slots + s] = slot_value(map[s], src + (size_t)j * srclen, 1, 0);      // one stream per QP: [q][slot]
  (void)BT;
}
// Compaction support: exchange the complete per-QP contents of slot pairs (slot =
// tile*BT + b).  Every array is [tile][len][BT].
__global__ void swap_plain_kernel(double *base, const int2 *pairs, int npairs, int len, int BT) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)npairs * len) return;
  const int2 pr = pairs[g / len];
  const int i = (int)(g % len);
  double *pa = base + ((size_t)(pr.x / BT) * len + i) * BT + pr.x % BT;
  double *pb = base + ((size_t)(pr.y / BT) * len + i) * BT + pr.y % BT;
  const double t = *pa; *pa = *pb; *pb = t;
}
__global__ void swap_int_kernel(int *base, const int2 *pairs, int npairs, int len, int BT) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)npairs * len) return;
  const int2 pr = pairs[g / len];
  const int i = (int)(g % len);
  int *pa = base + ((size_t)(pr.x / BT) * len + i) * BT + pr.x % BT;
  int *pb = base + ((size_t)(pr.y / BT) * len + i) * BT + pr.y % BT;
  const int t = *pa; *pa = *pb; *pb = t;
}
__global__ void swap_sched_kernel(double *base, const int2 *pairs, int npairs, SchedDev sd, int BT) {
  const size_t per = (size_t)sd.n_steps * 64;
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)npairs * per) return;
  const int2 pr = pairs[g / per];
  const size_t u = g % per;
  double *pa = base + (size_t)pr.x * per + u;        // one stream per slot
  double *pb = base + (size_t)pr.y * per + u;
  const double t = *pa; *pa = *pb; *pb = t;
  (void)BT;
}

// dst[q][i] = src[tile][i][b]
__global__ void deinterleave_kernel(const double *__restrict__ src, double *dst, int nq, int len, int BT) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)nq * len) return;
  const int q = (int)(g / len), i = (int)(g % len);
  dst[g] = src[((size_t)(q / BT) * len + i) * BT + (q % BT)];
}
__global__ void gather_status_kernel(const int *__restrict__ iscal, int32_t *status, int32_t *iters, int B, int BT) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= B) return;
  const int *t = iscal + (size_t)(q / BT) * IS_COUNT * BT;
  if (status) status[q] = t[IS_STATUS * BT + q % BT];
  if (iters) iters[q] = t[IS_ITER * BT + q % BT];
}
// Per-QP failure isolation (the KKT factor of a QP lost its inertia): the listed slots become kNonConvex - done, status -7,
// NaN solution, objective NaN, cold-started iterates ([EXT] store_solution of a status without solution) - and the batch
// goes on without them.  One workgroup per failed slot.
__global__ void fail_slots_kernel(KernelArgs a, const int *__restrict__ slots, int BT, int iter) {
  const int slot = slots[blockIdx.x], tid = threadIdx.x, nthr = blockDim.x;
  if (slot < 0) return;
  const size_t tile = (size_t)slot / BT, b = (size_t)slot % BT;
  const int qp = a.qp_of_slot ? a.qp_of_slot[slot] : slot;
  const double nanv = __builtin_nan("");
  for (int i = tid; i < a.n; i += nthr) {
    a.x[(tile * a.n + i) * BT + b] = 0.0;
    if (qp >= 0 && qp < a.B) a.x_out[(size_t)qp * a.n + i] = nanv;
  }
  for (int j = tid; j < a.m; j += nthr) {
    a.z[(tile * a.m + j) * BT + b] = 0.0; a.y[(tile * a.m + j) * BT + b] = 0.0;
    if (qp >= 0 && qp < a.B) a.y_out[(size_t)qp * a.m + j] = nanv;
  }
  if (tid == 0) {
    int *is = a.iscal + tile * IS_COUNT * BT;
    is[IS_DONE * BT + b] = 1; is[IS_STATUS * BT + b] = -7; is[IS_NEED_REFACTOR * BT + b] = 0; is[IS_ITER * BT + b] = iter;
    double *ds = a.dscal + tile * DS_COUNT * BT;
    ds[DS_OBJ * BT + b] = nanv; ds[DS_PRI_RES * BT + b] = nanv; ds[DS_DUA_RES * BT + b] = nanv;
  }
}
hipError_t launch_fail_slots(const KernelArgs &a, const int *slots, int nfail, int BT, int iter, hipStream_t st) {
  if (!nfail) return hipSuccess;
  hipLaunchKernelGGL(fail_slots_kernel, dim3(nfail), dim3(256), 0, st, a, slots, BT, iter);
  return hipGetLastError();
}
// ---- per-QP entry points (continuous batching, solver.hip): one workgroup per listed slot (slot = tile * BT + b) --------
// What fail_slots_kernel does to a slot, for callers that already sit in a workgroup of that slot.
__device__ __forceinline__ void fail_one_slot(const KernelArgs &a, int slot, int BT, int iter, int tid, int nthr) {
  const size_t tile = (size_t)slot / BT, b = (size_t)slot % BT;
  const double nanv = __builtin_nan("");
  for (int i = tid; i < a.n; i += nthr) { a.x[(tile * a.n + i) * BT + b] = 0.0; if (slot < a.B) a.x_out[(size_t)slot * a.n + i] = nanv; }
  for (int j = tid; j < a.m; j += nthr) {
    a.z[(tile * a.m + j) * BT + b] = 0.0; a.y[(tile * a.m + j) * BT + b] = 0.0;
    if (slot < a.B) a.y_out[(size_t)slot * a.m + j] = nanv;
  }
  if (tid == 0) {
    int *is = a.iscal + tile * IS_COUNT * BT;
    is[IS_DONE * BT + b] = 1; is[IS_STATUS * BT + b] = -7; is[IS_ITER * BT + b] = iter;
    double *ds = a.dscal + tile * DS_COUNT * BT;
    ds[DS_OBJ * BT + b] = nanv; ds[DS_PRI_RES * BT + b] = nanv; ds[DS_DUA_RES * BT + b] = nanv;
  }
}
// Solve() entry of the listed QPs ([EXT] osqp_solve: status unsolved, iteration count 0; warm_start off: cold iterates).
// A QP without a valid factor (the last refactorisation of its KKT matrix lost the inertia) ends at once as kNonConvex.
__global__ void start_slots_kernel(KernelArgs a, const int *__restrict__ slots, const int *__restrict__ clear, int BT, int cold) {
  const int slot = slots[blockIdx.x], tid = threadIdx.x, nthr = blockDim.x;
  if (slot < 0) return;
  const size_t tile = (size_t)slot / BT, b = (size_t)slot % BT;
  int *is = a.iscal + tile * IS_COUNT * BT;
  if (is[IS_NEED_REFACTOR * BT + b] < 0) {
    fail_one_slot(a, slot, BT, 0, tid, nthr);
    __syncthreads();
    if (tid == 0) { is[IS_PENDING * BT + b] = 0; __threadfence_system(); is[IS_EPOCH * BT + b] += 1; }      // (the epoch last: it validates the rest)
    return;
  }
  if (cold) {
    for (int i = tid; i < a.n; i += nthr) a.x[(tile * a.n + i) * BT + b] = 0.0;
    for (int j = tid; j < a.m; j += nthr) { a.z[(tile * a.m + j) * BT + b] = 0.0; a.y[(tile * a.m + j) * BT + b] = 0.0; }
  }
  __syncthreads();
  if (tid == 0) {
    // the slot stays idle (IS_DONE) until an advance launch activates it: this kernel may run next to an advance launch
    // that must not pick the QP up half-prepared.  Everything else first, the pending mark last.
    // (An advance launch that publishes this slot's flags in between must not make the host take the slot's previous,
    //  finished solve for the new one: the pending mark is up before the epoch moves.)
    is[IS_STATUS * BT + b] = -10; is[IS_ITER * BT + b] = 0; is[IS_NEED_REFACTOR * BT + b] = 0;
    is[IS_CUR * BT + b] = 0;
    if (clear[blockIdx.x]) is[IS_RHO_UPDATES * BT + b] = 0;
    __threadfence();
    __hip_atomic_store(&is[IS_PENDING * BT + b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    is[IS_EPOCH * BT + b] += 1;
  }
}
hipError_t launch_start_slots(const KernelArgs &a, const int *slots, const int *clear, int nslots, int BT, int cold, hipStream_t st) {
  if (!nslots) return hipSuccess;
  hipLaunchKernelGGL(start_slots_kernel, dim3(nslots), dim3(256), 0, st, a, slots, clear, BT, cold);
  return hipGetLastError();
}
// The iterates and scalars a fresh setup leaves (batch_setup_impl: zero x, y, z; rho = rho estimate = settings.rho; no rho
// updates; residuals / objective 0), and the slot idle until its solve is begun.  c / 1/c belong to the equilibration.
__global__ void fresh_slots_kernel(KernelArgs a, const int *__restrict__ slots, int BT, double rho0) {
  const int slot = slots[blockIdx.x], tid = threadIdx.x, nthr = blockDim.x;
  if (slot < 0) return;
  const size_t tile = (size_t)slot / BT, b = (size_t)slot % BT;
  for (int i = tid; i < a.n; i += nthr) { a.x[(tile * a.n + i) * BT + b] = 0.0; a.dx[(tile * a.n + i) * BT + b] = 0.0; }
  for (int j = tid; j < a.m; j += nthr) { a.z[(tile * a.m + j) * BT + b] = 0.0; a.y[(tile * a.m + j) * BT + b] = 0.0; a.dy[(tile * a.m + j) * BT + b] = 0.0; }
  if (tid == 0) {
    int *is = a.iscal + tile * IS_COUNT * BT;
    is[IS_STATUS * BT + b] = -10; is[IS_ITER * BT + b] = 0; is[IS_RHO_UPDATES * BT + b] = 0; is[IS_DONE * BT + b] = 1;
    is[IS_NEED_REFACTOR * BT + b] = 0; is[IS_CUR * BT + b] = 0; is[IS_PENDING * BT + b] = 0;
    double *ds = a.dscal + tile * DS_COUNT * BT;
    ds[DS_RHO * BT + b] = rho0; ds[DS_RHO_EST * BT + b] = rho0; ds[DS_PRI_RES * BT + b] = 0.0; ds[DS_DUA_RES * BT + b] = 0.0; ds[DS_OBJ * BT + b] = 0.0;
  }
}
hipError_t launch_fresh_slots(const KernelArgs &a, const int *slots, int nslots, int BT, double rho0, hipStream_t st) {
  if (!nslots) return hipSuccess;
  hipLaunchKernelGGL(fresh_slots_kernel, dim3(nslots), dim3(256), 0, st, a, slots, BT, rho0);
  return hipGetLastError();
}
// The refactorisation work list of a launch that nobody on the host has looked at: the slots whose rho changed in the
// check that has just run (flag 1), packed in front, -1 behind them.  One workgroup.
__global__ __launch_bounds__(1024) void worklist_kernel(const int *__restrict__ iscal, int *work, int nslots, int BT) {
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  for (int s = threadIdx.x; s < nslots; s += blockDim.x) work[s] = -1;
  __syncthreads();
  for (int s = threadIdx.x; s < nslots; s += blockDim.x)
    if (iscal[(size_t)(s / BT) * IS_COUNT * BT + IS_NEED_REFACTOR * BT + s % BT] == 1) work[atomicAdd(&s_cnt, 1)] = s;
  // (flag 1: a QP paused by advance_kernel - or one that ran into max_iter on a rho-update iteration: finished AND to be
  //  refactored, because the next solve of a warm-started solver continues from that factor)
}
hipError_t launch_worklist(const int *iscal, int *work, int nslots, int BT, hipStream_t st) {
  hipLaunchKernelGGL(worklist_kernel, dim3(1), dim3(1024), 0, st, iscal, work, nslots, BT);
  return hipGetLastError();
}
// After the refactorisations of the paused slots: flag 0 - the solve goes on with the next advance launch (pending mark 1);
// flag -1 - the new factor lost the inertia: the QP ends as kNonConvex ([EXT] osqp_solve: adapt_rho fails -> OSQP_NON_CVX)
// and keeps the flag until a refactorisation of it succeeds.
__global__ void resume_flagged_kernel(KernelArgs a, int BT) {
  const int slot = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const size_t tile = (size_t)slot / BT, b = (size_t)slot % BT;
  int *is = a.iscal + tile * IS_COUNT * BT;
  if (is[IS_PENDING * BT + b] != 2) return;
  const int flag = is[IS_NEED_REFACTOR * BT + b];
  if (flag == 1) return;                                 // (its refactorisation has not run yet)
  if (flag < 0) fail_one_slot(a, slot, BT, is[IS_CUR * BT + b], tid, nthr);
  __syncthreads();
  __threadfence();
  if (tid == 0) __hip_atomic_store(&is[IS_PENDING * BT + b], flag < 0 ? 0 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t launch_resume_flagged(const KernelArgs &a, int nslots, int BT, hipStream_t st) {
  if (!nslots) return hipSuccess;
  hipLaunchKernelGGL(resume_flagged_kernel, dim3(nslots), dim3(64), 0, st, a, BT);
  return hipGetLastError();
}
__global__ void copy_slot_streams_kernel(double *dst, const double *__restrict__ src, const int *__restrict__ slots, size_t per) {
  const int slot = slots[blockIdx.y];
  if (slot < 0) return;
  const size_t off = (size_t)slot * per;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) dst[off + e] = src[off + e];
}
__global__ void copy_flagged_streams_kernel(double *dst, const double *__restrict__ src, const int *__restrict__ flags, int want, size_t per) {
  const int slot = blockIdx.y;
  if (flags && (flags[slot] != 0) != (want != 0)) return;
  const size_t off = (size_t)slot * per;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) dst[off + e] = src[off + e];
}
hipError_t launch_copy_flagged_streams(double *dst, const double *src, const int *flags, int want, int nslots, size_t per, hipStream_t st) {
  if (!nslots || !per) return hipSuccess;
  const unsigned gx = (unsigned)std::min<size_t>(64, (per + 1023) / 1024);
  hipLaunchKernelGGL(copy_flagged_streams_kernel, dim3(gx, nslots), dim3(256), 0, st, dst, src, flags, want, per);
  return hipGetLastError();
}
hipError_t launch_copy_slot_streams(double *dst, const double *src, const int *slots, int nslots, size_t per, hipStream_t st) {
  if (!nslots || !per) return hipSuccess;
  const unsigned gx = (unsigned)std::min<size_t>(64, (per + 255) / 256);
  hipLaunchKernelGGL(copy_slot_streams_kernel, dim3(gx, nslots), dim3(256), 0, st, dst, src, slots, per);
  return hipGetLastError();
}
__global__ void copy_slot_rows_kernel(double *dst, const double *__restrict__ src, const int *__restrict__ slots, int len, int BT) {
  const int slot = slots[blockIdx.y];
  if (slot < 0) return;
  const size_t base = (size_t)(slot / BT) * len * BT + slot % BT;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) dst[base + (size_t)i * BT] = src[base + (size_t)i * BT];
}
hipError_t launch_copy_slot_rows(double *dst, const double *src, const int *slots, int nslots, int len, int BT, hipStream_t st) {
  if (!nslots || !len) return hipSuccess;
  const unsigned gx = (unsigned)std::min(16, (len + 255) / 256);
  hipLaunchKernelGGL(copy_slot_rows_kernel, dim3(gx, nslots), dim3(256), 0, st, dst, src, slots, len, BT);
  return hipGetLastError();
}
// rows that arrived list-major (row j of src belongs to QP ids[j]) kept QP-major: dst[ids[j]][:] = src[j][:]
__global__ void keep_rows_kernel(double *__restrict__ dst, const double *__restrict__ src, const int *__restrict__ ids, int len) {
  const int j = blockIdx.y;
  const size_t to = (size_t)ids[j] * len, from = (size_t)j * len;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) dst[to + i] = src[from + i];
}
hipError_t launch_keep_rows(double *dst, const double *src, const int *ids, int n_ids, int len, hipStream_t st) {
  if (!n_ids || !len) return hipSuccess;
  const unsigned gx = (unsigned)std::min(16, (len + 255) / 256);
  hipLaunchKernelGGL(keep_rows_kernel, dim3(gx, n_ids), dim3(256), 0, st, dst, src, ids, len);
  return hipGetLastError();
}
// bounds update on device: l,u <- E .* clip(l,u); flags a constraint-type change
__global__ void bounds_kernel(const double *__restrict__ gl, const double *__restrict__ gu, double *l, double *u,
                              const double *__restrict__ Esc, const double *__restrict__ rho_vec,
                              const double *__restrict__ dscal, int *changed, int B, int m, int BT, int scaling) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)B * m) return;
  const int q = (int)(g / m), i = (int)(g % m);
  const size_t e = ((size_t)(q / BT) * m + i) * BT + (q % BT);
  double lo = fmax(gl[g], -MI_INFTY), up = fmin(gu[g], MI_INFTY);
  if (lo > up) { atomicOr(changed, 2); return; }
  if (scaling) { lo *= Esc[e]; up *= Esc[e]; }
  l[e] = lo; u[e] = up;
  const double rho = dscal[((size_t)(q / BT) * DS_COUNT + DS_RHO) * BT + (q % BT)];
  double want;
  if (lo < -MI_INFTY * MI_MIN_SCALING && up > MI_INFTY * MI_MIN_SCALING) want = MI_RHO_MIN;
  else if (up - lo < 1e-4) want = 1e3 * rho;
  else want = rho;
  if (want != rho_vec[e]) atomicOr(changed, 1);
}

// ---- row E2 on the device: Ruiz equilibration of [[P, A'],[A, 0]] + cost normalisation after new A values ------------
// One workgroup per QP.  Same operations in the same order as host_core.cpp unscale_qp / scale_qp (and the oracle):
// infinity norms are maxima (exact in any order: atomic max on the bit patterns of non-negative doubles), every product is a
// separate multiplication, square roots and reciprocals are the correctly rounded ones, and the one sum (the mean column
// norm of P) is added up by one thread in index order - so the result equals the host's bit for bit.
__device__ __forceinline__ double ruiz_limit(double v) { v = v < 1e-4 ? 1.0 : v; return v > 1e4 ? 1e4 : v; }
__device__ __forceinline__ void ruiz_amax(double *p, double a) {
  atomicMax(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(a));
}
__global__ __launch_bounds__(512) void ruiz_kernel(RuizArgs a) {
  __shared__ double s_red[16];
  __shared__ double s_stage[2048];
  const int jq = blockIdx.x, qp = a.ids ? a.ids[jq] : jq;      // jq: position in the caller's list (= row of its QP-major arguments)
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int n = a.n, m = a.m, nnzP = a.nnzP, nnzA = a.nnzA, pa_len = nnzP + nnzA;
  const size_t tile = (size_t)(qp / a.BT), b = (size_t)(qp % a.BT), BT = (size_t)a.BT;
  auto H = [&](size_t len, size_t i) { return (tile * len + i) * BT + b; };
  double *dn = a.dn + (size_t)qp * n, *en = a.en + (size_t)qp * m;
  const int jr = a.raw_by_qp ? qp : jq;          // row of the raw A / bounds arguments
  const double *rawA = a.rawA + (size_t)jr * nnzA;
  // (dn / en are updated by atomics, which execute in L2: they are read and reset past the CU's L1 as well)
  auto ld = [](const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st0 = [](double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  double c = a.dscal[H(DS_COUNT, DS_C)];
  const double cinv0 = a.dscal[H(DS_COUNT, DS_CINV)];
  // ---- unscale P and q with the scaling in force; A is replaced; the bounds are replaced or unscaled
  if (a.fresh) {       // the QP as setup sees it: P and q as given then
    for (int k = tid; k < nnzP; k += nthr) a.pa_val[H(pa_len, k)] = a.rawP[(size_t)qp * nnzP + k];
    for (int j = tid; j < n; j += nthr) a.q[H(n, j)] = a.rawq[(size_t)qp * n + j];
  } else {
    for (int k = tid; k < nnzP; k += nthr) {
      double v = a.pa_val[H(pa_len, k)];
      v *= cinv0; v *= a.Dsc_inv[H(n, a.Prow[k])]; v *= a.Dsc_inv[H(n, a.Pcol[k])];
      a.pa_val[H(pa_len, k)] = v;
    }
    for (int j = tid; j < n; j += nthr) a.q[H(n, j)] *= cinv0 * a.Dsc_inv[H(n, j)];
  }
  for (int k = tid; k < nnzA; k += nthr) a.pa_val[H(pa_len, nnzP + k)] = rawA[k];
  for (int i = tid; i < m; i += nthr) {
    double lo, up;
    if (a.rawl) { lo = fmax(a.rawl[(size_t)jr * m + i], -MI_INFTY); up = fmin(a.rawu[(size_t)jr * m + i], MI_INFTY); }
    else { const double ei = a.Esc_inv[H(m, i)]; lo = a.l[H(m, i)] * ei; up = a.u[H(m, i)] * ei; }
    a.l[H(m, i)] = lo; a.u[H(m, i)] = up;
  }
  __syncthreads();
  for (int j = tid; j < n; j += nthr) a.Dsc[H(n, j)] = 1.0;
  for (int i = tid; i < m; i += nthr) a.Esc[H(m, i)] = 1.0;
  c = 1.0;
  __syncthreads();
  auto p_norms = [&]() {                  // dn = column infinity norms of the symmetric P (upper triangle stored)
    for (int j = tid; j < n; j += nthr) st0(&dn[j], 0.0);
    __syncthreads();
    for (int k = tid; k < nnzP; k += nthr) {
      const double v = fabs(a.pa_val[H(pa_len, k)]);
      const int i = a.Prow[k], j = a.Pcol[k];
      ruiz_amax(&dn[j], v);
      if (i != j) ruiz_amax(&dn[i], v);
    }
  };
  for (int it = 0; it < a.iters; it++) {
    for (int i = tid; i < m; i += nthr) st0(&en[i], 0.0);
    p_norms();
    for (int k = tid; k < nnzA; k += nthr) {
      const double v = fabs(a.pa_val[H(pa_len, nnzP + k)]);
      ruiz_amax(&dn[a.Acol[k]], v);
      ruiz_amax(&en[a.Arow[k]], v);
    }
    __syncthreads();
    for (int j = tid; j < n; j += nthr) st0(&dn[j], 1.0 / sqrt(ruiz_limit(ld(&dn[j]))));
    for (int i = tid; i < m; i += nthr) st0(&en[i], 1.0 / sqrt(ruiz_limit(ld(&en[i]))));
    __syncthreads();
    for (int k = tid; k < nnzP; k += nthr) { double v = a.pa_val[H(pa_len, k)]; v *= ld(&dn[a.Prow[k]]); v *= ld(&dn[a.Pcol[k]]); a.pa_val[H(pa_len, k)] = v; }
    for (int k = tid; k < nnzA; k += nthr) { double v = a.pa_val[H(pa_len, nnzP + k)]; v *= ld(&en[a.Arow[k]]); v *= ld(&dn[a.Acol[k]]); a.pa_val[H(pa_len, nnzP + k)] = v; }
    for (int j = tid; j < n; j += nthr) { const double d = ld(&dn[j]); a.q[H(n, j)] *= d; a.Dsc[H(n, j)] *= d; }
    for (int i = tid; i < m; i += nthr) a.Esc[H(m, i)] *= ld(&en[i]);
    __syncthreads();
    // cost normalisation: c = 1 / max(mean column norm of P, |q|_inf), both limited
    p_norms();
    double nq = 0.0;
    for (int j = tid; j < n; j += nthr) nq = fmax(nq, fabs(a.q[H(n, j)]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nq = fmax(nq, shfl_xor_d(nq, off));
    if (lane == 0) s_red[wave] = nq;
    __syncthreads();                       // (also: the norms of P are complete)
    // the mean column norm: added up in index order like the host does, by one thread, from LDS (staged by everybody)
    double mean = 0.0;
    for (int base = 0; base < n; base += 2048) {
      const int cnt = min(2048, n - base);
      for (int j = tid; j < cnt; j += nthr) s_stage[j] = ld(&dn[base + j]);
      __syncthreads();
      if (tid == 0) for (int j = 0; j < cnt; j++) mean += s_stage[j];
      __syncthreads();
    }
    if (tid == 0) {
      mean /= (double)n;
      double q1 = s_red[0];
      for (int w = 1; w < nw; w++) q1 = fmax(q1, s_red[w]);
      q1 = ruiz_limit(q1);
      double ct = ruiz_limit(fmax(mean, q1));
      s_red[15] = 1.0 / ct;
    }
    __syncthreads();
    const double ct = s_red[15];
    for (int k = tid; k < nnzP; k += nthr) a.pa_val[H(pa_len, k)] *= ct;
    for (int j = tid; j < n; j += nthr) a.q[H(n, j)] *= ct;
    c *= ct;
    __syncthreads();
  }
  for (int j = tid; j < n; j += nthr) a.Dsc_inv[H(n, j)] = 1.0 / a.Dsc[H(n, j)];
  for (int i = tid; i < m; i += nthr) {
    const double e = a.Esc[H(m, i)];
    a.Esc_inv[H(m, i)] = 1.0 / e;
    a.l[H(m, i)] *= e; a.u[H(m, i)] *= e;
  }
  if (tid == 0) { a.dscal[H(DS_COUNT, DS_C)] = c; a.dscal[H(DS_COUNT, DS_CINV)] = 1.0 / c; }
  for (int k = tid; k < pa_len; k += nthr) a.pa_out[(size_t)jq * pa_len + k] = a.pa_val[H(pa_len, k)];
}
// The same computation for QPs of up to 16 k entries and n + m <= 16 k (the GOMP QPs of BASELINE.md's batch configs, the
// random QPs of the headline): each thread keeps its share of the values of P and A - and their two norm slots, packed - in
// registers through all iterations, the norm vectors live in LDS (ds_max_u64 instead of L2 atomics); global memory is touched
// for the small vectors (q, D, E, bounds) only.  Operation for operation the kernel above (same bits).
// An entry of triu(P) at (r, c) has the slots (r, c) of dn, an entry of A at (r, c) the slots (n + r, c) of [dn ; en].
template <int VPT>
__global__ __launch_bounds__(512) void ruiz_reg_kernel(RuizArgs a) {
  extern __shared__ double rz_lds[];               // nv = [dn[n] ; en[m]]
  __shared__ double s_red[16];
  const int jq = blockIdx.x, qp = a.ids ? a.ids[jq] : jq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NT = 512, NW = NT / 64;
  const int n = a.n, m = a.m, nnzP = a.nnzP, nnzA = a.nnzA, pa_len = nnzP + nnzA;
  const size_t tile = (size_t)(qp / a.BT), b = (size_t)(qp % a.BT), BT = (size_t)a.BT;
  auto H = [&](size_t len, size_t i) { return (tile * len + i) * BT + b; };
  double *nv = rz_lds;
  const int jr = a.raw_by_qp ? qp : jq;
  const double *rawA = a.rawA + (size_t)jr * nnzA;
  auto amax = [](double *p, double v) { atomicMax(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v)); };
  double c = a.dscal[H(DS_COUNT, DS_C)];
  const double cinv0 = a.dscal[H(DS_COUNT, DS_CINV)];
  double v[VPT];
  unsigned slot[VPT];                              // row slot << 16 | column slot
#pragma unroll
  for (int i = 0; i < VPT; i++) {
    const int k = tid + i * NT;
    double x = 0.0;
    unsigned sl = 0;
    if (k < nnzP) {
      const int r = a.Prow[k], cc = a.Pcol[k];
      sl = ((unsigned)r << 16) | (unsigned)cc;
      if (a.fresh) x = a.rawP[(size_t)qp * nnzP + k];
      else { x = a.pa_val[H(pa_len, k)]; x *= cinv0; x *= a.Dsc_inv[H(n, r)]; x *= a.Dsc_inv[H(n, cc)]; }
    } else if (k < pa_len) {
      sl = ((unsigned)(n + a.Arow[k - nnzP]) << 16) | (unsigned)a.Acol[k - nnzP];
      x = rawA[k - nnzP];
    }
    v[i] = x; slot[i] = sl;
  }
  if (a.fresh) { for (int j = tid; j < n; j += NT) a.q[H(n, j)] = a.rawq[(size_t)qp * n + j]; }
  else { for (int j = tid; j < n; j += NT) a.q[H(n, j)] *= cinv0 * a.Dsc_inv[H(n, j)]; }
  for (int i = tid; i < m; i += NT) {
    double lo, up;
    if (a.rawl) { lo = fmax(a.rawl[(size_t)jr * m + i], -MI_INFTY); up = fmin(a.rawu[(size_t)jr * m + i], MI_INFTY); }
    else { const double ei = a.Esc_inv[H(m, i)]; lo = a.l[H(m, i)] * ei; up = a.u[H(m, i)] * ei; }
    a.l[H(m, i)] = lo; a.u[H(m, i)] = up;
  }
  __syncthreads();
  for (int j = tid; j < n; j += NT) a.Dsc[H(n, j)] = 1.0;
  for (int i = tid; i < m; i += NT) a.Esc[H(m, i)] = 1.0;
  c = 1.0;
  __syncthreads();
  // norms of the entries below `upto` (nnzP: P alone; pa_len: the whole KKT matrix) into the zeroed nv
  auto norms = [&](int upto) {
#pragma unroll
    for (int i = 0; i < VPT; i++) {
      const int k = tid + i * NT;
      if (k < upto) {
        const double x = fabs(v[i]);
        const unsigned s1 = slot[i] >> 16, s0 = slot[i] & 0xffffu;
        amax(&nv[s0], x);
        if (s1 != s0) amax(&nv[s1], x);
      }
    }
  };
  for (int it = 0; it < a.iters; it++) {
    for (int j = tid; j < n + m; j += NT) nv[j] = 0.0;
    __syncthreads();
    norms(pa_len);
    __syncthreads();
    for (int j = tid; j < n + m; j += NT) nv[j] = 1.0 / sqrt(ruiz_limit(nv[j]));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPT; i++) {
      const int k = tid + i * NT;
      if (k < pa_len) { double x = v[i]; x *= nv[slot[i] >> 16]; x *= nv[slot[i] & 0xffffu]; v[i] = x; }
    }
    for (int j = tid; j < n; j += NT) { const double d = nv[j]; a.q[H(n, j)] *= d; a.Dsc[H(n, j)] *= d; }
    for (int i = tid; i < m; i += NT) a.Esc[H(m, i)] *= nv[n + i];
    __syncthreads();
    for (int j = tid; j < n; j += NT) nv[j] = 0.0;
    __syncthreads();
    norms(nnzP);
    double nq = 0.0;
    for (int j = tid; j < n; j += NT) nq = fmax(nq, fabs(a.q[H(n, j)]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nq = fmax(nq, shfl_xor_d(nq, off));
    if (lane == 0) s_red[wave] = nq;
    __syncthreads();
    if (tid == 0) {
      double mean = 0.0;                     // added up in index order, like the host does
      for (int j = 0; j < n; j++) mean += nv[j];
      mean /= (double)n;
      double q1 = s_red[0];
      for (int w = 1; w < NW; w++) q1 = fmax(q1, s_red[w]);
      q1 = ruiz_limit(q1);
      const double ct = ruiz_limit(fmax(mean, q1));
      s_red[15] = 1.0 / ct;
    }
    __syncthreads();
    const double ct = s_red[15];
#pragma unroll
    for (int i = 0; i < VPT; i++) { const int k = tid + i * NT; if (k < nnzP) v[i] *= ct; }
    for (int j = tid; j < n; j += NT) a.q[H(n, j)] *= ct;
    c *= ct;
    __syncthreads();
  }
  for (int j = tid; j < n; j += NT) a.Dsc_inv[H(n, j)] = 1.0 / a.Dsc[H(n, j)];
  for (int i = tid; i < m; i += NT) {
    const double e = a.Esc[H(m, i)];
    a.Esc_inv[H(m, i)] = 1.0 / e;
    a.l[H(m, i)] *= e; a.u[H(m, i)] *= e;
  }
  if (tid == 0) { a.dscal[H(DS_COUNT, DS_C)] = c; a.dscal[H(DS_COUNT, DS_CINV)] = 1.0 / c; }
#pragma unroll
  for (int i = 0; i < VPT; i++) {
    const int k = tid + i * NT;
    if (k < pa_len) { a.pa_val[H(pa_len, k)] = v[i]; a.pa_out[(size_t)jq * pa_len + k] = v[i]; }
  }
}
template <int VPT>
static hipError_t launch_ruiz_reg(const RuizArgs &a, size_t lds, hipStream_t st) {
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&ruiz_reg_kernel<VPT>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ruiz_reg_kernel<VPT>, dim3(a.B), dim3(512), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_ruiz(const RuizArgs &a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  const long pa_len = (long)a.nnzP + a.nnzA;
  static const bool global_form = getenv("MI_OSQP_RUIZ_GLOBAL") != nullptr;      // (tests: the two forms give the same bits)
  if (!global_form && (long)a.n + a.m <= 16384 && pa_len <= 32 * 512) {
    const size_t lds = ((size_t)a.n + a.m) * sizeof(double);
    if (pa_len <= 4 * 512) return launch_ruiz_reg<4>(a, lds, st);
    if (pa_len <= 8 * 512) return launch_ruiz_reg<8>(a, lds, st);
    if (pa_len <= 16 * 512) return launch_ruiz_reg<16>(a, lds, st);
    return launch_ruiz_reg<32>(a, lds, st);
  }
  hipLaunchKernelGGL(ruiz_kernel, dim3(a.B), dim3(512), 0, st, a);
  return hipGetLastError();
}

// ---- GOMP re-linearisation on the device -------------------------------------------------------------------------------
// What ConstraintBuilder::withObstacles ([REF] src/constraints/constraint-builder.h:90-136) and GOMPSolver::isSolutionOK
// ([REF] src/gomp-solver.h:141-199) compute for one trajectory, for balls whose kinematics are built-in models: per ball and
// waypoint the position p = fk(q_w) and the 3 x D position Jacobian J(q_w); gripper balls get three rows J_axis q with
// bounds con - p_axis + J_axis q_w -+ radius, every (ball, line) pair one Z row - a bound when the ball collides with the
// line's vertical plane at that waypoint (HorizontalLine::hasCollision: close to it, or on opposite sides of it from a
// neighbouring waypoint), a dummy row otherwise - and the trajectory is accepted when every ball respects the box and is above
// (below) the lines it collides with.  One workgroup per trajectory, one thread per (ball, waypoint).
__device__ __forceinline__ void ur5e_point(const double *q, int frame, double *p, double *J /* 3 x 6 row-major */) {
  const double a[6] = {0.0, -0.425, -0.3922, 0.0, 0.0, 0.0}, d[6] = {0.1625, 0.0, 0.0, 0.1333, 0.0997, 0.0996};
  const double alpha[6] = {1.5707963267948966, 0.0, 0.0, 1.5707963267948966, -1.5707963267948966, 0.0};
  double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, o[3] = {0, 0, 0};
  double oj[7][3], zj[6][3];
  for (int i = 0; i < 6; i++) {
    for (int r = 0; r < 3; r++) { oj[i][r] = o[r]; zj[i][r] = R[r][2]; }
    const double ct = cos(q[i]), st = sin(q[i]), ca = cos(alpha[i]), sa = sin(alpha[i]);
    const double T[3][4] = {{ct, -st * ca, st * sa, a[i] * ct}, {st, ct * ca, -ct * sa, a[i] * st}, {0.0, sa, ca, d[i]}};
    double G[3][3], g[3];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) G[r][c] = R[r][0] * T[0][c] + R[r][1] * T[1][c] + R[r][2] * T[2][c];
      g[r] = R[r][0] * T[0][3] + R[r][1] * T[1][3] + R[r][2] * T[2][3] + o[r];
    }
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r][c] = G[r][c]; o[r] = g[r]; }
  }
  for (int r = 0; r < 3; r++) oj[6][r] = o[r];
  for (int r = 0; r < 3; r++) p[r] = oj[frame][r];
  for (int j = 0; j < 6; j++) {
    double col[3] = {0, 0, 0};
    if (j < frame) {
      const double r[3] = {p[0] - oj[j][0], p[1] - oj[j][1], p[2] - oj[j][2]};
      col[0] = zj[j][1] * r[2] - zj[j][2] * r[1];
      col[1] = zj[j][2] * r[0] - zj[j][0] * r[2];
      col[2] = zj[j][0] * r[1] - zj[j][1] * r[0];
    }
    for (int ax = 0; ax < 3; ax++) J[ax * 6 + j] = col[ax];
  }
}
#define MI_GOMP_MAXD 8
__device__ __forceinline__ void gomp_fk_jac(const GompBallDev &b, int D, const double *q, double *p, double *J) {
  for (int k = 0; k < 3 * D; k++) J[k] = 0.0;
  if (b.model == MI_GM_UR5E_FLANGE || b.model == MI_GM_UR5E_WRIST3 || b.model == MI_GM_UR5E_ELBOW) {
    double J6[18];
    ur5e_point(q, b.model == MI_GM_UR5E_FLANGE ? 6 : (b.model == MI_GM_UR5E_WRIST3 ? 5 : 2), p, J6);
    for (int ax = 0; ax < 3; ax++) for (int j = 0; j < 6 && j < D; j++) J[ax * D + j] = J6[ax * 6 + j];
  } else if (b.model == MI_GM_YAW_2LINK) {         // yaw q0, shoulder q1, elbow q2; links param[0], param[1], base height param[2]
    const double L1 = b.param[0], L2 = b.param[1], Z0 = b.param[2];
    const double c0 = cos(q[0]), s0 = sin(q[0]);
    const double r = L1 * cos(q[1]) + L2 * cos(q[1] + q[2]);
    const double dr1 = -L1 * sin(q[1]) - L2 * sin(q[1] + q[2]), dr2 = -L2 * sin(q[1] + q[2]);
    p[0] = r * c0; p[1] = r * s0; p[2] = Z0 + L1 * sin(q[1]) + L2 * sin(q[1] + q[2]);
    J[0 * D + 0] = -r * s0; J[0 * D + 1] = c0 * dr1; J[0 * D + 2] = c0 * dr2;
    J[1 * D + 0] = r * c0;  J[1 * D + 1] = s0 * dr1; J[1 * D + 2] = s0 * dr2;
    J[2 * D + 0] = 0.0;     J[2 * D + 1] = L1 * cos(q[1]) + L2 * cos(q[1] + q[2]); J[2 * D + 2] = L2 * cos(q[1] + q[2]);
  } else {                                         // MI_GM_TABLE (tests): p = (q0, q1, q2), J = the 3 x 3 table in param (D = 3)
    p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
    for (int ax = 0; ax < 3; ax++) for (int j = 0; j < 3 && j < D; j++) J[ax * D + j] = b.param[ax * 3 + j];
  }
}
__device__ __forceinline__ void gomp_dist_xy(const GompLineDev &ln, const double *P, double *dxy) {       // HorizontalLine::getDistanceVecXY
  double t = 0.0;
  for (int k = 0; k < 3; k++) t += (P[k] - ln.A[k]) * ln.D[k];
  dxy[0] = ln.A[0] + t * ln.D[0] - P[0]; dxy[1] = ln.A[1] + t * ln.D[1] - P[1];
}
__global__ __launch_bounds__(256) void gomp_relinearise_kernel(GompArgs g) {
  extern __shared__ double smem[];                 // xyz[n_balls][W][3]
  __shared__ int s_ok;
  const int jq = blockIdx.x, qp = g.ids[jq], tid = threadIdx.x;
  const int D = g.dims, W = g.W, NB = g.n_balls, NL = g.n_lines;
  const double *traj = g.traj + (size_t)jq * g.n;
  if (tid == 0) s_ok = 1;
  double p[3] = {0, 0, 0}, J[3 * MI_GOMP_MAXD], q[MI_GOMP_MAXD];
  // (one (ball, waypoint) pair per thread; more pairs than threads: the loop below repeats per chunk, with the positions of
  //  ALL pairs computed first - a waypoint's collision test looks at its neighbours)
  for (int e = tid; e < NB * W; e += blockDim.x) {
    const int w = e % W;
    for (int j = 0; j < D; j++) q[j] = traj[(size_t)w * D + j];
    gomp_fk_jac(g.balls[e / W], D, q, p, J);
    for (int k = 0; k < 3; k++) smem[(size_t)e * 3 + k] = p[k];
  }
  __syncthreads();
  int ok = 1;
  // pass 0: the acceptance test of the whole trajectory; pass 1: the rows (write_rows 1: always, 2: only when not accepted)
  for (int pass = 0; pass < 2; pass++) {
  bool wr = false;
  if (pass == 1) {
    if (!ok) atomicAnd(&s_ok, 0);
    __syncthreads();
    wr = g.write_rows == 1 || (g.write_rows == 2 && !s_ok);
    if (!wr) break;
  }
  for (int e = tid; e < NB * W; e += blockDim.x) {
    const int bi = e / W, w = e % W;
    const GompBallDev &ball = g.balls[bi];
    for (int j = 0; j < D; j++) q[j] = traj[(size_t)w * D + j];
    gomp_fk_jac(ball, D, q, p, J);
    const double *xyz = smem + (size_t)bi * W * 3;
    double Jq[3];
    for (int ax = 0; ax < 3; ax++) { double sacc = 0.0; for (int j = 0; j < D; j++) sacc += J[ax * D + j] * q[j]; Jq[ax] = sacc; }
    // this pair's first row: the balls before it (all their waypoints), then the waypoints before it of this ball
    int row = g.row0;
    for (int b2 = 0; b2 < bi; b2++) row += W * ((g.balls[b2].is_gripper ? 3 : 0) + NL);
    row += w * ((ball.is_gripper ? 3 : 0) + NL);
    auto put = [&](int r, int axis, double low, double upp) {
      if (!wr) return;
      const int *ai = g.aidx + (size_t)(r - g.row0) * D;
      for (int j = 0; j < D; j++) g.A[(size_t)qp * g.nnzA + ai[j]] = J[axis * D + j];
      g.l[(size_t)qp * g.m + r] = low + ball.radius;
      g.u[(size_t)qp * g.m + r] = upp - ball.radius;
    };
    if (ball.is_gripper) {
      for (int ax = 0; ax < 3; ax++) {
        const double lo = g.con_lo[ax] > -1e29 ? g.con_lo[ax] - p[ax] + Jq[ax] : -MI_INFTY;
        const double up = g.con_hi[ax] < 1e29 ? g.con_hi[ax] - p[ax] + Jq[ax] : MI_INFTY;
        put(row++, ax, lo, up);
        const double clo = g.con_lo[ax] > -1e29 ? g.con_lo[ax] : -MI_INFTY, cup = g.con_hi[ax] < 1e29 ? g.con_hi[ax] : MI_INFTY;
        if (!(clo - 1e-3 <= p[ax] - ball.radius && p[ax] + ball.radius <= cup + 1e-3)) ok = 0;
      }
    }
    for (int li = 0; li < NL; li++) {
      const GompLineDev &ln = g.lines[li];
      double dp[2], dn[2];
      gomp_dist_xy(ln, p, dp);
      bool coll = hypot(dp[0], dp[1]) < ball.radius;                                         // isClose
      if (!coll && w > 0) { gomp_dist_xy(ln, xyz + (size_t)(w - 1) * 3, dn); coll = dn[0] * dp[0] + dn[1] * dp[1] < 0; }
      if (!coll && w + 1 < W) { gomp_dist_xy(ln, xyz + (size_t)(w + 1) * 3, dn); coll = dp[0] * dn[0] + dp[1] * dn[1] < 0; }
      if (coll) {
        const double bound = (p[2] + (ln.A[2] - p[2])) - p[2] + Jq[2];                      // line[p][Z] - p[Z] + J q, in the host's order of operations
        if (ln.below) put(row++, 2, -MI_INFTY, bound); else put(row++, 2, bound, MI_INFTY);
        const bool above = ln.below ? (p[2] - ln.A[2]) <= -ball.radius + 1e-3 : (p[2] - ln.A[2]) >= ball.radius - 1e-3;
        if (!above) ok = 0;
      } else put(row++, 2, -MI_INFTY, MI_INFTY);                                            // dummy row: keeps the pattern constant
    }
  }
  }
  if (tid == 0) g.ok[jq] = s_ok;
}
hipError_t launch_gomp_relinearise(const GompArgs &g, hipStream_t st) {
  if (g.n_ids <= 0) return hipSuccess;
  if (g.dims > MI_GOMP_MAXD) return hipErrorInvalidValue;
  const size_t lds = (size_t)g.n_balls * g.W * 3 * sizeof(double);
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&gomp_relinearise_kernel), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gomp_relinearise_kernel, dim3(g.n_ids), dim3(256), lds, st, g);
  return hipGetLastError();
}

// --------------------------------------------------------------- launchers

// all iterate kernels are built for <= 512 threads per workgroup (256 VGPRs per lane
// hold the 16-step prefetch buffer); the host clamps `threads` accordingly
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size class instead of on every launch (it takes a
// runtime lock; a lone small QP launches two kernels per 25 iterations): the largest size set so far is remembered per
// kernel and device, a launch that needs more raises it.
static hipError_t ensure_dynamic_lds(const void *kern, size_t lds) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> have;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  size_t &cur = have[{kern, dev}];
  if (lds <= cur && cur != 0) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) cur = std::max<size_t>(lds, 1);
  return e;
}
#define MI_DISPATCH(KERNEL, ...)                                                                   \
  do {                                                                                             \
    hipError_t e_;                                                                                 \
    auto go = [&](auto kern) -> hipError_t {                                                       \
      e_ = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);                          \
      if (e_ != hipSuccess) return e_;                                                             \
      hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), lds, st, __VA_ARGS__);                  \
      return hipGetLastError();                                                                    \
    };                                                                                             \
    if (threads > 1024) return hipErrorInvalidValue;                                               \
    if (a.wide) {                 /* 32-bit index words: global vector, one QP per tile */         \
      if (!a.xs_global || BT != 1 || threads > 512) return hipErrorInvalidValue;                   \
      return go(&KERNEL<1, 512, true, true>);                                                      \
    }                                                                                              \
    if (threads > 512) {          /* 16 waves per tile: one workgroup per CU */                    \
      if (a.xs_global) return hipErrorInvalidValue;                                                \
      if (BT == 1) return go(&KERNEL<1, 1024, false>);                                             \
      if (BT == 2) return go(&KERNEL<2, 1024, false>);                                             \
      return hipErrorInvalidValue;                                                                 \
    }                                                                                              \
    if (a.xs_global) {                                                                             \
      if (BT == 1) return go(&KERNEL<1, 512, true>);                                               \
      if (BT == 2) return go(&KERNEL<2, 512, true>);                                               \
      return go(&KERNEL<4, 512, true>);                                                            \
    }                                                                                              \
    if (BT == 1) return go(&KERNEL<1, 512, false>);                                                \
    if (BT == 2) return go(&KERNEL<2, 512, false>);                                                \
    return go(&KERNEL<4, 512, false>);                                                             \
  } while (0)

// Workgroups of `threads` threads and `lds` bytes of dynamic LDS the device keeps resident at once for EVERY kernel that
// spins on the grid of a dataflow handle (iterate / check / kkt_solve, wide index words, one QP): the grid barrier and the
// "not yet" waits of those kernels only make progress when the whole grid is resident.  0: the query failed.
int max_coresident_groups(int threads, size_t lds, int n_cus) {
  int best = 1 << 30;
  auto q = [&](auto kern) {
    int nb = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { best = 0; return; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), threads, lds) != hipSuccess) { best = 0; return; }
    best = std::min(best, nb * n_cus);
  };
  q(&iterate_kernel<1, 512, true, true>);
  q(&check_kernel<1, 512, true, true>);
  q(&kkt_solve_kernel<1, 512, true, true>);
  (void)hipGetLastError();
  return best == (1 << 30) ? 0 : best;
}
// the same for the grouped refactorisation (factor_kernel<1> shared by G workgroups per QP)
int max_coresident_factor_groups(int threads, int n_cus) {
  int nb = 0;
  const size_t lds = factor_lds_bytes(1, threads);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&factor_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&factor_kernel<1>), threads, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return nb * n_cus;
}

hipError_t launch_advance(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st, int max_segments, int seg_len,
                          int *host_is, double *host_ds, unsigned *stop, unsigned seq, unsigned *counter, unsigned *host_done) {
  if (a.xs_global || a.wide || a.df || threads > 1024 || (threads > 512 && BT > 2)) return hipErrorInvalidValue;
  AdvanceArgs v{max_segments, seg_len, host_is, host_ds, stop, seq, counter, host_done};
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), lds, st, a, v);
    return hipGetLastError();
  };
  if (threads > 512) return BT == 1 ? go(&advance_kernel<1, 1024>) : go(&advance_kernel<2, 1024>);
  if (BT == 1) return go(&advance_kernel<1, 512>);
  if (BT == 2) return go(&advance_kernel<2, 512>);
  return go(&advance_kernel<4, 512>);
}

hipError_t launch_iterate(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st) {
  if (getenv("MI_OSQP_DEBUG_HIP")) fprintf(stderr, "[mi_osqp] launch_iterate: df %d wide %d xs_global %p BT %d tiles %d threads %d lds %zu groups %d bar %p\n", a.df, a.wide, (void *)a.xs_global, BT, tiles, threads, lds, a.mw_groups, (void *)a.mw_bar);
  if (a.df) { if (tiles != 1 || BT != 1 || !a.xs_global || !a.wide || !a.mw_bar || a.mw_groups < 1) return hipErrorInvalidValue; tiles = a.mw_groups; }
#ifdef MI_OSQP_DEBUG_BUILD
  if (a.df && debug_drop_group("iterate") && tiles > 1) tiles--;       // fault injection: a workgroup of the grid never shows up
#endif
  MI_DISPATCH(iterate_kernel, a);
}
hipError_t launch_check(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st) {
  if (a.df && a.mw_groups > 1) { if (tiles != 1 || BT != 1 || !a.xs_global || !a.wide || !a.mw_bar || !a.mw_scratch) return hipErrorInvalidValue; tiles = a.mw_groups; }
#ifdef MI_OSQP_DEBUG_BUILD
  if (a.df && a.mw_groups > 1 && debug_drop_group("check") && tiles > 1) tiles--;
#endif
  MI_DISPATCH(check_kernel, a);
}
hipError_t launch_spmv(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st,
                       const double *x, const double *y, double *Px, double *Aty, double *Ax) {
  MI_DISPATCH(spmv_kernel, a, x, y, Px, Aty, Ax);
}
hipError_t launch_kkt_solve(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st,
                            const double *rhs, double *sol) {
  if (a.df) { if (tiles != 1 || BT != 1 || !a.xs_global || !a.wide || !a.mw_bar || a.mw_groups < 1) return hipErrorInvalidValue; tiles = a.mw_groups; }
#ifdef MI_OSQP_DEBUG_BUILD
  if (a.df && debug_drop_group("kkt") && tiles > 1) tiles--;
#endif
  MI_DISPATCH(kkt_solve_kernel, a, rhs, sol);
}
hipError_t launch_kkt_trace(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st,
                            const double *rhs, double *sol, uint32_t *trace, uint32_t words) {
  if (BT != 2 || a.xs_global || threads != 512) return hipErrorInvalidValue;
  const size_t total = ((size_t)a.xs_len + 2 * (size_t)a.dt.k) * BT * sizeof(double) + (size_t)words * 4;
  if (total > 160 * 1024 - 512) return hipErrorInvalidValue;
  (void)lds;
#ifdef MI_OSQP_DEBUG_BUILD
  const bool waits = getenv("MI_OSQP_TRACE_WAITS") != nullptr;     // per-step ring-wait timing (slows every step down)
  auto kern = waits ? &kkt_trace_kernel<2> : &kkt_trace_kernel<1>;
#else
  auto kern = &kkt_trace_kernel<1>;
#endif
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), total, st, a, rhs, sol, trace, words);
  return hipGetLastError();
}
hipError_t launch_warm_start(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st, const double *x0) {
  MI_DISPATCH(warm_start_kernel, a, x0);
}

static inline unsigned nblk(size_t total, int bs) { return (unsigned)((total + bs - 1) / bs); }

hipError_t launch_interleave(const double *src, double *dst, const int *ids, int nq, int len, int BT, hipStream_t st) {
  if (!nq || !len) return hipSuccess;
  hipLaunchKernelGGL(interleave_kernel, dim3(nblk((size_t)nq * len, 256)), dim3(256), 0, st, src, dst, ids, nq, len, BT);
  return hipGetLastError();
}
hipError_t launch_scatter(const double *src, double *dst, const int *map, const int *ids, int nq, int srclen,
                          const SchedDev &sd, int BT, hipStream_t st) {
  if (!nq || !sd.n_slots) return hipSuccess;
  hipLaunchKernelGGL(scatter_kernel, dim3(nblk((size_t)nq * sd.n_slots, 256)), dim3(256), 0, st, src, dst, map, ids, nq, srclen, sd, BT);
  return hipGetLastError();
}
hipError_t launch_swap_plain(double *base, const int2 *pairs, int npairs, int len, int BT, hipStream_t st) {
  if (!npairs || !len) return hipSuccess;
  hipLaunchKernelGGL(swap_plain_kernel, dim3(nblk((size_t)npairs * len, 256)), dim3(256), 0, st, base, pairs, npairs, len, BT);
  return hipGetLastError();
}
hipError_t launch_swap_int(int *base, const int2 *pairs, int npairs, int len, int BT, hipStream_t st) {
  if (!npairs || !len) return hipSuccess;
  hipLaunchKernelGGL(swap_int_kernel, dim3(nblk((size_t)npairs * len, 256)), dim3(256), 0, st, base, pairs, npairs, len, BT);
  return hipGetLastError();
}
hipError_t launch_swap_sched(double *base, const int2 *pairs, int npairs, const SchedDev &sd, int BT, hipStream_t st) {
  if (!npairs) return hipSuccess;
  const size_t per = (size_t)sd.n_steps * 64;
  hipLaunchKernelGGL(swap_sched_kernel, dim3(nblk((size_t)npairs * per, 256)), dim3(256), 0, st, base, pairs, npairs, sd, BT);
  return hipGetLastError();
}
hipError_t launch_deinterleave(const double *src, double *dst, int nq, int len, int BT, hipStream_t st) {
  if (!nq || !len) return hipSuccess;
  hipLaunchKernelGGL(deinterleave_kernel, dim3(nblk((size_t)nq * len, 256)), dim3(256), 0, st, src, dst, nq, len, BT);
  return hipGetLastError();
}
hipError_t launch_gather_status(const int *iscal, int32_t *status, int32_t *iters, int B, int BT, hipStream_t st) {
  hipLaunchKernelGGL(gather_status_kernel, dim3(nblk((size_t)B, 256)), dim3(256), 0, st, iscal, status, iters, B, BT);
  return hipGetLastError();
}
hipError_t launch_bounds(const double *gl, const double *gu, double *l, double *u, const double *Esc,
                         const double *rho_vec, const double *dscal, int *changed, int B, int m, int BT,
                         int scaling, hipStream_t st) {
  if (!m) return hipSuccess;
  hipLaunchKernelGGL(bounds_kernel, dim3(nblk((size_t)B * m, 256)), dim3(256), 0, st, gl, gu, l, u, Esc, rho_vec, dscal, changed, B, m, BT, scaling);
  return hipGetLastError();
}

}  // namespace miosqp
