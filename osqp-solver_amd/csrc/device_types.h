// device_types.h -- POD shared by kernels.hip and solver.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace miosqp {

#define MI_CHUNK 16

// device view of one host_core.hpp Schedule (tables are shared by all tiles)
struct SchedDev {
  const uint32_t *step, *idxw;          // per step: descriptor (sched_format.h); per slot: index word
  const uint64_t *idxw64;               // wide index words instead of idxw (KernelArgs::wide)
  const uint32_t *lvl_pos, *tail_bar;   // (n_levels+1) x nw stream positions; per wave: barriers owed after the last step
  int n_phases, nw, n_levels;
  uint32_t n_steps, n_slots;
};

// device view of host_core.hpp DenseTail (k == 0: none)
struct DenseTailDev {
  int s, k, n_phases;
  uint32_t n_steps;
  const uint32_t *task;                             // 4 words per task
  const uint32_t *wave_task, *wave_step, *tail_bar;
};

// per-QP double scalars, laid out [tile][DS_COUNT][BT]
enum { DS_C = 0, DS_CINV, DS_RHO, DS_RHO_EST, DS_PRI_RES, DS_DUA_RES, DS_OBJ, DS_COUNT };
// per-QP int scalars, laid out [tile][IS_COUNT][BT]
// (IS_CUR: iterations of the QP's current solve so far, kept by check_kernel - the continuous entry points begin the solves
//  of a batch at different launches, and every QP counts its own iterations; IS_ITER: the count at which it finished)
// (continuous batching: the per-QP calls prepare a QP on a second stream while the others iterate.  IS_PENDING = 1: a solve
//  has been begun and joins the first advance launch that sees the mark; 2: the solve is paused for its refactorisation
//  (IS_NEED_REFACTOR = 1 until factor_kernel has run).  A pending / paused slot reads IS_DONE = 1, so every kernel leaves it
//  alone.  IS_EPOCH counts the solves begun in the slot: the host tells a finished solve from the previous one by it.)
enum { IS_STATUS = 0, IS_ITER, IS_RHO_UPDATES, IS_DONE, IS_NEED_REFACTOR, IS_CUR, IS_PENDING, IS_EPOCH, IS_COUNT };

struct KernelArgs {
  int n, m, N, B;
  SchedDev fwd, bwd, chk;
  const uint32_t *pinv;                 // natural index -> position in the permuted solve vector
  const uint32_t *xloc;                 // permuted row -> position of its forward result / backward input (host_core.hpp Analysis::xloc)
  // tile-interleaved value arrays: [tile][len][BT]
  const double *fwd_val, *bwd_val, *chk_val, *dinv;
  DenseTailDev dt;
  const double *dt_val;                 // per QP: the stream of the inverted Schur complement ([slot][dt.n_steps * 64])
  // The factor streams exist twice: the snapshot of the last setup / update (fwd_val0, bwd_val0, dt_val0) and the working copy
  // the refactorisations write.  use_work[slot] != 0: the QP's current factor is the working copy (it was refactored since the
  // last reset); else its streams are read straight from the snapshot - a reset clears the flags instead of copying gigabytes.
  const double *fwd_val0, *bwd_val0, *dt_val0;
  const int *use_work;                  // null: always the working copy
  double *x, *z, *y;
  const double *q, *l, *u, *rho_vec, *rho_inv, *Dsc, *Dsc_inv, *Esc, *Esc_inv;
  double *dx, *dy, *out1, *out2, *dscal;
  int *iscal;
  const int *qp_of_slot;                // slot (tile*BT + b) -> global QP id, -1 = empty (compaction during a solve)
  double *x_out, *y_out;                // QP-major [B][n], [B][m]
  double *xs_global;                    // non-null: the solve vector lives here ([tile][xs_len][BT]) instead of LDS
  int xs_len;                           // length of the solve vector: Analysis::Next >= n + m
  int wide;                             // 32-bit gather / row indices (Schedule::idxw64): vectors of 65 535 entries and more; needs xs_global, BT = 1
  // settings (row S)
  double sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, rho_tolerance;
  int check_termination, rho_interval, max_iter, scaled_termination, scaling, adaptive_rho;
  int iter_begin, iter_end;             // this launch runs iterations (iter_begin, iter_end]
  int op_out_lds;                       // spmv op: results staged in LDS (the launcher sized it)
  int info_at_end;                      // a check_kernel follows: store delta_x / delta_y of the last iteration
  // multi-workgroup mode (one large QP, global solve vector): mw_groups workgroups share the QP's triangular solves and
  // vector steps; mw_bar = {arrival count, generation, error flag} of their grid barrier (zero between launches)
  int mw_groups;
  unsigned *mw_bar;
  double *mw_scratch;                   // [8 slots][256 workgroups][16]: partial norms / sums of check_kernel
  // dataflow form of the triangular solves (Analysis::df; always with the wide index words and one QP): shadow offset of the
  // in-place results, flags per permuted row (bit 0 forward A step, bit 1 backward A step, bit 2 multi-row chunk)
  int df;
  unsigned df_shadow;
  const unsigned char *rflag;
  // per-QP entry points (continuous batching): sel[slot] = 0: the slot is not addressed by this launch, j + 1: it is, and
  // its input is row j of the launch's QP-major argument.  null = every QP, row = QP id.
  const int *sel;
};

// device block refactorisation (row E13); tables are host_core.hpp BlockFactor
struct FactorArgs {
  int n, m, N, B, nnzP, nnzK, pa_len, n_levels, force_all;
  int debug_skip;     // timing experiments only (MI_OSQP_FACTOR_SKIP): 1 rank-1 updates, 2 general updates, 4 diag, 8 trsm, 16 scatter
  uint32_t storage;
  SchedDev fwd, bwd;
  const uint32_t *blk, *lvl, *utask, *tri4, *dtask, *ttask, *asm_dst, *asm_src;
  const uint32_t *ubig;      // per level: where the small update tasks begin (BlockFactor::ubig)
  int *use_work;             // per slot: set for every QP refactored here (KernelArgs::use_work), or null
  const int32_t *fwd_srcblk, *bwd_srcblk;
  const double *pa_val, *l, *u, *dscal;
  double *rho_vec, *rho_inv, *Lblk, *Dl, *dinv_scratch, *fwd_val, *bwd_val, *dinv;
  int *iscal, *npos;
  int home_bt;        // QPs per tile of the per-QP arrays ([tile][len][home_bt]); the kernel's own BT (QPs per workgroup) may differ
  const int *work;    // non-null: work list of the slots (tile * BT + b) to refactor, packed BT per work tile, -1 = none;
                      // null: every slot of the batch (force_all) / the slots whose IS_NEED_REFACTOR flag is set
  double sigma;
  int dt_k;           // rows of the dense tail: their pivots are counted by dense_inverse_kernel, which then checks the inertia
  // the ONE QP of a handle shared by mw_groups workgroups (single large QPs; grid barriers between the phases of a level)
  int mw_groups;
  unsigned *mw_bar;
};

// dense tail: assembly of the Schur complement + its inversion (tail_kernel; one workgroup per refactored QP, after
// factor_kernel).  Tables: host_core.hpp DenseTail.
struct TailArgs {
  int n, N, s, k, kbt, home_bt;
  uint32_t storage, n_slots;
  int n_lt, n_ltcol;
  uint32_t n_quads;             // length of asm_q / 64
  int nh;                       // row tiles of one staged half of the panel
  uint32_t cs_doubles;          // LDS doubles reserved for the staged panel / the pivot block's image (Ps follows)
  const int *work;              // slot of every (work tile, lane class), -1 = none
  const uint32_t *lt_pos, *ltcol_col, *tile_tab, *wave_tiles, *diag_tile;
  const uint64_t *asm_q64;
  const int32_t *src_tile;      // per stream slot: tile-order offset in the scratch, MI_SRC_ZERO = 0 (host replay; unused by the kernel)
  const uint32_t *dt_task, *dt_task_step;       // the product's tasks (DenseTail::task) in stream order and the first stream step of each
  uint32_t n_tasks;
  const double *Lblk, *Dl;
  double *Sd, *dt_val, *dinv;
  int *npos, *iscal;
  unsigned long long *trace;    // null, or 8 clock sums per workgroup (MI_OSQP_TAIL_TRACE)
};
hipError_t launch_tail(const TailArgs &a, int nwork, size_t lds_asm, size_t lds, hipStream_t st);
hipError_t launch_factor(const FactorArgs &a, int BT, int tiles, int threads, hipStream_t st);
size_t factor_lds_bytes(int BT, int threads);
bool factor_fits_lds(const FactorArgs &a, int threads);      // the LDS-resident form of factor_kernel<1> applies (one QP per workgroup, no group sharing)

int max_coresident_groups(int threads, size_t lds, int n_cus);
int max_coresident_factor_groups(int threads, int n_cus);
hipError_t launch_iterate(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st);
hipError_t launch_check(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st);
// up to max_segments segments of seg_len iterations + check per tile in one launch (advance_kernel; LDS-resident tiles only);
// flags / solutions of the tiles it touched go to the pinned host images host_is / host_ds
// (counter: device word the tiles count themselves out on; host_done: pinned word that receives seq when the last tile has left)
hipError_t launch_advance(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st, int max_segments, int seg_len,
                          int *host_is, double *host_ds, unsigned *stop, unsigned seq, unsigned *counter, unsigned *host_done);
hipError_t launch_spmv(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st,
                       const double *x, const double *y, double *Px, double *Aty, double *Ax);
// fused P x / A'y / A x (spmv_fused_kernel): the compact P and A values of a tile staged in LDS once and used for all
// three products; row r of [P x ; A'y ; A x] = sum over ent[ptr[r] .. ptr[r+1]) of val[e & 0xFFFF] * [x ; y][e >> 16]
struct SpmvFused {
  const uint32_t *ptr, *ent;
  const double *pa_val;          // [tile][pa_len][BT]
  int pa_len;
  // ELL form of the same rows for the prefetching variant (ell != null): the rows sorted by length (descending) and cut
  // into passes of 512 rows; pass p holds its rows' entries as ell[ell_off[p] + k * 512 + thread], k < ell_k[p], padded
  // with entries that point at a zero value (position pa_len); rowid[pass * 512 + thread] = output row or 0xFFFFFFFF
  const uint32_t *ell, *rowid;
  int n_pass;
  uint32_t ell_off[4], ell_k[4];
};
hipError_t launch_spmv_fused(const KernelArgs &a, const SpmvFused &t, int BT, int tiles, int n_cus, hipStream_t st,
                             const double *x, const double *y, double *Px, double *Aty, double *Ax);
size_t spmv_fused_lds_bytes(int n, int m, int pa_len, int BT);
hipError_t launch_kkt_solve(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st,
                            const double *rhs, double *sol);
hipError_t launch_kkt_trace(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st,
                            const double *rhs, double *sol, uint32_t *trace, uint32_t words);
hipError_t launch_warm_start(const KernelArgs &a, int BT, int tiles, int threads, size_t lds, hipStream_t st,
                             const double *x0);
hipError_t launch_interleave(const double *src, double *dst, const int *ids, int nq, int len, int BT, hipStream_t st);
hipError_t launch_scatter(const double *src, double *dst, const int *map, const int *ids, int nq, int srclen,
                          const SchedDev &sd, int BT, hipStream_t st);
hipError_t launch_swap_plain(double *base, const int2 *pairs, int npairs, int len, int BT, hipStream_t st);
hipError_t launch_swap_int(int *base, const int2 *pairs, int npairs, int len, int BT, hipStream_t st);
hipError_t launch_swap_sched(double *base, const int2 *pairs, int npairs, const SchedDev &sd, int BT, hipStream_t st);
hipError_t launch_deinterleave(const double *src, double *dst, int nq, int len, int BT, hipStream_t st);
// Row E2 on the device (ruiz_kernel): new raw A values (and bounds) of every QP of the batch -> unscale, Ruiz rescale,
// scaled bounds; one workgroup per QP, the arithmetic and its order are those of host_core.cpp scale_qp / unscale_qp.
struct RuizArgs {
  int n, m, nnzP, nnzA, B, BT, iters;
  const int *ids;                            // non-null: workgroup j serves QP ids[j] (B = length of the list); rawA / rawl / rawu / pa_out are indexed by j
  int raw_by_qp;                             // 1: rawA / rawl / rawu are indexed by the QP's number instead (arrays kept per QP on the device)
  int fresh;                                 // 1: equilibrate from the raw P and q kept since setup (rawP / rawq, [QP][nnzP], [QP][n]) instead of
  const double *rawP, *rawq;                 //    unscaling the values in force: bit for bit what setup computes for (P, q, rawA, rawl, rawu)
  const int32_t *Prow, *Pcol, *Arow, *Acol;   // per entry of triu(P) / A: row, column
  const double *rawA;                        // [B][nnzA] new values of A (natural CSC order)
  const double *rawl, *rawu;                 // [B][m] new bounds, or null: keep the bounds (unscale, rescale)
  double *pa_val, *q, *Dsc, *Dsc_inv, *Esc, *Esc_inv, *l, *u, *dscal;      // tile-interleaved state of the handle
  double *dn, *en;                           // scratch [B][n], [B][m]
  double *pa_out;                            // [B][nnzP + nnzA]: the scaled values once more, QP-major (for the check-stream scatter)
};
hipError_t launch_ruiz(const RuizArgs &a, hipStream_t st);
// ---- per-QP entry points (continuous batching; solver.hip "continuous")
// begin a solve of the listed slots: status unsolved, own iteration count 0, next epoch, pending until an advance launch
// activates it; slots whose factor is invalid (IS_NEED_REFACTOR < 0) end at once as kNonConvex.
// clear[j] != 0: the count of rho updates restarts at 0.
hipError_t launch_start_slots(const KernelArgs &a, const int *slots, const int *clear, int nslots, int BT, int cold, hipStream_t st);
// the state a fresh setup leaves in the listed slots: zero iterates, rho = rho0, no rho updates, idle
hipError_t launch_fresh_slots(const KernelArgs &a, const int *slots, int nslots, int BT, double rho0, hipStream_t st);
// work[0 .. nslots) = the slots whose rho changed (IS_NEED_REFACTOR = 1: paused, or finished at max_iter; any order), then -1
hipError_t launch_worklist(const int *iscal, int *work, int nslots, int BT, hipStream_t st);
// paused slots: resume (flag 0) or, when the refactorisation lost the inertia (flag -1), kNonConvex at their own iteration count
hipError_t launch_resume_flagged(const KernelArgs &a, int nslots, int BT, hipStream_t st);
// dst[slot] = src[slot] for the listed slots: [slot][per] streams / [tile][len][BT] interleaved arrays
hipError_t launch_copy_slot_streams(double *dst, const double *src, const int *slots, int nslots, size_t per, hipStream_t st);
hipError_t launch_keep_rows(double *dst, const double *src, const int *ids, int n_ids, int len, hipStream_t st);
// dst[slot][:] = src[slot][:] for the slots whose flag is set (want = 1) / clear (want = 0); flags null: all
hipError_t launch_copy_flagged_streams(double *dst, const double *src, const int *flags, int want, int nslots, size_t per, hipStream_t st);
hipError_t launch_copy_slot_rows(double *dst, const double *src, const int *slots, int nslots, int len, int BT, hipStream_t st);
// ---- GOMP re-linearisation on the device (solver.hip "gomp scene"): ConstraintBuilder::withObstacles + isSolutionOK for
// built-in kinematic models ([REF] src/constraints/constraint-builder.h:90-136, src/gomp-solver.h:141-199)
enum { MI_GM_UR5E_FLANGE = 1, MI_GM_UR5E_WRIST3 = 2, MI_GM_UR5E_ELBOW = 3, MI_GM_YAW_2LINK = 4, MI_GM_TABLE = 5 };
struct GompBallDev { int model, is_gripper; double radius; double param[12]; };
struct GompLineDev { double D[3], A[3]; int below, pad; };
struct GompArgs {
  int dims, W, n_balls, n_lines, n, m, nnzA, n_ids, row0, write_rows;
  const int *ids;                // the listed QPs
  const GompBallDev *balls;
  const GompLineDev *lines;
  double con_lo[3], con_hi[3];   // work-space box of the gripper balls (+-1e30 = none)
  const int *aidx;               // [3-D row][dims]: position of that row's entry in column nthPos(w) + j of A's value array
  const double *traj;            // [n_ids][n]: the trajectories to linearise around (row = position in the list)
  double *A, *l, *u;             // the kept raw constraint data, [B][nnzA] / [B][m] by QP: the 3-D rows are rewritten
  int *ok;                       // [n_ids]: isSolutionOK of the trajectory
};
hipError_t launch_gomp_relinearise(const GompArgs &g, hipStream_t st);
hipError_t launch_gather_status(const int *iscal, int32_t *status, int32_t *iters, int B, int BT, hipStream_t st);
hipError_t launch_fail_slots(const KernelArgs &a, const int *slots, int nfail, int BT, int iter, hipStream_t st);
hipError_t launch_bounds(const double *gl, const double *gu, double *l, double *u, const double *Esc,
                         const double *rho_vec, const double *dscal, int *changed, int B, int m, int BT,
                         int scaling, hipStream_t st);

}  // namespace miosqp
