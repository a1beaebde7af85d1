/*
 * mi_osqp.h -- C-ABI of the MI355X-native OSQP ADMM core.
 *
 * This is the drop-in boundary for the solver path of ZPP-Robotics/OSQP-Solver:
 * every entry point names the reference interface it replaces ([REF] = path
 * under /root/reference, file:line).  The reference reaches its solver only
 * through class QPSolver ([REF] src/osqp-wrapper.h:12-60), whose four methods
 * wrap google/osqp-cpp calls; a maintainer binds this library there (see
 * INTEGRATION.md for the exact stub).
 *
 * Conventions
 *   - plain C, no torch / Eigen types; all arrays caller-owned and copied at
 *     the call (matches the reference: its OsqpInstance dies at the end of the
 *     constructor, [REF] src/osqp-wrapper.h:18-31);
 *   - sparse matrices are CSC with 64-bit indices = Eigen::SparseMatrix<double,
 *     ColMajor, long long> ([REF] src/utils.h:12);
 *   - +-1e30 means "unbounded" ([REF] src/constraints/constraints.h:11);
 *   - every function returns an mi_osqp_error (0 = ok) unless stated; nothing
 *     aborts and nothing falls back to a CPU solve: without a usable gfx950
 *     device setup fails with MI_OSQP_ERR_DEVICE.
 */
#ifndef MI_OSQP_H
#define MI_OSQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MI_OSQP_OK = 0,
  MI_OSQP_ERR_INVALID_DATA = 1,      /* dims, l>u, bad CSC      -> osqp-cpp InvalidArgument */
  MI_OSQP_ERR_INVALID_SETTINGS = 2,
  MI_OSQP_ERR_PATTERN_CHANGED = 3,   /* UpdateConstraintMatrix with a new pattern */
  MI_OSQP_ERR_NONCONVEX = 4,         /* KKT inertia wrong / zero pivot */
  MI_OSQP_ERR_DEVICE = 5,            /* HIP error or no gfx950 device */
  MI_OSQP_ERR_NULL = 6,
  MI_OSQP_ERR_ALLOC = 7
} mi_osqp_error;

/* exit codes = osqp::OsqpExitCode as consumed by the reference
 * ([REF] src/utils.h:11; src/gomp-solver.h:40,46-49,68,72,79;
 *  examples/solver-example.cpp:71,94).  Values follow the osqp-cpp enum order. */
typedef enum {
  MI_OSQP_EXIT_OPTIMAL = 0,
  MI_OSQP_EXIT_PRIMAL_INFEASIBLE = 1,
  MI_OSQP_EXIT_DUAL_INFEASIBLE = 2,
  MI_OSQP_EXIT_OPTIMAL_INACCURATE = 3,
  MI_OSQP_EXIT_PRIMAL_INFEASIBLE_INACCURATE = 4,
  MI_OSQP_EXIT_DUAL_INFEASIBLE_INACCURATE = 5,
  MI_OSQP_EXIT_MAX_ITERATIONS = 6,
  MI_OSQP_EXIT_INTERRUPTED = 7,
  MI_OSQP_EXIT_TIME_LIMIT_REACHED = 8,
  MI_OSQP_EXIT_NON_CONVEX = 9,
  MI_OSQP_EXIT_UNKNOWN = 10
} mi_osqp_exit_code;

/* OsqpSettings subset that influences the iterates.  The reference sets only
 * `verbose` ([REF] src/osqp-wrapper.h:26-27), so defaults = osqp 0.6.x defaults. */
typedef struct {
  double  rho;                    /* 0.1  */
  double  sigma;                  /* 1e-6 */
  int64_t scaling;                /* 10   */
  int64_t adaptive_rho;           /* 1    */
  int64_t adaptive_rho_interval;  /* 0 = auto -> 4*check_termination (deterministic) */
  double  adaptive_rho_tolerance; /* 5    */
  int64_t max_iter;               /* 4000 */
  double  eps_abs;                /* 1e-3 */
  double  eps_rel;                /* 1e-3 */
  double  eps_prim_inf;           /* 1e-4 */
  double  eps_dual_inf;           /* 1e-4 */
  double  alpha;                  /* 1.6  */
  int64_t scaled_termination;     /* 0    */
  int64_t check_termination;      /* 25   */
  int64_t warm_start;             /* 1    */
  int64_t verbose;                /* 0 here; the reference passes 1 (log only) */
} mi_osqp_settings;

typedef struct {
  int64_t iter;
  int64_t status_val;   /* raw osqp status (1 solved, -2 max iter, ...) */
  int64_t exit_code;    /* mi_osqp_exit_code */
  double  obj_val;
  double  pri_res;
  double  dua_res;
  int64_t rho_updates;
  double  rho_estimate;
  double  rho;
} mi_osqp_info;

/* analysis / schedule statistics (DESIGN.md quotes these) */
typedef struct {
  int64_t n, m, N, batch, tile, n_tiles;
  int64_t nnz_P_triu, nnz_A, nnz_KKT, nnz_L;
  int64_t n_supernodes, n_blocks;
  int64_t fwd_levels, bwd_levels, fwd_slots, bwd_slots, chk_slots;
  int64_t lds_bytes, threads_per_block;
  int64_t dense_tail_rows, dense_tail_slots;   /* trailing rows served by the inverted Schur complement (0 = none), its stream slots */
  double  setup_seconds_host, setup_seconds_factor, setup_seconds_upload;
  int64_t nnz_L_before_tail;   /* entries of L in the columns before the dense tail (= nnz_L without one): what the two sweeps stream */
  int64_t solve_groups, solve_group_threads;   /* large single QP: workgroups x threads that share its sweeps (0 = one workgroup); never more than the device keeps resident */
} mi_osqp_stats;

typedef struct mi_osqp_solver mi_osqp_solver; /* one QP  */
typedef struct mi_osqp_batch  mi_osqp_batch;  /* B QPs, ONE shared sparsity pattern */

void        mi_osqp_default_settings(mi_osqp_settings *s);
const char *mi_osqp_exit_code_name(int64_t exit_code);  /* replaces osqp::ToString(OsqpExitCode) */
const char *mi_osqp_error_name(int64_t err);
const char *mi_osqp_version(void);
const char *mi_osqp_last_error(void);   /* text of the last MI_OSQP_ERR_DEVICE / _ALLOC on this thread */

/* ------------------------------------------------------------------ single QP
 * Replaces QPSolver::QPSolver(const QPConstraints&, const QPMatrixSparse&)
 * = OsqpSolver::Init ([REF] src/osqp-wrapper.h:16-31).  P may hold both
 * triangles (the reference's triDiagonalMatrix does, [REF] src/utils.h:53-61);
 * the upper one is used.  q may be NULL (= 0, [REF] src/osqp-wrapper.h:22). */
int mi_osqp_setup(mi_osqp_solver **out, int64_t n, int64_t m,
                  const int64_t *P_colptr, const int64_t *P_rowidx, const double *P_val,
                  const double *q,
                  const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                  const double *l, const double *u, const mi_osqp_settings *settings);
/* QPSolver::update, first half = OsqpSolver::UpdateConstraintMatrix
 * ([REF] src/osqp-wrapper.h:36): same pattern required. */
int mi_osqp_update_A(mi_osqp_solver *h, const int64_t *A_colptr, const int64_t *A_rowidx,
                     const double *A_val);
/* QPSolver::update, second half = OsqpSolver::SetBounds ([REF] src/osqp-wrapper.h:40). */
int mi_osqp_update_bounds(mi_osqp_solver *h, const double *l, const double *u);
/* QPSolver::update as ONE call ([REF] src/osqp-wrapper.h:33-43: UpdateConstraintMatrix, then SetBounds): the state it
 * leaves is that of mi_osqp_update_A followed by mi_osqp_update_bounds, with one numeric refactorisation instead of two
 * when the new bounds change a row's type (equality / inequality / free).  Nothing is changed when l > u somewhere or the
 * pattern differs. */
int mi_osqp_update_A_bounds(mi_osqp_solver *h, const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                            const double *l, const double *u);
/* QPSolver::setWarmStart = SetPrimalWarmStart ([REF] src/osqp-wrapper.h:45-49). */
int mi_osqp_warm_start_x(mi_osqp_solver *h, const double *x);
/* QPSolver::solve = Solve ([REF] src/osqp-wrapper.h:52); returns error code,
 * exit code and residuals in *info (may be NULL). */
int mi_osqp_solve(mi_osqp_solver *h, mi_osqp_info *info);
/* primal_solution() ([REF] src/osqp-wrapper.h:53) / dual_solution(); NaN-filled
 * when the exit code carries no solution. */
int mi_osqp_get_primal(mi_osqp_solver *h, double *x_out);
int mi_osqp_get_dual(mi_osqp_solver *h, double *y_out);
int mi_osqp_get_stats(mi_osqp_solver *h, mi_osqp_stats *st);
void mi_osqp_free(mi_osqp_solver *h);

/* --------------------------------------------------------------------- batch
 * B independent QPs sharing ONE sparsity pattern (the GOMP situation: the
 * reference keeps A's pattern constant on purpose, [REF]
 * src/constraints/constraint-builder.h:112-116).  Values/bounds are QP-major:
 * P_val[B][nnzP], q[B][n] (or NULL), A_val[B][nnzA], l/u[B][m].
 * device < 0 selects the current HIP device. */
int mi_osqp_batch_setup(mi_osqp_batch **out, int64_t B, int64_t n, int64_t m,
                        const int64_t *P_colptr, const int64_t *P_rowidx, const double *P_val,
                        const double *q,
                        const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                        const double *l, const double *u, const mi_osqp_settings *settings,
                        int64_t device);
int mi_osqp_batch_update_A(mi_osqp_batch *h, const int64_t *A_colptr, const int64_t *A_rowidx,
                           const double *A_val);
int mi_osqp_batch_update_bounds(mi_osqp_batch *h, const double *l, const double *u);
int mi_osqp_batch_update_A_bounds(mi_osqp_batch *h, const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                                  const double *l, const double *u);      /* see mi_osqp_update_A_bounds */
int mi_osqp_batch_warm_start_x(mi_osqp_batch *h, const double *x);
/* Blocking solve of all B QPs (the ADMM iterate runs on the GPU). */
int mi_osqp_batch_solve(mi_osqp_batch *h);
int mi_osqp_batch_get_primal(mi_osqp_batch *h, double *x_out /*[B][n]*/);
int mi_osqp_batch_get_dual(mi_osqp_batch *h, double *y_out /*[B][m]*/);
int mi_osqp_batch_get_info(mi_osqp_batch *h, mi_osqp_info *info /*[B]*/);
int mi_osqp_batch_get_stats(mi_osqp_batch *h, mi_osqp_stats *st);
/* the elimination order the analysis chose for the KKT matrix [[P + sigma I, A'], [A, -1/rho]]: kkt_perm[k] = natural index
 * (0 .. n-1 variables, n .. n+m-1 constraint rows) eliminated k-th; n + m entries.  (Lets a CPU checker factor the same
 * large KKT matrix without a minimum-degree ordering of its own.) */
int mi_osqp_batch_get_ordering(mi_osqp_batch *h, int64_t *kkt_perm);
void mi_osqp_batch_free(mi_osqp_batch *h);
/* Compute the pattern analysis (ordering, symbolic factor, schedules) that a later mi_osqp_batch_setup / mi_osqp_setup of B
 * QPs with this sparsity pattern on this device will need, into the process-wide analysis cache.  Thread-safe; blocking (call
 * it from a spare host thread); a setup that arrives meanwhile waits for it.  The reference builds a fresh QPSolver per horizon
 * ([REF] src/gomp-solver.h:61-65): a planner that knows its horizons up front takes the analyses off its critical path. */
int mi_osqp_prefetch_analysis(int64_t B, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi,
                              const int64_t *Ap, const int64_t *Ai, int64_t device);
/* Freed handles leave their device buffers (at most 8 GiB), pinned host buffers (at most 1 GiB) and stream / event sets in
 * process-wide caches for the next setup - the GOMP drivers build one solver per horizon segment; this returns them to
 * the HIP runtime. */
void mi_osqp_release_device_cache(void);

/* Device-resident I/O (HBM pointers on the solver's device; `stream` is a
 * hipStream_t passed as void*, NULL = default stream).
 * d_l/d_u: [B][m] doubles, unscaled; the E-scaling and the constraint-type
 * check run on the device (no refactor happens unless a type changes, in which
 * case the call returns through the host path transparently). */
int mi_osqp_batch_update_bounds_device(mi_osqp_batch *h, const double *d_l, const double *d_u, void *stream);
/* QPSolver::update ([REF] src/osqp-wrapper.h:33-43) with the new A values ([B][nnzA], CSC order of setup's pattern) and bounds
 * ([B][m]) in HBM; `stream`: the stream that wrote them (NULL: none pending) */
int mi_osqp_batch_update_A_bounds_device(mi_osqp_batch *h, const double *d_Av, const double *d_l, const double *d_u, void *stream);
/* Solve and leave x[B][n] (and optionally status[B]/iters[B], int32) in HBM. */
int mi_osqp_batch_solve_device(mi_osqp_batch *h, double *d_x_out, int32_t *d_status, int32_t *d_iters, void *stream);
/* Back to the state right after setup (or after the last update that refactored): cold-start every QP, restore rho,
 * the rho vectors and the factor of that moment, forget the count of rho updates.  A planner that builds the same
 * solver again and again (one per horizon segment and run, [REF] src/gomp-solver.h:61) may keep the handle instead:
 * reset + update_bounds + warm_start gives bitwise the results of a fresh setup with the same P and A.  The bench uses it
 * so that repeated steps do identical work. */
int mi_osqp_batch_reset(mi_osqp_batch *h);
/* Totals of the last solve: ADMM iterations summed over QPs, iterate launches (= segments),
 * seconds in iterate_kernel (HIP events), in device refactorisations and in the
 * compaction swaps (wall, including their synchronisation). */
int mi_osqp_batch_last_solve_stats(mi_osqp_batch *h, int64_t *total_iters, int64_t *kernel_launches,
                                   double *device_seconds, double *refactor_seconds, int64_t *refactor_count,
                                   double *compact_seconds);

/* ------------------------------------------------------ continuous batching
 * The reference's SQP loop is per trajectory: solve -> check -> re-linearise -> update -> solve again
 * ([REF] src/gomp-solver.h:70-88), with a fresh QPSolver per horizon segment ([REF] src/gomp-solver.h:61-65).  The QPs of a
 * batch therefore do not finish together, and a planner must not wait for the slowest of them.  These entry points address
 * single QPs of a batch handle (`ids`: n_ids QP numbers in [0, B); per-QP arguments are QP-major in the order of `ids`) and
 * advance whatever is iterating without blocking:
 *
 *     reinit_some / update_A_bounds_some / warm_start_x_some   new data for QPs that are not iterating
 *     solve_begin_some                                         Solve() entry of those QPs (own iteration count from 0)
 *     advance(n_segments)                                      enqueue ONE launch in which every iterating QP runs up to
 *                                                              n_segments segments of L iterations + check, L =
 *                                                              gcd(check_termination, adaptive_rho_interval, max_iter) = 25 by
 *                                                              default - with the checks / rho updates a blocking solve of
 *                                                              its own would see at those iterations.  With n_segments > 1 the
 *                                                              launch ends at the first segment boundary after ANY QP of the
 *                                                              handle has finished (so that the caller can react to it), and
 *                                                              a QP whose rho changes pauses until the refactorisation that
 *                                                              follows the launch in stream order
 *     poll(wait, ...)                                          which QPs finished in the oldest advance not polled yet
 *     get_primal_some / get_dual_some / get_info_some          results of finished QPs (host memory, no device access)
 *
 * Every QP takes exactly the iterations of a mi_osqp_batch_solve of its own: same exit code, iteration count, rho updates
 * and solution, bit for bit.  Nothing here waits for the device except poll() (and a full staging ring); at most two
 * advances may be waiting for their poll().  A blocking mi_osqp_batch_* call ends the continuous mode of the handle (solves
 * in flight are forgotten).  Not available for handles whose solve vector does not fit LDS (large single QPs).
 *
 * reinit_some = QPSolver::QPSolver for those QPs ([REF] src/osqp-wrapper.h:16-31) with the P and q given at setup and new
 * A values / bounds: equilibration from the raw data, rho = settings.rho, zero iterates, no rho updates - the state
 * mi_osqp_batch_setup leaves for that QP, bit for bit, without analysis or allocation.
 * update_A_bounds_some = QPSolver::update ([REF] src/osqp-wrapper.h:33-43) for those QPs. */
int mi_osqp_batch_reinit_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids,
                              const double *A_val /*[n_ids][nnzA]*/, const double *l /*[n_ids][m]*/, const double *u);
int mi_osqp_batch_update_A_bounds_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids,
                                       const double *A_val, const double *l, const double *u);
int mi_osqp_batch_warm_start_x_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, const double *x /*[n_ids][n]*/);
int mi_osqp_batch_solve_begin_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids);
int mi_osqp_batch_advance(mi_osqp_batch *h, int64_t n_segments);
/* wait != 0: block until the oldest unpolled advance has run; wait == 0: *n_finished = -1 when it has not.  ids_out receives
 * the QPs that finished in it (capacity >= B is always enough; too small: error, *n_finished = the number, nothing consumed). */
int mi_osqp_batch_poll(mi_osqp_batch *h, int64_t wait, int64_t *n_finished, int64_t *ids_out, int64_t capacity);
int mi_osqp_batch_get_primal_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, double *x_out /*[n_ids][n]*/);
int mi_osqp_batch_get_dual_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, double *y_out /*[n_ids][m]*/);
int mi_osqp_batch_get_info_some(mi_osqp_batch *h, int64_t n_ids, const int64_t *ids, mi_osqp_info *info /*[n_ids]*/);
int64_t mi_osqp_batch_running(mi_osqp_batch *h);      /* QPs whose solve has been begun and not been reported by poll() */

/* ------------------------------------------- GOMP re-linearisation on the device
 * The step before and the step after the solver path in the reference's SQP loop - the re-linearised 3-D / obstacle rows
 * of the constraint matrix (ConstraintBuilder::withObstacles, [REF] src/constraints/constraint-builder.h:90-136) and the
 * feasibility check that ends the loop (GOMPSolver::isSolutionOK, [REF] src/gomp-solver.h:141-199) - for collision balls
 * whose kinematics are built-in models (the reference binds arbitrary host callbacks, [REF] src/utils.h:21-22,33-42; those
 * stay on the host path of include/mi_osqp/gomp.hpp).  A scene belongs to one batch handle whose QPs have the reference's
 * GOMP row layout ([REF] constraint-builder.h:34-44) for `dims` joints and `waypoints` waypoints; it keeps the raw
 * constraint data of every QP on the device, so an SQP step moves one trajectory to the device and one flag back. */
typedef enum {
  MI_GOMP_MODEL_UR5E_FLANGE = 1,   /* UR5e (published DH parameters, include/mi_osqp/ur5e_kinematics.hpp): tool flange */
  MI_GOMP_MODEL_UR5E_WRIST3 = 2,   /*   wrist-3 joint (forward_kinematics_6_back)                                      */
  MI_GOMP_MODEL_UR5E_ELBOW = 3,    /*   elbow joint                                                                    */
  MI_GOMP_MODEL_YAW_2LINK = 4,     /* 3 joints: yaw, shoulder, elbow; param = {link 1, link 2, base height}            */
  MI_GOMP_MODEL_TABLE = 5          /* 3 joints: p = (q0, q1, q2), constant 3 x 3 Jacobian in param (known-answer tests) */
} mi_gomp_model;
typedef struct { int32_t model; int32_t is_gripper; double radius; double param[12]; } mi_gomp_ball;   /* = RobotBall */
typedef struct { double dir[2]; double point[3]; int32_t below; int32_t reserved; } mi_gomp_line;       /* = HorizontalLine */
typedef struct mi_gomp_scene mi_gomp_scene;
/* con_lo / con_hi: the work-space box of the gripper balls (con_3d), +-1e30 or NULL = none */
int mi_gomp_scene_create(mi_gomp_scene **out, mi_osqp_batch *h, int64_t dims, int64_t waypoints,
                         int64_t n_balls, const mi_gomp_ball *balls, int64_t n_lines, const mi_gomp_line *lines,
                         const double *con_lo, const double *con_hi);
void mi_gomp_scene_free(mi_gomp_scene *sc);          /* before mi_osqp_batch_free of its handle; one scene per handle */
/* While a scene exists, mi_osqp_batch_reinit_some / _update_A_bounds_some of its handle also keep the QPs' raw constraint
 * data (as ConstraintBuilder::build() produced it) in the scene, so nothing extra is needed when a trajectory enters the
 * handle.  set_rows writes that copy directly ([n_ids][nnzA], [n_ids][m]); get_rows reads it back (tests). */
int mi_gomp_scene_set_rows(mi_gomp_scene *sc, int64_t n_ids, const int64_t *ids, const double *A_val, const double *l, const double *u);
int mi_gomp_scene_get_rows(mi_gomp_scene *sc, int64_t id, double *A_val, double *l, double *u);       /* (tests) */
/* withObstacles(con_3d, x) on the kept rows of the listed QPs (x: [n_ids][n] trajectories, host); ok_out[j] = isSolutionOK(x_j) */
int mi_gomp_assemble_some(mi_gomp_scene *sc, int64_t n_ids, const int64_t *ids, const double *x, int32_t *ok_out);
/* one SQP step of the listed, finished QPs ([REF] src/gomp-solver.h:79-87): ok_out[j] = isSolutionOK(x_j); the QPs whose
 * trajectory is not acceptable are re-linearised around it and updated (QPSolver::update from the device-resident rows)
 * and are ready for mi_osqp_batch_solve_begin_some */
int mi_gomp_relinearise_some(mi_gomp_scene *sc, int64_t n_ids, const int64_t *ids, const double *x, int32_t *ok_out);

/* ---------------------------------------------------------- multi-GPU batch
 * The batch is the shard axis across the GPUs of a node (the runs of a planner are independent,
 * [REF] src/gomp-solver.h:38-55): the B QPs are cut into n_devices contiguous blocks (the first B % n_devices one QP
 * longer), block k lives on HIP device devices[k] (devices == NULL: 0 .. n_devices-1; a device may be listed more than
 * once - two shards then share it), every call fans out over one long-lived worker thread + stream per shard and joins.  There is
 * no data-path collective; results come back QP-major in the order of the whole batch.  Arguments as for
 * mi_osqp_batch_*; the return value is the first shard error (0 = ok). */
typedef struct mi_osqp_multi mi_osqp_multi;
int mi_osqp_multi_batch_setup(mi_osqp_multi **out, int64_t n_devices, const int64_t *devices,
                              int64_t B, int64_t n, int64_t m,
                              const int64_t *P_colptr, const int64_t *P_rowidx, const double *P_val,
                              const double *q,
                              const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                              const double *l, const double *u, const mi_osqp_settings *settings);
int mi_osqp_multi_batch_update_A(mi_osqp_multi *h, const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val);
int mi_osqp_multi_batch_update_bounds(mi_osqp_multi *h, const double *l, const double *u);
int mi_osqp_multi_batch_update_A_bounds(mi_osqp_multi *h, const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                                        const double *l, const double *u);
int mi_osqp_multi_batch_warm_start_x(mi_osqp_multi *h, const double *x);
int mi_osqp_multi_batch_solve(mi_osqp_multi *h);
/* The same without waiting (every shard has one long-lived worker thread; all calls of this section run on them): solve_async
 * returns once the shards have their job, wait() joins them and returns the first shard error.  Any other multi-batch call
 * joins a pending solve first. */
int mi_osqp_multi_batch_solve_async(mi_osqp_multi *h);
int mi_osqp_multi_batch_wait(mi_osqp_multi *h);
int mi_osqp_multi_batch_get_primal(mi_osqp_multi *h, double *x_out /*[B][n]*/);
int mi_osqp_multi_batch_get_dual(mi_osqp_multi *h, double *y_out /*[B][m]*/);
int mi_osqp_multi_batch_get_info(mi_osqp_multi *h, mi_osqp_info *info /*[B]*/);
/* shard k: its device, its QP range [begin, end) and its single-device handle (owned by the multi handle) */
int64_t mi_osqp_multi_batch_shards(mi_osqp_multi *h);
int mi_osqp_multi_batch_shard(mi_osqp_multi *h, int64_t k, int64_t *device, int64_t *begin, int64_t *end, mi_osqp_batch **handle);
void mi_osqp_multi_batch_free(mi_osqp_multi *h);

/* ------------------------------------------------- the path's kernels as ops
 * (parity tests and roofline measurements call these; all pointers HBM)
 * KKT-structured SpMV on the scaled data: from x[B][n], y[B][m] compute
 * Px[B][n], Aty[B][n], Ax[B][m]  (rows E11/E14 of SURVEY 8(a)). */
int mi_osqp_batch_spmv(mi_osqp_batch *h, const double *d_x, const double *d_y,
                       double *d_Px, double *d_Aty, double *d_Ax, void *stream);
/* One KKT solve per QP with the current factor: sol = K^-1 rhs, [B][n+m]
 * (row E7: permute, level-scheduled L solve, D^-1, L' solve, un-permute). */
int mi_osqp_batch_kkt_solve(mi_osqp_batch *h, const double *d_rhs, double *d_sol, void *stream);
/* Row E13 as an op: rebuild every QP's KKT factor ON THE DEVICE from the current
 * scaled data and rho vector (batched block LDL'), scatter it into the solve
 * schedules.  solve() calls the same kernel for the QPs whose rho changed. */
int mi_osqp_batch_refactor_device(mi_osqp_batch *h);
/* average duration (ms) of the last `iterate` launches measured with HIP events
 * on the launch stream, and their count (bench.py's roofline leg). */
int mi_osqp_batch_kernel_time(mi_osqp_batch *h, double *avg_ms, int64_t *launches);
/* the same for the refactorisation kernels (row E13) since the last call: summed durations (ms, HIP events on the
 * launch stream) of factor_kernel and of dense_inverse_kernel, their launches and the QPs they refactored. */
int mi_osqp_batch_refactor_time(mi_osqp_batch *h, double *factor_ms, double *tail_ms, int64_t *launches, int64_t *qps);
/* the largest of those refactorisations (most QPs; a solve also holds small ones for stragglers): its QPs and the
 * durations of factor_kernel and of the dense-tail kernels (tail_assemble_kernel + tail_kernel). Reset by refactor_time. */
int mi_osqp_batch_refactor_peak(mi_osqp_batch *h, int64_t *qps, double *factor_ms, double *tail_ms);

/* --------------------------------------------------- host-only diagnostics
 * No GPU needed: analyse a pattern + one value set and replay the DEVICE
 * schedules on the host (a sequential interpreter of the same task tables) so
 * that the schedule builder is testable in CPU-only CI.  Not a solve path. */
int mi_osqp_debug_host_kkt_solve(int64_t n, int64_t m,
                                 const int64_t *P_colptr, const int64_t *P_rowidx, const double *P_val,
                                 const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                                 const double *l, const double *u, const mi_osqp_settings *settings,
                                 int64_t tile, const double *rhs /*[n+m], scaled space*/,
                                 double *sol_schedule /*[n+m]*/, double *sol_direct /*[n+m]*/,
                                 mi_osqp_stats *st);

/* Host interpreter of the DEVICE block refactorisation (row E13) against the
 * host left-looking factor: max relative differences of L and D^-1, and
 * counts[4] = {blocks, triples, storage doubles per QP, levels}. */
int mi_osqp_debug_host_block_factor(int64_t n, int64_t m,
                                    const int64_t *P_colptr, const int64_t *P_rowidx, const double *P_val,
                                    const int64_t *A_colptr, const int64_t *A_rowidx, const double *A_val,
                                    const double *l, const double *u, const mi_osqp_settings *settings,
                                    double *max_rel_diff_L, double *max_rel_diff_Dinv, int64_t *counts);

/* Device diagnostics (tile 2, LDS mode only): one KKT solve of the whole batch with
 * per-phase / per-wave shader-clock stamps of two tiles (first, middle).
 * which = 0: run the traced solve; trace receives 2 * dims[3] words (see kkt_trace_kernel).
 * which = 1 / 2: copy the forward / backward phase table (dims[0|1] rows of 4*dims[2]+1 words) instead.
 * dims[4] = {forward phases, backward phases, waves per tile, trace words per tile}. */
int mi_osqp_debug_trace_kkt_solve(mi_osqp_batch *h, int32_t which, const double *d_rhs, double *d_sol,
                                  uint32_t *out, int64_t out_capacity_words, int64_t *dims);

#ifdef __cplusplus
}
#endif
#endif /* MI_OSQP_H */
