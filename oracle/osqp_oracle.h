/*
 * oracle/osqp_oracle.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement, in plain C, of the OSQP 0.6.x ADMM algorithm that the
 * reference reaches through `#include <osqp++.h>`
 * ([REF] /root/reference/src/osqp-wrapper.h:6,18-28,36,40,46,52-53).
 *
 * PARITY UNPINNED: the algorithm lives in third-party code that is NOT under
 * /root/reference (google/osqp-cpp, unpinned HEAD, fetched by
 * [REF] src/CMakeLists.txt:25-30; transitively osqp/osqp v0.6.x + QDLDL +
 * SuiteSparse-AMD).  None of it is in this container and the reference's own
 * tests never touch the solver ([REF] tests/test.cpp:1-2 include only the
 * constraint builder).  This file therefore restates the *published*
 * algorithm (Stellato et al., Math. Prog. Comp. 2020; osqp 0.6.x behaviour
 * as summarised in SURVEY.md section 8(a) rows S, E1-E14, X) and is pinned
 * by problem-intrinsic optimality checks (oracle/kkt_check.py) instead of
 * by reference outputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#ifndef OSQP_ORACLE_H
#define OSQP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef long long oq_int;   /* = Eigen StorageIndex `long long`, [REF] src/utils.h:12 */
typedef double    oq_float;

/* status values, osqp 0.6.x constants.h [EXT]; row X of SURVEY 8(a) */
enum {
  OQ_SOLVED                       = 1,
  OQ_SOLVED_INACCURATE            = 2,
  OQ_PRIMAL_INFEASIBLE_INACCURATE = 3,
  OQ_DUAL_INFEASIBLE_INACCURATE   = 4,
  OQ_MAX_ITER_REACHED             = -2,
  OQ_PRIMAL_INFEASIBLE            = -3,
  OQ_DUAL_INFEASIBLE              = -4,
  OQ_NON_CVX                      = -7,
  OQ_UNSOLVED                     = -10
};

/* settings in force at the reference boundary: only `verbose` is set there
 * ([REF] src/osqp-wrapper.h:26-27), so every field below defaults to the
 * osqp 0.6.x default (row S). */
typedef struct {
  oq_float rho;                    /* 0.1   */
  oq_float sigma;                  /* 1e-6  */
  oq_int   scaling;                /* 10    */
  oq_int   adaptive_rho;           /* 1     */
  oq_int   adaptive_rho_interval;  /* 0 = "auto"; resolved DETERMINISTICALLY to
                                      4*check_termination (upstream's
                                      non-PROFILING rule); upstream's PROFILING
                                      build uses wall-clock and is irreproducible */
  oq_float adaptive_rho_tolerance; /* 5     */
  oq_int   max_iter;               /* 4000  */
  oq_float eps_abs;                /* 1e-3  */
  oq_float eps_rel;                /* 1e-3  */
  oq_float eps_prim_inf;           /* 1e-4  */
  oq_float eps_dual_inf;           /* 1e-4  */
  oq_float alpha;                  /* 1.6   */
  oq_int   scaled_termination;     /* 0     */
  oq_int   check_termination;      /* 25    */
  oq_int   warm_start;             /* 1     */
} oq_settings;

typedef struct {
  oq_int   iter;
  oq_int   status_val;
  oq_float obj_val;
  oq_float pri_res;
  oq_float dua_res;
  oq_int   rho_updates;
  oq_float rho_estimate;
  oq_float rho;        /* rho in force after the solve */
  oq_int   nnz_L;      /* strictly-lower entries of the factor */
} oq_info;

typedef struct oq_work oq_work;

void oq_default_settings(oq_settings *s);

/* osqp_setup as driven by OsqpSolver::Init ([REF] src/osqp-wrapper.h:18-28).
 * P may hold both triangles (the reference's generator emits both,
 * [REF] src/utils.h:53-61); only the upper one is used.  q may be NULL
 * (= 0, [REF] src/osqp-wrapper.h:22).  Returns NULL on failure with *err:
 * 1 data validation, 2 settings validation, 4 non-convex (wrong inertia). */
oq_work *oq_setup(oq_int n, oq_int m,
                  const oq_int *Pp, const oq_int *Pi, const oq_float *Px,
                  const oq_float *q,
                  const oq_int *Ap, const oq_int *Ai, const oq_float *Ax,
                  const oq_float *l, const oq_float *u,
                  const oq_settings *settings, oq_int *err);

/* The same with a caller-supplied elimination order of the KKT matrix [[P + sigma I, A'], [A, -1/rho]] (kkt_perm[k] = natural
 * index, 0 .. n+m-1, eliminated k-th; NULL = the exact minimum degree of oq_setup): test infrastructure for problems whose
 * minimum-degree ordering would take minutes (BASELINE config 5 at its literal size).  The ordering changes round-off only. */
oq_work *oq_setup_ordered(oq_int n, oq_int m,
                          const oq_int *Pp, const oq_int *Pi, const oq_float *Px,
                          const oq_float *q,
                          const oq_int *Ap, const oq_int *Ai, const oq_float *Ax,
                          const oq_float *l, const oq_float *u,
                          const oq_settings *settings, const oq_int *kkt_perm, oq_int *err);

/* OsqpSolver::Solve ([REF] src/osqp-wrapper.h:52). Returns status_val. */
oq_int oq_solve(oq_work *w);

/* primal_solution()/dual_solution() ([REF] src/osqp-wrapper.h:53). NaN-filled
 * when the status carries no solution. */
void oq_get_solution(const oq_work *w, oq_float *x, oq_float *y);
void oq_get_info(const oq_work *w, oq_info *info);

/* UpdateConstraintMatrix ([REF] src/osqp-wrapper.h:36): same pattern, new
 * values.  Ap/Ai are compared with the stored pattern; mismatch -> 1. */
oq_int oq_update_A(oq_work *w, const oq_int *Ap, const oq_int *Ai,
                   const oq_float *Ax);
/* SetBounds ([REF] src/osqp-wrapper.h:40): l<=u required, else 1. */
oq_int oq_update_bounds(oq_work *w, const oq_float *l, const oq_float *u);
/* SetPrimalWarmStart ([REF] src/osqp-wrapper.h:46). */
oq_int oq_warm_start_x(oq_work *w, const oq_float *x);

void oq_cleanup(oq_work *w);

/* factor introspection for tests: perm[N], Lp[N+1], Li, Lx, Dinv[N] */
oq_int oq_kkt_dim(const oq_work *w);
void   oq_get_factor(const oq_work *w, oq_int *perm, oq_int *Lp, oq_int *Li,
                     oq_float *Lx, oq_float *Dinv);
/* one KKT solve  K sol = rhs  with the current factor (length n+m) */
void   oq_kkt_solve(const oq_work *w, const oq_float *rhs, oq_float *sol);

/* cpu_baseline helper: B independent QPs with ONE shared pattern and per-QP
 * values (arrays are [B][nnz] / [B][n] / [B][m], row-major).  Each QP runs
 * setup + solve exactly as above; `threads` OpenMP threads over the batch.
 * Outputs x[B][n], status[B], iters[B]; times in seconds (wall). */
oq_int oq_batch_solve(oq_int B, oq_int n, oq_int m,
                      const oq_int *Pp, const oq_int *Pi, const oq_float *Px,
                      const oq_float *q,
                      const oq_int *Ap, const oq_int *Ai, const oq_float *Ax,
                      const oq_float *l, const oq_float *u,
                      const oq_settings *settings, oq_int threads,
                      oq_float *x, oq_int *status, oq_int *iters,
                      oq_float *setup_seconds, oq_float *solve_seconds);

#ifdef __cplusplus
}
#endif
#endif
