/*
 * oracle/osqp_oracle.c -- TEST INFRASTRUCTURE ONLY.  See osqp_oracle.h.
 *
 * PARITY UNPINNED (header of osqp_oracle.h explains why).  Every function
 * names the reference call site it serves ([REF] = file under
 * /root/reference) and the upstream routine whose published behaviour it
 * restates ([EXT] = osqp 0.6.x / QDLDL, not in this container).
 *
 * Scalar, single-threaded per QP, exactly how upstream OSQP+QDLDL runs one
 * problem.  The fill-reducing ordering is an exact minimum-degree ordering
 * written here (SuiteSparse AMD is [EXT] and absent); any permutation gives
 * the same KKT solution up to round-off, so it does not affect parity.
 */
#include "osqp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OQ_INFTY        1e30   /* = INF of [REF] src/constraints/constraints.h:11 */
#define OQ_NAN          (NAN)
#define OQ_RHO_MIN      1e-6
#define OQ_RHO_MAX      1e6
#define OQ_RHO_EQ_OVER_INEQ 1e3
#define OQ_RHO_TOL      1e-4
#define OQ_MIN_SCALING  1e-4
#define OQ_MAX_SCALING  1e4
#define OQ_DIVISION_TOL (1.0 / OQ_INFTY)
#define OQ_ADAPTIVE_RHO_MULTIPLE_TERMINATION 4
#define OQ_ADAPTIVE_RHO_FIXED 100

#define MAXF(a, b) (((a) > (b)) ? (a) : (b))
#define MINF(a, b) (((a) < (b)) ? (a) : (b))

typedef struct {
  oq_int    nrow, ncol;
  oq_int   *p;   /* ncol+1 */
  oq_int   *i;   /* nnz    */
  oq_float *x;   /* nnz    */
} csc;

struct oq_work {
  oq_int n, m;
  csc P;              /* upper triangle, scaled in place */
  csc A;              /* scaled in place */
  oq_float *q, *l, *u;
  /* scaling (E2) */
  oq_float *D, *Dinv, *E, *Einv, c, cinv;
  oq_float *D_temp, *D_temp_A, *E_temp;
  /* rho (E3) */
  oq_float *rho_vec, *rho_inv_vec;
  oq_int   *constr_type;
  /* iterates */
  oq_float *x, *z, *y, *x_prev, *z_prev, *xz_tilde;
  oq_float *Ax, *Px, *Aty, *delta_x, *delta_y, *Adelta_x, *Pdelta_x, *Atdelta_y;
  oq_float *sol_x, *sol_y;
  /* linear system (E4, E5, E7) */
  oq_int N;
  csc K;              /* upper-triangular KKT, natural order */
  oq_int *PtoK, *AtoK, *rhotoK;
  oq_int *perm, *pinv;
  csc Kp;             /* permuted upper-triangular KKT */
  oq_int *KtoKp;
  oq_int *etree, *Lnz, *Lp, *Li;
  oq_float *Lx, *Dd, *Ddinv;
  oq_int *iwork; unsigned char *bwork; oq_float *fwork;
  oq_float *bp, *sol;
  oq_settings settings;
  oq_info info;
};

/* ---------------------------------------------------------------- utils */

static void *xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

static oq_float vec_norm_inf(const oq_float *v, oq_int n) {
  oq_float r = 0.0;
  for (oq_int k = 0; k < n; k++) { oq_float a = fabs(v[k]); if (a > r) r = a; }
  return r;
}
static oq_float vec_scaled_norm_inf(const oq_float *s, const oq_float *v, oq_int n) {
  oq_float r = 0.0;
  for (oq_int k = 0; k < n; k++) { oq_float a = fabs(s[k] * v[k]); if (a > r) r = a; }
  return r;
}
static oq_float vec_prod(const oq_float *a, const oq_float *b, oq_int n) {
  oq_float r = 0.0;
  for (oq_int k = 0; k < n; k++) r += a[k] * b[k];
  return r;
}

/* y (+)= A x ; plus_eq: 0 overwrite, 1 add, -1 subtract  [EXT mat_vec] */
static void mat_vec(const csc *A, const oq_float *x, oq_float *y, int plus_eq) {
  if (!plus_eq) for (oq_int r = 0; r < A->nrow; r++) y[r] = 0.0;
  if (A->p[A->ncol] == 0) return;
  for (oq_int j = 0; j < A->ncol; j++)
    for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) {
      if (plus_eq == -1) y[A->i[k]] -= A->x[k] * x[j];
      else               y[A->i[k]] += A->x[k] * x[j];
    }
}
/* y (+)= A' x ; skip_diag skips i==j entries  [EXT mat_tpose_vec] */
static void mat_tpose_vec(const csc *A, const oq_float *x, oq_float *y,
                          int plus_eq, int skip_diag) {
  if (!plus_eq) for (oq_int c = 0; c < A->ncol; c++) y[c] = 0.0;
  if (A->p[A->ncol] == 0) return;
  for (oq_int j = 0; j < A->ncol; j++)
    for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) {
      if (skip_diag && A->i[k] == j) continue;
      if (plus_eq == -1) y[j] -= A->x[k] * x[A->i[k]];
      else               y[j] += A->x[k] * x[A->i[k]];
    }
}
/* 1/2 x' P x with P upper triangular  [EXT quad_form] */
static oq_float quad_form(const csc *P, const oq_float *x) {
  oq_float r = 0.0;
  for (oq_int j = 0; j < P->ncol; j++)
    for (oq_int k = P->p[j]; k < P->p[j + 1]; k++) {
      oq_int i = P->i[k];
      if (i == j)      r += 0.5 * P->x[k] * x[i] * x[i];
      else if (i < j)  r += P->x[k] * x[i] * x[j];
    }
  return r;
}

static int csc_alloc(csc *M, oq_int nrow, oq_int ncol, oq_int nnz) {
  M->nrow = nrow; M->ncol = ncol;
  M->p = (oq_int *)xcalloc((size_t)ncol + 1, sizeof(oq_int));
  M->i = (oq_int *)xcalloc((size_t)nnz, sizeof(oq_int));
  M->x = (oq_float *)xcalloc((size_t)nnz, sizeof(oq_float));
  return (M->p && M->i && M->x) ? 0 : 1;
}
static void csc_free(csc *M) { free(M->p); free(M->i); free(M->x); M->p = M->i = NULL; M->x = NULL; }

/* ------------------------------------------------------------ scaling E2 */

static void limit_scaling(oq_float *v, oq_int n) {
  for (oq_int k = 0; k < n; k++) {
    v[k] = v[k] < OQ_MIN_SCALING ? 1.0 : v[k];
    v[k] = v[k] > OQ_MAX_SCALING ? OQ_MAX_SCALING : v[k];
  }
}
static void inf_norm_cols_sym_triu(const csc *P, oq_float *e) {
  for (oq_int j = 0; j < P->ncol; j++) e[j] = 0.0;
  for (oq_int j = 0; j < P->ncol; j++)
    for (oq_int k = P->p[j]; k < P->p[j + 1]; k++) {
      oq_int i = P->i[k]; oq_float a = fabs(P->x[k]);
      if (a > e[j]) e[j] = a;
      if (i != j && a > e[i]) e[i] = a;
    }
}
static void inf_norm_cols(const csc *A, oq_float *e) {
  for (oq_int j = 0; j < A->ncol; j++) {
    e[j] = 0.0;
    for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) { oq_float a = fabs(A->x[k]); if (a > e[j]) e[j] = a; }
  }
}
static void inf_norm_rows(const csc *A, oq_float *e) {
  for (oq_int r = 0; r < A->nrow; r++) e[r] = 0.0;
  for (oq_int j = 0; j < A->ncol; j++)
    for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) { oq_float a = fabs(A->x[k]); if (a > e[A->i[k]]) e[A->i[k]] = a; }
}
static void premult_diag(csc *A, const oq_float *d) {
  for (oq_int j = 0; j < A->ncol; j++) for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[A->i[k]];
}
static void postmult_diag(csc *A, const oq_float *d) {
  for (oq_int j = 0; j < A->ncol; j++) for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[j];
}

/* Ruiz equilibration + cost scaling.  [EXT scale_data], SURVEY row E2;
 * runs inside OsqpSolver::Init ([REF] src/osqp-wrapper.h:28) and again inside
 * UpdateConstraintMatrix ([REF] src/osqp-wrapper.h:36). */
static void scale_data(oq_work *w) {
  oq_int n = w->n, m = w->m;
  w->c = 1.0;
  for (oq_int k = 0; k < n; k++) { w->D[k] = 1.0; w->Dinv[k] = 1.0; }
  for (oq_int k = 0; k < m; k++) { w->E[k] = 1.0; w->Einv[k] = 1.0; }
  for (oq_int it = 0; it < w->settings.scaling; it++) {
    inf_norm_cols_sym_triu(&w->P, w->D_temp);
    inf_norm_cols(&w->A, w->D_temp_A);
    for (oq_int k = 0; k < n; k++) w->D_temp[k] = MAXF(w->D_temp[k], w->D_temp_A[k]);
    inf_norm_rows(&w->A, w->E_temp);
    limit_scaling(w->D_temp, n);
    limit_scaling(w->E_temp, m);
    for (oq_int k = 0; k < n; k++) w->D_temp[k] = 1.0 / sqrt(w->D_temp[k]);
    for (oq_int k = 0; k < m; k++) w->E_temp[k] = 1.0 / sqrt(w->E_temp[k]);
    premult_diag(&w->P, w->D_temp);  postmult_diag(&w->P, w->D_temp);
    premult_diag(&w->A, w->E_temp);  postmult_diag(&w->A, w->D_temp);
    for (oq_int k = 0; k < n; k++) w->q[k] *= w->D_temp[k];
    for (oq_int k = 0; k < n; k++) w->D[k] *= w->D_temp[k];
    for (oq_int k = 0; k < m; k++) w->E[k] *= w->E_temp[k];
    /* cost normalisation */
    inf_norm_cols_sym_triu(&w->P, w->D_temp);
    oq_float c_temp = 0.0;
    for (oq_int k = 0; k < n; k++) c_temp += w->D_temp[k];
    c_temp /= (oq_float)n;
    oq_float nq = vec_norm_inf(w->q, n);
    limit_scaling(&nq, 1);
    c_temp = MAXF(c_temp, nq);
    limit_scaling(&c_temp, 1);
    c_temp = 1.0 / c_temp;
    for (oq_int k = 0; k < w->P.p[n]; k++) w->P.x[k] *= c_temp;
    for (oq_int k = 0; k < n; k++) w->q[k] *= c_temp;
    w->c *= c_temp;
  }
  w->cinv = 1.0 / w->c;
  for (oq_int k = 0; k < n; k++) w->Dinv[k] = 1.0 / w->D[k];
  for (oq_int k = 0; k < m; k++) w->Einv[k] = 1.0 / w->E[k];
  for (oq_int k = 0; k < m; k++) { w->l[k] *= w->E[k]; w->u[k] *= w->E[k]; }
}

/* [EXT unscale_data] -- used by the A-update path (row E13). */
static void unscale_data(oq_work *w) {
  oq_int n = w->n, m = w->m;
  for (oq_int k = 0; k < w->P.p[n]; k++) w->P.x[k] *= w->cinv;
  premult_diag(&w->P, w->Dinv); postmult_diag(&w->P, w->Dinv);
  for (oq_int k = 0; k < n; k++) w->q[k] *= w->cinv * w->Dinv[k];
  premult_diag(&w->A, w->Einv); postmult_diag(&w->A, w->Dinv);
  for (oq_int k = 0; k < m; k++) { w->l[k] *= w->Einv[k]; w->u[k] *= w->Einv[k]; }
}

/* --------------------------------------------------------------- rho E3 */

/* [EXT set_rho_vec] on the SCALED bounds. */
static void set_rho_vec(oq_work *w) {
  w->settings.rho = MINF(MAXF(w->settings.rho, OQ_RHO_MIN), OQ_RHO_MAX);
  for (oq_int k = 0; k < w->m; k++) {
    if (w->l[k] < -OQ_INFTY * OQ_MIN_SCALING && w->u[k] > OQ_INFTY * OQ_MIN_SCALING) {
      w->constr_type[k] = -1; w->rho_vec[k] = OQ_RHO_MIN;
    } else if (w->u[k] - w->l[k] < OQ_RHO_TOL) {
      w->constr_type[k] = 1;  w->rho_vec[k] = OQ_RHO_EQ_OVER_INEQ * w->settings.rho;
    } else {
      w->constr_type[k] = 0;  w->rho_vec[k] = w->settings.rho;
    }
    w->rho_inv_vec[k] = 1.0 / w->rho_vec[k];
  }
}

/* ------------------------------------------------- ordering (stands in for AMD) */

typedef struct { oq_int len, cap; oq_int *v; } ilist;

static void ilist_push(ilist *L, oq_int a) {
  if (L->len == L->cap) { L->cap = L->cap ? 2 * L->cap : 8; L->v = (oq_int *)realloc(L->v, (size_t)L->cap * sizeof(oq_int)); }
  L->v[L->len++] = a;
}
static int cmp_int(const void *a, const void *b) {
  oq_int x = *(const oq_int *)a, y = *(const oq_int *)b; return (x > y) - (x < y);
}

/* Exact minimum-degree ordering of the graph of K+K' (smallest index breaks
 * ties).  perm[k] = natural index eliminated k-th.  Plays the role of
 * [EXT amd_l_order] in row E5. */
static void order_min_degree(oq_int N, const csc *K, oq_int *perm) {
  ilist *adj = (ilist *)xcalloc((size_t)N, sizeof(ilist));
  for (oq_int j = 0; j < N; j++)
    for (oq_int k = K->p[j]; k < K->p[j + 1]; k++) {
      oq_int i = K->i[k];
      if (i != j) { ilist_push(&adj[i], j); ilist_push(&adj[j], i); }
    }
  for (oq_int v = 0; v < N; v++) {           /* sort + unique */
    qsort(adj[v].v, (size_t)adj[v].len, sizeof(oq_int), cmp_int);
    oq_int o = 0;
    for (oq_int k = 0; k < adj[v].len; k++) if (!o || adj[v].v[o - 1] != adj[v].v[k]) adj[v].v[o++] = adj[v].v[k];
    adj[v].len = o;
  }
  unsigned char *dead = (unsigned char *)xcalloc((size_t)N, 1);
  oq_int *tmp = (oq_int *)xcalloc((size_t)N, sizeof(oq_int));
  oq_int alive = N;
  for (oq_int step = 0; step < N; ) {
    oq_int best = -1, bestdeg = N + 1;
    for (oq_int v = 0; v < N; v++) if (!dead[v] && adj[v].len < bestdeg) { best = v; bestdeg = adj[v].len; }
    if (bestdeg == alive - 1) {               /* remaining graph is a clique */
      for (oq_int v = 0; v < N; v++) if (!dead[v]) perm[step++] = v;
      break;
    }
    oq_int v = best;
    perm[step++] = v; dead[v] = 1; alive--;
    ilist S = adj[v];
    for (oq_int a = 0; a < S.len; a++) {
      oq_int u = S.v[a];
      /* adj[u] <- (adj[u] U S) \ {u, v}, both sorted */
      oq_int ia = 0, ib = 0, o = 0;
      ilist *Lu = &adj[u];
      while (ia < Lu->len || ib < S.len) {
        oq_int x;
        if (ib >= S.len || (ia < Lu->len && Lu->v[ia] < S.v[ib])) x = Lu->v[ia++];
        else if (ia >= Lu->len || S.v[ib] < Lu->v[ia]) x = S.v[ib++];
        else { x = Lu->v[ia]; ia++; ib++; }
        if (x != u && x != v) tmp[o++] = x;
      }
      if (o > Lu->cap) { Lu->cap = o + o / 2 + 4; Lu->v = (oq_int *)realloc(Lu->v, (size_t)Lu->cap * sizeof(oq_int)); }
      memcpy(Lu->v, tmp, (size_t)o * sizeof(oq_int));
      Lu->len = o;
    }
  }
  for (oq_int v = 0; v < N; v++) free(adj[v].v);
  free(adj); free(dead); free(tmp);
}

/* ---------------------------------------------------- KKT assembly E4 */

/* Upper-triangular CSC of [[P+sigma I, A'],[A, -diag(1/rho)]] with value
 * maps for in-place updates.  [EXT form_KKT]. */
static int form_KKT(oq_work *w) {
  oq_int n = w->n, m = w->m, N = n + m;
  const csc *P = &w->P, *A = &w->A;
  oq_int *cnt = (oq_int *)xcalloc((size_t)N + 1, sizeof(oq_int));
  for (oq_int j = 0; j < n; j++) {
    int has_diag = 0;
    for (oq_int k = P->p[j]; k < P->p[j + 1]; k++) { if (P->i[k] == j) has_diag = 1; cnt[j]++; }
    if (!has_diag) cnt[j]++;
  }
  for (oq_int j = 0; j < n; j++) for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) cnt[n + A->i[k]]++;
  for (oq_int r = 0; r < m; r++) cnt[n + r]++;
  oq_int nnz = 0;
  for (oq_int j = 0; j < N; j++) nnz += cnt[j];
  if (csc_alloc(&w->K, N, N, nnz)) { free(cnt); return 1; }
  w->PtoK = (oq_int *)xcalloc((size_t)P->p[n], sizeof(oq_int));
  w->AtoK = (oq_int *)xcalloc((size_t)A->p[n], sizeof(oq_int));
  w->rhotoK = (oq_int *)xcalloc((size_t)m, sizeof(oq_int));
  csc *K = &w->K;
  K->p[0] = 0;
  for (oq_int j = 0; j < N; j++) K->p[j + 1] = K->p[j] + cnt[j];
  oq_int *nxt = cnt;                      /* reuse as fill pointers */
  for (oq_int j = 0; j < N; j++) nxt[j] = K->p[j];
  for (oq_int j = 0; j < n; j++) {
    int has_diag = 0;
    for (oq_int k = P->p[j]; k < P->p[j + 1]; k++) {
      oq_int pos = nxt[j]++;
      K->i[pos] = P->i[k];
      K->x[pos] = P->x[k] + (P->i[k] == j ? w->settings.sigma : 0.0);
      if (P->i[k] == j) has_diag = 1;
      w->PtoK[k] = pos;
    }
    if (!has_diag) { oq_int pos = nxt[j]++; K->i[pos] = j; K->x[pos] = w->settings.sigma; }
  }
  for (oq_int j = 0; j < n; j++)
    for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) {
      oq_int pos = nxt[n + A->i[k]]++;
      K->i[pos] = j; K->x[pos] = A->x[k]; w->AtoK[k] = pos;
    }
  for (oq_int r = 0; r < m; r++) {
    oq_int pos = nxt[n + r]++;
    K->i[pos] = n + r; K->x[pos] = -w->rho_inv_vec[r]; w->rhotoK[r] = pos;
  }
  free(cnt);
  return 0;
}

/* [EXT update_KKT_P / update_KKT_A / update_KKT_param2] */
static void refresh_KKT_values(oq_work *w) {
  oq_int n = w->n, m = w->m;
  for (oq_int j = 0; j < n; j++)
    for (oq_int k = w->P.p[j]; k < w->P.p[j + 1]; k++)
      w->K.x[w->PtoK[k]] = w->P.x[k] + (w->P.i[k] == j ? w->settings.sigma : 0.0);
  for (oq_int k = 0; k < w->A.p[n]; k++) w->K.x[w->AtoK[k]] = w->A.x[k];
  for (oq_int r = 0; r < m; r++) w->K.x[w->rhotoK[r]] = -w->rho_inv_vec[r];
}

/* Kp = K(perm,perm), upper triangle kept.  [EXT csc_symperm]. */
static int permute_KKT(oq_work *w) {
  oq_int N = w->N; const csc *K = &w->K;
  oq_int *cnt = (oq_int *)xcalloc((size_t)N + 1, sizeof(oq_int));
  for (oq_int j = 0; j < N; j++)
    for (oq_int k = K->p[j]; k < K->p[j + 1]; k++) {
      oq_int i2 = w->pinv[K->i[k]], j2 = w->pinv[j];
      cnt[i2 > j2 ? i2 : j2]++;
    }
  if (csc_alloc(&w->Kp, N, N, K->p[N])) { free(cnt); return 1; }
  w->KtoKp = (oq_int *)xcalloc((size_t)K->p[N], sizeof(oq_int));
  w->Kp.p[0] = 0;
  for (oq_int j = 0; j < N; j++) w->Kp.p[j + 1] = w->Kp.p[j] + cnt[j];
  for (oq_int j = 0; j < N; j++) cnt[j] = w->Kp.p[j];
  for (oq_int j = 0; j < N; j++)
    for (oq_int k = K->p[j]; k < K->p[j + 1]; k++) {
      oq_int i2 = w->pinv[K->i[k]], j2 = w->pinv[j];
      oq_int c = i2 > j2 ? i2 : j2, r = i2 > j2 ? j2 : i2;
      oq_int pos = cnt[c]++;
      w->Kp.i[pos] = r; w->Kp.x[pos] = K->x[k]; w->KtoKp[k] = pos;
    }
  free(cnt);
  return 0;
}

/* ------------------------------------------------------ LDL' (QDLDL) E5 */

/* elimination tree + column counts of L.  [EXT QDLDL_etree]. Returns nnz(L) or -1. */
static oq_int ldl_etree(oq_int N, const csc *U, oq_int *work, oq_int *Lnz, oq_int *etree) {
  for (oq_int i = 0; i < N; i++) { work[i] = 0; Lnz[i] = 0; etree[i] = -1; if (U->p[i] == U->p[i + 1]) return -1; }
  for (oq_int j = 0; j < N; j++) {
    work[j] = j;
    for (oq_int p = U->p[j]; p < U->p[j + 1]; p++) {
      oq_int i = U->i[p];
      if (i > j) return -1;
      while (work[i] != j) {
        if (etree[i] == -1) etree[i] = j;
        Lnz[i]++;
        work[i] = j;
        i = etree[i];
      }
    }
  }
  oq_int s = 0;
  for (oq_int i = 0; i < N; i++) s += Lnz[i];
  return s;
}

/* up-looking numeric LDL'.  [EXT QDLDL_factor].  Returns #positive pivots or -1. */
static oq_int ldl_factor(oq_work *w) {
  oq_int N = w->N; const csc *U = &w->Kp;
  oq_int *Lp = w->Lp, *Li = w->Li; oq_float *Lx = w->Lx, *D = w->Dd, *Dinv = w->Ddinv;
  unsigned char *mark = w->bwork;
  oq_int *yIdx = w->iwork, *ebuf = w->iwork + N, *nextSpace = w->iwork + 2 * N;
  oq_float *yVals = w->fwork;
  oq_int positive = 0;
  Lp[0] = 0;
  for (oq_int i = 0; i < N; i++) {
    Lp[i + 1] = Lp[i] + w->Lnz[i];
    mark[i] = 0; yVals[i] = 0.0; D[i] = 0.0; nextSpace[i] = Lp[i];
  }
  for (oq_int k = 0; k < N; k++) {
    oq_int nnzY = 0;
    for (oq_int p = U->p[k]; p < U->p[k + 1]; p++) {
      oq_int b = U->i[p];
      if (b == k) { D[k] = U->x[p]; continue; }
      yVals[b] = U->x[p];
      oq_int nx = b;
      if (!mark[nx]) {
        mark[nx] = 1; ebuf[0] = nx; oq_int nE = 1;
        nx = w->etree[b];
        while (nx != -1 && nx < k) {
          if (mark[nx]) break;
          mark[nx] = 1; ebuf[nE++] = nx; nx = w->etree[nx];
        }
        while (nE) yIdx[nnzY++] = ebuf[--nE];
      }
    }
    for (oq_int t = nnzY - 1; t >= 0; t--) {
      oq_int c = yIdx[t];
      oq_int end = nextSpace[c];
      oq_float yc = yVals[c];
      for (oq_int j = Lp[c]; j < end; j++) yVals[Li[j]] -= Lx[j] * yc;
      Li[end] = k;
      Lx[end] = yc * Dinv[c];
      D[k] -= yc * Lx[end];
      nextSpace[c]++;
      yVals[c] = 0.0; mark[c] = 0;
    }
    if (D[k] == 0.0) return -1;
    if (D[k] > 0.0) positive++;
    Dinv[k] = 1.0 / D[k];
  }
  return positive;
}

/* K sol = b in place on the permuted vector.  [EXT QDLDL_solve], row E7. */
static void ldl_solve_inplace(const oq_work *w, oq_float *b) {
  oq_int N = w->N;
  for (oq_int i = 0; i < N; i++) {
    oq_float v = b[i];
    for (oq_int j = w->Lp[i]; j < w->Lp[i + 1]; j++) b[w->Li[j]] -= w->Lx[j] * v;
  }
  for (oq_int i = 0; i < N; i++) b[i] *= w->Ddinv[i];
  for (oq_int i = N - 1; i >= 0; i--) {
    oq_float v = b[i];
    for (oq_int j = w->Lp[i]; j < w->Lp[i + 1]; j++) v -= w->Lx[j] * b[w->Li[j]];
    b[i] = v;
  }
}

/* refresh permuted values and re-factor. Returns 0 ok, 4 wrong inertia/zero pivot */
static oq_int refactor(oq_work *w) {
  for (oq_int k = 0; k < w->K.p[w->N]; k++) w->Kp.x[w->KtoKp[k]] = w->K.x[k];
  oq_int pos = ldl_factor(w);
  if (pos < 0 || pos != w->n) return 4;
  return 0;
}

/* given_perm: a fill-reducing ordering supplied by the caller (perm[k] = natural KKT index eliminated k-th) instead of the
 * exact minimum degree below, whose cost is quadratic in the fill - the only way to set this checker up for a 4 10^5-row KKT
 * matrix in seconds.  Any permutation yields the same KKT solutions up to round-off ([EXT]: upstream takes AMD's). */
static oq_int init_linsys(oq_work *w, const oq_int *given_perm) {
  oq_int N = w->N = w->n + w->m;
  if (form_KKT(w)) return 1;
  w->perm = (oq_int *)xcalloc((size_t)N, sizeof(oq_int));
  w->pinv = (oq_int *)xcalloc((size_t)N, sizeof(oq_int));
  if (given_perm) {
    for (oq_int k = 0; k < N; k++) w->pinv[k] = -1;
    for (oq_int k = 0; k < N; k++) {
      if (given_perm[k] < 0 || given_perm[k] >= N || w->pinv[given_perm[k]] >= 0) return 1;      /* not a permutation */
      w->perm[k] = given_perm[k]; w->pinv[given_perm[k]] = k;
    }
  } else order_min_degree(N, &w->K, w->perm);
  for (oq_int k = 0; k < N; k++) w->pinv[w->perm[k]] = k;
  if (permute_KKT(w)) return 1;
  w->etree = (oq_int *)xcalloc((size_t)N, sizeof(oq_int));
  w->Lnz = (oq_int *)xcalloc((size_t)N, sizeof(oq_int));
  w->iwork = (oq_int *)xcalloc((size_t)3 * N, sizeof(oq_int));
  w->bwork = (unsigned char *)xcalloc((size_t)N, 1);
  w->fwork = (oq_float *)xcalloc((size_t)N, sizeof(oq_float));
  oq_int nnzL = ldl_etree(N, &w->Kp, w->iwork, w->Lnz, w->etree);
  if (nnzL < 0) return 1;
  w->info.nnz_L = nnzL;
  w->Lp = (oq_int *)xcalloc((size_t)N + 1, sizeof(oq_int));
  w->Li = (oq_int *)xcalloc((size_t)nnzL, sizeof(oq_int));
  w->Lx = (oq_float *)xcalloc((size_t)nnzL, sizeof(oq_float));
  w->Dd = (oq_float *)xcalloc((size_t)N, sizeof(oq_float));
  w->Ddinv = (oq_float *)xcalloc((size_t)N, sizeof(oq_float));
  w->bp = (oq_float *)xcalloc((size_t)N, sizeof(oq_float));
  w->sol = (oq_float *)xcalloc((size_t)N, sizeof(oq_float));
  oq_int pos = ldl_factor(w);
  if (pos < 0 || pos != w->n) return 4;
  return 0;
}

/* [EXT solve_linsys_qdldl]: sol = K^-1 b ; b[0:n] = x~ ; b[n:] += rho^-1 .* nu */
static void solve_linsys(oq_work *w, oq_float *b) {
  oq_int n = w->n, m = w->m, N = w->N;
  for (oq_int k = 0; k < N; k++) w->bp[k] = b[w->perm[k]];
  ldl_solve_inplace(w, w->bp);
  for (oq_int k = 0; k < N; k++) w->sol[w->perm[k]] = w->bp[k];
  for (oq_int j = 0; j < n; j++) b[j] = w->sol[j];
  for (oq_int j = 0; j < m; j++) b[n + j] += w->rho_inv_vec[j] * w->sol[n + j];
}

/* ------------------------------------------------------------- settings */

void oq_default_settings(oq_settings *s) {
  s->rho = 0.1; s->sigma = 1e-6; s->scaling = 10; s->adaptive_rho = 1;
  s->adaptive_rho_interval = 0; s->adaptive_rho_tolerance = 5.0;
  s->max_iter = 4000; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->alpha = 1.6;
  s->scaled_termination = 0; s->check_termination = 25; s->warm_start = 1;
}

static int validate_settings(const oq_settings *s) {
  if (s->scaling < 0) return 1;
  if (s->adaptive_rho != 0 && s->adaptive_rho != 1) return 1;
  if (s->adaptive_rho_interval < 0) return 1;
  if (s->adaptive_rho_tolerance < 1.0) return 1;
  if (s->max_iter <= 0) return 1;
  if (s->rho <= 0.0 || s->sigma <= 0.0) return 1;
  if (s->eps_abs < 0.0 || s->eps_rel < 0.0) return 1;
  if (s->eps_abs == 0.0 && s->eps_rel == 0.0) return 1;
  if (s->eps_prim_inf <= 0.0 || s->eps_dual_inf <= 0.0) return 1;
  if (s->alpha <= 0.0 || s->alpha >= 2.0) return 1;
  if (s->scaled_termination != 0 && s->scaled_termination != 1) return 1;
  if (s->check_termination < 0) return 1;
  if (s->warm_start != 0 && s->warm_start != 1) return 1;
  return 0;
}

/* ---------------------------------------------------------------- setup */

static void cold_start(oq_work *w) {
  memset(w->x, 0, (size_t)w->n * sizeof(oq_float));
  memset(w->z, 0, (size_t)w->m * sizeof(oq_float));
  memset(w->y, 0, (size_t)w->m * sizeof(oq_float));
}

static void reset_info(oq_info *info) { info->status_val = OQ_UNSOLVED; info->rho_updates = 0; }

oq_work *oq_setup(oq_int n, oq_int m,
                  const oq_int *Pp, const oq_int *Pi, const oq_float *Px,
                  const oq_float *q,
                  const oq_int *Ap, const oq_int *Ai, const oq_float *Ax,
                  const oq_float *l, const oq_float *u,
                  const oq_settings *settings, oq_int *err) {
  return oq_setup_ordered(n, m, Pp, Pi, Px, q, Ap, Ai, Ax, l, u, settings, NULL, err);
}

oq_work *oq_setup_ordered(oq_int n, oq_int m,
                          const oq_int *Pp, const oq_int *Pi, const oq_float *Px,
                          const oq_float *q,
                          const oq_int *Ap, const oq_int *Ai, const oq_float *Ax,
                          const oq_float *l, const oq_float *u,
                          const oq_settings *settings, const oq_int *kkt_perm, oq_int *err) {
  oq_int e_dummy; if (!err) err = &e_dummy; *err = 0;
  /* E1: validation ([EXT] osqp-cpp Init + validate_data) */
  if (n <= 0 || m < 0 || !Pp || !Ap || !settings) { *err = 1; return NULL; }
  if (validate_settings(settings)) { *err = 2; return NULL; }
  if (Pp[0] != 0 || Ap[0] != 0) { *err = 1; return NULL; }
  for (oq_int j = 0; j < n; j++) {
    if (Pp[j + 1] < Pp[j] || Ap[j + 1] < Ap[j]) { *err = 1; return NULL; }
    for (oq_int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] < 0 || Pi[k] >= n) { *err = 1; return NULL; }
    for (oq_int k = Ap[j]; k < Ap[j + 1]; k++) if (Ai[k] < 0 || Ai[k] >= m) { *err = 1; return NULL; }
  }
  for (oq_int k = 0; k < m; k++) if (l[k] > u[k]) { *err = 1; return NULL; }

  oq_work *w = (oq_work *)xcalloc(1, sizeof(oq_work));
  w->n = n; w->m = m; w->settings = *settings;
  /* P: keep the upper triangle only ([EXT] osqp-cpp triangularView<Upper>) */
  oq_int nnzP = 0;
  for (oq_int j = 0; j < n; j++) for (oq_int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] <= j) nnzP++;
  csc_alloc(&w->P, n, n, nnzP);
  nnzP = 0;
  for (oq_int j = 0; j < n; j++) {
    w->P.p[j] = nnzP;
    for (oq_int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] <= j) { w->P.i[nnzP] = Pi[k]; w->P.x[nnzP] = Px[k]; nnzP++; }
  }
  w->P.p[n] = nnzP;
  csc_alloc(&w->A, m, n, Ap[n]);
  memcpy(w->A.p, Ap, ((size_t)n + 1) * sizeof(oq_int));
  memcpy(w->A.i, Ai, (size_t)Ap[n] * sizeof(oq_int));
  memcpy(w->A.x, Ax, (size_t)Ap[n] * sizeof(oq_float));
#define VN(name, len) w->name = (oq_float *)xcalloc((size_t)(len), sizeof(oq_float))
  VN(q, n); VN(l, m); VN(u, m);
  VN(D, n); VN(Dinv, n); VN(E, m); VN(Einv, m); VN(D_temp, n); VN(D_temp_A, n); VN(E_temp, m);
  VN(rho_vec, m); VN(rho_inv_vec, m);
  VN(x, n); VN(z, m); VN(y, m); VN(x_prev, n); VN(z_prev, m); VN(xz_tilde, n + m);
  VN(Ax, m); VN(Px, n); VN(Aty, n); VN(delta_x, n); VN(delta_y, m);
  VN(Adelta_x, m); VN(Pdelta_x, n); VN(Atdelta_y, n); VN(sol_x, n); VN(sol_y, m);
#undef VN
  w->constr_type = (oq_int *)xcalloc((size_t)m, sizeof(oq_int));
  if (q) memcpy(w->q, q, (size_t)n * sizeof(oq_float));
  for (oq_int k = 0; k < m; k++) {          /* clip to +-INFTY ([EXT] osqp-cpp Init) */
    w->l[k] = MAXF(l[k], -OQ_INFTY); w->u[k] = MINF(u[k], OQ_INFTY);
  }
  for (oq_int k = 0; k < n; k++) { w->D[k] = w->Dinv[k] = 1.0; }
  for (oq_int k = 0; k < m; k++) { w->E[k] = w->Einv[k] = 1.0; }
  w->c = w->cinv = 1.0;
  if (w->settings.scaling) scale_data(w);
  set_rho_vec(w);
  oq_int rc = init_linsys(w, kkt_perm);
  if (rc) { *err = rc; oq_cleanup(w); return NULL; }
  w->info.status_val = OQ_UNSOLVED; w->info.iter = 0; w->info.rho_updates = 0;
  w->info.rho_estimate = w->settings.rho;
  /* deterministic resolution of the "auto" interval (see osqp_oracle.h) */
  if (w->settings.adaptive_rho && !w->settings.adaptive_rho_interval) {
    w->settings.adaptive_rho_interval = w->settings.check_termination
        ? OQ_ADAPTIVE_RHO_MULTIPLE_TERMINATION * w->settings.check_termination
        : OQ_ADAPTIVE_RHO_FIXED;
  }
  return w;
}

void oq_cleanup(oq_work *w) {
  if (!w) return;
  csc_free(&w->P); csc_free(&w->A); csc_free(&w->K); csc_free(&w->Kp);
  free(w->q); free(w->l); free(w->u); free(w->D); free(w->Dinv); free(w->E); free(w->Einv);
  free(w->D_temp); free(w->D_temp_A); free(w->E_temp); free(w->rho_vec); free(w->rho_inv_vec);
  free(w->constr_type); free(w->x); free(w->z); free(w->y); free(w->x_prev); free(w->z_prev);
  free(w->xz_tilde); free(w->Ax); free(w->Px); free(w->Aty); free(w->delta_x); free(w->delta_y);
  free(w->Adelta_x); free(w->Pdelta_x); free(w->Atdelta_y); free(w->sol_x); free(w->sol_y);
  free(w->PtoK); free(w->AtoK); free(w->rhotoK); free(w->perm); free(w->pinv); free(w->KtoKp);
  free(w->etree); free(w->Lnz); free(w->Lp); free(w->Li); free(w->Lx); free(w->Dd); free(w->Ddinv);
  free(w->iwork); free(w->bwork); free(w->fwork); free(w->bp); free(w->sol);
  free(w);
}

/* ------------------------------------------------------ ADMM steps E6-E10 */

static void update_xz_tilde(oq_work *w) {       /* [EXT compute_rhs + solve], rows E6,E7 */
  oq_int n = w->n, m = w->m;
  for (oq_int i = 0; i < n; i++) w->xz_tilde[i] = w->settings.sigma * w->x_prev[i] - w->q[i];
  for (oq_int i = 0; i < m; i++) w->xz_tilde[n + i] = w->z_prev[i] - w->rho_inv_vec[i] * w->y[i];
  solve_linsys(w, w->xz_tilde);
}
static void update_x(oq_work *w) {              /* row E8 */
  oq_float a = w->settings.alpha;
  for (oq_int i = 0; i < w->n; i++) w->x[i] = a * w->xz_tilde[i] + (1.0 - a) * w->x_prev[i];
  for (oq_int i = 0; i < w->n; i++) w->delta_x[i] = w->x[i] - w->x_prev[i];
}
static void update_z(oq_work *w) {              /* row E9 */
  oq_float a = w->settings.alpha; oq_int n = w->n;
  for (oq_int i = 0; i < w->m; i++) {
    w->z[i] = a * w->xz_tilde[n + i] + (1.0 - a) * w->z_prev[i] + w->rho_inv_vec[i] * w->y[i];
  }
  for (oq_int i = 0; i < w->m; i++) w->z[i] = MINF(MAXF(w->z[i], w->l[i]), w->u[i]);
}
static void update_y(oq_work *w) {              /* row E10 */
  oq_float a = w->settings.alpha; oq_int n = w->n;
  for (oq_int i = 0; i < w->m; i++) {
    w->delta_y[i] = w->rho_vec[i] * (a * w->xz_tilde[n + i] + (1.0 - a) * w->z_prev[i] - w->z[i]);
    w->y[i] += w->delta_y[i];
  }
}

/* ------------------------------------------------ residuals / info E11 */

static oq_float compute_obj_val(const oq_work *w, const oq_float *x) {
  oq_float v = quad_form(&w->P, x) + vec_prod(w->q, x, w->n);
  if (w->settings.scaling) v *= w->cinv;
  return v;
}
static oq_float compute_pri_res(oq_work *w) {
  mat_vec(&w->A, w->x, w->Ax, 0);
  for (oq_int i = 0; i < w->m; i++) w->z_prev[i] = w->Ax[i] - w->z[i];   /* z_prev is scratch here */
  if (w->settings.scaling && !w->settings.scaled_termination) return vec_scaled_norm_inf(w->Einv, w->z_prev, w->m);
  return vec_norm_inf(w->z_prev, w->m);
}
static oq_float compute_dua_res(oq_work *w) {
  oq_int n = w->n;
  memcpy(w->x_prev, w->q, (size_t)n * sizeof(oq_float));               /* x_prev is scratch here */
  mat_vec(&w->P, w->x, w->Px, 0);
  mat_tpose_vec(&w->P, w->x, w->Px, 1, 1);
  for (oq_int i = 0; i < n; i++) w->x_prev[i] += w->Px[i];
  if (w->m > 0) {
    mat_tpose_vec(&w->A, w->y, w->Aty, 0, 0);
    for (oq_int i = 0; i < n; i++) w->x_prev[i] += w->Aty[i];
  }
  if (w->settings.scaling && !w->settings.scaled_termination) return w->cinv * vec_scaled_norm_inf(w->Dinv, w->x_prev, n);
  return vec_norm_inf(w->x_prev, n);
}
static void update_info(oq_work *w, oq_int iter) {
  w->info.iter = iter;
  w->info.obj_val = compute_obj_val(w, w->x);
  w->info.pri_res = (w->m == 0) ? 0.0 : compute_pri_res(w);
  w->info.dua_res = compute_dua_res(w);
}

/* ------------------------------------------------------ termination E12 */

static oq_float compute_pri_tol(const oq_work *w, oq_float eps_abs, oq_float eps_rel) {
  oq_float mx;
  if (w->settings.scaling && !w->settings.scaled_termination) {
    mx = vec_scaled_norm_inf(w->Einv, w->z, w->m);
    mx = MAXF(mx, vec_scaled_norm_inf(w->Einv, w->Ax, w->m));
  } else {
    mx = MAXF(vec_norm_inf(w->z, w->m), vec_norm_inf(w->Ax, w->m));
  }
  return eps_abs + eps_rel * mx;
}
static oq_float compute_dua_tol(const oq_work *w, oq_float eps_abs, oq_float eps_rel) {
  oq_float mx; oq_int n = w->n;
  if (w->settings.scaling && !w->settings.scaled_termination) {
    mx = vec_scaled_norm_inf(w->Dinv, w->q, n);
    mx = MAXF(mx, vec_scaled_norm_inf(w->Dinv, w->Aty, n));
    mx = MAXF(mx, vec_scaled_norm_inf(w->Dinv, w->Px, n));
    mx *= w->cinv;
  } else {
    mx = vec_norm_inf(w->q, n);
    mx = MAXF(mx, vec_norm_inf(w->Aty, n));
    mx = MAXF(mx, vec_norm_inf(w->Px, n));
  }
  return eps_abs + eps_rel * mx;
}

static int is_primal_infeasible(oq_work *w, oq_float eps) {
  oq_int m = w->m;
  for (oq_int i = 0; i < m; i++) {
    if (w->u[i] > OQ_INFTY * OQ_MIN_SCALING) {
      if (w->l[i] < -OQ_INFTY * OQ_MIN_SCALING) w->delta_y[i] = 0.0;
      else w->delta_y[i] = MINF(w->delta_y[i], 0.0);
    } else if (w->l[i] < -OQ_INFTY * OQ_MIN_SCALING) {
      w->delta_y[i] = MAXF(w->delta_y[i], 0.0);
    }
  }
  oq_float norm_dy;
  if (w->settings.scaling && !w->settings.scaled_termination) {
    for (oq_int i = 0; i < m; i++) w->Adelta_x[i] = w->E[i] * w->delta_y[i];
    norm_dy = vec_norm_inf(w->Adelta_x, m);
  } else norm_dy = vec_norm_inf(w->delta_y, m);
  if (norm_dy > OQ_DIVISION_TOL) {
    oq_float lhs = 0.0;
    for (oq_int i = 0; i < m; i++)
      lhs += w->u[i] * MAXF(w->delta_y[i], 0.0) + w->l[i] * MINF(w->delta_y[i], 0.0);
    if (lhs < -eps * norm_dy) {
      mat_tpose_vec(&w->A, w->delta_y, w->Atdelta_y, 0, 0);
      if (w->settings.scaling && !w->settings.scaled_termination)
        for (oq_int i = 0; i < w->n; i++) w->Atdelta_y[i] *= w->Dinv[i];
      return vec_norm_inf(w->Atdelta_y, w->n) < eps * norm_dy;
    }
  }
  return 0;
}

static int is_dual_infeasible(oq_work *w, oq_float eps) {
  oq_int n = w->n, m = w->m;
  oq_float norm_dx, cost_scaling;
  if (w->settings.scaling && !w->settings.scaled_termination) {
    norm_dx = vec_scaled_norm_inf(w->D, w->delta_x, n); cost_scaling = w->c;
  } else { norm_dx = vec_norm_inf(w->delta_x, n); cost_scaling = 1.0; }
  if (norm_dx > OQ_DIVISION_TOL) {
    if (vec_prod(w->q, w->delta_x, n) < -cost_scaling * eps * norm_dx) {
      mat_vec(&w->P, w->delta_x, w->Pdelta_x, 0);
      mat_tpose_vec(&w->P, w->delta_x, w->Pdelta_x, 1, 1);
      if (w->settings.scaling && !w->settings.scaled_termination)
        for (oq_int i = 0; i < n; i++) w->Pdelta_x[i] *= w->Dinv[i];
      if (vec_norm_inf(w->Pdelta_x, n) < cost_scaling * eps * norm_dx) {
        mat_vec(&w->A, w->delta_x, w->Adelta_x, 0);
        if (w->settings.scaling && !w->settings.scaled_termination)
          for (oq_int i = 0; i < m; i++) w->Adelta_x[i] *= w->Einv[i];
        for (oq_int i = 0; i < m; i++) {
          if ((w->u[i] < OQ_INFTY * OQ_MIN_SCALING && w->Adelta_x[i] > eps * norm_dx) ||
              (w->l[i] > -OQ_INFTY * OQ_MIN_SCALING && w->Adelta_x[i] < -eps * norm_dx)) return 0;
        }
        return 1;
      }
    }
  }
  return 0;
}

static int check_termination(oq_work *w, int approximate) {
  oq_float eps_abs = w->settings.eps_abs, eps_rel = w->settings.eps_rel;
  oq_float eps_pinf = w->settings.eps_prim_inf, eps_dinf = w->settings.eps_dual_inf;
  int prim_res_check = 0, dual_res_check = 0, prim_inf_check = 0, dual_inf_check = 0;
  if (w->info.pri_res > OQ_INFTY || w->info.dua_res > OQ_INFTY) {
    w->info.status_val = OQ_NON_CVX; w->info.obj_val = OQ_NAN; return 1;
  }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; eps_pinf *= 10; eps_dinf *= 10; }
  if (w->m == 0) prim_res_check = 1;
  else {
    oq_float eps_prim = compute_pri_tol(w, eps_abs, eps_rel);
    if (w->info.pri_res < eps_prim) prim_res_check = 1;
    else prim_inf_check = is_primal_infeasible(w, eps_pinf);
  }
  oq_float eps_dual = compute_dua_tol(w, eps_abs, eps_rel);
  if (w->info.dua_res < eps_dual) dual_res_check = 1;
  else dual_inf_check = is_dual_infeasible(w, eps_dinf);

  if (prim_res_check && dual_res_check) {
    w->info.status_val = approximate ? OQ_SOLVED_INACCURATE : OQ_SOLVED; return 1;
  } else if (prim_inf_check) {
    w->info.status_val = approximate ? OQ_PRIMAL_INFEASIBLE_INACCURATE : OQ_PRIMAL_INFEASIBLE;
    if (w->settings.scaling && !w->settings.scaled_termination)
      for (oq_int i = 0; i < w->m; i++) w->delta_y[i] *= w->E[i];
    w->info.obj_val = OQ_INFTY; return 1;
  } else if (dual_inf_check) {
    w->info.status_val = approximate ? OQ_DUAL_INFEASIBLE_INACCURATE : OQ_DUAL_INFEASIBLE;
    if (w->settings.scaling && !w->settings.scaled_termination)
      for (oq_int i = 0; i < w->n; i++) w->delta_x[i] *= w->D[i];
    w->info.obj_val = -OQ_INFTY; return 1;
  }
  return 0;
}

/* -------------------------------------------------------- adaptive rho E13 */

static oq_float compute_rho_estimate(const oq_work *w) {
  oq_int n = w->n, m = w->m;
  oq_float pri = vec_norm_inf(w->z_prev, m);      /* scaled residual vectors left by update_info */
  oq_float dua = vec_norm_inf(w->x_prev, n);
  oq_float pn = MAXF(vec_norm_inf(w->z, m), vec_norm_inf(w->Ax, m));
  pri /= (pn + OQ_DIVISION_TOL);
  oq_float dn = MAXF(vec_norm_inf(w->q, n), vec_norm_inf(w->Aty, n));
  dn = MAXF(dn, vec_norm_inf(w->Px, n));
  dua /= (dn + OQ_DIVISION_TOL);
  oq_float est = w->settings.rho * sqrt(pri / dua);
  est = MINF(MAXF(est, OQ_RHO_MIN), OQ_RHO_MAX);
  return est;
}

static oq_int update_rho(oq_work *w, oq_float rho_new) {   /* [EXT osqp_update_rho] */
  w->settings.rho = MINF(MAXF(rho_new, OQ_RHO_MIN), OQ_RHO_MAX);
  for (oq_int i = 0; i < w->m; i++) {
    if (w->constr_type[i] == 0) { w->rho_vec[i] = w->settings.rho; w->rho_inv_vec[i] = 1.0 / w->settings.rho; }
    else if (w->constr_type[i] == 1) { w->rho_vec[i] = OQ_RHO_EQ_OVER_INEQ * w->settings.rho; w->rho_inv_vec[i] = 1.0 / w->rho_vec[i]; }
  }
  for (oq_int r = 0; r < w->m; r++) w->K.x[w->rhotoK[r]] = -w->rho_inv_vec[r];
  return refactor(w);
}

static oq_int adapt_rho(oq_work *w) {
  oq_float rho_new = compute_rho_estimate(w);
  w->info.rho_estimate = rho_new;
  if (rho_new > w->settings.rho * w->settings.adaptive_rho_tolerance ||
      rho_new < w->settings.rho / w->settings.adaptive_rho_tolerance) {
    oq_int rc = update_rho(w, rho_new);
    w->info.rho_updates += 1;
    return rc;
  }
  return 0;
}

/* ------------------------------------------------------------ solve (R2) */

static int has_solution(const oq_info *info) {
  return info->status_val != OQ_PRIMAL_INFEASIBLE && info->status_val != OQ_PRIMAL_INFEASIBLE_INACCURATE &&
         info->status_val != OQ_DUAL_INFEASIBLE && info->status_val != OQ_DUAL_INFEASIBLE_INACCURATE &&
         info->status_val != OQ_NON_CVX;
}

static void store_solution(oq_work *w) {        /* row E14 */
  if (has_solution(&w->info)) {
    for (oq_int i = 0; i < w->n; i++) w->sol_x[i] = w->x[i];
    for (oq_int i = 0; i < w->m; i++) w->sol_y[i] = w->y[i];
    if (w->settings.scaling) {
      for (oq_int i = 0; i < w->n; i++) w->sol_x[i] *= w->D[i];
      for (oq_int i = 0; i < w->m; i++) w->sol_y[i] *= w->E[i] * w->cinv;
    }
  } else {
    for (oq_int i = 0; i < w->n; i++) w->sol_x[i] = OQ_NAN;
    for (oq_int i = 0; i < w->m; i++) w->sol_y[i] = OQ_NAN;
    cold_start(w);
  }
}

/* [EXT osqp_solve] as reached from QPSolver::solve ([REF] src/osqp-wrapper.h:51-54). */
oq_int oq_solve(oq_work *w) {
  oq_int iter; int can_check = 0;
  w->info.status_val = OQ_UNSOLVED;   /* a fresh Solve() starts unsolved (update_* reset it upstream too) */
  if (!w->settings.warm_start) cold_start(w);
  for (iter = 1; iter <= w->settings.max_iter; iter++) {
    oq_float *t;
    t = w->x; w->x = w->x_prev; w->x_prev = t;
    t = w->z; w->z = w->z_prev; w->z_prev = t;
    update_xz_tilde(w);
    update_x(w);
    update_z(w);
    update_y(w);
    can_check = w->settings.check_termination && (iter % w->settings.check_termination == 0);
    if (can_check) {
      update_info(w, iter);
      if (check_termination(w, 0)) break;
    }
    if (w->settings.adaptive_rho && w->settings.adaptive_rho_interval &&
        (iter % w->settings.adaptive_rho_interval == 0)) {
      if (!can_check) update_info(w, iter);
      if (adapt_rho(w)) { w->info.status_val = OQ_NON_CVX; break; }
    }
  }
  if (!can_check) {
    update_info(w, iter - 1);
    check_termination(w, 0);
  }
  if (w->info.status_val == OQ_UNSOLVED) {
    if (!check_termination(w, 1)) w->info.status_val = OQ_MAX_ITER_REACHED;
  }
  if (has_solution(&w->info)) w->info.obj_val = compute_obj_val(w, w->x);
  w->info.rho_estimate = compute_rho_estimate(w);
  store_solution(w);
  w->info.rho = w->settings.rho;
  return w->info.status_val;
}

void oq_get_solution(const oq_work *w, oq_float *x, oq_float *y) {
  if (x) memcpy(x, w->sol_x, (size_t)w->n * sizeof(oq_float));
  if (y) memcpy(y, w->sol_y, (size_t)w->m * sizeof(oq_float));
}
void oq_get_info(const oq_work *w, oq_info *info) { *info = w->info; info->rho = w->settings.rho; }

/* --------------------------------------------------------- updates R3, R4 */

oq_int oq_update_A(oq_work *w, const oq_int *Ap, const oq_int *Ai, const oq_float *Ax) {
  oq_int n = w->n;
  /* [EXT] osqp-cpp VerifySameSparsity -> InvalidArgument -> the reference throws
   * ([REF] src/osqp-wrapper.h:36-38) */
  for (oq_int j = 0; j <= n; j++) if (Ap[j] != w->A.p[j]) return 1;
  for (oq_int k = 0; k < Ap[n]; k++) if (Ai[k] != w->A.i[k]) return 1;
  if (w->settings.scaling) unscale_data(w);
  memcpy(w->A.x, Ax, (size_t)Ap[n] * sizeof(oq_float));
  if (w->settings.scaling) scale_data(w);
  refresh_KKT_values(w);
  reset_info(&w->info);
  return refactor(w) ? 4 : 0;
}

oq_int oq_update_bounds(oq_work *w, const oq_float *l, const oq_float *u) {
  oq_int m = w->m;
  for (oq_int i = 0; i < m; i++) if (l[i] > u[i]) return 1;
  for (oq_int i = 0; i < m; i++) { w->l[i] = MAXF(l[i], -OQ_INFTY); w->u[i] = MINF(u[i], OQ_INFTY); }
  if (w->settings.scaling) for (oq_int i = 0; i < m; i++) { w->l[i] *= w->E[i]; w->u[i] *= w->E[i]; }
  reset_info(&w->info);
  /* [EXT update_rho_vec]: re-derive constraint types, refactor if any changed */
  int changed = 0;
  for (oq_int i = 0; i < m; i++) {
    if (w->l[i] < -OQ_INFTY * OQ_MIN_SCALING && w->u[i] > OQ_INFTY * OQ_MIN_SCALING) {
      if (w->constr_type[i] != -1) { w->constr_type[i] = -1; w->rho_vec[i] = OQ_RHO_MIN; w->rho_inv_vec[i] = 1.0 / OQ_RHO_MIN; changed = 1; }
    } else if (w->u[i] - w->l[i] < OQ_RHO_TOL) {
      if (w->constr_type[i] != 1) { w->constr_type[i] = 1; w->rho_vec[i] = OQ_RHO_EQ_OVER_INEQ * w->settings.rho; w->rho_inv_vec[i] = 1.0 / w->rho_vec[i]; changed = 1; }
    } else {
      if (w->constr_type[i] != 0) { w->constr_type[i] = 0; w->rho_vec[i] = w->settings.rho; w->rho_inv_vec[i] = 1.0 / w->settings.rho; changed = 1; }
    }
  }
  if (changed) {
    for (oq_int r = 0; r < m; r++) w->K.x[w->rhotoK[r]] = -w->rho_inv_vec[r];
    return refactor(w) ? 4 : 0;
  }
  return 0;
}

oq_int oq_warm_start_x(oq_work *w, const oq_float *x) {   /* [EXT osqp_warm_start_x], row E14 */
  w->settings.warm_start = 1;
  for (oq_int i = 0; i < w->n; i++) w->x[i] = x[i];
  if (w->settings.scaling) for (oq_int i = 0; i < w->n; i++) w->x[i] *= w->Dinv[i];
  mat_vec(&w->A, w->x, w->z, 0);
  return 0;
}

/* ----------------------------------------------------------- introspection */

oq_int oq_kkt_dim(const oq_work *w) { return w->N; }
void oq_get_factor(const oq_work *w, oq_int *perm, oq_int *Lp, oq_int *Li, oq_float *Lx, oq_float *Dinv) {
  oq_int N = w->N;
  if (perm) memcpy(perm, w->perm, (size_t)N * sizeof(oq_int));
  if (Lp) memcpy(Lp, w->Lp, ((size_t)N + 1) * sizeof(oq_int));
  if (Li) memcpy(Li, w->Li, (size_t)w->Lp[N] * sizeof(oq_int));
  if (Lx) memcpy(Lx, w->Lx, (size_t)w->Lp[N] * sizeof(oq_float));
  if (Dinv) memcpy(Dinv, w->Ddinv, (size_t)N * sizeof(oq_float));
}
void oq_kkt_solve(const oq_work *w, const oq_float *rhs, oq_float *sol) {
  oq_int N = w->N;
  oq_float *bp = (oq_float *)xcalloc((size_t)N, sizeof(oq_float));
  for (oq_int k = 0; k < N; k++) bp[k] = rhs[w->perm[k]];
  ldl_solve_inplace(w, bp);
  for (oq_int k = 0; k < N; k++) sol[w->perm[k]] = bp[k];
  free(bp);
}

/* ------------------------------------------------------------ batch driver */

static double now_s(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

oq_int oq_batch_solve(oq_int B, oq_int n, oq_int m,
                      const oq_int *Pp, const oq_int *Pi, const oq_float *Px,
                      const oq_float *q,
                      const oq_int *Ap, const oq_int *Ai, const oq_float *Ax,
                      const oq_float *l, const oq_float *u,
                      const oq_settings *settings, oq_int threads,
                      oq_float *x, oq_int *status, oq_int *iters,
                      oq_float *setup_seconds, oq_float *solve_seconds) {
  oq_int nnzP = Pp[n], nnzA = Ap[n];
  oq_work **ws = (oq_work **)xcalloc((size_t)B, sizeof(oq_work *));
  oq_int fail = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads((int)threads);
#else
  (void)threads;
#endif
  double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : fail)
  for (oq_int b = 0; b < B; b++) {
    oq_int err = 0;
    ws[b] = oq_setup(n, m, Pp, Pi, Px + b * nnzP, q ? q + b * n : NULL,
                     Ap, Ai, Ax + b * nnzA, l + b * m, u + b * m, settings, &err);
    if (!ws[b]) fail += 1;
  }
  double t1 = now_s();
#pragma omp parallel for schedule(dynamic, 1)
  for (oq_int b = 0; b < B; b++) {
    if (!ws[b]) { status[b] = OQ_UNSOLVED; iters[b] = 0; continue; }
    status[b] = oq_solve(ws[b]);
    iters[b] = ws[b]->info.iter;
    oq_get_solution(ws[b], x + b * n, NULL);
  }
  double t2 = now_s();
  for (oq_int b = 0; b < B; b++) oq_cleanup(ws[b]);
  free(ws);
  if (setup_seconds) *setup_seconds = t1 - t0;
  if (solve_seconds) *solve_seconds = t2 - t1;
  return fail;
}
