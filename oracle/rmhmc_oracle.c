/*
 * rmhmc_oracle.c — CPU restatement of the RMHMC hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP library: a plain-C restatement of
 * emilemathieu/RiemannHamiltonianMonteCarlo  code/rmhmc.py:13-201  exporting the
 * C-ABI of include/rmhmc.h.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product path never does.
 *
 * Pinning: the oracle is checked in tests/test_oracle_golden.py against golden
 * vectors captured from the reference itself (tests/golden/make_golden.py imports
 * /root/reference/code/rmhmc.py under sys.settrace and records its locals).
 *
 * Two algorithm variants, selected by RMHMC_FLAG_ORACLE_LITERAL:
 *   literal      forms the DxDxD tensor InvGdG[d] = G^-1 dG/dw_d and uses LU
 *                inverse / LU solve exactly where rmhmc.py uses np.linalg.inv /
 *                np.linalg.solve; recomputes the set-up block every transition.
 *   matrix-free  never forms the tensor: tr(G^-1 dG_d) = sum_n c_n h_n x_nd with
 *                h_n = x_n' G^-1 x_n, and u' dG_d u = sum_n c_n (x_n'u)^2 x_nd,
 *                c = v(1-2p); Cholesky instead of LU; caches the point record
 *                (G, chol, G^-1, grad, trace) instead of recomputing it.  This is
 *                the algorithm the HIP kernels implement.
 *
 * Build:  make -C oracle      (gcc -O2 -fopenmp -shared -fPIC)
 */
#include "../include/rmhmc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI2 6.283185307179586476925286766559

struct rmhmc_ctx {
  int64_t M;
  int32_t D;
  int64_t n;
  uint32_t flags;
  double alpha;
  double *X; /* M*D row-major copy */
  double *t; /* M */
  int have_data;
  char err[256];
  /* stateful chain API */
  int chains_ready;
  uint64_t seed;
  int64_t chain_offset;
  int32_t L, K;
  double eps;
  struct chain *chains;
};

static char g_err[256] = "";

static int fail(rmhmc_ctx *ctx, int code, const char *msg) {
  snprintf(ctx ? ctx->err : g_err, 256, "%s", msg);
  return code;
}

const char *rmhmc_version(void) { return "rmhmc-oracle-cpu 0.1 (C restatement of code/rmhmc.py)"; }
const char *rmhmc_last_error(const rmhmc_ctx *ctx) { return ctx ? ctx->err : g_err; }

int rmhmc_device_info(rmhmc_ctx *ctx, char *buf, size_t len) {
  int th = 1;
#ifdef _OPENMP
  th = omp_get_max_threads();
#endif
  (void)ctx;
  snprintf(buf, len, "cpu oracle, %d OpenMP threads", th);
  return RMHMC_OK;
}

int rmhmc_create(rmhmc_ctx **out, int32_t device_id, int64_t M, int32_t D, int64_t n_chains,
                 int32_t dtype, uint32_t flags) {
  (void)device_id;
  if (!out || M <= 0 || D <= 0 || n_chains <= 0) return fail(NULL, RMHMC_ERR_INVALID, "bad shape");
  if (dtype != RMHMC_F64) return fail(NULL, RMHMC_ERR_UNSUPPORTED, "only float64");
  rmhmc_ctx *c = (rmhmc_ctx *)calloc(1, sizeof(*c));
  if (!c) return fail(NULL, RMHMC_ERR_NOMEM, "oom");
  c->M = M; c->D = D; c->n = n_chains; c->flags = flags; c->alpha = 100.0;
  c->X = (double *)malloc(sizeof(double) * M * D);
  c->t = (double *)malloc(sizeof(double) * M);
  if (!c->X || !c->t) { free(c->X); free(c->t); free(c); return fail(NULL, RMHMC_ERR_NOMEM, "oom"); }
  *out = c;
  return RMHMC_OK;
}

int rmhmc_set_data(rmhmc_ctx *ctx, const double *X, const double *t, double alpha) {
  if (!ctx || !X || !t || !(alpha > 0)) return fail(ctx, RMHMC_ERR_INVALID, "set_data: bad argument");
  memcpy(ctx->X, X, sizeof(double) * ctx->M * ctx->D);
  memcpy(ctx->t, t, sizeof(double) * ctx->M);
  ctx->alpha = alpha;
  ctx->have_data = 1;
  ctx->chains_ready = 0;
  return RMHMC_OK;
}

/* ------------------------------------------------------------------------ */
/* small dense helpers                                                      */
/* ------------------------------------------------------------------------ */

/* lower Cholesky, A (D*D, full symmetric) -> L (lower, upper part zeroed).
 * np.linalg.cholesky, rmhmc.py:60,171.  Returns 0 ok, 1 if a pivot is <=0/NaN
 * (L is then filled with NaN so that every derived quantity is NaN => reject). */
static int chol_lower(int D, const double *A, double *L) {
  memset(L, 0, sizeof(double) * D * D);
  for (int j = 0; j < D; j++) {
    double s = A[j * D + j];
    for (int k = 0; k < j; k++) s -= L[j * D + k] * L[j * D + k];
    if (!(s > 0.0)) {
      for (int i = 0; i < D * D; i++) L[i] = NAN;
      return 1;
    }
    double ljj = sqrt(s);
    L[j * D + j] = ljj;
    for (int i = j + 1; i < D; i++) {
      double a = A[i * D + j];
      for (int k = 0; k < j; k++) a -= L[i * D + k] * L[j * D + k];
      L[i * D + j] = a / ljj;
    }
  }
  return 0;
}

/* x = (L L')^-1 b */
static void chol_solve(int D, const double *L, const double *b, double *x) {
  for (int i = 0; i < D; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * D + k] * x[k];
    x[i] = s / L[i * D + i];
  }
  for (int i = D - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < D; k++) s -= L[k * D + i] * x[k];
    x[i] = s / L[i * D + i];
  }
}

/* Ginv = (L L')^-1, full symmetric */
static void chol_inverse(int D, const double *L, double *Ginv, double *tmp /* D */) {
  double *e = tmp;
  for (int j = 0; j < D; j++) {
    memset(e, 0, sizeof(double) * D);
    e[j] = 1.0;
    chol_solve(D, L, e, &Ginv[j * D]); /* column j == row j by symmetry */
  }
  for (int i = 0; i < D; i++)
    for (int j = i + 1; j < D; j++) {
      double m = 0.5 * (Ginv[i * D + j] + Ginv[j * D + i]);
      Ginv[i * D + j] = Ginv[j * D + i] = m;
    }
}

static double half_logdet(int D, const double *L) {
  double s = 0;
  for (int i = 0; i < D; i++) s += log(L[i * D + i]);
  return s;
}

/* LU with partial pivoting (dgetrf-style), in place; piv[D]. returns 1 if singular */
static int lu_factor(int D, double *A, int *piv) {
  int sing = 0;
  for (int k = 0; k < D; k++) {
    int p = k;
    double best = fabs(A[k * D + k]);
    for (int i = k + 1; i < D; i++)
      if (fabs(A[i * D + k]) > best) { best = fabs(A[i * D + k]); p = i; }
    piv[k] = p;
    if (p != k)
      for (int j = 0; j < D; j++) { double tmp = A[k * D + j]; A[k * D + j] = A[p * D + j]; A[p * D + j] = tmp; }
    double d = A[k * D + k];
    if (d == 0.0 || d != d) { sing = 1; continue; }
    for (int i = k + 1; i < D; i++) {
      double m = A[i * D + k] / d;
      A[i * D + k] = m;
      for (int j = k + 1; j < D; j++) A[i * D + j] -= m * A[k * D + j];
    }
  }
  return sing;
}
static void lu_solve(int D, const double *LU, const int *piv, double *b) {
  for (int k = 0; k < D; k++) { int p = piv[k]; if (p != k) { double tmp = b[k]; b[k] = b[p]; b[p] = tmp; } }
  for (int i = 0; i < D; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= LU[i * D + k] * b[k]; b[i] = s; }
  for (int i = D - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < D; k++) s -= LU[i * D + k] * b[k]; b[i] = s / LU[i * D + i]; }
}
/* np.linalg.solve(G, b)  (rmhmc.py:113,121): dgesv */
static void lu_solve_copy(int D, const double *G, const double *b, double *x, double *work /* D*D */, int *piv) {
  memcpy(work, G, sizeof(double) * D * D);
  lu_factor(D, work, piv);
  memcpy(x, b, sizeof(double) * D);
  lu_solve(D, work, piv, x);
}
/* np.linalg.inv(G) (rmhmc.py:58,138): dgesv against the identity */
static void lu_inverse(int D, const double *G, double *Ginv, double *work /* D*D */, int *piv, double *col /* D */) {
  memcpy(work, G, sizeof(double) * D * D);
  lu_factor(D, work, piv);
  for (int j = 0; j < D; j++) {
    memset(col, 0, sizeof(double) * D);
    col[j] = 1.0;
    lu_solve(D, work, piv, col);
    for (int i = 0; i < D; i++) Ginv[i * D + j] = col[i];
  }
}

static double norm2(int D, const double *x) {
  double s = 0;
  for (int i = 0; i < D; i++) s += x[i] * x[i];
  return sqrt(s);
}
static void matvec(int D, const double *A, const double *x, double *y) {
  for (int i = 0; i < D; i++) {
    double s = 0;
    for (int j = 0; j < D; j++) s += A[i * D + j] * x[j];
    y[i] = s;
  }
}
static double dot(int D, const double *a, const double *b) {
  double s = 0;
  for (int i = 0; i < D; i++) s += a[i] * b[i];
  return s;
}

/* ------------------------------------------------------------------------ */
/* the reference's inline "callbacks"                                       */
/* ------------------------------------------------------------------------ */

/* C1: log-joint.  rmhmc.py:31-34,166-169; tools.LogNormPDF tools.py:10-14.
 * Naive log(1+exp(f)) on purpose: overflows to +inf for f>709 like the reference. */
static double log_joint(const rmhmc_ctx *c, const double *w) {
  const int D = c->D;
  double ll = 0;
  for (int64_t n = 0; n < c->M; n++) {
    double f = dot(D, &c->X[n * D], w);
    ll += f * c->t[n] - log(1.0 + exp(f));
  }
  double lp = 0;
  for (int d = 0; d < D; d++) lp += -0.5 * log(PI2 * c->alpha) - w[d] * w[d] / (2.0 * c->alpha);
  return ll + lp;
}

/* C2: gradient  X'(t - e^f/(1+e^f)) - w/alpha.  rmhmc.py:99-100,140
 * (the e^f/(1+e^f) form: NaN for f>709, like the reference). */
static void gradient(const rmhmc_ctx *c, const double *w, double *g) {
  const int D = c->D;
  for (int d = 0; d < D; d++) g[d] = 0;
  for (int64_t n = 0; n < c->M; n++) {
    const double *x = &c->X[n * D];
    double f = dot(D, x, w), e = exp(f);
    double r = c->t[n] - e / (1.0 + e);
    for (int d = 0; d < D; d++) g[d] += r * x[d];
  }
  for (int d = 0; d < D; d++) g[d] -= w[d] / c->alpha;
}

/* C3: metric  G = X' diag(v) X + I/alpha, v = p(1-p), p = 1/(1+e^-f).
 * rmhmc.py:51-57,116-119,134-137.  Optionally returns c_n = v_n(1-2p_n). */
static void metric(const rmhmc_ctx *c, const double *w, double *G, double *cvec) {
  const int D = c->D;
  memset(G, 0, sizeof(double) * D * D);
  for (int64_t n = 0; n < c->M; n++) {
    const double *x = &c->X[n * D];
    double f = dot(D, x, w);
    double p = 1.0 / (1.0 + exp(-f));
    double v = p * (1.0 - p);
    if (cvec) cvec[n] = v * (1.0 - 2.0 * p);
    for (int a = 0; a < D; a++) {
      double va = v * x[a];
      for (int b = a; b < D; b++) G[a * D + b] += va * x[b];
    }
  }
  for (int a = 0; a < D; a++) {
    G[a * D + a] += 1.0 / c->alpha;
    for (int b = a + 1; b < D; b++) G[b * D + a] = G[a * D + b];
  }
}

/* C4 matrix-free: tr_d = sum_n c_n h_n x_nd, h_n = x_n' Ginv x_n  (== np.trace(InvG.dot(GDeriv)),
 * rmhmc.py:67-77) */
static void trace_term_mf(const rmhmc_ctx *c, const double *cvec, const double *Ginv, double *tr, double *tmp /* D */) {
  const int D = c->D;
  for (int d = 0; d < D; d++) tr[d] = 0;
  for (int64_t n = 0; n < c->M; n++) {
    const double *x = &c->X[n * D];
    matvec(D, Ginv, x, tmp);
    double h = dot(D, x, tmp) * cvec[n];
    for (int d = 0; d < D; d++) tr[d] += h * x[d];
  }
}
/* C4 matrix-free: quad_d = u' dG_d u = sum_n c_n (x_n'u)^2 x_nd  (LastTerm = quad/2, rmhmc.py:104-107) */
static void quad_term_mf(const rmhmc_ctx *c, const double *cvec, const double *u, double *q) {
  const int D = c->D;
  for (int d = 0; d < D; d++) q[d] = 0;
  for (int64_t n = 0; n < c->M; n++) {
    const double *x = &c->X[n * D];
    double s = dot(D, x, u);
    double r = cvec[n] * s * s;
    for (int d = 0; d < D; d++) q[d] += r * x[d];
  }
}

/* C4 literal: InvGdG[d] = InvG . (X' diag(v(1-2p)x_d) X), Trace[d].  rmhmc.py:64-77,142-156 */
static void tensor_literal(const rmhmc_ctx *c, const double *w, const double *InvG, double *T /* D^3 */, double *tr, double *GD /* D*D */) {
  const int D = c->D;
  double *cv = (double *)malloc(sizeof(double) * c->M);
  for (int64_t n = 0; n < c->M; n++) {
    double f = dot(D, &c->X[n * D], w);
    double p = 1.0 / (1.0 + exp(-f));
    cv[n] = p * (1.0 - p) * (1.0 - 2.0 * p);
  }
  for (int d = 0; d < D; d++) {
    memset(GD, 0, sizeof(double) * D * D);
    for (int64_t n = 0; n < c->M; n++) {
      const double *x = &c->X[n * D];
      double z1 = cv[n] * x[d];
      for (int a = 0; a < D; a++) {
        double za = z1 * x[a];
        for (int b = 0; b < D; b++) GD[a * D + b] += za * x[b];
      }
    }
    double *Td = &T[(size_t)d * D * D];
    double trd = 0;
    for (int i = 0; i < D; i++)
      for (int j = 0; j < D; j++) {
        double s = 0;
        for (int k = 0; k < D; k++) s += InvG[i * D + k] * GD[k * D + j];
        Td[i * D + j] = s;
        if (i == j) trd += s;
      }
    tr[d] = trd;
  }
  free(cv);
}

/* ------------------------------------------------------------------------ */
/* point record + workspace                                                 */
/* ------------------------------------------------------------------------ */
typedef struct {
  double *w, *grad, *tr, *G, *L, *Ginv, *cvec; /* D, D, D, D*D, D*D, D*D, M */
  double *T;                                   /* D^3, literal only */
  double ljl, hld;
  int status;
} point_t;

typedef struct {
  double *G2, *L2, *tmpDD, *a, *b, *u, *u0, *PM, *Pw, *q, *tmpD;
  int *piv;
} work_t;

static void point_alloc(const rmhmc_ctx *c, point_t *pt, int literal) {
  const int D = c->D;
  pt->w = (double *)calloc(D, sizeof(double));
  pt->grad = (double *)calloc(D, sizeof(double));
  pt->tr = (double *)calloc(D, sizeof(double));
  pt->G = (double *)calloc((size_t)D * D, sizeof(double));
  pt->L = (double *)calloc((size_t)D * D, sizeof(double));
  pt->Ginv = (double *)calloc((size_t)D * D, sizeof(double));
  pt->cvec = (double *)calloc(c->M, sizeof(double));
  pt->T = literal ? (double *)calloc((size_t)D * D * D, sizeof(double)) : NULL;
  pt->ljl = pt->hld = 0;
  pt->status = 0;
}
static void point_free(point_t *pt) {
  free(pt->w); free(pt->grad); free(pt->tr); free(pt->G); free(pt->L); free(pt->Ginv); free(pt->cvec); free(pt->T);
}
static void point_copy(const rmhmc_ctx *c, point_t *dst, const point_t *src) {
  const int D = c->D;
  memcpy(dst->w, src->w, sizeof(double) * D);
  memcpy(dst->grad, src->grad, sizeof(double) * D);
  memcpy(dst->tr, src->tr, sizeof(double) * D);
  memcpy(dst->G, src->G, sizeof(double) * D * D);
  memcpy(dst->L, src->L, sizeof(double) * D * D);
  memcpy(dst->Ginv, src->Ginv, sizeof(double) * D * D);
  memcpy(dst->cvec, src->cvec, sizeof(double) * c->M);
  if (dst->T && src->T) memcpy(dst->T, src->T, sizeof(double) * D * D * D);
  dst->ljl = src->ljl; dst->hld = src->hld; dst->status = src->status;
}
static void work_alloc(const rmhmc_ctx *c, work_t *k) {
  const int D = c->D;
  k->G2 = (double *)calloc((size_t)D * D, sizeof(double));
  k->L2 = (double *)calloc((size_t)D * D, sizeof(double));
  k->tmpDD = (double *)calloc((size_t)D * D, sizeof(double));
  k->a = (double *)calloc(D, sizeof(double)); k->b = (double *)calloc(D, sizeof(double));
  k->u = (double *)calloc(D, sizeof(double)); k->u0 = (double *)calloc(D, sizeof(double));
  k->PM = (double *)calloc(D, sizeof(double)); k->Pw = (double *)calloc(D, sizeof(double));
  k->q = (double *)calloc(D, sizeof(double)); k->tmpD = (double *)calloc(D, sizeof(double));
  k->piv = (int *)calloc(D, sizeof(int));
}
static void work_free(work_t *k) {
  free(k->G2); free(k->L2); free(k->tmpDD); free(k->a); free(k->b); free(k->u); free(k->u0);
  free(k->PM); free(k->Pw); free(k->q); free(k->tmpD); free(k->piv);
}

/* Evaluate everything the sampler needs at pt->w.
 * matrix-free: G, chol, hld, Ginv (from chol), grad, ljl, cvec, trace term.
 * literal: G, LU inverse, chol (for hld / momentum draw), grad, ljl, tensor + trace. */
static void point_eval(const rmhmc_ctx *c, point_t *pt, work_t *k, int literal) {
  const int D = c->D;
  metric(c, pt->w, pt->G, pt->cvec);
  pt->status = chol_lower(D, pt->G, pt->L) ? RMHMC_ST_NOT_PD : 0;
  pt->hld = half_logdet(D, pt->L);
  gradient(c, pt->w, pt->grad);
  pt->ljl = log_joint(c, pt->w);
  if (literal) {
    lu_inverse(D, pt->G, pt->Ginv, k->tmpDD, k->piv, k->tmpD);
    tensor_literal(c, pt->w, pt->Ginv, pt->T, pt->tr, k->G2);
  } else {
    chol_inverse(D, pt->L, pt->Ginv, k->tmpD);
    trace_term_mf(c, pt->cvec, pt->Ginv, pt->tr, k->tmpD);
  }
}

/* LastTerm_d = 0.5 * PM' InvGdG[d] InvG PM   (rmhmc.py:104-107,158-161) */
static void last_term(const rmhmc_ctx *c, const point_t *pt, const double *PM, double *last, work_t *k, int literal) {
  const int D = c->D;
  matvec(D, pt->Ginv, PM, k->u);
  if (literal) {
    for (int d = 0; d < D; d++) {
      const double *Td = &pt->T[(size_t)d * D * D];
      /* PM' Td u */
      double s = 0;
      for (int i = 0; i < D; i++) {
        double r = 0;
        for (int j = 0; j < D; j++) r += Td[i * D + j] * k->u[j];
        s += PM[i] * r;
      }
      last[d] = 0.5 * s;
    }
  } else {
    quad_term_mf(c, pt->cvec, k->u, last);
    for (int d = 0; d < D; d++) last[d] *= 0.5;
  }
}

/* One generalised leapfrog step (rmhmc.py:96-163).  pt: record at the current
 * w (in/out: replaced by the record at the new w); p in/out. */
static void leapfrog_step(const rmhmc_ctx *c, point_t *pt, double *p, double tau, double eps, int K, work_t *k, int literal, int *status) {
  const int D = c->D;
  const double h = tau * eps / 2.0;
  /* implicit momentum half step, K fixed-point iterations, rmhmc.py:102-110 */
  memcpy(k->PM, p, sizeof(double) * D);
  for (int it = 0; it < K; it++) {
    last_term(c, pt, k->PM, k->q, k, literal);
    for (int d = 0; d < D; d++) k->PM[d] = p[d] + h * (pt->grad[d] - 0.5 * pt->tr[d] + k->q[d]);
  }
  memcpy(p, k->PM, sizeof(double) * D);
  /* implicit position step, rmhmc.py:113-123 */
  if (literal) lu_solve_copy(D, pt->G, p, k->u0, k->tmpDD, k->piv);
  else chol_solve(D, pt->L, p, k->u0);
  memcpy(k->Pw, pt->w, sizeof(double) * D);
  for (int it = 0; it < K; it++) {
    if (literal) {
      metric(c, k->Pw, k->G2, NULL); /* rmhmc.py:116-119 (recomputed even for it==0) */
      lu_solve_copy(D, k->G2, p, k->u, k->tmpDD, k->piv);
    } else if (it == 0) {
      memcpy(k->u, k->u0, sizeof(double) * D); /* G(Pw^0) == G(w): same expression, same input */
    } else {
      metric(c, k->Pw, k->G2, NULL);
      if (chol_lower(D, k->G2, k->L2)) *status |= RMHMC_ST_NOT_PD;
      chol_solve(D, k->L2, p, k->u);
    }
    for (int d = 0; d < D; d++) k->Pw[d] = pt->w[d] + h * (k->u0[d] + k->u[d]);
  }
  memcpy(pt->w, k->Pw, sizeof(double) * D);
  /* position guard, rmhmc.py:125-130 */
  if (c->flags & RMHMC_FLAG_GUARDS) {
    double nw = norm2(D, pt->w);
    if (nw > 10.0) {
      for (int d = 0; d < D; d++) pt->w[d] /= nw * 3.0;
      *status |= RMHMC_ST_GUARD_W;
    }
  }
  /* explicit momentum half step at the new w, rmhmc.py:134-163 */
  point_eval(c, pt, k, literal);
  *status |= pt->status;
  last_term(c, pt, p, k->q, k, literal);
  for (int d = 0; d < D; d++) p[d] += h * (pt->grad[d] - 0.5 * pt->tr[d] + k->q[d]);
  for (int d = 0; d < D; d++)
    if (!isfinite(pt->w[d]) || !isfinite(p[d])) *status |= RMHMC_ST_NONFINITE;
}

/* momentum draw + guard, rmhmc.py:80-87 */
static void draw_momentum(const rmhmc_ctx *c, const double *L, const double *z, double *p, int *status) {
  const int D = c->D;
  if (c->flags & RMHMC_FLAG_MOMENTUM_LT) { /* (z L)' = L' z */
    for (int j = 0; j < D; j++) { double s = 0; for (int i = j; i < D; i++) s += L[i * D + j] * z[i]; p[j] = s; }
  } else { /* L z : Cov = G, as BLR_RMHMC.m:231,249 with upper chol */
    for (int i = 0; i < D; i++) { double s = 0; for (int j = 0; j <= i; j++) s += L[i * D + j] * z[j]; p[i] = s; }
  }
  if (c->flags & RMHMC_FLAG_GUARDS) {
    double np_ = norm2(D, p);
    if (np_ > 100.0) { for (int d = 0; d < D; d++) p[d] /= np_ * 25.0; *status |= RMHMC_ST_GUARD_P; }
  }
}

/* H = -LJL + 0.5 log|G| + 0.5 p' G^-1 p   (rmhmc.py:171-172,175-176) */
static double hamiltonian(const rmhmc_ctx *c, const point_t *pt, const double *p, work_t *k) {
  matvec(c->D, pt->Ginv, p, k->tmpD);
  return -pt->ljl + pt->hld + 0.5 * dot(c->D, p, k->tmpD);
}

typedef struct {
  int accepted, nsteps, dir, status;
  double H_cur, H_prop, hld_prop;
} trans_out_t;

/* One transition (rmhmc.py:47-184) from the cached record `cur`; `trj` is scratch.
 * Literal mode re-evaluates the set-up block like the reference does (:50-77). */
static void transition(const rmhmc_ctx *c, point_t *cur, point_t *trj, double *p, const double *z, double u_len,
                       double g_dir, double u_acc, int L, double eps, int K, work_t *k, int literal, trans_out_t *o,
                       double *p0 /* D scratch */) {
  const int D = c->D;
  int status = 0;
  if (literal) point_eval(c, cur, k, 1);
  point_copy(c, trj, cur);
  draw_momentum(c, cur->L, z, p, &status);
  memcpy(p0, p, sizeof(double) * D);
  int nsteps = (int)ceil(u_len * L);       /* rmhmc.py:89 */
  double tau = (g_dir > 0.5) ? 1.0 : -1.0; /* rmhmc.py:90-93 */
  for (int s = 0; s < nsteps; s++) leapfrog_step(c, trj, p, tau, eps, K, k, literal, &status);
  double Hp = hamiltonian(c, trj, p, k);
  double Hc = hamiltonian(c, cur, p0, k);
  if (literal) { /* rmhmc.py:175 uses the stored CurrentLJL; identical value */ }
  double ratio = -Hp + Hc;
  int acc = (ratio > 0) || (ratio > log(u_acc)); /* rmhmc.py:181 */
  o->accepted = acc; o->nsteps = nsteps; o->dir = (int)tau; o->status = status;
  o->H_cur = Hc; o->H_prop = Hp; o->hld_prop = trj->hld;
}

/* ------------------------------------------------------------------------ */
/* unit entry points                                                        */
/* ------------------------------------------------------------------------ */
#define NEED_DATA(ctx) if (!(ctx) || !(ctx)->have_data) return fail(ctx, RMHMC_ERR_INVALID, "set_data not called")
#define LITERAL(ctx) (((ctx)->flags & RMHMC_FLAG_ORACLE_LITERAL) != 0)

int rmhmc_log_posterior(rmhmc_ctx *ctx, const double *w, double *ljl_out) {
  NEED_DATA(ctx);
  const int D = ctx->D;
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) ljl_out[c] = log_joint(ctx, &w[c * D]);
  return RMHMC_OK;
}

int rmhmc_metric(rmhmc_ctx *ctx, const double *w, double *G_out, double *half_logdet_out, double *grad_out) {
  NEED_DATA(ctx);
  const int D = ctx->D;
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    double *G = (double *)malloc(sizeof(double) * D * D), *L = (double *)malloc(sizeof(double) * D * D);
    metric(ctx, &w[c * D], G, NULL);
    if (G_out) memcpy(&G_out[c * D * D], G, sizeof(double) * D * D);
    if (half_logdet_out) { chol_lower(D, G, L); half_logdet_out[c] = half_logdet(D, L); }
    if (grad_out) gradient(ctx, &w[c * D], &grad_out[c * D]);
    free(G); free(L);
  }
  return RMHMC_OK;
}

int rmhmc_metric_terms(rmhmc_ctx *ctx, const double *w, const double *p, double *trace_out, double *quad_out) {
  NEED_DATA(ctx);
  const int D = ctx->D;
  const int lit = LITERAL(ctx);
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    point_t pt; work_t k;
    point_alloc(ctx, &pt, lit); work_alloc(ctx, &k);
    memcpy(pt.w, &w[c * D], sizeof(double) * D);
    point_eval(ctx, &pt, &k, lit);
    if (trace_out) memcpy(&trace_out[c * D], pt.tr, sizeof(double) * D);
    if (p && quad_out) {
      last_term(ctx, &pt, &p[c * D], k.q, &k, lit);
      for (int d = 0; d < D; d++) quad_out[c * D + d] = 2.0 * k.q[d];
    }
    point_free(&pt); work_free(&k);
  }
  return RMHMC_OK;
}

int rmhmc_leapfrog(rmhmc_ctx *ctx, double *w, double *p, double eps, const int32_t *dir, const int32_t *nsteps,
                   int32_t K, double *half_logdet_out, int32_t *status_out) {
  NEED_DATA(ctx);
  const int D = ctx->D;
  const int lit = LITERAL(ctx);
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    point_t pt; work_t k;
    point_alloc(ctx, &pt, lit); work_alloc(ctx, &k);
    memcpy(pt.w, &w[c * D], sizeof(double) * D);
    point_eval(ctx, &pt, &k, lit);
    int st = pt.status;
    for (int s = 0; s < nsteps[c]; s++) leapfrog_step(ctx, &pt, &p[c * D], (double)dir[c], eps, K, &k, lit, &st);
    memcpy(&w[c * D], pt.w, sizeof(double) * D);
    if (half_logdet_out) half_logdet_out[c] = pt.hld;
    if (status_out) status_out[c] = st;
    point_free(&pt); work_free(&k);
  }
  return RMHMC_OK;
}

int rmhmc_transition(rmhmc_ctx *ctx, double *w, const double *z, const double *u_len, const double *g_dir,
                     const double *u_acc, int32_t L, double eps, int32_t K, int32_t *accepted_out,
                     int32_t *nsteps_out, double *H_cur_out, double *H_prop_out, double *w_prop_out,
                     double *p_prop_out, double *half_logdet_prop_out, int32_t *status_out) {
  NEED_DATA(ctx);
  const int D = ctx->D;
  const int lit = LITERAL(ctx);
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    point_t cur, trj; work_t k; trans_out_t o;
    point_alloc(ctx, &cur, lit); point_alloc(ctx, &trj, lit); work_alloc(ctx, &k);
    double *p = (double *)calloc(D, sizeof(double)), *p0 = (double *)calloc(D, sizeof(double));
    memcpy(cur.w, &w[c * D], sizeof(double) * D);
    if (!lit) point_eval(ctx, &cur, &k, 0);
    transition(ctx, &cur, &trj, p, &z[c * D], u_len[c], g_dir[c], u_acc[c], L, eps, K, &k, lit, &o, p0);
    if (o.accepted) memcpy(&w[c * D], trj.w, sizeof(double) * D);
    if (accepted_out) accepted_out[c] = o.accepted;
    if (nsteps_out) nsteps_out[c] = o.nsteps;
    if (H_cur_out) H_cur_out[c] = o.H_cur;
    if (H_prop_out) H_prop_out[c] = o.H_prop;
    if (w_prop_out) memcpy(&w_prop_out[c * D], trj.w, sizeof(double) * D);
    if (p_prop_out) memcpy(&p_prop_out[c * D], p, sizeof(double) * D);
    if (half_logdet_prop_out) half_logdet_prop_out[c] = o.hld_prop;
    if (status_out) status_out[c] = o.status;
    free(p); free(p0);
    point_free(&cur); point_free(&trj); work_free(&k);
  }
  return RMHMC_OK;
}

/* ------------------------------------------------------------------------ */
/* counter-based RNG shared (by specification) with the HIP library          */
/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11).  key = seed, counter =
 * (chain lo, chain hi, iteration, block).  Blocks 0..ceil(D/2)-1 give the D
 * momentum normals by Box-Muller (2 per block); block 0x40000000 gives
 * (u_len, u_acc); block 0x40000001 gives g_dir. */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}
static void rng_block(uint64_t seed, uint64_t chain, uint32_t iter, uint32_t block, double *U0, double *U1) {
  uint32_t c[4] = {(uint32_t)chain, (uint32_t)(chain >> 32), iter, block};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  *U0 = u53(c[0], c[1]);
  *U1 = u53(c[2], c[3]);
}
static void rng_draws(uint64_t seed, uint64_t chain, uint32_t iter, int D, double *z, double *u_len, double *g_dir, double *u_acc) {
  for (int j = 0; 2 * j < D; j++) {
    double U0, U1;
    rng_block(seed, chain, iter, (uint32_t)j, &U0, &U1);
    double R = sqrt(-2.0 * log(U0));
    z[2 * j] = R * cos(PI2 * U1);
    if (2 * j + 1 < D) z[2 * j + 1] = R * sin(PI2 * U1);
  }
  double U0, U1;
  rng_block(seed, chain, iter, 0x40000000u, u_len, u_acc);
  rng_block(seed, chain, iter, 0x40000001u, &U0, &U1);
  *g_dir = sqrt(-2.0 * log(U0)) * cos(PI2 * U1);
}
/* exported for tests: raw Philox block (known-answer tests) and the draw recipe */
void rmhmc_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  philox4x32_10(c, key[0], key[1]);
  memcpy(out, c, sizeof(c));
}
void rmhmc_oracle_draws(uint64_t seed, uint64_t chain, uint32_t iter, int32_t D, double *z, double *u3 /* len,dir,acc */) {
  rng_draws(seed, chain, iter, D, z, &u3[0], &u3[1], &u3[2]);
}

/* ------------------------------------------------------------------------ */
/* bulk entry points                                                        */
/* ------------------------------------------------------------------------ */
struct chain {
  point_t cur, trj;
  work_t k;
  double *p, *p0, *z;
  double H_cur, tau;
  int steps_left;
  int64_t iter, accepted, steps_done;
  int status;
};

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int rmhmc_sample(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                 int64_t chain_offset, const double *theta0, double *samples_out, int64_t *accept_out,
                 int64_t *steps_out, double *seconds_out) {
  NEED_DATA(ctx);
  if (burn_in >= n_iter || burn_in < 0 || L < 1 || K < 0) return fail(ctx, RMHMC_ERR_INVALID, "need 0 <= burn_in < n_iter, L>=1");
  const int D = ctx->D;
  const int lit = LITERAL(ctx);
  const int64_t S = n_iter - burn_in;
  double t_post = 0;
  int64_t *acc_all = (int64_t *)calloc(ctx->n, sizeof(int64_t));
  /* Two phases so that the timer covers exactly the post-burn-in transitions of every chain
   * (TimeTaken, rmhmc.py:194-198).  Between the phases a chain is represented by its position only
   * (row 0 of its samples); its point record is re-evaluated, which is deterministic. */
  for (int phase = 0; phase < 2; phase++) {
    double t0 = now_s();
#pragma omp parallel for schedule(dynamic)
    for (int64_t c = 0; c < ctx->n; c++) {
      point_t cur, trj;
      work_t k;
      point_alloc(ctx, &cur, lit); point_alloc(ctx, &trj, lit); work_alloc(ctx, &k);
      double *p = (double *)calloc(D, sizeof(double)), *p0 = (double *)calloc(D, sizeof(double)), *z = (double *)calloc(D + 1, sizeof(double));
      int64_t it0, it1, steps = 0;
      if (phase == 0) {
        for (int d = 0; d < D; d++) cur.w[d] = theta0 ? theta0[c * D + d] : 1e-3; /* rmhmc.py:27 */
        it0 = 0; it1 = burn_in + 1;
      } else {
        memcpy(cur.w, &samples_out[(c * S + 0) * D], sizeof(double) * D);
        it0 = burn_in + 1; it1 = n_iter;
      }
      if (!lit) point_eval(ctx, &cur, &k, 0);
      for (int64_t it = it0; it < it1; it++) {
        double u_len, g_dir, u_acc;
        trans_out_t o;
        rng_draws(seed, (uint64_t)(chain_offset + c), (uint32_t)it, D, z, &u_len, &g_dir, &u_acc);
        transition(ctx, &cur, &trj, p, z, u_len, g_dir, u_acc, L, eps, K, &k, lit, &o, p0);
        if (o.accepted) { point_copy(ctx, &cur, &trj); acc_all[c]++; }
        if (it >= burn_in) memcpy(&samples_out[(c * S + (it - burn_in)) * D], cur.w, sizeof(double) * D);
        if (it > burn_in) steps += o.nsteps;
      }
      if (steps_out && phase == 1) steps_out[c] = steps;
      free(p); free(p0); free(z);
      point_free(&cur); point_free(&trj); work_free(&k);
    }
    if (phase == 1) t_post = now_s() - t0;
  }
  if (accept_out) memcpy(accept_out, acc_all, sizeof(int64_t) * ctx->n);
  free(acc_all);
  if (steps_out && n_iter == burn_in + 1)
    for (int64_t c = 0; c < ctx->n; c++) steps_out[c] = 0;
  if (seconds_out) *seconds_out = t_post;
  return RMHMC_OK;
}

static void chain_begin(rmhmc_ctx *ctx, struct chain *ch, int64_t gid) {
  double u_len, g_dir, u_acc;
  rng_draws(ctx->seed, (uint64_t)gid, (uint32_t)ch->iter, ctx->D, ch->z, &u_len, &g_dir, &u_acc);
  point_copy(ctx, &ch->trj, &ch->cur);
  ch->status = 0;
  draw_momentum(ctx, ch->cur.L, ch->z, ch->p, &ch->status);
  memcpy(ch->p0, ch->p, sizeof(double) * ctx->D);
  ch->steps_left = (int)ceil(u_len * ctx->L);
  ch->tau = (g_dir > 0.5) ? 1.0 : -1.0;
  ch->H_cur = hamiltonian(ctx, &ch->cur, ch->p0, &ch->k);
}
static void chain_end(rmhmc_ctx *ctx, struct chain *ch, int64_t gid) {
  double u_len, g_dir, u_acc;
  rng_draws(ctx->seed, (uint64_t)gid, (uint32_t)ch->iter, ctx->D, ch->z, &u_len, &g_dir, &u_acc);
  double Hp = hamiltonian(ctx, &ch->trj, ch->p, &ch->k);
  double ratio = -Hp + ch->H_cur;
  if ((ratio > 0) || (ratio > log(u_acc))) { point_copy(ctx, &ch->cur, &ch->trj); ch->accepted++; }
  ch->iter++;
}

int rmhmc_chains_init(rmhmc_ctx *ctx, const double *theta0, uint64_t seed, int64_t chain_offset, int32_t L, double eps, int32_t K) {
  NEED_DATA(ctx);
  if (LITERAL(ctx)) return fail(ctx, RMHMC_ERR_UNSUPPORTED, "chains_* API is matrix-free only");
  const int D = ctx->D;
  if (!ctx->chains) {
    ctx->chains = (struct chain *)calloc(ctx->n, sizeof(struct chain));
    for (int64_t c = 0; c < ctx->n; c++) {
      struct chain *ch = &ctx->chains[c];
      point_alloc(ctx, &ch->cur, 0); point_alloc(ctx, &ch->trj, 0); work_alloc(ctx, &ch->k);
      ch->p = (double *)calloc(D, sizeof(double)); ch->p0 = (double *)calloc(D, sizeof(double)); ch->z = (double *)calloc(D + 1, sizeof(double));
    }
  }
  ctx->seed = seed; ctx->chain_offset = chain_offset; ctx->L = L; ctx->eps = eps; ctx->K = K;
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    struct chain *ch = &ctx->chains[c];
    for (int d = 0; d < D; d++) ch->cur.w[d] = theta0 ? theta0[c * D + d] : 1e-3;
    point_eval(ctx, &ch->cur, &ch->k, 0);
    ch->iter = 0; ch->accepted = 0; ch->steps_done = 0; ch->steps_left = 0;
  }
  ctx->chains_ready = 1;
  return RMHMC_OK;
}

int rmhmc_chains_run(rmhmc_ctx *ctx, int64_t n_steps) {
  if (!ctx || !ctx->chains_ready) return fail(ctx, RMHMC_ERR_INVALID, "chains_init not called");
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    struct chain *ch = &ctx->chains[c];
    const int64_t gid = ctx->chain_offset + c;
    for (int64_t s = 0; s < n_steps; s++) {
      if (ch->steps_left == 0) chain_begin(ctx, ch, gid);
      leapfrog_step(ctx, &ch->trj, ch->p, ch->tau, ctx->eps, ctx->K, &ch->k, 0, &ch->status);
      ch->steps_done++;
      if (--ch->steps_left == 0) chain_end(ctx, ch, gid);
    }
  }
  return RMHMC_OK;
}

int rmhmc_chains_state(rmhmc_ctx *ctx, double *w_out, int64_t *iters_out, int64_t *accept_out) {
  if (!ctx || !ctx->chains_ready) return fail(ctx, RMHMC_ERR_INVALID, "chains_init not called");
  for (int64_t c = 0; c < ctx->n; c++) {
    if (w_out) memcpy(&w_out[c * ctx->D], ctx->chains[c].cur.w, sizeof(double) * ctx->D);
    if (iters_out) iters_out[c] = ctx->chains[c].iter;
    if (accept_out) accept_out[c] = ctx->chains[c].accepted;
  }
  return RMHMC_OK;
}

int rmhmc_chains_restore(rmhmc_ctx *ctx, const int64_t *iters, const int64_t *accepted) {
  if (!ctx || !ctx->chains_ready || !iters || !accepted) return fail(ctx, RMHMC_ERR_INVALID, "chains_restore: call chains_init first");
  for (int64_t c = 0; c < ctx->n; c++) {
    if (iters[c] < 0 || accepted[c] < 0) return fail(ctx, RMHMC_ERR_INVALID, "chains_restore: negative counter");
    ctx->chains[c].iter = iters[c];
    ctx->chains[c].accepted = accepted[c];
    ctx->chains[c].steps_left = 0;
  }
  return RMHMC_OK;
}

int rmhmc_kernel_time(rmhmc_ctx *ctx, const char *which, double *seconds_out, int64_t *launches_out) {
  (void)which;
  if (seconds_out) *seconds_out = 0;
  if (launches_out) *launches_out = 0;
  return fail(ctx, RMHMC_ERR_UNSUPPORTED, "no kernels in the CPU oracle");
}

/* ------------------------------------------------------------------------ */
/* plain HMC, identity mass: restatement of code/hmc.py:12-99                */
/* ------------------------------------------------------------------------ */
/* One transition from w (log joint ljl_w).  hmc.py:41-80.  Returns accepted; w_prop/p_prop = end of the
 * trajectory.  The gradient is re-evaluated at both ends of every step exactly like the reference. */
static int hmc_transition(const rmhmc_ctx *c, const double *w, double ljl_w, const double *z, double u_len, double u_acc,
                          int L, double eps, double *w_prop, double *p_prop, double *g /* D scratch */, int *nsteps_out,
                          double *Hc_out, double *Hp_out, double *ljl_prop_out) {
  const int D = c->D;
  memcpy(w_prop, w, sizeof(double) * D);
  memcpy(p_prop, z, sizeof(double) * D);                 /* Mass = I: p = z (hmc.py:41) */
  const double Hc = -ljl_w + 0.5 * dot(D, z, z);          /* hmc.py:72 */
  const int nsteps = (int)ceil(u_len * L);                /* hmc.py:48 */
  for (int s = 0; s < nsteps; s++) {
    gradient(c, w_prop, g);                               /* hmc.py:52-53 */
    for (int d = 0; d < D; d++) p_prop[d] += eps / 2 * g[d];
    int nan = 0;
    for (int d = 0; d < D; d++) if (p_prop[d] != p_prop[d]) nan = 1;
    if (nan) break;                                       /* hmc.py:56-57 */
    for (int d = 0; d < D; d++) w_prop[d] += eps * p_prop[d];
    gradient(c, w_prop, g);                               /* hmc.py:60-61 */
    for (int d = 0; d < D; d++) p_prop[d] += eps / 2 * g[d];
  }
  const double ljl_p = log_joint(c, w_prop);              /* hmc.py:64-67 */
  const double Hp = -ljl_p + 0.5 * dot(D, p_prop, p_prop); /* hmc.py:69 */
  const double ratio = -Hp + Hc;
  *nsteps_out = nsteps; *Hc_out = Hc; *Hp_out = Hp; *ljl_prop_out = ljl_p;
  return (ratio > 0) || (ratio > log(u_acc));             /* hmc.py:77 */
}

int rmhmc_hmc_transition(rmhmc_ctx *ctx, double *w, const double *z, const double *u_len, const double *u_acc, int32_t L,
                         double eps, int32_t *accepted_out, int32_t *nsteps_out, double *H_cur_out, double *H_prop_out,
                         double *w_prop_out, double *p_prop_out) {
  NEED_DATA(ctx);
  const int D = ctx->D;
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    double *wp = (double *)malloc(sizeof(double) * D), *pp = (double *)malloc(sizeof(double) * D), *g = (double *)malloc(sizeof(double) * D);
    int ns; double Hc, Hp, lj;
    const int acc = hmc_transition(ctx, &w[c * D], log_joint(ctx, &w[c * D]), &z[c * D], u_len[c], u_acc[c], L, eps, wp, pp, g, &ns, &Hc, &Hp, &lj);
    if (acc) memcpy(&w[c * D], wp, sizeof(double) * D);
    if (accepted_out) accepted_out[c] = acc;
    if (nsteps_out) nsteps_out[c] = ns;
    if (H_cur_out) H_cur_out[c] = Hc;
    if (H_prop_out) H_prop_out[c] = Hp;
    if (w_prop_out) memcpy(&w_prop_out[c * D], wp, sizeof(double) * D);
    if (p_prop_out) memcpy(&p_prop_out[c * D], pp, sizeof(double) * D);
    free(wp); free(pp); free(g);
  }
  return RMHMC_OK;
}

int rmhmc_hmc_sample(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, uint64_t seed, int64_t chain_offset,
                     const double *theta0, double *samples_out, int64_t *accept_out, int64_t *steps_out, double *seconds_out) {
  NEED_DATA(ctx);
  if (burn_in >= n_iter || burn_in < 0 || L < 1) return fail(ctx, RMHMC_ERR_INVALID, "need 0 <= burn_in < n_iter, L>=1");
  const int D = ctx->D;
  const int64_t S = n_iter - burn_in;
  double t_post = 0;
  int64_t *acc_all = (int64_t *)calloc(ctx->n, sizeof(int64_t));
  for (int phase = 0; phase < 2; phase++) {  /* timer covers the post-burn-in transitions only (hmc.py:92-96) */
    double t0 = now_s();
#pragma omp parallel for schedule(dynamic)
    for (int64_t c = 0; c < ctx->n; c++) {
      double *w = (double *)calloc(D, sizeof(double)), *wp = (double *)calloc(D, sizeof(double)), *pp = (double *)calloc(D, sizeof(double));
      double *g = (double *)calloc(D, sizeof(double)), *z = (double *)calloc(D + 1, sizeof(double));
      int64_t it0, it1, steps = 0;
      if (phase == 0) {
        for (int d = 0; d < D; d++) w[d] = theta0 ? theta0[c * D + d] : 0.0; /* hmc.py:27 */
        it0 = 0; it1 = burn_in + 1;
      } else {
        memcpy(w, &samples_out[(c * S + 0) * D], sizeof(double) * D);
        it0 = burn_in + 1; it1 = n_iter;
      }
      double ljl = log_joint(ctx, w);
      for (int64_t it = it0; it < it1; it++) {
        double u_len, g_dir, u_acc, Hc, Hp, lj; int ns;
        rng_draws(seed, (uint64_t)(chain_offset + c), (uint32_t)it, D, z, &u_len, &g_dir, &u_acc);
        if (hmc_transition(ctx, w, ljl, z, u_len, u_acc, L, eps, wp, pp, g, &ns, &Hc, &Hp, &lj)) {
          memcpy(w, wp, sizeof(double) * D); ljl = lj; acc_all[c]++;
        }
        if (it >= burn_in) memcpy(&samples_out[(c * S + (it - burn_in)) * D], w, sizeof(double) * D);
        if (it > burn_in) steps += ns;
      }
      if (steps_out && phase == 1) steps_out[c] = steps;
      free(w); free(wp); free(pp); free(g); free(z);
    }
    if (phase == 1) t_post = now_s() - t0;
  }
  if (accept_out) memcpy(accept_out, acc_all, sizeof(int64_t) * ctx->n);
  free(acc_all);
  if (steps_out && n_iter == burn_in + 1)
    for (int64_t c = 0; c < ctx->n; c++) steps_out[c] = 0;
  if (seconds_out) *seconds_out = t_post;
  return RMHMC_OK;
}

/* ------------------------------------------------------------------------ */
/* ESS: restatement of tools.CalculateESS (tools.py:32-74) without the FFT    */
/* ------------------------------------------------------------------------ */
/* series x[s*stride], s < S.  Autocovariances are computed lag by lag (the FFT of tools.ac, tools.py:21-30,
 * yields the same sums for every lag it does not wrap); Gamma_j = rho_2j + rho_2j+1 (tools.py:46-50), running
 * minimum (:54-60), sum of the positive prefix (:62-67), floor at 1 (:70-71), ESS = S / MonoEst (:73). */
/* nfft = 0: linear autocovariances (ac.m:78); nfft > 0: the circular ones of tools.py:21-30 with period nfft = nextpow2(S)+1 */
static void ess_series(const double *x, int64_t S, int64_t stride, double *ess, double *mean_out, double *var_out, int64_t nfft) {
  double m = 0;
  for (int64_t s = 0; s < S; s++) m += x[s * stride];
  m /= (double)S;
  double *xc = (double *)malloc(sizeof(double) * S);
  double c0 = 0;
  for (int64_t s = 0; s < S; s++) { xc[s] = x[s * stride] - m; c0 += xc[s] * xc[s]; }
  if (mean_out) *mean_out = m;
  if (var_out) *var_out = c0 / (double)S;
  double prev = INFINITY, sum = 0;
  const int64_t half = S / 2; /* floor((MaxLag+1)/2) with MaxLag = S-1 */
  for (int64_t j = 0; j < half; j++) {
    double a = 0, b = 0;
    for (int64_t s = 0; s + 2 * j < S; s++) a += xc[s] * xc[s + 2 * j];
    for (int64_t s = 0; s + 2 * j + 1 < S; s++) b += xc[s] * xc[s + 2 * j + 1];
    if (nfft > 0) {
      const int64_t w0 = nfft - 2 * j, w1 = nfft - 2 * j - 1;
      if (j > 0) for (int64_t s = 0; s + w0 < S; s++) a += xc[s] * xc[s + w0];
      for (int64_t s = 0; s + w1 < S; s++) b += xc[s] * xc[s + w1];
    }
    double g = (a + b) / c0;
    if (g > prev) g = prev;
    if (!(g > 0)) break;
    sum += g; prev = g;
  }
  double mono = -1.0 + 2.0 * sum;
  if (mono < 1) mono = 1;
  *ess = (c0 > 0) ? (double)S / mono : NAN;
  free(xc);
}

static int64_t ess_nfft(const rmhmc_ctx *ctx, int64_t S) {
  if (!(ctx->flags & RMHMC_FLAG_ESS_WRAP)) return 0;
  int64_t n = 1;
  while (n < S) n *= 2; /* tools.py:16-19 */
  return n + 1;         /* tools.py:23 */
}

int rmhmc_ess(rmhmc_ctx *ctx, const double *samples, int64_t n, int64_t S, int32_t P, double *ess_out) {
  if (!ctx || !samples || !ess_out || n < 1 || S < 2 || P < 1) return fail(ctx, RMHMC_ERR_INVALID, "ess: bad argument");
  const int64_t nfft = ess_nfft(ctx, S);
#pragma omp parallel for schedule(dynamic)
  for (int64_t i = 0; i < n * P; i++) ess_series(&samples[(i / P) * S * P + (i % P)], S, P, &ess_out[i], NULL, NULL, nfft);
  return RMHMC_OK;
}

int rmhmc_sample_stats(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                       int64_t chain_offset, const double *theta0, double *mean_out, double *var_out, double *ess_out,
                       int64_t *accept_out, int64_t *steps_out, double *seconds_out) {
  NEED_DATA(ctx);
  if (burn_in >= n_iter || burn_in < 0) return fail(ctx, RMHMC_ERR_INVALID, "need 0 <= burn_in < n_iter");
  const int64_t S = n_iter - burn_in;
  const int D = ctx->D;
  double *smp = (double *)malloc(sizeof(double) * ctx->n * S * D);
  int rc = rmhmc_sample(ctx, n_iter, burn_in, L, eps, K, seed, chain_offset, theta0, smp, accept_out, steps_out, seconds_out);
  if (rc == RMHMC_OK) {
#pragma omp parallel for schedule(dynamic)
    for (int64_t i = 0; i < ctx->n * D; i++) {
      double e, m, v;
      if (S >= 2) ess_series(&smp[(i / D) * S * D + (i % D)], S, D, &e, &m, &v, ess_nfft(ctx, S));
      else { m = smp[i]; v = 0; e = NAN; }
      if (ess_out) ess_out[i] = e;
      if (mean_out) mean_out[i] = m;
      if (var_out) var_out[i] = v;
    }
  }
  free(smp);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* simplified manifold MALA: restatement of BLR_mMALA_Simp.m:175-290          */
/* ------------------------------------------------------------------------ */
typedef struct { double *w, *grad, *G, *L, *Ginv; double ljl, hld; } mpoint_t;
static void mpoint_alloc(int D, mpoint_t *q) {
  q->w = (double *)calloc(D, sizeof(double)); q->grad = (double *)calloc(D, sizeof(double));
  q->G = (double *)calloc((size_t)D * D, sizeof(double)); q->L = (double *)calloc((size_t)D * D, sizeof(double));
  q->Ginv = (double *)calloc((size_t)D * D, sizeof(double));
}
static void mpoint_free(mpoint_t *q) { free(q->w); free(q->grad); free(q->G); free(q->L); free(q->Ginv); }
/* `grad` holds the vector the drift is built from: the gradient for the simplified sampler; for the full one
 * (RMHMC_FLAG_MMALA_FULL, BLR_mMALA.m:198-214,231-233) gradient minus trace term, because
 *   sum_d G^-1 dG_d G^-1 e_d = sum_n c_n h_n G^-1 x_n = G^-1 tr   (dG_d = sum_n c_n x_nd x_n x_n', h_n = x_n'G^-1 x_n)
 * so that Mean = w + eps/2 G^-1 grad - eps G^-1 tr + eps/2 G^-1 tr = w + eps/2 G^-1 (grad - tr). */
static void mpoint_eval(const rmhmc_ctx *c, mpoint_t *q, double *tmp) {  /* BLR_mMALA_Simp.m:186-199,229-243 */
  const int full = (c->flags & RMHMC_FLAG_MMALA_FULL) != 0;
  double *cvec = full ? (double *)malloc(sizeof(double) * c->M) : NULL;
  metric(c, q->w, q->G, cvec);
  chol_lower(c->D, q->G, q->L);
  q->hld = half_logdet(c->D, q->L);
  chol_inverse(c->D, q->L, q->Ginv, tmp);
  gradient(c, q->w, q->grad);
  if (full) {
    double *tr = (double *)malloc(sizeof(double) * c->D);
    trace_term_mf(c, cvec, q->Ginv, tr, tmp);
    for (int d = 0; d < c->D; d++) q->grad[d] -= tr[d];
    free(tr); free(cvec);
  }
  q->ljl = log_joint(c, q->w);
}
static void mpoint_copy(int D, mpoint_t *d, const mpoint_t *s) {
  memcpy(d->w, s->w, sizeof(double) * D); memcpy(d->grad, s->grad, sizeof(double) * D);
  memcpy(d->G, s->G, sizeof(double) * D * D); memcpy(d->L, s->L, sizeof(double) * D * D); memcpy(d->Ginv, s->Ginv, sizeof(double) * D * D);
  d->ljl = s->ljl; d->hld = s->hld;
}
/* one transition; cur in/out (replaced on acceptance); returns accepted */
static int mmala_transition(const rmhmc_ctx *c, mpoint_t *cur, mpoint_t *prop, const double *z, double u_acc, double eps,
                            double *a, double *b, double *ratio_out) {
  const int D = c->D;
  /* drift and proposal (:217-219): w' = w + eps/2 G^-1 grad + sqrt(eps) L^-T z, i.e. N(mean, eps G^-1); L^-T z = G^-1 (L z) */
  matvec(D, cur->Ginv, cur->grad, a);
  for (int i = 0; i < D; i++) { double s = 0; for (int j = 0; j <= i; j++) s += cur->L[i * D + j] * z[j]; b[i] = s; }
  for (int i = 0; i < D; i++) prop->w[i] = cur->w[i] + 0.5 * eps * a[i];
  matvec(D, cur->Ginv, b, a);
  for (int i = 0; i < D; i++) prop->w[i] += sqrt(eps) * a[i];
  /* log q(w'|w) up to the constant -(D/2) log eps shared by both directions (:227): (w'-mean)'(G/eps)(w'-mean) = z'z */
  const double q_fwd = cur->hld - 0.5 * dot(D, z, z);
  mpoint_eval(c, prop, a);
  /* log q(w|w') (:245-247) */
  matvec(D, prop->Ginv, prop->grad, a);
  for (int i = 0; i < D; i++) b[i] = prop->w[i] + 0.5 * eps * a[i] - cur->w[i];
  double yy = 0;
  for (int j = 0; j < D; j++) { double s = 0; for (int i = j; i < D; i++) s += prop->L[i * D + j] * b[i]; yy += s * s; } /* |L'^T d|^2 = d'G'd */
  const double q_rev = prop->hld - yy / (2.0 * eps);
  const double ratio = prop->ljl + q_rev - cur->ljl - q_fwd; /* :251 */
  if (ratio_out) *ratio_out = ratio;
  const int acc = (ratio > 0) || (ratio > log(u_acc));
  if (acc) mpoint_copy(D, cur, prop);
  return acc;
}

int rmhmc_mmala_transition(rmhmc_ctx *ctx, double *w, const double *z, const double *u_acc, double eps, int32_t *accepted_out,
                           double *ratio_out, double *w_prop_out) {
  NEED_DATA(ctx);
  const int D = ctx->D;
#pragma omp parallel for schedule(dynamic)
  for (int64_t c = 0; c < ctx->n; c++) {
    mpoint_t cur, prop; mpoint_alloc(D, &cur); mpoint_alloc(D, &prop);
    double *a = (double *)calloc(D, sizeof(double)), *b = (double *)calloc(D, sizeof(double)), r;
    memcpy(cur.w, &w[c * D], sizeof(double) * D);
    mpoint_eval(ctx, &cur, a);
    const int acc = mmala_transition(ctx, &cur, &prop, &z[c * D], u_acc[c], eps, a, b, &r);
    memcpy(&w[c * D], cur.w, sizeof(double) * D);
    if (accepted_out) accepted_out[c] = acc;
    if (ratio_out) ratio_out[c] = r;
    if (w_prop_out) memcpy(&w_prop_out[c * D], prop.w, sizeof(double) * D);
    free(a); free(b); mpoint_free(&cur); mpoint_free(&prop);
  }
  return RMHMC_OK;
}

int rmhmc_mmala_sample(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, double eps, uint64_t seed, int64_t chain_offset,
                       const double *theta0, double *samples_out, int64_t *accept_out, double *seconds_out) {
  NEED_DATA(ctx);
  if (burn_in >= n_iter || burn_in < 0 || !(eps > 0)) return fail(ctx, RMHMC_ERR_INVALID, "need 0 <= burn_in < n_iter, eps > 0");
  const int D = ctx->D;
  const int64_t S = n_iter - burn_in;
  double t_post = 0;
#pragma omp parallel for schedule(dynamic) reduction(max : t_post)
  for (int64_t c = 0; c < ctx->n; c++) {
    mpoint_t cur, prop; mpoint_alloc(D, &cur); mpoint_alloc(D, &prop);
    double *a = (double *)calloc(D, sizeof(double)), *b = (double *)calloc(D, sizeof(double)), *z = (double *)calloc(D + 1, sizeof(double));
    for (int d = 0; d < D; d++) cur.w[d] = theta0 ? theta0[c * D + d] : 0.0; /* BLR_mMALA_Simp.m:175 */
    mpoint_eval(ctx, &cur, a);
    int64_t acc = 0;
    double t0 = 0;
    for (int64_t it = 0; it < n_iter; it++) {
      double u_len, g_dir, u_acc;
      rng_draws(seed, (uint64_t)(chain_offset + c), (uint32_t)it, D, z, &u_len, &g_dir, &u_acc);
      acc += mmala_transition(ctx, &cur, &prop, z, u_acc, eps, a, b, NULL);
      if (it >= burn_in) memcpy(&samples_out[(c * S + (it - burn_in)) * D], cur.w, sizeof(double) * D);
      if (it == burn_in) t0 = now_s();
    }
    const double dt = now_s() - t0;
    if (dt > t_post) t_post = dt;
    if (accept_out) accept_out[c] = acc;
    free(a); free(b); free(z); mpoint_free(&cur); mpoint_free(&prop);
  }
  if (seconds_out) *seconds_out = t_post;
  return RMHMC_OK;
}

void rmhmc_destroy(rmhmc_ctx *ctx) {
  if (!ctx) return;
  if (ctx->chains) {
    for (int64_t c = 0; c < ctx->n; c++) {
      struct chain *ch = &ctx->chains[c];
      point_free(&ch->cur); point_free(&ch->trj); work_free(&ch->k);
      free(ch->p); free(ch->p0); free(ch->z);
    }
    free(ctx->chains);
  }
  free(ctx->X); free(ctx->t); free(ctx);
}

/* Device-resident write-out of include/rmhmc.h: the oracle has no device, so "device" memory is host memory and the
 * _dev entry points are the host ones (the world-size-2 gloo tests drive the multi-GPU host logic through them). */
int rmhmc_chains_state_dev(rmhmc_ctx *ctx, double *w_dev, int64_t *iters_dev, int64_t *accept_dev) {
  return rmhmc_chains_state(ctx, w_dev, iters_dev, accept_dev);
}
int rmhmc_sample_dev(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K, uint64_t seed,
                     int64_t chain_offset, const double *theta0, double *samples_dev, int64_t *accept_dev,
                     int64_t *steps_dev, double *seconds_out) {
  return rmhmc_sample(ctx, n_iter, burn_in, L, eps, K, seed, chain_offset, theta0, samples_dev, accept_dev, steps_dev, seconds_out);
}
int rmhmc_sample_stats_dev(rmhmc_ctx *ctx, int64_t n_iter, int64_t burn_in, int32_t L, double eps, int32_t K,
                           uint64_t seed, int64_t chain_offset, const double *theta0, double *mean_dev,
                           double *var_dev, double *ess_dev, int64_t *accept_dev, int64_t *steps_dev,
                           double *seconds_out) {
  return rmhmc_sample_stats(ctx, n_iter, burn_in, L, eps, K, seed, chain_offset, theta0, mean_dev, var_dev, ess_dev, accept_dev,
                            steps_dev, seconds_out);
}

int rmhmc_int8_certificate(rmhmc_ctx *ctx, double *bound_out, int32_t *active_out) {
  if (!ctx) return RMHMC_ERR_INVALID;
  if (bound_out) *bound_out = 0.0;   /* the oracle is float64 throughout */
  if (active_out) *active_out = 0;
  return RMHMC_OK;
}

int rmhmc_set_progress(rmhmc_ctx *ctx, rmhmc_progress_fn fn, int64_t first, int64_t every, void *user) {
  (void)fn; (void)first; (void)every; (void)user;   /* the oracle runs its chains to the end in parallel loops and never reports */
  return ctx ? RMHMC_OK : RMHMC_ERR_INVALID;
}

/* Tuning options of the HIP library (include/rmhmc.h): scheduling switches that have no meaning for the serial restatement.  Accepted
 * and ignored, so that a test can hand both libraries the same option set. */
int rmhmc_create_opts(rmhmc_ctx **out, int32_t device_id, int64_t M, int32_t D, int64_t n_chains, int32_t dtype, uint32_t flags,
                      const rmhmc_option *opts, int32_t n_opts) {
  if (n_opts < 0 || (n_opts > 0 && !opts)) return fail(NULL, RMHMC_ERR_INVALID, "create_opts: bad option array");
  return rmhmc_create(out, device_id, M, D, n_chains, dtype, flags);
}
int rmhmc_set_option(rmhmc_ctx *ctx, const char *key, int64_t value) {
  (void)value;
  return (ctx && key) ? RMHMC_OK : RMHMC_ERR_INVALID;
}
int rmhmc_get_option(rmhmc_ctx *ctx, const char *key, int64_t *value_out) {
  if (!ctx || !key || !value_out) return RMHMC_ERR_INVALID;
  *value_out = 0;
  return RMHMC_OK;
}
int rmhmc_options(rmhmc_ctx *ctx, char *buf, size_t len) {
  if (!ctx || !buf || !len) return RMHMC_ERR_INVALID;
  buf[0] = 0;
  return RMHMC_OK;
}
