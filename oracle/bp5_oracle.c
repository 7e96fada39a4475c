/* CPU ORACLE (test infrastructure, NOT product code) -- C/OpenMP restatement of the BP5 hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PARITY UNPINNED by the reference (no golden vectors exist there; its arithmetic lives in
 * deal.II 9.2.0-pre, absent from /root/reference and from this image); pinned against
 * oracle/bp5_oracle.py, which is pinned by the known-answer tests of SURVEY.md Appendix A.7.
 *
 * It is a sum-factorised, OpenMP-over-cells implementation of the same mathematics and serves
 * as the "CPU path timed on the host cores" (labelled: CPU restatement, not deal.II).
 *
 * Reference call sites followed (read as text):
 *   merged metric planes/order ............ bp5/step-64.cu:84-114
 *   cell operator sequence ................ bp5/step-64.cu:147-194 (cell offset fixed)
 *   local numbering i + n(j + n k) ........ bp5/fe_evaluation_gl.h:139-142
 *   vmult = cell loop + Dirichlet copy .... bp5/step-64.cu:263-276
 *   RHS ................................... bp5/step-64.cu:372-418
 *   plain PCG ............................. bp5/step-64.cu:428-453 (deal.II SolverCG)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 10

/* ------------------------------------------------------------------ 1-D tables */
static void legendre(int n, double x, double *P, double *dP)
{ /* P_n(x), P_n'(x) on [-1,1] */
  double p0 = 1.0, p1 = x, d0 = 0.0, d1 = 1.0;
  if (n == 0) { *P = 1.0; *dP = 0.0; return; }
  for (int k = 2; k <= n; ++k) {
    double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    double d2 = d0 + (2 * k - 1) * p1;
    p0 = p1; p1 = p2; d0 = d1; d1 = d2;
  }
  *P = p1; *dP = d1;
}

static void gauss01(int n, double *x, double *w)
{
  for (int i = 0; i < n; ++i) {
    double z = -cos(M_PI * (i + 0.75) / (n + 0.5)), P, dP;
    for (int it = 0; it < 100; ++it) {
      legendre(n, z, &P, &dP);
      double dz = P / dP; z -= dz;
      if (fabs(dz) < 1e-16) break;
    }
    legendre(n, z, &P, &dP);
    x[i] = z; w[i] = 2.0 / ((1.0 - z * z) * dP * dP);
  }
  for (int i = 0; i < n / 2; ++i) { /* antisymmetrise */
    double a = 0.5 * (x[i] - x[n - 1 - i]); x[i] = a; x[n - 1 - i] = -a;
    double b = 0.5 * (w[i] + w[n - 1 - i]); w[i] = b; w[n - 1 - i] = b;
  }
  if (n % 2) x[n / 2] = 0.0;
  for (int i = 0; i < n; ++i) { x[i] = 0.5 * (x[i] + 1.0); w[i] *= 0.5; }
}

static void gll01(int n, double *x, double *w)
{ /* roots of (1-z^2) P'_{n-1}(z) */
  int m = n - 1;
  x[0] = -1.0; x[n - 1] = 1.0;
  for (int i = 1; i < n - 1; ++i) {
    double z = -cos(M_PI * i / m);
    for (int it = 0; it < 100; ++it) {
      double P, dP; legendre(m, z, &P, &dP);
      /* f = P'_m, f' = P''_m = (2 z P' - m(m+1) P)/(1-z^2) */
      double ddP = (2.0 * z * dP - m * (m + 1.0) * P) / (1.0 - z * z);
      double dz = dP / ddP; z -= dz;
      if (fabs(dz) < 1e-16) break;
    }
    x[i] = z;
  }
  for (int i = 0; i < n / 2; ++i) { double a = 0.5 * (x[i] - x[n - 1 - i]); x[i] = a; x[n - 1 - i] = -a; }
  if (n % 2) x[n / 2] = 0.0;
  for (int i = 0; i < n; ++i) {
    double P, dP; legendre(m, x[i], &P, &dP);
    w[i] = 2.0 / (m * (m + 1.0) * P * P);
  }
  for (int i = 0; i < n; ++i) { x[i] = 0.5 * (x[i] + 1.0); w[i] *= 0.5; }
}

/* N[q*n+i] = phi_i(x_q), D[q*n+i] = phi_i'(x_q) */
static void lagrange(int n, const double *nodes, const double *pts, double *N, double *D)
{
  for (int q = 0; q < n; ++q)
    for (int i = 0; i < n; ++i) {
      double den = 1.0, num = 1.0, s = 0.0;
      for (int m = 0; m < n; ++m) if (m != i) { den *= nodes[i] - nodes[m]; num *= pts[q] - nodes[m]; }
      for (int l = 0; l < n; ++l) {
        if (l == i) continue;
        double t = 1.0;
        for (int m = 0; m < n; ++m) if (m != i && m != l) t *= pts[q] - nodes[m];
        s += t;
      }
      N[q * n + i] = num / den; D[q * n + i] = s / den;
    }
}

/* quadrature: 0 = Gauss(p+1), 1 = GLL(p+1) */
int orc_tables(int p, int quadrature, double *nodes, double *pts, double *w, double *N, double *D)
{
  int n = p + 1; double wn[MAXN];
  if (p < 1 || n > MAXN) return 1;
  gll01(n, nodes, wn);
  if (quadrature == 1) gll01(n, pts, w); else gauss01(n, pts, w);
  lagrange(n, nodes, pts, N, D);
  if (quadrature == 1) for (int q = 0; q < n; ++q) for (int i = 0; i < n; ++i) N[q * n + i] = (q == i);
  return 0;
}

/* ------------------------------------------------------------------ tensor kernels
 * arrays are [k][j][i]; contraction along dim (0 = x fastest): out[..q..] = sum_i M[q*n+i] in[..i..]
 * transpose=1 uses M[i*n+q] (integration). */
static inline void contract(int n, int dim, int transpose, const double *M, const double *in, double *out, int add)
{
  int s = dim == 0 ? 1 : dim == 1 ? n : n * n;
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      int base = dim == 0 ? n * (b + n * a) : dim == 1 ? b + n * n * a : b + n * a;
      for (int q = 0; q < n; ++q) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) acc += (transpose ? M[i * n + q] : M[q * n + i]) * in[base + i * s];
        if (add) out[base + q * s] += acc; else out[base + q * s] = acc;
      }
    }
}

static void cell_grad(int n, const double *N, const double *D, const double *u, double *g0, double *g1, double *g2,
                      double *t0, double *t1)
{
  contract(n, 0, 0, D, u, t0, 0); contract(n, 1, 0, N, t0, t1, 0); contract(n, 2, 0, N, t1, g0, 0);
  contract(n, 0, 0, N, u, t0, 0); contract(n, 1, 0, D, t0, t1, 0); contract(n, 2, 0, N, t1, g1, 0);
  contract(n, 1, 0, N, t0, t1, 0); contract(n, 2, 0, D, t1, g2, 0);
}

static void cell_interp(int n, const double *N, const double *u, double *v, double *t0, double *t1)
{
  contract(n, 0, 0, N, u, t0, 0); contract(n, 1, 0, N, t0, t1, 0); contract(n, 2, 0, N, t1, v, 0);
}

static void cell_integrate_grad(int n, const double *N, const double *D, const double *g0, const double *g1,
                                const double *g2, double *y, double *t0, double *t1)
{
  contract(n, 2, 1, N, g0, t0, 0); contract(n, 1, 1, N, t0, t1, 0); contract(n, 0, 1, D, t1, y, 0);
  contract(n, 2, 1, N, g1, t0, 0); contract(n, 1, 1, D, t0, t1, 0); contract(n, 0, 1, N, t1, y, 1);
  contract(n, 2, 1, D, g2, t0, 0); contract(n, 1, 1, N, t0, t1, 0); contract(n, 0, 1, N, t1, y, 1);
}

static double kappa_eval(int mode, double x, double y, double z)
{ return mode == 1 ? 10.0 / (0.05 + 2.0 * (x * x + y * y + z * z)) : 1.0; }

/* ------------------------------------------------------------------ geometry
 * coords: [n_dofs][3]; outputs (any may be NULL):
 *   coef [6][n_cells][nq]  planes 00,11,22,01,02,12  = kappa*JxW*(K K^T)
 *   K    [n_cells][nq][3][3] (K[d][e] = d xi_d/d x_e),  JxW [n_cells][nq] */
int orc_geometry(int p, int quadrature, uint32_t n_cells, const uint32_t *l2g, const double *coords,
                 int kappa_mode, double *coef, double *Kout, double *JxWout)
{
  int n = p + 1, nq = n * n * n;
  double nodes[MAXN], pts[MAXN], w[MAXN], N[MAXN * MAXN], D[MAXN * MAXN];
  if (orc_tables(p, quadrature, nodes, pts, w, N, D)) return 1;
#pragma omp parallel
  {
    double *X = (double *)malloc(sizeof(double) * nq * 17);
    double *g = X + 3 * nq, *xq = g + 9 * nq, *t0 = xq + 3 * nq, *t1 = t0 + nq;
#pragma omp for schedule(static)
    for (int64_t c = 0; c < (int64_t)n_cells; ++c) {
      for (int e = 0; e < 3; ++e) {
        for (int i = 0; i < nq; ++i) X[e * nq + i] = coords[3 * (size_t)l2g[c * nq + i] + e];
        cell_grad(n, N, D, X + e * nq, g + (3 * e + 0) * nq, g + (3 * e + 1) * nq, g + (3 * e + 2) * nq, t0, t1);
        cell_interp(n, N, X + e * nq, xq + e * nq, t0, t1);
      }
      for (int q = 0; q < nq; ++q) {
        double J[3][3], K[3][3];
        for (int e = 0; e < 3; ++e) for (int d = 0; d < 3; ++d) J[e][d] = g[(3 * e + d) * nq + q];
        double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                     J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        double id = 1.0 / det;
        K[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id; K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
        K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id; K[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
        K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
        K[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id; K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
        K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
        int qi = q % n, qj = (q / n) % n, qk = q / (n * n);
        double jxw = fabs(det) * w[qi] * w[qj] * w[qk];
        if (Kout) for (int d = 0; d < 3; ++d) for (int e = 0; e < 3; ++e) Kout[((size_t)c * nq + q) * 9 + 3 * d + e] = K[d][e];
        if (JxWout) JxWout[(size_t)c * nq + q] = jxw;
        if (coef) {
          double s = jxw * kappa_eval(kappa_mode, xq[q], xq[nq + q], xq[2 * nq + q]);
          static const int pd[6] = {0, 1, 2, 0, 0, 1}, pe[6] = {0, 1, 2, 1, 2, 2};
          for (int pl = 0; pl < 6; ++pl) {
            int d = pd[pl], e = pe[pl];
            coef[((size_t)pl * n_cells + c) * nq + q] = s * (K[d][0] * K[e][0] + K[d][1] * K[e][1] + K[d][2] * K[e][2]);
          }
        }
      }
    }
    free(X);
  }
  return 0;
}

/* ------------------------------------------------------------------ operator
 * Deterministic scatter: per-cell results go to an E-vector, then each DoF sums its slots in a
 * fixed order through a transpose map built once (orc_plan_*). */
typedef struct {
  int p, quadrature, n, nq;
  uint32_t n_cells, n_dofs;
  const uint32_t *l2g;
  uint32_t *t_off;   /* [n_dofs+1] */
  uint32_t *t_slot;  /* [n_cells*nq] slot ids grouped by DoF */
  double *evec;      /* [n_cells*nq] */
  double N[MAXN * MAXN], D[MAXN * MAXN], w[MAXN], nodes[MAXN], pts[MAXN];
} orc_plan;

orc_plan *orc_plan_create(int p, int quadrature, uint32_t n_cells, uint32_t n_dofs, const uint32_t *l2g)
{
  orc_plan *P = (orc_plan *)calloc(1, sizeof(orc_plan));
  P->p = p; P->quadrature = quadrature; P->n = p + 1; P->nq = P->n * P->n * P->n;
  P->n_cells = n_cells; P->n_dofs = n_dofs; P->l2g = l2g;
  if (orc_tables(p, quadrature, P->nodes, P->pts, P->w, P->N, P->D)) { free(P); return NULL; }
  size_t ns = (size_t)n_cells * P->nq;
  P->t_off = (uint32_t *)calloc((size_t)n_dofs + 1, sizeof(uint32_t));
  P->t_slot = (uint32_t *)malloc(ns * sizeof(uint32_t));
  P->evec = (double *)malloc(ns * sizeof(double));
  for (size_t s = 0; s < ns; ++s) P->t_off[l2g[s] + 1]++;
  for (uint32_t i = 0; i < n_dofs; ++i) P->t_off[i + 1] += P->t_off[i];
  uint32_t *cur = (uint32_t *)malloc((size_t)n_dofs * sizeof(uint32_t));
  memcpy(cur, P->t_off, (size_t)n_dofs * sizeof(uint32_t));
  for (size_t s = 0; s < ns; ++s) P->t_slot[cur[l2g[s]]++] = (uint32_t)s;
  free(cur);
  return P;
}

void orc_plan_destroy(orc_plan *P)
{ if (!P) return; free(P->t_off); free(P->t_slot); free(P->evec); free(P); }

/* dst = sum_cells P^T B^T S B P src ; coef layout [6][n_cells][nq] */
void orc_apply(orc_plan *P, const double *coef, const double *src, double *dst)
{
  const int n = P->n, nq = P->nq;
  const size_t plane = (size_t)P->n_cells * nq;
#pragma omp parallel
  {
    double buf[8 * MAXN * MAXN * MAXN];
    double *u = buf, *g0 = u + nq, *g1 = g0 + nq, *g2 = g1 + nq, *t0 = g2 + nq, *t1 = t0 + nq;
#pragma omp for schedule(static)
    for (int64_t c = 0; c < (int64_t)P->n_cells; ++c) {
      const uint32_t *idx = P->l2g + c * nq;
      for (int i = 0; i < nq; ++i) u[i] = src[idx[i]];
      cell_grad(n, P->N, P->D, u, g0, g1, g2, t0, t1);
      const double *S = coef + (size_t)c * nq;
      for (int q = 0; q < nq; ++q) {
        double a = g0[q], b = g1[q], d = g2[q];
        double s00 = S[q], s11 = S[plane + q], s22 = S[2 * plane + q], s01 = S[3 * plane + q], s02 = S[4 * plane + q],
               s12 = S[5 * plane + q];
        g0[q] = s00 * a + s01 * b + s02 * d;
        g1[q] = s01 * a + s11 * b + s12 * d;
        g2[q] = s02 * a + s12 * b + s22 * d;
      }
      cell_integrate_grad(n, P->N, P->D, g0, g1, g2, P->evec + (size_t)c * nq, t0, t1);
    }
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t)P->n_dofs; ++i) {
      double acc = 0.0;
      for (uint32_t s = P->t_off[i]; s < P->t_off[i + 1]; ++s) acc += P->evec[P->t_slot[s]];
      dst[i] = acc;
    }
  }
}

/* PoissonOperator::vmult: cell loop + dst[c] = src[c] on constrained DoFs */
void orc_vmult(orc_plan *P, const double *coef, const uint32_t *constrained, uint32_t n_constrained, const double *src,
               double *dst)
{
  orc_apply(P, coef, src, dst);
  for (uint32_t i = 0; i < n_constrained; ++i) dst[constrained[i]] = src[constrained[i]];
}

/* b_i = sum_q phi_i(x_q) JxW(q), Gauss(p+1); constrained rows zero */
int orc_rhs(int p, uint32_t n_cells, uint32_t n_dofs, const uint32_t *l2g, const double *coords, const uint32_t *constrained,
            uint32_t n_constrained, double *b)
{
  int n = p + 1, nq = n * n * n;
  double nodes[MAXN], pts[MAXN], w[MAXN], N[MAXN * MAXN], D[MAXN * MAXN];
  if (orc_tables(p, 0, nodes, pts, w, N, D)) return 1;
  double *JxW = (double *)malloc(sizeof(double) * (size_t)n_cells * nq);
  orc_geometry(p, 0, n_cells, l2g, coords, 0, NULL, NULL, JxW);
  memset(b, 0, sizeof(double) * n_dofs);
  double y[MAXN * MAXN * MAXN], t0[MAXN * MAXN * MAXN], t1[MAXN * MAXN * MAXN];
  for (size_t c = 0; c < n_cells; ++c) {
    contract(n, 2, 1, N, JxW + c * nq, t0, 0); contract(n, 1, 1, N, t0, t1, 0); contract(n, 0, 1, N, t1, y, 0);
    for (int i = 0; i < nq; ++i) b[l2g[c * nq + i]] += y[i];
  }
  for (uint32_t i = 0; i < n_constrained; ++i) b[constrained[i]] = 0.0;
  free(JxW);
  return 0;
}

static double dot(size_t n, const double *a, const double *b)
{
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) s += a[i] * b[i];
  return s;
}

/* deal.II SolverCG recurrence with identity preconditioner (diag == 1), x0 = 0.
 * IterationNumberControl: stop at res <= tol or k == max_iter. */
int orc_cg_plain(orc_plan *P, const double *coef, const uint32_t *constrained, uint32_t n_constrained, const double *b,
                 double *x, int max_iter, double tol, int *iters, double *res_out)
{
  size_t n = P->n_dofs;
  double *g = (double *)malloc(3 * n * sizeof(double)), *d = g + n, *h = d + n;
  for (size_t i = 0; i < n; ++i) { x[i] = 0.0; g[i] = -b[i]; d[i] = b[i]; }
  double gh = dot(n, g, g), res = sqrt(gh);
  int k = 0;
  if (res > tol)
    for (;;) {
      ++k;
      orc_vmult(P, coef, constrained, n_constrained, d, h);
      double alpha = gh / dot(n, d, h);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < (int64_t)n; ++i) { x[i] += alpha * d[i]; g[i] += alpha * h[i]; }
      double gg = dot(n, g, g);
      res = sqrt(gg);
      if (res <= tol || k == max_iter) break;
      double beta = gg / gh; gh = gg;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < (int64_t)n; ++i) d[i] = beta * d[i] - g[i];
    }
  *iters = k; *res_out = res;
  free(g);
  return 0;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
