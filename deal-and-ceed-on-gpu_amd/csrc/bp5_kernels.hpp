// Device kernels of the BP5 hot path, hand-written for gfx950 (CDNA4, wave64).
//
//   apply_pencil_kernel   fused operator  dst += P^T B^T S B P src   (LocalPoissonOperator,
//                         bp5/step-64.cu:147-194 + FEEvaluation read/evaluate/integrate/distribute,
//                         bp5/fe_evaluation_gl.h:128-250)
//   metric/rhs/l2 kernels setup + post-processing on a generic 3-D thread layout
//                         (JacobianFunctor bp5/step-64.cu:84-114, assemble_rhs :372-418, :602-616)
//   blas1 / cg kernels    vector updates and fused dot products with device-resident scalars
//                         (bp5/solver.h:48-336, deal.II SolverCG)
//
// Design of the fused operator ("pencil" kernel)
//   * a cell is worked on by n^2 lanes (n = p+1); each lane owns one 1-D pencil of n values in
//     registers, so every 1-D contraction is an in-register n x n mat-vec whose matrix entries
//     are wave-uniform (kernel-argument tables -> SGPR operands);
//   * between the z-, y- and x-contractions the pencils are re-oriented through a padded LDS
//     tile (write n / read n per field) -- LDS traffic is 10n..11n doubles per lane instead of
//     the 18 n per lane of the one-thread-per-point scheme of the reference;
//   * a team is TW waves; with TW == 1 all synchronisation is wave-local (no s_barrier);
//   * gather / scatter happen in the z-owner orientation so consecutive lanes touch
//     consecutive local DoFs; the six metric planes are streamed in the x-owner orientation
//     from a layout permuted to match (see bp5.h: bp5_mf_compute_merged_metric).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bp5 {

// ------------------------------------------------------------------------------------ helpers
template <int n>
struct ShapeArg { // passed by value as a kernel argument: uniform -> scalar loads
  double N[n * n];
  double D[n * n];
};

struct ApplyArgs {
  const uint32_t *l2g;
  const double *coef;
  const double *src;
  double *dst;
  uint64_t plane_stride; // n_cells_total * n^3
  uint32_t cell_begin, cell_end;
  uint32_t n_teams;      // teams needed for the range
  uint32_t teams_per_xcd;
};

// In-register n x n mat-vec with wave-uniform matrix entries.  The 1-D tables are symmetric
// under x -> 1-x:  N[q][i] = N[n-1-q][n-1-i],  D[q][i] = -D[n-1-q][n-1-i]  (enforced bitwise by
// the host, bp5_host.cpp), so only the first half of each table is ever read: 26 instead of 50
// doubles at p = 4, which keeps the tables in SGPRs without spilling.
// out[q] (+)= sum_i M[q][i] in[i]   (TR: M[i][q]);  ANTI selects the antisymmetric table D.
template <int n, bool TR, bool ANTI, bool ADD>
__device__ __forceinline__ void mv_sym(const double *__restrict__ M, const double (&in)[n], double (&out)[n])
{
#pragma unroll
  for (int q = 0; q < n; ++q) {
    double acc = ADD ? out[q] : 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const int r = TR ? i : q, c = TR ? q : i;
      const int f = r * n + c, g = (n - 1 - r) * n + (n - 1 - c);
      const double m = (f <= g) ? M[f] : (ANTI ? -M[g] : M[g]);
      if (!ADD && i == 0) acc = m * in[0];
      else acc = fma(m, in[i], acc);
    }
    out[q] = acc;
  }
}
#define MV_N(M, in, out) mv_sym<n, false, false, false>(M, in, out)
#define MV_D(M, in, out) mv_sym<n, false, true, false>(M, in, out)
#define MV_NT(M, in, out) mv_sym<n, true, false, false>(M, in, out)
#define MV_DT(M, in, out) mv_sym<n, true, true, false>(M, in, out)
#define MV_NT_ADD(M, in, out) mv_sym<n, true, false, true>(M, in, out)
#define MV_DT_ADD(M, in, out) mv_sym<n, true, true, true>(M, in, out)

template <int TW>
__device__ __forceinline__ void team_sync()
{
  if constexpr (TW == 1) {
    // one wave: LDS operations of a wave execute in issue order; only the compiler must be
    // kept from moving LDS accesses across this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// LDS tile strides in doubles (tools/lds_stride_search.py): RS row, PS plane, CS cell slot
template <int n, int LPC> struct LdsLayout { static constexpr int RS = n, PS = n * n + 1, CS = 3 * n * (n * n + 1) + 1; };
template <> struct LdsLayout<2, 4> { static constexpr int RS = 2, PS = 5, CS = 36; };
template <> struct LdsLayout<2, 16> { static constexpr int RS = 2, PS = 4, CS = 24; };
template <> struct LdsLayout<3, 9> { static constexpr int RS = 3, PS = 18, CS = 169; };
template <> struct LdsLayout<3, 16> { static constexpr int RS = 3, PS = 12, CS = 112; };
template <> struct LdsLayout<4, 16> { static constexpr int RS = 4, PS = 19, CS = 240; };
template <> struct LdsLayout<5, 25> { static constexpr int RS = 5, PS = 25, CS = 377; };
template <> struct LdsLayout<5, 32> { static constexpr int RS = 5, PS = 25, CS = 375; };
template <> struct LdsLayout<6, 36> { static constexpr int RS = 6, PS = 36, CS = 648; };
template <> struct LdsLayout<7, 49> { static constexpr int RS = 7, PS = 52, CS = 1092; };
template <> struct LdsLayout<8, 64> { static constexpr int RS = 9, PS = 72, CS = 1728; };

__device__ __forceinline__ void atomic_add_f64(double *p, double v)
{
  // hardware global_atomic_add_f64 (no CAS loop); result unused
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------ fused operator
// P degree, COLL: quadrature == GLL (N == I), TW waves per team, LPC lanes per cell slot,
// TPB teams per block (TW > 1 requires TPB == 1), PF: prefetch all six planes before evaluate
template <int P, bool COLL, int TW, int LPC, int TPB, bool PF>
__global__ void __launch_bounds__(64 * TW * TPB) apply_pencil_kernel(ApplyArgs a, ShapeArg<P + 1> sh)
{
  constexpr int n = P + 1, n2 = n * n, n3 = n2 * n;
  constexpr int TEAM = 64 * TW;
  constexpr int CPT = TEAM / LPC;
  static_assert(LPC >= n2 && CPT >= 1, "lanes per cell");
  static_assert(TW == 1 || TPB == 1, "block-wide barrier needs one team per block");
  using L = LdsLayout<n, LPC>;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int tid = threadIdx.x;
  const int team_in_block = tid / TEAM;
  const int t = tid - team_in_block * TEAM;
  const int c = t / LPC, ab = t - c * LPC;
  // XCD-aware mapping: blocks are dealt round-robin over the 8 XCDs, give each XCD a
  // contiguous range of teams so x/y-neighbour cells share one L2 (speed only)
  const uint32_t blk = (blockIdx.x & 7u) * a.teams_per_xcd + (blockIdx.x >> 3);
  const uint32_t team = blk * TPB + team_in_block;
  const uint64_t cell_raw = (uint64_t)a.cell_begin + (uint64_t)team * CPT + c;
  const bool active = (ab < n2) && (c < CPT) && (team < a.n_teams) && (cell_raw < a.cell_end);
  // idle lanes (tail of the wave / of the cell range) mirror a valid lane: every load below is
  // unconditional and in bounds; only LDS writes and the final atomics are predicated
  const uint64_t cell = cell_raw < a.cell_end ? cell_raw : (uint64_t)a.cell_end - 1;
  const int abm = ab < n2 ? ab : ab % n2;
  const int a_ = abm % n, b_ = abm / n;
  double *T = lds + (team_in_block * CPT + (c < CPT ? c : 0)) * L::CS;
#define TL(f, k, j, i) T[(f) * (n * L::PS) + (k) * L::PS + (j) * L::RS + (i)]

  // ---- gather (z-owner: a_ = i, b_ = j; registers hold k)
  uint32_t idx[n];
  double u[n];
  const uint32_t *l2g_c = a.l2g + cell * n3 + abm;
#pragma unroll
  for (int k = 0; k < n; ++k) idx[k] = l2g_c[k * n2];
#pragma unroll
  for (int k = 0; k < n; ++k) u[k] = a.src[idx[k]];

  // ---- metric planes (x-owner: a_ = j, b_ = k; registers hold i), layout [c][cell][i][j+n k]
  const double *cf = a.coef + cell * n3 + abm;
  double S[PF ? 6 : 1][n];
  if constexpr (PF) {
#pragma unroll
    for (int pl = 0; pl < 6; ++pl)
#pragma unroll
      for (int i = 0; i < n; ++i) S[pl][i] = cf[pl * a.plane_stride + i * n2];
  }

  double g0[n], g1[n], g2[n];
  if constexpr (!COLL) {
    // z-pass in registers
    double aN[n], aD[n];
    MV_N(sh.N, u, aN);
    MV_D(sh.D, u, aD);
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) { TL(0, k, b_, a_) = aN[k]; TL(1, k, b_, a_) = aD[k]; }
    }
    team_sync<TW>();
    // y-owner: a_ = i, b_ = k
    double vN[n], vD[n];
#pragma unroll
    for (int j = 0; j < n; ++j) { vN[j] = TL(0, b_, j, a_); vD[j] = TL(1, b_, j, a_); }
    double c1[n], c2[n], c3[n];
    MV_N(sh.N, vN, c1);
    MV_D(sh.D, vN, c2);
    MV_N(sh.N, vD, c3);
    team_sync<TW>();
    if (active) {
#pragma unroll
      for (int j = 0; j < n; ++j) { TL(0, b_, j, a_) = c1[j]; TL(1, b_, j, a_) = c2[j]; TL(2, b_, j, a_) = c3[j]; }
    }
    team_sync<TW>();
    // x-owner: a_ = j, b_ = k
    double r1[n], r2[n], r3[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      r1[i] = TL(0, b_, a_, i);
      r2[i] = TL(1, b_, a_, i);
      r3[i] = TL(2, b_, a_, i);
    }
    MV_D(sh.D, r1, g0);
    MV_N(sh.N, r2, g1);
    MV_N(sh.N, r3, g2);
  } else {
    // collocation: g0 = Dx u, g1 = Dy u, g2 = Dz u
    double gz[n];
    MV_D(sh.D, u, gz);
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) { TL(0, k, b_, a_) = u[k]; TL(2, k, b_, a_) = gz[k]; }
    }
    team_sync<TW>();
    double vN[n], c2[n];
#pragma unroll
    for (int j = 0; j < n; ++j) vN[j] = TL(0, b_, j, a_);
    MV_D(sh.D, vN, c2);
    if (active) {
#pragma unroll
      for (int j = 0; j < n; ++j) TL(1, b_, j, a_) = c2[j];
    }
    team_sync<TW>();
    double r1[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      r1[i] = TL(0, b_, a_, i);
      g1[i] = TL(1, b_, a_, i);
      g2[i] = TL(2, b_, a_, i);
    }
    MV_D(sh.D, r1, g0);
  }

  // ---- quadrature-point operation: t = S ghat (symmetric 3x3, bp5/step-64.cu:166-177)
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s00, s11, s22, s01, s02, s12;
    if constexpr (PF) {
      s00 = S[0][i]; s11 = S[1][i]; s22 = S[2][i]; s01 = S[3][i]; s02 = S[4][i]; s12 = S[5][i];
    } else {
      s00 = cf[0 * a.plane_stride + i * n2];
      s11 = cf[1 * a.plane_stride + i * n2];
      s22 = cf[2 * a.plane_stride + i * n2];
      s01 = cf[3 * a.plane_stride + i * n2];
      s02 = cf[4 * a.plane_stride + i * n2];
      s12 = cf[5 * a.plane_stride + i * n2];
    }
    const double x0 = g0[i], x1 = g1[i], x2 = g2[i];
    g0[i] = s00 * x0 + s01 * x1 + s02 * x2;
    g1[i] = s01 * x0 + s11 * x1 + s12 * x2;
    g2[i] = s02 * x0 + s12 * x1 + s22 * x2;
  }

  // ---- integrate (transpose sequence)
  double y[n];
  if constexpr (!COLL) {
    double e1[n], e2[n], e3[n];
    MV_DT(sh.D, g0, e1);
    MV_NT(sh.N, g1, e2);
    MV_NT(sh.N, g2, e3);
    team_sync<TW>();
    if (active) {
#pragma unroll
      for (int i = 0; i < n; ++i) { TL(0, b_, a_, i) = e1[i]; TL(1, b_, a_, i) = e2[i]; TL(2, b_, a_, i) = e3[i]; }
    }
    team_sync<TW>();
    double w1[n], w2[n], w3[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
      w1[j] = TL(0, b_, j, a_);
      w2[j] = TL(1, b_, j, a_);
      w3[j] = TL(2, b_, j, a_);
    }
    double f1[n], f2[n];
    MV_NT(sh.N, w1, f1);
    MV_DT_ADD(sh.D, w2, f1);
    MV_NT(sh.N, w3, f2);
    team_sync<TW>();
    if (active) {
#pragma unroll
      for (int j = 0; j < n; ++j) { TL(0, b_, j, a_) = f1[j]; TL(1, b_, j, a_) = f2[j]; }
    }
    team_sync<TW>();
    double z1[n], z2[n];
#pragma unroll
    for (int k = 0; k < n; ++k) { z1[k] = TL(0, k, b_, a_); z2[k] = TL(1, k, b_, a_); }
    MV_NT(sh.N, z1, y);
    MV_DT_ADD(sh.D, z2, y);
  } else {
    double e1[n];
    MV_DT(sh.D, g0, e1);
    team_sync<TW>();
    if (active) {
#pragma unroll
      for (int i = 0; i < n; ++i) { TL(0, b_, a_, i) = e1[i]; TL(1, b_, a_, i) = g1[i]; TL(2, b_, a_, i) = g2[i]; }
    }
    team_sync<TW>();
    double w1[n], w2[n];
#pragma unroll
    for (int j = 0; j < n; ++j) { w1[j] = TL(0, b_, j, a_); w2[j] = TL(1, b_, j, a_); }
    MV_DT_ADD(sh.D, w2, w1);
    if (active) { // each y-owner lane rewrites only the column it has just read
#pragma unroll
      for (int j = 0; j < n; ++j) TL(0, b_, j, a_) = w1[j];
    }
    team_sync<TW>();
    double z2[n];
#pragma unroll
    for (int k = 0; k < n; ++k) { y[k] = TL(0, k, b_, a_); z2[k] = TL(2, k, b_, a_); }
    MV_DT_ADD(sh.D, z2, y);
  }

  // ---- scatter-add (distribute_local_to_global, bp5/fe_evaluation_gl.h:170-180)
  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) atomic_add_f64(a.dst + idx[k], y[k]);
  }
#undef TL
}

// ------------------------------------------------------------------------------------ generic 3-D layout
// One cell per block, one thread per (i,j,k); used by setup/post-processing kernels only.
// tab: device table block [N (n*n) | D (n*n) | w (n)].
template <int n>
struct Cell3 {
  static constexpr int n2 = n * n, n3 = n * n * n;
  // v = (Mz (x) My (x) Mx) src at this thread's point; TR uses the transposed matrices
  template <bool TR>
  static __device__ double tensor3(const double *Mx, const double *My, const double *Mz, const double *src, double *t1,
                                   double *t2, int i, int j, int k)
  {
    double acc = 0.0;
    for (int m = 0; m < n; ++m) acc += (TR ? Mx[m * n + i] : Mx[i * n + m]) * src[m + n * (j + n * k)];
    t1[i + n * (j + n * k)] = acc;
    __syncthreads();
    acc = 0.0;
    for (int m = 0; m < n; ++m) acc += (TR ? My[m * n + j] : My[j * n + m]) * t1[i + n * (m + n * k)];
    t2[i + n * (j + n * k)] = acc;
    __syncthreads();
    acc = 0.0;
    for (int m = 0; m < n; ++m) acc += (TR ? Mz[m * n + k] : Mz[k * n + m]) * t2[i + n * (j + n * m)];
    __syncthreads();
    return acc;
  }
};

struct GeomOut {
  double *coef;          // permuted merged metric or NULL
  uint64_t plane_stride; // n_cells * n3
  double *inv_jac;       // 9 planes of n_cells*pad, or NULL
  double *JxW;           // n_cells*pad or NULL
  double *q_points;      // 3 planes of n_cells*pad or NULL
  uint32_t pad;
  uint64_t geo_plane;    // n_cells * pad
};

__device__ __forceinline__ double kappa_eval(int mode, double x, double y, double z)
{
  return mode == 1 ? 10.0 / (0.05 + 2.0 * (x * x + y * y + z * z)) : 1.0;
}

// Jacobian at this thread's q-point from the cell's nodal coordinates (LDS X[3][n3])
template <int n>
__device__ void cell_jacobian(const double *tab, const double *X, double *t1, double *t2, int i, int j, int k, double (&J)[3][3],
                              double (&xq)[3])
{
  const double *N = tab, *D = tab + n * n;
  constexpr int n3 = n * n * n;
  for (int e = 0; e < 3; ++e) {
    J[e][0] = Cell3<n>::template tensor3<false>(D, N, N, X + e * n3, t1, t2, i, j, k);
    J[e][1] = Cell3<n>::template tensor3<false>(N, D, N, X + e * n3, t1, t2, i, j, k);
    J[e][2] = Cell3<n>::template tensor3<false>(N, N, D, X + e * n3, t1, t2, i, j, k);
    xq[e] = Cell3<n>::template tensor3<false>(N, N, N, X + e * n3, t1, t2, i, j, k);
  }
}

__device__ __forceinline__ double invert3(const double (&J)[3][3], double (&K)[3][3])
{
  const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                     J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  const double id = 1.0 / det;
  K[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id; K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id; K[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
  K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  K[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id; K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
  K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
  return det;
}

// == JacobianFunctor (bp5/step-64.cu:84-114) fused with the geometry part of MatrixFree::reinit
template <int n>
__global__ void __launch_bounds__(n *n *n) geometry_kernel(const uint32_t *l2g, const double *coords, const double *tab,
                                                          int kappa_mode, uint32_t n_cells, GeomOut o)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  __shared__ double X[3 * n3], t1[n3], t2[n3];
  const int i = threadIdx.x, j = threadIdx.y, k = threadIdx.z;
  const int q = i + n * (j + n * k);
  for (uint64_t cell = blockIdx.x; cell < n_cells; cell += gridDim.x) {
    const uint32_t g = l2g[cell * n3 + q];
    for (int e = 0; e < 3; ++e) X[e * n3 + q] = coords[3 * (uint64_t)g + e];
    __syncthreads();
    double J[3][3], K[3][3], xq[3];
    cell_jacobian<n>(tab, X, t1, t2, i, j, k, J, xq);
    const double det = invert3(J, K);
    const double *w = tab + 2 * n2;
    const double jxw = fabs(det) * w[i] * w[j] * w[k];
    if (o.coef) {
      const double s = jxw * kappa_eval(kappa_mode, xq[0], xq[1], xq[2]);
      double *c = o.coef + cell * n3 + (uint64_t)i * n2 + (j + n * k); // permuted: x slowest
      c[0 * o.plane_stride] = s * (K[0][0] * K[0][0] + K[0][1] * K[0][1] + K[0][2] * K[0][2]);
      c[1 * o.plane_stride] = s * (K[1][0] * K[1][0] + K[1][1] * K[1][1] + K[1][2] * K[1][2]);
      c[2 * o.plane_stride] = s * (K[2][0] * K[2][0] + K[2][1] * K[2][1] + K[2][2] * K[2][2]);
      c[3 * o.plane_stride] = s * (K[0][0] * K[1][0] + K[0][1] * K[1][1] + K[0][2] * K[1][2]);
      c[4 * o.plane_stride] = s * (K[0][0] * K[2][0] + K[0][1] * K[2][1] + K[0][2] * K[2][2]);
      c[5 * o.plane_stride] = s * (K[1][0] * K[2][0] + K[1][1] * K[2][1] + K[1][2] * K[2][2]);
    }
    if (o.inv_jac) {
      const uint64_t at = cell * o.pad + q;
      for (int d = 0; d < 3; ++d)
        for (int e = 0; e < 3; ++e) o.inv_jac[(uint64_t)(3 * d + e) * o.geo_plane + at] = K[d][e];
      o.JxW[at] = jxw;
      for (int e = 0; e < 3; ++e) o.q_points[(uint64_t)e * o.geo_plane + at] = xq[e];
    }
    __syncthreads();
  }
}

// permute merged metric between the device layout (x slowest) and the reference layout
template <int n>
__global__ void metric_permute_kernel(const double *in, double *out, uint64_t total /*6*n_cells*n3*/)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  for (uint64_t o = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; o < total; o += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t cellplane = o / n3;
    const int q = (int)(o - cellplane * n3); // reference index qi + n(qj + n qk)
    const int qi = q % n, rest = q / n;      // rest = qj + n qk
    out[o] = in[cellplane * n3 + (uint64_t)qi * n2 + rest];
  }
}

// b_i = sum_q phi_i(x_q) JxW(q), Gauss tables (assemble_rhs, bp5/step-64.cu:372-418)
template <int n>
__global__ void __launch_bounds__(n *n *n) rhs_kernel(const uint32_t *l2g, const double *coords, const double *tab_gauss,
                                                     uint32_t n_cells, double *b)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  __shared__ double X[3 * n3], t1[n3], t2[n3], v[n3];
  const int i = threadIdx.x, j = threadIdx.y, k = threadIdx.z;
  const int q = i + n * (j + n * k);
  for (uint64_t cell = blockIdx.x; cell < n_cells; cell += gridDim.x) {
    const uint32_t g = l2g[cell * n3 + q];
    for (int e = 0; e < 3; ++e) X[e * n3 + q] = coords[3 * (uint64_t)g + e];
    __syncthreads();
    double J[3][3], K[3][3], xq[3];
    cell_jacobian<n>(tab_gauss, X, t1, t2, i, j, k, J, xq);
    const double det = invert3(J, K);
    const double *w = tab_gauss + 2 * n2;
    v[q] = fabs(det) * w[i] * w[j] * w[k];
    __syncthreads();
    const double *N = tab_gauss;
    const double y = Cell3<n>::template tensor3<true>(N, N, N, v, t1, t2, i, j, k);
    atomic_add_f64(b + g, y);
    __syncthreads();
  }
}

// sum_cells sum_q (u_h(x_q))^2 JxW  -> *out (atomic)
template <int n>
__global__ void __launch_bounds__(n *n *n) l2norm_kernel(const uint32_t *l2g, const double *coords, const double *tab_gauss,
                                                        uint32_t n_cells, const double *u, double *out)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  __shared__ double X[3 * n3], t1[n3], t2[n3], v[n3];
  const int i = threadIdx.x, j = threadIdx.y, k = threadIdx.z;
  const int q = i + n * (j + n * k);
  double acc = 0.0;
  for (uint64_t cell = blockIdx.x; cell < n_cells; cell += gridDim.x) {
    const uint32_t g = l2g[cell * n3 + q];
    for (int e = 0; e < 3; ++e) X[e * n3 + q] = coords[3 * (uint64_t)g + e];
    v[q] = u[g];
    __syncthreads();
    double J[3][3], K[3][3], xq[3];
    cell_jacobian<n>(tab_gauss, X, t1, t2, i, j, k, J, xq);
    const double det = invert3(J, K);
    const double *w = tab_gauss + 2 * n2;
    const double *N = tab_gauss;
    const double uq = Cell3<n>::template tensor3<false>(N, N, N, v, t1, t2, i, j, k);
    acc += uq * uq * fabs(det) * w[i] * w[j] * w[k];
    __syncthreads();
  }
  t1[q] = acc;
  __syncthreads();
  if (q == 0) {
    double s = 0.0;
    for (int m = 0; m < n3; ++m) s += t1[m];
    atomic_add_f64(out, s);
  }
}

// ------------------------------------------------------------------------------------ small vector kernels
__global__ void copy_constrained_kernel(const uint32_t *cdofs, uint32_t n, const double *src, double *dst)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t c = cdofs[i]; dst[c] = src[c]; }
}
__global__ void set_constrained_kernel(const uint32_t *cdofs, uint32_t n, double val, double *dst)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[cdofs[i]] = val;
}
__global__ void pack_kernel(const uint32_t *idx, uint32_t n, const double *v, double *buf)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = v[idx[i]];
}
__global__ void unpack_add_kernel(const uint32_t *idx, uint32_t n, const double *buf, double *v)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[idx[i]] += buf[i]; // indices of one neighbour are distinct
}

constexpr int VB = 256;      // threads per block of streaming kernels
constexpr int MAXBLK = 2048; // grid cap; also the length of each partial-sum row

// mode 0: y = value ; 1: y += a x ; 2: y = a x ; 3: y = s y + a x
template <int MODE>
__global__ void __launch_bounds__(VB) vec_kernel(double *y, const double *x, double s, double a, size_t n)
{
  const size_t stride = (size_t)gridDim.x * VB * 2;
  for (size_t i = ((size_t)blockIdx.x * VB + threadIdx.x) * 2; i < n; i += stride) {
    if (i + 1 < n) {
      double2 yv = *reinterpret_cast<double2 *>(y + i);
      double2 xv = MODE ? *reinterpret_cast<const double2 *>(x + i) : double2{0, 0};
      if (MODE == 0) { yv.x = s; yv.y = s; }
      if (MODE == 1) { yv.x += a * xv.x; yv.y += a * xv.y; }
      if (MODE == 2) { yv.x = a * xv.x; yv.y = a * xv.y; }
      if (MODE == 3) { yv.x = s * yv.x + a * xv.x; yv.y = s * yv.y + a * xv.y; }
      *reinterpret_cast<double2 *>(y + i) = yv;
    } else {
      if (MODE == 0) y[i] = s;
      if (MODE == 1) y[i] += a * x[i];
      if (MODE == 2) y[i] = a * x[i];
      if (MODE == 3) y[i] = s * y[i] + a * x[i];
    }
  }
}

// block reduction of K running sums; wave64 shuffles then LDS across waves (fixed order ->
// bitwise reproducible for a fixed grid)
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&acc)[K], double *partials /*[K][MAXBLK]*/)
{
  __shared__ double red[K][VB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < VB / 64; ++w) s += red[threadIdx.x][w];
    partials[threadIdx.x * MAXBLK + blockIdx.x] = s;
  }
}

// sums partials[K][nblk] in a fixed tree order into out[K]
template <int K>
__global__ void __launch_bounds__(VB) finalize_kernel(const double *partials, int nblk, double *out, const int *state)
{
  if (state && state[0]) return;
  __shared__ double red[VB];
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += VB) s += partials[k * MAXBLK + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = VB / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = red[0];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(VB) dot_kernel(const double *x, const double *y, size_t n, double *partials)
{
  double acc[1] = {0.0};
  const size_t stride = (size_t)gridDim.x * VB * 2;
  for (size_t i = ((size_t)blockIdx.x * VB + threadIdx.x) * 2; i < n; i += stride) {
    if (i + 1 < n) {
      const double2 a = *reinterpret_cast<const double2 *>(x + i), b = *reinterpret_cast<const double2 *>(y + i);
      acc[0] += a.x * b.x + a.y * b.y;
    } else
      acc[0] += x[i] * y[i];
  }
  block_reduce_store<1>(acc, partials);
}

// ------------------------------------------------------------------------------------ CG kernels
// Device-resident solver state.  sc[] doubles, st[] ints.
enum { SC_GH = 0, SC_DH, SC_GG, SC_GDG, SC_ALPHA, SC_BETA, SC_ALPHA_OLD, SC_BETA_OLD, SC_RES, SC_RES0, SC_TOL, SC_R0 /* 7 merged dots R0..R6 */,
       SC_COUNT = SC_R0 + 7 };
enum { ST_DONE = 0, ST_ITER, ST_PENDING, ST_MAXIT, ST_BREAKDOWN, ST_COUNT };

// ---- plain CG (deal.II SolverCG, Appendix A.5)
// init: g = -b ; d = b (= -D g with D = 1) ; x = 0 ; partial sums of g.g and g.Dg
__global__ void __launch_bounds__(VB) cg_init_kernel(const double *b, const double *diag, double *x, double *g, double *d,
                                                    size_t n, double *partials)
{
  double acc[2] = {0.0, 0.0};
  const size_t stride = (size_t)gridDim.x * VB;
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += stride) {
    const double gi = -b[i], hi = diag ? diag[i] * gi : gi;
    x[i] = 0.0; g[i] = gi; d[i] = -hi;
    acc[0] += gi * gi; acc[1] += gi * hi;
  }
  block_reduce_store<2>(acc, partials);
}
// after init: res0 = sqrt(gg); gh = gDg; done if res0 <= tol
__global__ void cg_init_control_kernel(double *sc, int *st)
{
  sc[SC_RES0] = sc[SC_RES] = sqrt(sc[SC_GG]);
  sc[SC_GH] = sc[SC_GDG];
  st[ST_ITER] = 0; st[ST_PENDING] = 0; st[ST_BREAKDOWN] = 0;
  st[ST_DONE] = (sc[SC_RES] <= sc[SC_TOL]) ? 1 : 0;
}
// x += alpha d ; g += alpha h ; partial sums g.g, g.Dg   with alpha = gh / dh
__global__ void __launch_bounds__(VB) cg_update_kernel(double *x, double *g, const double *d, const double *h, const double *diag,
                                                      size_t n, const double *sc, const int *st, double *partials)
{
  if (st[ST_DONE]) return;
  const double alpha = sc[SC_GH] / sc[SC_DH];
  double acc[2] = {0.0, 0.0};
  const size_t stride = (size_t)gridDim.x * VB * 2;
  for (size_t i = ((size_t)blockIdx.x * VB + threadIdx.x) * 2; i < n; i += stride) {
    if (i + 1 < n) {
      double2 xv = *reinterpret_cast<double2 *>(x + i), gv = *reinterpret_cast<double2 *>(g + i);
      const double2 dv = *reinterpret_cast<const double2 *>(d + i), hv = *reinterpret_cast<const double2 *>(h + i);
      xv.x += alpha * dv.x; xv.y += alpha * dv.y;
      gv.x += alpha * hv.x; gv.y += alpha * hv.y;
      *reinterpret_cast<double2 *>(x + i) = xv;
      *reinterpret_cast<double2 *>(g + i) = gv;
      const double z0 = diag ? diag[i] * gv.x : gv.x, z1 = diag ? diag[i + 1] * gv.y : gv.y;
      acc[0] += gv.x * gv.x + gv.y * gv.y;
      acc[1] += gv.x * z0 + gv.y * z1;
    } else {
      const double xi = x[i] + alpha * d[i], gi = g[i] + alpha * h[i];
      x[i] = xi; g[i] = gi;
      acc[0] += gi * gi; acc[1] += gi * (diag ? diag[i] * gi : gi);
    }
  }
  block_reduce_store<2>(acc, partials);
}
// res = sqrt(gg); ++it; stop test; beta = gDg / gh ; gh = gDg
__global__ void cg_control_kernel(double *sc, int *st)
{
  if (st[ST_DONE]) return;
  const double dh = sc[SC_DH];
  if (!(dh == dh) || dh == 0.0) { st[ST_BREAKDOWN] = 1; st[ST_DONE] = 1; return; }
  const double res = sqrt(sc[SC_GG]);
  sc[SC_RES] = res;
  sc[SC_ALPHA] = sc[SC_GH] / dh;
  const int it = ++st[ST_ITER];
  if (res <= sc[SC_TOL] || it >= st[ST_MAXIT]) { st[ST_DONE] = 1; return; }
  sc[SC_BETA] = sc[SC_GDG] / sc[SC_GH];
  sc[SC_GH] = sc[SC_GDG];
}
// d = beta d - D g
__global__ void __launch_bounds__(VB) cg_direction_kernel(double *d, const double *g, const double *diag, size_t n, const double *sc,
                                                         const int *st)
{
  if (st[ST_DONE]) return;
  const double beta = sc[SC_BETA];
  const size_t stride = (size_t)gridDim.x * VB * 2;
  for (size_t i = ((size_t)blockIdx.x * VB + threadIdx.x) * 2; i < n; i += stride) {
    if (i + 1 < n) {
      double2 dv = *reinterpret_cast<double2 *>(d + i);
      const double2 gv = *reinterpret_cast<const double2 *>(g + i);
      dv.x = beta * dv.x - (diag ? diag[i] * gv.x : gv.x);
      dv.y = beta * dv.y - (diag ? diag[i + 1] * gv.y : gv.y);
      *reinterpret_cast<double2 *>(d + i) = dv;
    } else
      d[i] = beta * d[i] - (diag ? diag[i] * g[i] : g[i]);
  }
}

// ---- merged CG (SolverCGFullMerge, bp5/solver.h), schedule fixed
// MODE 0: update_a0 (solver.h:48-72)    p = -D r ; v = 0
// MODE 1: update_a<false> (:74-104)     r += alpha v ; p = beta p - D r ; v = 0
// MODE 2: update_a1 (:106-140)          x += (alpha + ao/bo) p + (ao/bo) D r_old ; then as MODE 1
// If the solve finished in the previous iteration with an x update pending, the kernel performs
// the epilogue (solver.h:510-526) instead: odd it: x += alpha p ; even it: update_c (:315-336).
template <int MODE>
__global__ void __launch_bounds__(VB) cgm_update_kernel(double *p, double *r, double *v, double *x, const double *diag, size_t n,
                                                       const double *sc, const int *st)
{
  const bool done = st[ST_DONE];
  if (done && !st[ST_PENDING]) return;
  const double alpha = sc[SC_ALPHA], beta = sc[SC_BETA];
  const double aob = (MODE == 2 || done) ? sc[SC_ALPHA_OLD] / sc[SC_BETA_OLD] : 0.0;
  const bool epi_odd = done && (st[ST_ITER] & 1);
  const size_t stride = (size_t)gridDim.x * VB;
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += stride) {
    const double di = diag ? diag[i] : 1.0;
    if (done) {
      if (epi_odd) x[i] += alpha * p[i];
      else x[i] += (alpha + aob) * p[i] + aob * di * r[i];
      continue;
    }
    if (MODE == 0) {
      p[i] = -di * r[i];
      v[i] = 0.0;
    } else {
      const double r_old = r[i], p_old = p[i];
      if (MODE == 2) x[i] += (alpha + aob) * p_old + aob * di * r_old;
      const double rn = r_old + alpha * v[i];
      r[i] = rn;
      p[i] = beta * p_old - di * rn;
      v[i] = 0.0;
    }
  }
}
// update_b (solver.h:142-311): [p.v, v.v, r.v, r.r, r.Dv, v.Dv, r.Dr]
__global__ void __launch_bounds__(VB) cgm_dots_kernel(const double *p, const double *r, const double *v, const double *diag, size_t n,
                                                     const int *st, double *partials)
{
  if (st[ST_DONE]) return;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * VB;
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += stride) {
    const double pi = p[i], ri = r[i], vi = v[i], di = diag ? diag[i] : 1.0;
    acc[0] += pi * vi; acc[1] += vi * vi; acc[2] += ri * vi; acc[3] += ri * ri;
    acc[4] += ri * di * vi; acc[5] += vi * di * vi; acc[6] += ri * di * ri;
  }
  block_reduce_store<7>(acc, partials);
}
// scalars of one merged iteration (solver.h:496-506,533)
__global__ void cgm_control_kernel(double *sc, int *st)
{
  if (st[ST_DONE]) { st[ST_PENDING] = 0; return; }
  const double *R = sc + SC_R0;
  if (!(R[0] == R[0]) || R[0] == 0.0) { st[ST_BREAKDOWN] = 1; st[ST_DONE] = 1; return; }
  sc[SC_ALPHA_OLD] = sc[SC_ALPHA];
  sc[SC_BETA_OLD] = sc[SC_BETA];
  const double alpha = R[6] / R[0];
  sc[SC_ALPHA] = alpha;
  const double r2 = R[3] + 2.0 * alpha * R[2] + alpha * alpha * R[1];
  const double res = sqrt(r2 > 0.0 ? r2 : 0.0);
  sc[SC_RES] = res;
  const int it = ++st[ST_ITER];
  if (res <= sc[SC_TOL] || it >= st[ST_MAXIT]) { st[ST_DONE] = 1; st[ST_PENDING] = 1; return; }
  sc[SC_BETA] = alpha * (R[4] + alpha * R[5]) / R[6];
}
// merged init: r = -b, x = 0, partial r.r
__global__ void __launch_bounds__(VB) cgm_init_kernel(const double *b, double *x, double *r, double *p, double *v, size_t n,
                                                     double *partials)
{
  double acc[2] = {0.0, 0.0};
  const size_t stride = (size_t)gridDim.x * VB;
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += stride) {
    const double ri = -b[i];
    x[i] = 0.0; r[i] = ri; p[i] = 0.0; v[i] = 0.0;
    acc[0] += ri * ri;
  }
  acc[1] = acc[0];
  block_reduce_store<2>(acc, partials);
}
__global__ void cgm_init_control_kernel(double *sc, int *st)
{
  sc[SC_RES0] = sc[SC_RES] = sqrt(sc[SC_GG]);
  sc[SC_ALPHA] = sc[SC_BETA] = sc[SC_ALPHA_OLD] = sc[SC_BETA_OLD] = 0.0;
  st[ST_ITER] = 0; st[ST_PENDING] = 0; st[ST_BREAKDOWN] = 0;
  st[ST_DONE] = (sc[SC_RES] <= sc[SC_TOL]) ? 1 : 0;
}

} // namespace bp5
