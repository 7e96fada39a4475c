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
#include <type_traits>

namespace bp5 {

// ------------------------------------------------------------------------------------ helpers
template <int n>
struct ShapeArg { // passed by value as a kernel argument: uniform -> scalar loads
  double N[n * n];
  double D[n * n];
};

// ---------------------------------------------------------------------------------- device layout of per-q-point data
// A cell's n^3 values of one plane (merged metric, affine scalar plane) are consumed in the orientation where lane
// ab = qj + n qk owns the x-pencil qi = 0..n-1.  "Pair layout": the pencil values are stored as pairs (qi, qi+1), lane
// after lane, so that a lane fetches 16 bytes per load and a wave's load is one contiguous run (n^2 * 16 bytes per cell);
// the unpaired last qi of an odd n follows as n^2 single values:
//   offset(qi, ab) = (qi/2) * 2 n^2 + 2 ab + (qi & 1)      for qi < 2 (n/2)
//                  = (n/2) * 2 n^2 + ab                     for the last qi of odd n
// Measured on the block kernel (p = 4): 18 instead of 30 metric load instructions per lane and cell, -4.5 % kernel time.
template <int n>
__host__ __device__ constexpr int coef_off(int qi, int ab)
{
  return (qi < 2 * (n / 2)) ? (qi / 2) * (2 * n * n) + 2 * ab + (qi & 1) : (n / 2) * (2 * n * n) + ab;
}
typedef double bp5_d2u __attribute__((ext_vector_type(2), aligned(8))); // 16-byte load from an 8-byte aligned address
// the pencil of lane ab from one cell plane (base = plane + cell * n^3)
template <int n, bool NT = false>
__device__ __forceinline__ void load_pencil(const double *base, int ab, double (&S)[n])
{
#pragma unroll
  for (int m = 0; m < n / 2; ++m) {
    const bp5_d2u *q = reinterpret_cast<const bp5_d2u *>(base + m * (2 * n * n) + 2 * ab);
    const bp5_d2u v = NT ? __builtin_nontemporal_load(q) : *q;
    S[2 * m] = v.x;
    S[2 * m + 1] = v.y;
  }
  if constexpr (n & 1) S[n - 1] = NT ? __builtin_nontemporal_load(base + (n / 2) * (2 * n * n) + ab) : base[(n / 2) * (2 * n * n) + ab];
}

// pieces of load_pencil: pair m (qi = 2m, 2m + 1) and the unpaired last qi of an odd n -- the rolling metric prefetch of the block kernel
// (BlockPass::ROLL) refills a pencil piece by piece, each piece right after the quadrature-point loop has consumed it
template <int n, bool NT = false>
__device__ __forceinline__ void load_pencil_pair(const double *base, int ab, int m, double &s0, double &s1)
{
  const bp5_d2u *q = reinterpret_cast<const bp5_d2u *>(base + m * (2 * n * n) + 2 * ab);
  const bp5_d2u v = NT ? __builtin_nontemporal_load(q) : *q;
  s0 = v.x;
  s1 = v.y;
}
template <int n, bool NT = false>
__device__ __forceinline__ double load_pencil_tail(const double *base, int ab)
{
  return NT ? __builtin_nontemporal_load(base + (n / 2) * (2 * n * n) + ab) : base[(n / 2) * (2 * n * n) + ab];
}

struct ApplyArgs {
  const uint32_t *l2g;
  const double *coef;
  const double *src;
  double *dst;
  uint64_t plane_stride; // doubles between two planes of one cell: n_cells_total * n^3 (plane-major) or n^3 (cell-major)
  uint64_t cell_stride;  // doubles between two cells of one plane: n^3 (plane-major, affine scalar plane) or 6 n^3 (cell-major)
  uint32_t cell_begin, cell_end;
  uint32_t n_teams;      // teams needed for the range
  uint32_t teams_per_xcd;
  const double *gcell;   // affine mode: [6][n_cells] per-cell K K^T (planes 00,11,22,01,02,12); coef = 1 scalar plane
  uint32_t n_cells_total;
  // hanging nodes (builds with ABL & 2097152): per-cell constraint mask (bp5.h BP5_HANG_*) and the two 1-D interpolation
  // matrices I[h][a][b] = phi_b(xi_a / 2 + h / 2)
  const uint32_t *hang_mask;
  const double *hang_I;
};

// constraint-mask decoding shared by every kernel that gathers or scatters through local_to_global (bp5.h: BP5_HANG_*):
// bits 0-2 the face normal to x / y / z is constrained (any subset), bits 3-5 the cell's position in its parent along x / y / z
// (locates the constrained faces / edges: xi = 0 or 1), bits 6-8 the same position as "upper half" selector of the interpolation
// ALONG x / y / z, bits 9-11 the edge along x / y / z at the corner named by bits 3-5 is constrained on its own.
// The fix-up is one 1-D interpolation per direction d on the cell-local lines along d that lie on a constrained face tangential
// to d or on the constrained edge along d.
__device__ __forceinline__ bool hang_dir_any(uint32_t m, int d)
{
  const int e1 = d == 0 ? 1 : 0, e2 = d == 2 ? 1 : 2;
  return ((m >> e1) | (m >> e2) | (m >> (9 + d))) & 1u;
}
__device__ __forceinline__ bool hang_on_line(uint32_t m, int d, int i, int j, int k, int last)
{
  const int idx[3] = {i, j, k};
  const int e1 = d == 0 ? 1 : 0, e2 = d == 2 ? 1 : 2;
  const bool s1 = idx[e1] == (int)((m >> (3 + e1)) & 1u) * last, s2 = idx[e2] == (int)((m >> (3 + e2)) & 1u) * last;
  return (((m >> e1) & 1u) && s1) || (((m >> e2) & 1u) && s2) || (((m >> (9 + d)) & 1u) && s1 && s2);
}
#define BP5_HANG_ANY 0xe07u // FACE and EDGE bits

// In-register n x n mat-vec with wave-uniform matrix entries.  The 1-D tables are symmetric
// under x -> 1-x:  N[q][i] = N[n-1-q][n-1-i],  D[q][i] = -D[n-1-q][n-1-i]  (enforced bitwise by
// the host, bp5_host.cpp), so only the first half of each table is ever read: 26 instead of 50
// doubles at p = 4, which keeps the tables in SGPRs without spilling.
// out[q] (+)= sum_i M[q][i] in[i]   (TR: M[i][q]);  ANTI selects the antisymmetric table D.
// n >= EO_MIN_N (p >= 6): even-odd form of the same product.  With e_i = in_i + in_{n-1-i}, o_i = in_i - in_{n-1-i} a
// table that is (anti)symmetric under reversal of both indices needs ceil(n/2) x floor(n/2) + floor(n/2)^2 multiply-adds
// instead of n^2 (n = 7: 25 + 12 additions against 49; n = 9: 41 + 16 against 81) -- the deterministic block kernel is
// ALU-bound in evaluate / integrate at these degrees (profiles/r2: p = 6 cycle stamps; measured on one box, on / off: block kernel
// p = 6 1.358 / 1.471 ms, p = 7 1.154 / 1.236 ms, pencil kernel p = 8 1.085 / 1.190 ms; p = 5 (n = 6) loses: 1.716 / 1.527 ms).  Table layout (host: pack_even_odd in
// bp5_device.hpp), c = ceil(n/2), h = floor(n/2), m = (n-1)/2:  P[c][h], S[h][h], mid[c] with
//   E[q][i] = (M[q][i] + M[q][n-1-i]) / 2,  O[q][i] = (M[q][i] - M[q][n-1-i]) / 2,  mid[q] = M[q][m];
//   symmetric table: P = E, S = O;  antisymmetric table: P = O, S = E.
#ifndef BP5_EO_MIN_N
#define BP5_EO_MIN_N 7
#endif
constexpr int EO_MIN_N = BP5_EO_MIN_N;
template <int n, bool TR, bool ANTI, bool ADD>
__device__ __forceinline__ void mv_even_odd(const double *__restrict__ M, const double (&in)[n], double (&out)[n])
{
  constexpr int c = (n + 1) / 2, h = n / 2, m = (n - 1) / 2;
  constexpr bool odd = (n & 1) != 0;
  const double *__restrict__ P = M, *__restrict__ S = M + c * h, *__restrict__ mid = M + c * h + h * h;
  double e[h], o[h];
#pragma unroll
  for (int i = 0; i < h; ++i) {
    e[i] = in[i] + in[n - 1 - i];
    o[i] = in[i] - in[n - 1 - i];
  }
  if constexpr (!TR) {
#pragma unroll
    for (int q = 0; q < h; ++q) {
      double pv = P[q * h] * (ANTI ? o[0] : e[0]), sv = S[q * h] * (ANTI ? e[0] : o[0]);
#pragma unroll
      for (int i = 1; i < h; ++i) {
        pv = fma(P[q * h + i], ANTI ? o[i] : e[i], pv);
        sv = fma(S[q * h + i], ANTI ? e[i] : o[i], sv);
      }
      if constexpr (odd) {
        if constexpr (ANTI) sv = fma(mid[q], in[m], sv);
        else pv = fma(mid[q], in[m], pv);
      }
      const double r0 = pv + sv, r1 = pv - sv; // symmetric: A = pv, B = sv: A - B; antisymmetric: A = sv, B = pv: B - A
      out[q] = ADD ? out[q] + r0 : r0;
      out[n - 1 - q] = ADD ? out[n - 1 - q] + r1 : r1;
    }
    if constexpr (odd) {
      double pv = P[m * h] * (ANTI ? o[0] : e[0]);
#pragma unroll
      for (int i = 1; i < h; ++i) pv = fma(P[m * h + i], ANTI ? o[i] : e[i], pv);
      if constexpr (!ANTI) pv = fma(mid[m], in[m], pv);
      out[m] = ADD ? out[m] + pv : pv;
    }
  } else {
#pragma unroll
    for (int i = 0; i < h; ++i) {
      double av = P[i] * e[0], bv = S[i] * o[0];
#pragma unroll
      for (int q = 1; q < h; ++q) {
        av = fma(P[q * h + i], e[q], av);
        bv = fma(S[q * h + i], o[q], bv);
      }
      if constexpr (odd) av = fma(P[m * h + i], in[m], av);
      const double r0 = av + bv, r1 = ANTI ? bv - av : av - bv;
      out[i] = ADD ? out[i] + r0 : r0;
      out[n - 1 - i] = ADD ? out[n - 1 - i] + r1 : r1;
    }
    if constexpr (odd) {
      double v = mid[0] * (ANTI ? o[0] : e[0]);
#pragma unroll
      for (int q = 1; q < h; ++q) v = fma(mid[q], ANTI ? o[q] : e[q], v);
      if constexpr (!ANTI) v = fma(mid[m], in[m], v);
      out[m] = ADD ? out[m] + v : v;
    }
  }
}

template <int n, bool TR, bool ANTI, bool ADD>
__device__ __forceinline__ void mv_sym(const double *__restrict__ M, const double (&in)[n], double (&out)[n])
{
  if constexpr (n >= EO_MIN_N) {
    mv_even_odd<n, TR, ANTI, ADD>(M, in, out);
    return;
  }
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
template <> struct LdsLayout<7, 64> { static constexpr int RS = 7, PS = 52, CS = 1092; };
template <> struct LdsLayout<6, 64> { static constexpr int RS = 6, PS = 36, CS = 648; };
template <> struct LdsLayout<8, 64> { static constexpr int RS = 9, PS = 72, CS = 1728; };

__device__ __forceinline__ void atomic_add_f64(double *p, double v)
{
  // hardware global_atomic_add_f64 (no CAS loop); result unused
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Hanging-node fix-up of one cell's local vector in the pencil layout (lane (a_, b_) = (i, j) holds k = 0..n-1 in registers):
// the entries on the constrained face hold the COARSE face's DoF values and become the fine face's nodal values (TR: the adjoint,
// applied to the cell's result before the scatter) -- resolve_hanging_nodes at bp5/fe_evaluation_gl.h:150-151,167-168.  Every lane
// of the team runs the exchange (team-wide syncs), only flagged cells change values.  T: the cell's first LDS tile.
template <int n, int TW, bool TR, typename L>
__device__ __forceinline__ void pencil_hang_resolve(uint32_t m, const double *__restrict__ I, double (&u)[n], double *T, int a_, int b_, bool active)
{
  const bool on = active && (m & BP5_HANG_ANY) != 0;
#define TH(k, j, i) T[(k) * L::PS + (j) * L::RS + (i)]
  for (int d = 0; d < 3; ++d) { // every lane of the team takes every sweep (cells of one team differ in their masks)
    const double *M = I + ((m >> (6 + d)) & 1u) * n * n;
    team_sync<TW>();
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) TH(k, b_, a_) = u[k];
    }
    team_sync<TW>();
    if (on && hang_dir_any(m, d)) {
#pragma unroll
      for (int k = 0; k < n; ++k) {
        if (!hang_on_line(m, d, a_, b_, k, n - 1)) continue;
        const int idx[3] = {a_, b_, k};
        double acc = 0.0;
        for (int q = 0; q < n; ++q) {
          int e[3] = {a_, b_, k};
          e[d] = q;
          const double w = TR ? M[q * n + idx[d]] : M[idx[d] * n + q];
          acc += w * TH(e[2], e[1], e[0]);
        }
        u[k] = acc;
      }
    }
  }
  team_sync<TW>();
#undef TH
}

// ------------------------------------------------------------------------------------ fused operator
// P degree, COLL: quadrature == GLL (N == I), TW waves per team, LPC lanes per cell slot,
// TPB teams per block (TW > 1 requires TPB == 1), PF: prefetch all six planes before evaluate
// ABL (timing-only ablation builds, wrong results): bit0 skip the scatter atomics, bit1 skip the
// metric loads, bit2 skip the src gather, bit3 skip the contractions
// ABL bit 512: ask the compiler for 4 waves per SIMD (<= 128 VGPRs)
template <int P, bool COLL, int TW, int LPC, int TPB, bool PF, int ABL = 0>
__global__ void __launch_bounds__(64 * TW * TPB, (ABL & 512) ? (TW * TPB) : 1) apply_pencil_kernel(ApplyArgs a, ShapeArg<P + 1> sh)
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
  for (int k = 0; k < n; ++k) idx[k] = (ABL & 256) ? __builtin_nontemporal_load(l2g_c + k * n2) : l2g_c[k * n2];
#pragma unroll
  for (int k = 0; k < n; ++k) u[k] = (ABL & 4) ? 1e-9 * idx[k] : a.src[idx[k]];
  uint32_t hmask = 0;
  if constexpr ((ABL & 2097152) != 0) { // hanging nodes: coarse-face values -> this cell's own face nodes
    hmask = a.hang_mask[cell];
    pencil_hang_resolve<n, TW, false, L>(hmask, a.hang_I, u, T, a_, b_, active);
  }

  if constexpr (ABL & 131072) { // timing only: the E-vector-sized write stream issued at the START of the workgroup
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) a.dst[cell * n3 + k * n2 + abm] = u[k];
    }
  }
  // ---- metric planes (x-owner: a_ = j, b_ = k; registers hold i), layout [c][cell][i][j+n k]
  const double *cf = a.coef + cell * a.cell_stride; // cell base; lane offsets through coef_off / load_pencil
  constexpr bool AFFINE = (ABL & 1024) != 0; // affine geometry: one scalar plane + six per-cell numbers
  // HELM: step-64's Helmholtz operator (grad v, grad u) + (v, a u) (step-64/step-64.cu:154-160,201-219): evaluate(true, true) /
  // integrate(true, true) cost ONE more 1-D contraction each way (the value path shares the y- and z-contractions with the x-derivative),
  // submit_value(a * get_value()) one more plane: a(x_q) JxW behind the six merged planes
  constexpr bool HELM = (ABL & 8388608) != 0;
  static_assert(!HELM || (PF && !AFFINE), "Helmholtz build: prefetched planes, six-plane geometry");
  constexpr int NPL = HELM ? 7 : 6;
  double S[(PF && !AFFINE) ? NPL : 1][n];
  double um[HELM ? n : 1]; // HELM: u at the quadrature points of this lane's x-pencil, then a JxW u
  double Gc[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (AFFINE) {
    load_pencil<n>(cf, abm, S[0]);
#pragma unroll
    for (int pl = 0; pl < 6; ++pl) Gc[pl] = a.gcell[(uint64_t)pl * a.n_cells_total + cell];
  } else if constexpr (PF) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      if constexpr (ABL & 2) {
#pragma unroll
        for (int i = 0; i < n; ++i) S[pl][i] = 1.0 + pl + i + 1e-3 * abm;
      } else
        load_pencil<n, (ABL & 256) != 0>(cf + pl * a.plane_stride, abm, S[pl]);
    }
  }

  double g0[n], g1[n], g2[n];
  if constexpr (ABL & 8) {
#pragma unroll
    for (int i = 0; i < n; ++i) { g0[i] = u[i]; g1[i] = 2.0 * u[i]; g2[i] = 3.0 * u[i]; }
  } else if constexpr (!COLL) {
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
    if constexpr (HELM) MV_N(sh.N, r1, um); // value_at_quad_pts: (N x N x N) u
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
    if constexpr (HELM) {
#pragma unroll
      for (int i = 0; i < n; ++i) um[i] = r1[i]; // collocation: the nodal values are the values at the quadrature points
    }
  }

  // ---- quadrature-point operation: t = S ghat (symmetric 3x3, bp5/step-64.cu:166-177)
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s00, s11, s22, s01, s02, s12;
    if constexpr (AFFINE) {
      const double sc = S[0][i];
      s00 = sc * Gc[0]; s11 = sc * Gc[1]; s22 = sc * Gc[2]; s01 = sc * Gc[3]; s02 = sc * Gc[4]; s12 = sc * Gc[5];
    } else if constexpr (PF) {
      s00 = S[0][i]; s11 = S[1][i]; s22 = S[2][i]; s01 = S[3][i]; s02 = S[4][i]; s12 = S[5][i];
    } else {
      s00 = cf[0 * a.plane_stride + coef_off<n>(i, abm)];
      s11 = cf[1 * a.plane_stride + coef_off<n>(i, abm)];
      s22 = cf[2 * a.plane_stride + coef_off<n>(i, abm)];
      s01 = cf[3 * a.plane_stride + coef_off<n>(i, abm)];
      s02 = cf[4 * a.plane_stride + coef_off<n>(i, abm)];
      s12 = cf[5 * a.plane_stride + coef_off<n>(i, abm)];
    }
    const double x0 = g0[i], x1 = g1[i], x2 = g2[i];
    g0[i] = s00 * x0 + s01 * x1 + s02 * x2;
    g1[i] = s01 * x0 + s11 * x1 + s12 * x2;
    g2[i] = s02 * x0 + s12 * x1 + s22 * x2;
    if constexpr (HELM) um[i] *= S[6][i]; // submit_value(coef * get_value(q)) with JxW folded into the plane (step-64/step-64.cu:158)
  }

  // ---- integrate (transpose sequence)
  double y[n];
  if constexpr (ABL & 8) {
#pragma unroll
    for (int i = 0; i < n; ++i) y[i] = g0[i] + g1[i] + g2[i];
  } else if constexpr (!COLL) {
    double e1[n], e2[n], e3[n];
    MV_DT(sh.D, g0, e1);
    if constexpr (HELM) MV_NT_ADD(sh.N, um, e1); // integrate_value shares the y- and z-contractions with the x-derivative
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
    if constexpr (HELM) {
#pragma unroll
      for (int i = 0; i < n; ++i) e1[i] += um[i];
    }
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

  if constexpr ((ABL & 2097152) != 0) pencil_hang_resolve<n, TW, true, L>(hmask, a.hang_I, y, T, a_, b_, active); // adjoint, before the scatter
  // ---- scatter-add (distribute_local_to_global, bp5/fe_evaluation_gl.h:170-180)
  if constexpr (ABL & 1) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < n; ++k) acc += y[k];
    if (acc == 1.2345e300) a.dst[idx[0]] = acc; // keeps y live, never taken
  } else if constexpr (ABL & 4096) { // timing only: 64-bit integer atomics on the bit pattern
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k)
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.dst) + idx[k], (unsigned long long)__double_as_longlong(y[k]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if constexpr (ABL & 8192) { // timing only: two f32 atomics per value (hi/lo words)
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) {
        float *pf = reinterpret_cast<float *>(a.dst + idx[k]);
        __hip_atomic_fetch_add(pf, (float)y[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else if constexpr (ABL & 64) { // timing only: plain scattered stores instead of atomics
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) a.dst[idx[k]] = y[k];
    }
  } else if constexpr (ABL & 262144) { // timing only: full-line stores: lane-contiguous, 512 B per wave instruction
#pragma unroll
    for (int k = 0; k < n; ++k) a.dst[((uint64_t)k * a.n_teams + team) * TEAM + t] = y[k];
  } else if constexpr (ABL & 128) { // timing only: contiguous E-vector stores (a.dst must hold n_cells*n^3 doubles)
    if (active) {
#pragma unroll
      for (int k = 0; k < ((ABL & 65536) ? 1 : n); ++k) {
        if constexpr (ABL & 16384) __builtin_nontemporal_store(y[k], a.dst + cell * n3 + k * n2 + abm);
        else a.dst[cell * n3 + k * n2 + abm] = y[k];
      }
    }
  } else if constexpr ((ABL & 32) != 0) {
    // Cell-interior entries (1 <= i, j, k <= n - 2: (n - 2)^3 of n^3, 47 % at p = 8) belong to this cell alone: a plain store instead of a memory-side
    // atomic -- the kernel is bound by the COUNT of its atomics.  Correct on any mesh; it pays only where these DoFs do not share cache
    // lines with atomically updated ones ("an atomic to a line that is dirty in L2 forces its write-back": stores mixed into such lines cost 21 % at p = 4),
    // i.e. with the cell-interior DoFs numbered contiguously ahead of all others (bp5_mesh_desc.dof_numbering = 2; bp5_mf_create detects the property)
    if (active) {
      const bool lane_interior = a_ >= 1 && a_ <= n - 2 && b_ >= 1 && b_ <= n - 2;
#pragma unroll
      for (int k = 0; k < n; ++k) {
        if (lane_interior && k >= 1 && k <= n - 2) a.dst[idx[k]] = y[k];
        else atomic_add_f64(a.dst + idx[k], y[k]);
      }
    }
  } else if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) atomic_add_f64(a.dst + idx[k], y[k]);
  }
#undef TL
}

// ------------------------------------------------------------------------------------ fused operator, z-marching
// Same per-cell arithmetic and launch shape as apply_pencil_kernel, but a workgroup walks CPT chains of
// cells along their local z direction (host plan: build_march_plan).  A lane's pencil runs along z, so the
// contribution to the shared face  y[n-1]  stays in a register and is added to  y[0]  of the next cell, and
// u[n-1] is reused as the next u[0]: n^2 (n-1) instead of n^3 gathers and -- the point -- memory-side
// atomics per cell (the default kernel is bound by their count: DESIGN.md 5).
struct MarchPlan {
  const uint32_t *team_off; // [n_teams+1] in steps
  const uint32_t *entries;  // [steps * CPT]: cell | bit31 idle | bit30 linked to the previous step
  uint32_t n_teams, teams_per_xcd;
};

template <int P, bool COLL, int TW, int LPC, bool PF, int ABL = 0>
__global__ void __launch_bounds__(64 * TW) apply_march_kernel(ApplyArgs a, MarchPlan mp, ShapeArg<P + 1> sh)
{
  constexpr int n = P + 1, n2 = n * n, n3 = n2 * n;
  constexpr int TEAM = 64 * TW;
  constexpr int CPT = TEAM / LPC;
  static_assert(LPC >= n2 && CPT >= 1, "lanes per cell");
  using L = LdsLayout<n, LPC>;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int t = threadIdx.x;
  const int c = t / LPC, ab = t - c * LPC;
  const uint32_t team = (blockIdx.x & 7u) * mp.teams_per_xcd + (blockIdx.x >> 3);
  if (team >= mp.n_teams) return; // whole workgroup
  const bool lane_ok = (ab < n2) && (c < CPT);
  const int slot = c < CPT ? c : 0;
  const int abm = ab < n2 ? ab : ab % n2;
  const int a_ = abm % n, b_ = abm / n;
  double *T = lds + slot * L::CS;
#define TL(f, k, j, i) T[(f) * (n * L::PS) + (k) * L::PS + (j) * L::RS + (i)]

  const uint32_t s0 = mp.team_off[team], s1 = mp.team_off[team + 1];
  uint32_t e_next = mp.entries[(uint64_t)s0 * CPT + slot];
  uint32_t e_next2 = mp.entries[(uint64_t)(s0 + 1 < s1 ? s0 + 1 : s0) * CPT + slot];
  double u_carry = 0.0, y_carry = 0.0;
  uint32_t idx_carry = 0;
  // software pipeline: idxN / uN hold layers 1..n-1 of the NEXT step's cell (layer 0 too when it starts a chain)
  uint32_t idxN[n];
  double uN[n];
  {
    const uint32_t *l2g_n = a.l2g + (uint64_t)(e_next & 0x3fffffffu) * n3 + abm;
#pragma unroll
    for (int k = 0; k < n; ++k) idxN[k] = l2g_n[k * n2];
#pragma unroll
    for (int k = 0; k < n; ++k) uN[k] = a.src[idxN[k]];
  }
  for (uint32_t st = s0; st < s1; ++st) {
    const uint32_t e = e_next;
    e_next = e_next2;
    e_next2 = st + 2 < s1 ? mp.entries[(uint64_t)(st + 2) * CPT + slot] : 0x80000000u;
    if (st + 1 >= s1) e_next = 0x80000000u | (e & 0x3fffffffu);
    const bool active = lane_ok && !(e >> 31);
    const bool linked = (e >> 30) & 1u;                              // k = 0 layer == previous cell's k = n-1 layer
    const bool next_linked = !(e_next >> 31) && ((e_next >> 30) & 1u); // this cell's k = n-1 layer is continued
    const uint64_t cell = e & 0x3fffffffu;

    // ---- gather (z-owner: a_ = i, b_ = j; registers hold k); a linked cell reuses the carried face
    uint32_t idx[n];
    double u[n];
    idx[0] = linked ? idx_carry : idxN[0];
    u[0] = linked ? u_carry : uN[0];
#pragma unroll
    for (int k = 1; k < n; ++k) { idx[k] = idxN[k]; u[k] = uN[k]; }
    u_carry = u[n - 1];
    idx_carry = idx[n - 1];
    // indices of the next step's cell (its layer 0 is only needed when it starts a new chain, but loading it
    // unconditionally keeps the lanes converged)
    {
      const uint32_t *l2g_n = a.l2g + (uint64_t)(e_next & 0x3fffffffu) * n3 + abm;
#pragma unroll
      for (int k = 0; k < n; ++k) idxN[k] = l2g_n[k * n2];
    }

    // ---- metric planes (x-owner: a_ = j, b_ = k; registers hold i), layout [c][cell][i][j+n k]
    const double *cf = a.coef + cell * a.cell_stride; // cell base; lane offsets through coef_off / load_pencil
    constexpr bool AFFINE = (ABL & 1024) != 0; // affine geometry: one scalar plane + six per-cell numbers
    double S[(PF && !AFFINE) ? 6 : 1][n];
    double Gc[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (AFFINE) {
      load_pencil<n>(cf, abm, S[0]);
#pragma unroll
      for (int pl = 0; pl < 6; ++pl) Gc[pl] = a.gcell[(uint64_t)pl * a.n_cells_total + cell];
    } else if constexpr (PF) {
#pragma unroll
      for (int pl = 0; pl < 6; ++pl) {
        if constexpr (ABL & 2) {
#pragma unroll
          for (int i = 0; i < n; ++i) S[pl][i] = 1.0 + pl + i + 1e-3 * abm;
        } else
          load_pencil<n, (ABL & 256) != 0>(cf + pl * a.plane_stride, abm, S[pl]);
      }
    }

    double g0[n], g1[n], g2[n];
    if constexpr (ABL & 8) {
#pragma unroll
      for (int i = 0; i < n; ++i) { g0[i] = u[i]; g1[i] = 2.0 * u[i]; g2[i] = 3.0 * u[i]; }
    } else if constexpr (!COLL) {
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

    // the next step's indices have landed: start its src gather
#pragma unroll
    for (int k = 0; k < n; ++k) uN[k] = a.src[idxN[k]];

    // ---- quadrature-point operation: t = S ghat (symmetric 3x3, bp5/step-64.cu:166-177)
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s00, s11, s22, s01, s02, s12;
      if constexpr (AFFINE) {
        const double sc = S[0][i];
        s00 = sc * Gc[0]; s11 = sc * Gc[1]; s22 = sc * Gc[2]; s01 = sc * Gc[3]; s02 = sc * Gc[4]; s12 = sc * Gc[5];
      } else if constexpr (PF) {
        s00 = S[0][i]; s11 = S[1][i]; s22 = S[2][i]; s01 = S[3][i]; s02 = S[4][i]; s12 = S[5][i];
      } else {
        s00 = cf[0 * a.plane_stride + coef_off<n>(i, abm)];
        s11 = cf[1 * a.plane_stride + coef_off<n>(i, abm)];
        s22 = cf[2 * a.plane_stride + coef_off<n>(i, abm)];
        s01 = cf[3 * a.plane_stride + coef_off<n>(i, abm)];
        s02 = cf[4 * a.plane_stride + coef_off<n>(i, abm)];
        s12 = cf[5 * a.plane_stride + coef_off<n>(i, abm)];
      }
      const double x0 = g0[i], x1 = g1[i], x2 = g2[i];
      g0[i] = s00 * x0 + s01 * x1 + s02 * x2;
      g1[i] = s01 * x0 + s11 * x1 + s12 * x2;
      g2[i] = s02 * x0 + s12 * x1 + s22 * x2;
    }

    // ---- integrate (transpose sequence)
    double y[n];
    if constexpr (ABL & 8) {
#pragma unroll
      for (int i = 0; i < n; ++i) y[i] = g0[i] + g1[i] + g2[i];
    } else if constexpr (!COLL) {
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


    // ---- scatter: the carried face of the previous cell joins k = 0; the k = n-1 layer is held back when
    //      the chain continues
    if constexpr (ABL & 1) { // timing only: no scatter
      double sacc = y_carry;
#pragma unroll
      for (int k = 0; k < n; ++k) sacc += y[k];
      if (sacc == 1.2345e300) a.dst[idx[0]] = sacc;
    } else if (active) {
      if (linked) y[0] += y_carry;
#pragma unroll
      for (int k = 0; k < n - 1; ++k) atomic_add_f64(a.dst + idx[k], y[k]);
      if (!next_linked) atomic_add_f64(a.dst + idx[n - 1], y[n - 1]);
    }
    y_carry = y[n - 1];
  }
#undef TL
}

// ------------------------------------------------------------------------------------ fused operator, team-assembled
// Same arithmetic as apply_pencil_kernel, but gather and scatter go through the team's LDS:
//   * the host builds, per team of CPT consecutive cells, the sorted list of the distinct DoFs the
//     team touches (tp.dofs) and for every local DoF its 16-bit position in that list (tp.pos);
//   * gather: the team reads src through the sorted list (contiguous runs -> coalesced) into LDS,
//     lanes pick their values by position;
//   * scatter: lanes accumulate into an LDS vector (ds_add_f64), then the team issues ONE global
//     atomic per distinct DoF, consecutive lanes on consecutive sorted indices, so a 64-byte
//     atomic request carries up to 8 contributions instead of ~3 and shared faces inside the team
//     are pre-summed.  (v1 was bound by the chip-wide atomic request rate: profiles/r1.)
// The staging vectors alias the transpose tiles (they are live before / after the tile phase).
struct TeamPlan {
  const uint32_t *off;        // [n_teams_total + 1]
  const uint32_t *dofs;       // sorted distinct local DoF indices per team; bit 31: touched by this team only
  const uint16_t *pos;        // [n_cells * n^3] position of each local DoF in its team's list
  const uint8_t *cell_round;  // [n_cells] accumulation round of the cell inside its team
  const uint8_t *team_rounds; // [n_teams_total] number of rounds
  double *partial;            // [off[n_teams_total]] per-(team, DoF) partial sums of shared DoFs
};

// Scatter modes of the team kernel
//   SC_ATOMIC     dst += via one global atomic per distinct DoF (any cell range)
//   SC_OWNER_SET  team-exclusive DoFs: dst = sum (plain store); shared DoFs: partial slab
//   SC_OWNER_ADD  same with dst += for the exclusive DoFs
// The OWNER modes need every team of the plan to run in the launch; combine_kernel then finishes
// the shared DoFs in a fixed order, so the whole operator is free of atomics and bitwise
// reproducible.
//   SC_OWNER_SET_ATOMIC / SC_OWNER_ADD_ATOMIC  exclusive DoFs as above, shared DoFs by one global atomic per
//                 (group, DoF): no partial slab and no combine pass; the caller zeroes the shared DoFs first
//                 in SET mode.  Order-dependent rounding on group-surface DoFs only.
enum { SC_ATOMIC = 0, SC_OWNER_SET = 1, SC_OWNER_ADD = 2, SC_OWNER_SET_ATOMIC = 3, SC_OWNER_ADD_ATOMIC = 4 };

__device__ __forceinline__ void lds_add_f64(double *p, double v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int P, bool COLL, int TW, int LPC, bool PF, int SCATTER, int ABL = 0>
__global__ void __launch_bounds__(64 * TW) apply_team_kernel(ApplyArgs a, TeamPlan tp, ShapeArg<P + 1> sh)
{
  constexpr int n = P + 1, n2 = n * n, n3 = n2 * n;
  constexpr int TEAM = 64 * TW;
  constexpr int CPT = TEAM / LPC;
  constexpr int MAXU = (CPT * n3 + TEAM - 1) / TEAM; // list entries per thread
  static_assert(LPC >= n2 && CPT >= 1, "lanes per cell");
  using L = LdsLayout<n, LPC>;
  static_assert(CPT * L::CS >= CPT * n3, "staging vectors must fit in the tile region");
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int t = threadIdx.x;
  const int c = t / LPC, ab = t - c * LPC;
  const uint32_t blk = (blockIdx.x & 7u) * a.teams_per_xcd + (blockIdx.x >> 3);
  if (blk >= a.n_teams) return;                      // whole block: no barrier is skipped by a subset
  const uint32_t team = a.cell_begin / CPT + blk;    // teams are aligned to multiples of CPT cells
  const uint64_t cell_raw = (uint64_t)team * CPT + c;
  const bool active = (ab < n2) && (c < CPT) && (cell_raw >= a.cell_begin) && (cell_raw < a.cell_end);
  const uint64_t cell = cell_raw < a.cell_end ? cell_raw : (uint64_t)a.cell_end - 1;
  const int abm = ab < n2 ? ab : ab % n2;
  const int a_ = abm % n, b_ = abm / n;
  double *T = lds + (c < CPT ? c : 0) * L::CS;
#define TL(f, k, j, i) T[(f) * (n * L::PS) + (k) * L::PS + (j) * L::RS + (i)]

  // ---- stage the team's distinct src values in LDS (sorted indices: contiguous runs)
  const uint32_t o0 = tp.off[team];
  const int m = (int)(tp.off[team + 1] - o0);
  uint32_t gidx[MAXU];
#pragma unroll
  for (int r = 0; r < MAXU; ++r) {
    const int i = t + r * TEAM;
    gidx[r] = i < m ? tp.dofs[o0 + i] : 0u;
  }
  const int my_round = tp.cell_round[cell];
  const int n_rounds = tp.team_rounds[team];
  // ---- metric planes (issued early; consumed after the evaluate phase)
  const double *cf = a.coef + cell * a.cell_stride; // cell base; lane offsets through coef_off / load_pencil
  constexpr bool AFFINE = (ABL & 1024) != 0;
  double S[(PF && !AFFINE) ? 6 : 1][n];
  double Gc[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (AFFINE) {
    load_pencil<n>(cf, abm, S[0]);
#pragma unroll
    for (int pl = 0; pl < 6; ++pl) Gc[pl] = a.gcell[(uint64_t)pl * a.n_cells_total + cell];
  } else if constexpr (PF) {
#pragma unroll
    for (int pl = 0; pl < 6; ++pl) load_pencil<n>(cf + pl * a.plane_stride, abm, S[pl]);
  }
  uint16_t ps[n];
  const uint16_t *pos_c = tp.pos + cell * n3 + abm;
#pragma unroll
  for (int k = 0; k < n; ++k) ps[k] = pos_c[k * n2];
  double u[n];
  if constexpr (ABL & 32) { // direct gather through local_to_global (L2-served re-reads), no LDS staging
    const uint32_t *l2g_c = a.l2g + cell * n3 + abm;
    uint32_t idx[n];
#pragma unroll
    for (int k = 0; k < n; ++k) idx[k] = l2g_c[k * n2];
#pragma unroll
    for (int k = 0; k < n; ++k) u[k] = a.src[idx[k]];
  } else if constexpr (ABL & 4) {
#pragma unroll
    for (int k = 0; k < n; ++k) u[k] = 1e-9 * (ps[k] + gidx[k % MAXU]);
  } else {
#pragma unroll
    for (int r = 0; r < MAXU; ++r) {
      const int i = t + r * TEAM;
      if (i < m) lds[i] = a.src[gidx[r] & 0x7fffffffu];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < n; ++k) u[k] = lds[ps[k]];
    __syncthreads();
  }

  double g0[n], g1[n], g2[n];
  if constexpr (!COLL) {
    double aN[n], aD[n];
    MV_N(sh.N, u, aN);
    MV_D(sh.D, u, aD);
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) { TL(0, k, b_, a_) = aN[k]; TL(1, k, b_, a_) = aD[k]; }
    }
    __syncthreads();
    double vN[n], vD[n];
#pragma unroll
    for (int j = 0; j < n; ++j) { vN[j] = TL(0, b_, j, a_); vD[j] = TL(1, b_, j, a_); }
    double c1[n], c2[n], c3[n];
    MV_N(sh.N, vN, c1);
    MV_D(sh.D, vN, c2);
    MV_N(sh.N, vD, c3);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < n; ++j) { TL(0, b_, j, a_) = c1[j]; TL(1, b_, j, a_) = c2[j]; TL(2, b_, j, a_) = c3[j]; }
    }
    __syncthreads();
    double r1[n], r2[n], r3[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { r1[i] = TL(0, b_, a_, i); r2[i] = TL(1, b_, a_, i); r3[i] = TL(2, b_, a_, i); }
    MV_D(sh.D, r1, g0);
    MV_N(sh.N, r2, g1);
    MV_N(sh.N, r3, g2);
  } else {
    double gz[n];
    MV_D(sh.D, u, gz);
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) { TL(0, k, b_, a_) = u[k]; TL(2, k, b_, a_) = gz[k]; }
    }
    __syncthreads();
    double vN[n], c2[n];
#pragma unroll
    for (int j = 0; j < n; ++j) vN[j] = TL(0, b_, j, a_);
    MV_D(sh.D, vN, c2);
    if (active) {
#pragma unroll
      for (int j = 0; j < n; ++j) TL(1, b_, j, a_) = c2[j];
    }
    __syncthreads();
    double r1[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { r1[i] = TL(0, b_, a_, i); g1[i] = TL(1, b_, a_, i); g2[i] = TL(2, b_, a_, i); }
    MV_D(sh.D, r1, g0);
  }

#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s00, s11, s22, s01, s02, s12;
    if constexpr (AFFINE) {
      const double sc = S[0][i];
      s00 = sc * Gc[0]; s11 = sc * Gc[1]; s22 = sc * Gc[2]; s01 = sc * Gc[3]; s02 = sc * Gc[4]; s12 = sc * Gc[5];
    } else if constexpr (PF) {
      s00 = S[0][i]; s11 = S[1][i]; s22 = S[2][i]; s01 = S[3][i]; s02 = S[4][i]; s12 = S[5][i];
    } else {
      s00 = cf[0 * a.plane_stride + coef_off<n>(i, abm)]; s11 = cf[1 * a.plane_stride + coef_off<n>(i, abm)]; s22 = cf[2 * a.plane_stride + coef_off<n>(i, abm)];
      s01 = cf[3 * a.plane_stride + coef_off<n>(i, abm)]; s02 = cf[4 * a.plane_stride + coef_off<n>(i, abm)]; s12 = cf[5 * a.plane_stride + coef_off<n>(i, abm)];
    }
    const double x0 = g0[i], x1 = g1[i], x2 = g2[i];
    g0[i] = s00 * x0 + s01 * x1 + s02 * x2;
    g1[i] = s01 * x0 + s11 * x1 + s12 * x2;
    g2[i] = s02 * x0 + s12 * x1 + s22 * x2;
  }

  double y[n];
  if constexpr (!COLL) {
    double e1[n], e2[n], e3[n];
    MV_DT(sh.D, g0, e1);
    MV_NT(sh.N, g1, e2);
    MV_NT(sh.N, g2, e3);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int i = 0; i < n; ++i) { TL(0, b_, a_, i) = e1[i]; TL(1, b_, a_, i) = e2[i]; TL(2, b_, a_, i) = e3[i]; }
    }
    __syncthreads();
    double w1[n], w2[n], w3[n];
#pragma unroll
    for (int j = 0; j < n; ++j) { w1[j] = TL(0, b_, j, a_); w2[j] = TL(1, b_, j, a_); w3[j] = TL(2, b_, j, a_); }
    double f1[n], f2[n];
    MV_NT(sh.N, w1, f1);
    MV_DT_ADD(sh.D, w2, f1);
    MV_NT(sh.N, w3, f2);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < n; ++j) { TL(0, b_, j, a_) = f1[j]; TL(1, b_, j, a_) = f2[j]; }
    }
    __syncthreads();
    double z1[n], z2[n];
#pragma unroll
    for (int k = 0; k < n; ++k) { z1[k] = TL(0, k, b_, a_); z2[k] = TL(1, k, b_, a_); }
    MV_NT(sh.N, z1, y);
    MV_DT_ADD(sh.D, z2, y);
  } else {
    double e1[n];
    MV_DT(sh.D, g0, e1);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int i = 0; i < n; ++i) { TL(0, b_, a_, i) = e1[i]; TL(1, b_, a_, i) = g1[i]; TL(2, b_, a_, i) = g2[i]; }
    }
    __syncthreads();
    double w1[n], w2[n];
#pragma unroll
    for (int j = 0; j < n; ++j) { w1[j] = TL(0, b_, j, a_); w2[j] = TL(1, b_, j, a_); }
    MV_DT_ADD(sh.D, w2, w1);
    if (active) {
#pragma unroll
      for (int j = 0; j < n; ++j) TL(0, b_, j, a_) = w1[j];
    }
    __syncthreads();
    double z2[n];
#pragma unroll
    for (int k = 0; k < n; ++k) { y[k] = TL(0, k, b_, a_); z2[k] = TL(2, k, b_, a_); }
    MV_DT_ADD(sh.D, z2, y);
  }

  // ---- team-level assembly in LDS.  Cells of one round share no DoF (host colouring), so
  //      plain read-modify-write is race-free and the summation order is fixed.
  if constexpr (ABL & 1) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < n; ++k) acc += y[k];
    if (acc == 1.2345e300) a.dst[gidx[0] & 0x7fffffffu] = acc + my_round + n_rounds; // keeps y live, never taken
    return;
  }
  __syncthreads(); // every tile read is done: the region becomes the accumulator
#pragma unroll
  for (int r = 0; r < MAXU; ++r) {
    const int i = t + r * TEAM;
    if (i < m) lds[i] = 0.0;
  }
  __syncthreads();
  for (int rd = 0; rd < n_rounds; ++rd) {
    if (active && my_round == rd) {
#pragma unroll
      for (int k = 0; k < n; ++k) lds[ps[k]] += y[k];
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < MAXU; ++r) {
    const int i = t + r * TEAM;
    if (i < m) {
      const double v = lds[i];
      const uint32_t g = gidx[r];
      if constexpr (SCATTER == SC_ATOMIC) {
        atomic_add_f64(a.dst + (g & 0x7fffffffu), v);
      } else {
        if (g & 0x80000000u) {
          if constexpr (SCATTER == SC_OWNER_SET) a.dst[g & 0x7fffffffu] = v;
          else a.dst[g & 0x7fffffffu] += v;
        } else
          tp.partial[o0 + i] = v;
      }
    }
  }
#undef TL
}

// ------------------------------------------------------------------------------------ fused operator, block-assembled
// One 256-thread workgroup owns a compact brick of cells (e.g. 4x4x4) and walks it in passes of
// CPT cells.  Contributions are summed into an LDS accumulator that spans the brick's distinct
// DoFs; after the last pass DoFs touched by this brick only are stored with plain stores and
// the brick-surface DoFs go to a per-brick partial slab that combine_kernel sums in a fixed
// order: no global atomics, no zero-fill of dst, bitwise reproducible.
// Build flags (ABL) select the shape.  First version: three transpose tiles per cell and a second register set
// (double-buffered prefetch of the next pass: indices, gathered src values, all six metric planes), two workgroups per
// CU.  Default at p = 4 (variant 56 = 2048 | 8192 | 16384 | 262144), all measured steps in profiles/r1/README.md:
//   2048    single register set: 151-167 VGPRs -> three waves per SIMD
//   8192    ONE transpose tile per cell, used field after field (32 lanes per cell: tile syncs are wave-local and
//           free) -> tiles + 17^3 accumulator + run tables = 49.5 KB -> three workgroups per CU
//   16384   run-length write-out: the brick's sorted DoF list as <= 128 runs in LDS, 16-byte stores, no list loads
//   262144  packed indices: one u16 (run << 9 | offset) per cell-local DoF gives both the accumulator slot and the
//           DoF to gather through that run table; local_to_global is not read
// With the metric in the pair layout (coef_off) a lane issues 18 + 3 + 5 + 1 load instructions per cell.

// the same pair layout for the block plan's per-cell index arrays (z-pencil of lane ab = i + n j, entries k = 0..n-1)
template <int n, typename T>
__device__ __forceinline__ void load_pencil_idx(const T *base, int ab, T (&v)[n])
{
  typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
#pragma unroll
  for (int m = 0; m < n / 2; ++m) {
    const pair_t q = *reinterpret_cast<const pair_t *>(base + m * (2 * n * n) + 2 * ab);
    v[2 * m] = q.x;
    v[2 * m + 1] = q.y;
  }
  if constexpr (n & 1) v[n - 1] = base[(n / 2) * (2 * n * n) + ab];
}

// Explicit wait for this wave's outstanding LDS operations.  Needed in front of a workgroup barrier that sits at a
// loop header: the compiler (ROCm 7.2 clang) emits the `s_waitcnt lgkmcnt(0)` of __syncthreads() on the fall-through
// path only, so an LDS write at the end of the loop body reaches the barrier through the back-edge still in flight,
// and a wave on another SIMD can pass the barrier and read-modify-write the same word first (lost update; seen as
// rare wrong sums in multi-round passes at full size).  tools/check_lds_barrier.py checks the ISA for this pattern.
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// degrees whose block-kernel cells span waves (n^2 lanes per cell: p = 2, 5, 8) and exchange their tiles through the workgroup barrier: two
// tiles used alternately halve the barriers of a pass (BlockPass::PP) where the second tile still fits the LDS share of the workgroup
#ifndef BP5_PINGPONG_P2
#define BP5_PINGPONG_P2 1
#endif
#ifndef BP5_PINGPONG_P5
#define BP5_PINGPONG_P5 0
#endif
#ifndef BP5_PINGPONG_P8
#define BP5_PINGPONG_P8 1
#endif
constexpr bool block_pingpong(int degree) { return degree == 2 ? BP5_PINGPONG_P2 != 0 : degree == 5 ? BP5_PINGPONG_P5 != 0 : degree == 8 ? BP5_PINGPONG_P8 != 0 : false; }

// tile strides of the block kernel (doubles): LdsLayout's, except for the cell shapes that span waves (p = 2, 5, 8), where the bank conflicts of
// all four waves of the workgroup count (tools/lds_stride_search.py --block; weighted LDS cycles per pass against LdsLayout's strides:
// n = 9: 363 / 551, n = 3: 397 / 805 with two tiles, n = 6: 399 / 445).  ONE: one tile, SLOT: cell slot stride of the sequential-tile
// builds (PP: two tiles)
template <int n, int LPC, bool PP>
struct BlockLayout {
  using B = LdsLayout<n, LPC>;
  static constexpr int RS = B::RS, PS = B::PS, CS = B::CS, ONE = n * PS + 3, SLOT = (PP ? 2 : 1) * ONE;
};
template <> struct BlockLayout<9, 81, true> { static constexpr int RS = 9, PS = 81, CS = 3 * 9 * 81 + 1, ONE = 732, SLOT = 1465; };
template <> struct BlockLayout<3, 9, true> { static constexpr int RS = 4, PS = 19, CS = 3 * 3 * 19 + 1, ONE = 60, SLOT = 121; };
template <> struct BlockLayout<6, 36, false> { static constexpr int RS = 7, PS = 42, CS = 3 * 6 * 42 + 1, ONE = 252, SLOT = 252; };

constexpr int BLOCK_MAX_RUNS = 128;
// run table entry "first DoF": bit 31 = the run's DoFs are touched by this block only (owner stores), bit 30 = Dirichlet DoFs
// (builds with fused dot products write src there: copy_constrained_values folded into the write-out); DoF indices < 2^30
constexpr uint32_t BLOCK_DOF_MASK = 0x3fffffffu, BLOCK_DOF_CONSTRAINED = 0x40000000u;
constexpr int PARTIAL_STRIDE = 8192;   // row length of the partial-sum array d_partials[8][PARTIAL_STRIDE] (all reducing kernels)
constexpr int BLOCK_LATTICE_WORDS = 64; // per block: 27 entity slots, 27 entity DoFs, dims | flag, face-carry words, padding (bp5_device.hip: detect_lattice_blocks)
// face carry (builds with ABL & 268435456): when block g + 1 of the plan is the neighbour of block g across a face whose interior DoFs the two
// share with nobody else, and one workgroup walks both, the face's partial sums stay in LDS from the write-out of g to the write-out of
// g + 1, which stores p(g + 1) + p(g) as an owner store -- the bits of the combine pass's p(g) + p(g + 1) without the slab round trip.
// Lattice words of block g: [55] = face DoFs << 16 | first list slot of the face g can hand to g + 1 (0: none), [56] = the same for the face g
// shares with g - 1 (set when [55] of g - 1 is).
constexpr int BLOCK_CARRY_MAX = 225;    // face DoFs that can be carried (a 4x4 face at p = 4: 15 x 15)
constexpr int BLOCK_PACK_OFF_BITS = 9;                              // packed index = run << 9 | offset (runs are cut at 512 entries)
constexpr int BLOCK_PACK_MAX_RUNS = 1 << (16 - BLOCK_PACK_OFF_BITS); // 128 == BLOCK_MAX_RUNS
static_assert(BLOCK_PACK_MAX_RUNS <= BLOCK_MAX_RUNS, "the LDS run table holds every packable run");
struct BlockPlan {
  const uint32_t *pass_cell;  // [n_passes * CPT] cell id per slot; bit 31: idle slot (id still valid)
  const uint32_t *pass_off;   // [n_blocks+1] first pass of each block
  const uint32_t *off;        // [n_blocks+1] offsets into dofs / partial
  const uint32_t *dofs;       // sorted distinct DoFs per block; bit 31: touched by this block only
  const uint16_t *pos;        // [n_cells*n^3] position of each local DoF in its block's list, pair layout (coef_off(k, i + n j))
  const uint32_t *gidx;       // [n_cells*n^3] local_to_global in the same pair layout
  // packed form (builds with ABL & 262144): ONE u16 per cell-local DoF = run << 9 | offset in the run (runs are cut at
  // 512 entries and where the Dirichlet flag changes, at most 128 per block -- a boundary brick of a slab mesh with its ghost rows has ~70); list slot = run_slot[run] + offset, DoF = run_dof[run] + offset, both from the
  // block's run table in LDS -- the local_to_global stream is not read at all
  const uint16_t *packed;     // [n_cells*n^3], pair layout
  // lattice blocks (structured bricks, recognised and verified on the host: detect_lattice_blocks): list slot and DoF of every cell-local
  // entry follow in closed form from the cell's position in its block and the block's 27 entity bases -- the packed stream is not read
  const uint32_t *lattice;    // [n_blocks][BLOCK_LATTICE_WORDS] or NULL: 27 entity slots, 27 entity DoFs, bx | by << 8 | bz << 16 | 1 << 31 (0: packed block)
  const uint16_t *cell_pos;   // [n_cells] cx | cy << 4 | cz << 8 (lattice blocks)
  const uint8_t *cell_round;  // [n_cells] accumulation round inside the pass (0 when conflict-free)
  const uint8_t *blk_rounds;  // [n_blocks] rounds needed by the block's passes (normally 1)
  double *partial;            // [off[n_blocks]]
  uint32_t n_blocks, n_wg;    // blocks of this launch; persistent workgroups, n_wg a multiple of 8
  uint32_t blk_begin;         // first block of this launch (block-aligned cell ranges; 0 for the whole mesh)
  const uint32_t *wg_block;   // [n_parts][n_wg+1] first block of every persistent workgroup: ranges balanced by estimated cost
  // boundary-first launches (halo exchange under the interior bricks): n_parts == 2 -- every workgroup walks its share of the
  // ghost-touching bricks (part 0) before its interior bricks (part 1), as ONE sequence of passes (the software pipeline runs across
  // the seam), and counts itself in at *signal once its part 0 is written out; the communication stream waits for the count
  // (hipStreamWaitValue64), combines the ghost rows and sends them
  uint32_t n_parts;           // 1 or 2
  unsigned long long *signal; // NULL: nobody waits
  // run-length form of dofs (builds with ABL & 16384): run r of block b covers the list slots [runs[2r], runs[2r+2]) and
  // the consecutive DoFs starting at runs[2r+1] (bit 31 as in dofs); at most BLOCK_MAX_RUNS runs per block
  const uint32_t *run_off;    // [n_blocks+1]
  const uint32_t *runs;       // [2 * run_off[n_blocks]]
  uint32_t max_list;          // longest block list (the accumulator's size in LDS)
  uint32_t carry;             // face-carry builds: 1 = carry (the combine tables of this launch leave the carried faces out), 0 = every shared DoF to the slab
  unsigned long long *stamps; // diagnostic builds only: [n_wg][16] cycle sums per phase (never read by kernels)
  // builds with ABL & 1048576 (fused CG dot products, SolverCGFullMerge's update_b, bp5/solver.h:142-311): src == p, dst == v
  const double *cg_r;         // residual vector r
  double *dot_partials;       // [7][PARTIAL_STRIDE] row k, column = dot_col0 + workgroup: p.v, v.v, r.v, r.r (rows 4-6 = rows 2, 1, 3: D == 1)
  uint32_t dot_col0;          // first column of this launch (the boundary-first schedule runs the bricks in two launches)
  uint32_t n_owned;           // dot products run over owned entries only
  const int *cg_state;        // st[ST_DONE] != 0: the solve has stopped, the launch is a no-op (iterate frozen)
};

// register set of one pass (cell ids, positions, gathered values, metric)
template <int n, bool AFFINE, int NPL = 6, bool ROLL = false>
struct PassRegs {
  uint16_t ps[n];
  double u[n];
  double S[(AFFINE || ROLL) ? 1 : NPL][ROLL ? 1 : n]; // affine: one scalar plane ... (NPL = 7: the Helmholtz build's mass plane behind the six merged ones; ROLL: the metric lives in ONE register set shared by all passes)
  double Gc[AFFINE ? 6 : 1];   // ... and the cell's constant K K^T
  uint32_t idx[n];
  uint32_t ent; // pass_cell entry
  uint32_t mask; // hanging-node builds: the cell's constraint mask (BP5_HANG_*)
  uint32_t cpos; // lattice blocks: the cell's position in its block
  int round;
  bool active;
};

// cycle stamp for diagnostic builds (MI355X guide: "In-kernel stamps"): one asm statement, fenced
__device__ __forceinline__ unsigned long long stamp_now()
{
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
// wave priority per phase of a pass (probes: -DBP5_PRIO_Q=k ... ; -1: no instruction)
#ifndef BP5_PRIO_Q
#define BP5_PRIO_Q -1
#endif
#ifndef BP5_PRIO_I
#define BP5_PRIO_I -1
#endif
#ifndef BP5_PRIO_A
#define BP5_PRIO_A -1
#endif
#ifndef BP5_PRIO_END
#define BP5_PRIO_END -1
#endif
#define BP5_PRIO_PHASE(k) if constexpr ((k) >= 0) __builtin_amdgcn_s_setprio((k) < 0 ? 0 : (k));
#define BP5_STAMP(slot)                                                                                            \
  if constexpr (ABL & 4096) {                                                                                      \
    const unsigned long long now_ = stamp_now();                                                                   \
    ph[slot] += now_ - tprev;                                                                                      \
    tprev = now_;                                                                                                  \
  }

#ifndef BP5_WAVE_PACK
#define BP5_WAVE_PACK 1
#endif
template <int P, bool COLL, int LPC, int SCATTER, int ABL>
struct BlockPass {
  static constexpr int n = P + 1, n2 = n * n, n3 = n2 * n;
  static constexpr int TEAM = 256;
  static constexpr int CPT = TEAM / LPC;
  static constexpr bool AFFINE = (ABL & 1024) != 0;
  // SINGLE: the metric of a pass is loaded at the top of that pass (as in apply_pencil_kernel) and only the
  // indices / gathered values are prefetched one pass ahead: ~60 fewer VGPRs -> three workgroups per CU
  static constexpr bool SINGLE = (ABL & 2048) != 0;
  // HELM: step-64's Helmholtz operator, see apply_pencil_kernel (one more contraction each way, one more plane: a(x_q) JxW)
  static constexpr bool HELM = (ABL & 8388608) != 0;
  static constexpr int NPL = HELM ? 7 : 6;
  static_assert(!HELM || (!AFFINE && (ABL & 8192) != 0), "Helmholtz build: six-plane geometry, sequential tiles");
  // HANG: 2:1 refined meshes -- resolve_hanging_nodes after the gather and its adjoint before the accumulation into the brick vector
  // (bp5/fe_evaluation_gl.h:150-151,167-168), through the cell's transpose tile like apply_pencil_kernel; a pass without a flagged
  // cell skips both (one ballot / one barrier-with-or per pass)
  static constexpr bool HANG = (ABL & 2097152) != 0;
  static_assert(!HANG || (ABL & 8192) != 0, "hanging-node build: sequential tiles");
  // ROLL: rolling metric prefetch.  The metric of pass q + 1 is loaded INTO THE REGISTERS OF PASS q's METRIC, piece by piece, each piece right
  // after the quadrature-point loop of pass q has consumed it: the six planes of the next pass are in flight during integrate, accumulation,
  // write-out and the next evaluate -- a whole pass ahead -- at the register cost of the single-buffered build (three workgroups per CU).
  // Without it a wave has loads in flight during evaluate only.  MEASURED, NOT ADOPTED (profiles/r4 d_*): held for a whole pass the six planes cost
  // 50 more VGPRs than the three-waves-per-SIMD budget has (210-221 against 168); at two workgroups per CU the kernel takes 2.72 against 2.40 ms, capped
  // at 168 registers it spills and takes 4.73 -- this kernel lives on its twelve waves per CU, not on its prefetch depth.  Kept for libbp5_timing.so
  static constexpr bool ROLL = (ABL & 67108864) != 0;
  static_assert(!ROLL || (((ABL & 2048) != 0) && ((ABL & 8192) != 0) && !AFFINE), "rolling prefetch: single-buffered build, sequential tiles, plane geometry");
  using R = PassRegs<n, AFFINE, NPL, ROLL>;
  // all lanes of a cell slot sit in one wave when LPC divides 64: the tile exchanges then need no block barrier
  // WPACK (round 4, p = 2): 9 lanes per cell do not divide a wave, but SEVEN whole cells fit one (63 lanes) and 4 x 7 = 28 = 256 / 9 cells fill the pass
  // all the same: with the cells packed wave by wave no cell spans two waves, every tile exchange is wave-local again (no workgroup barrier, no second
  // tile) -- the kernel maps thread -> (cell slot, lane) accordingly (apply_block_kernel)
  static constexpr bool WPACK = (BP5_WAVE_PACK != 0) && (64 % LPC != 0) && (4 * (64 / LPC) == TEAM / LPC);
  static constexpr bool WAVE_LOCAL = (64 % LPC == 0) || WPACK;
  // SEQ: the transposes go through ONE field tile per cell, field after field (wave-local syncs are free), so a
  // workgroup needs a third of the tile memory: 4x4x4 accumulator + tiles = 47 KB -> three workgroups per CU
  static constexpr bool PACK = (ABL & 262144) != 0; // packed (run, offset) indices, decoded through the LDS run table
  // LATT: EVERY block of the plan is a lattice block (bp5_device.hip: detect_lattice_blocks) -- list slots and DoFs in closed form from the
  // cell's position in its block, no per-DoF index stream at all (a build of its own: the headline kernel sits at the register budget of
  // three waves per SIMD, and this path needs fewer registers than the packed one, both together more)
  static constexpr bool LATT = (ABL & 16777216) != 0;
  static_assert(!LATT || PACK, "lattice build: on top of the packed shape (run tables for the write-out)");
  // STAGE: the brick's src values are staged ONCE in an LDS array indexed like the accumulator (block start: coalesced loads
  // along the runs of the brick's sorted DoF list); the cells gather from LDS -- no global gather instructions, every src
  // entry of a brick crosses HBM/L2 once instead of once per cell that touches it
  static constexpr bool STAGE = (ABL & 524288) != 0;
  static_assert(!STAGE || (PACK && (ABL & 16384)), "LDS-staged src needs packed indices and the run table");
  static constexpr bool SEQ = (ABL & 8192) != 0;
  // (p = 8: 81 lanes per cell span two waves; the tile exchanges then use the workgroup barrier -- correct, every lane reaches every sync)
  // PP: cells that span waves exchange their tiles through the WORKGROUP barrier (20 barriers per pass with one tile: write, barrier, read,
  // barrier, field after field).  Two tiles used alternately need no barrier behind the reads -- the next field goes to the OTHER tile, and
  // by the time a tile is written again every lane has passed the barrier of the field in between: half the barriers per pass
  static constexpr bool PP = SEQ && !WAVE_LOCAL && block_pingpong(P);
  using L = BlockLayout<n, LPC, PP>;
  static constexpr int TILE_ONE = L::ONE;
  static constexpr int TILE_CS = SEQ ? L::SLOT : L::CS; // doubles per cell slot
  static_assert((n - 1) * (L::PS + L::RS + 1) < L::ONE, "a tile holds every (k, j, i) entry");
  static_assert(L::SLOT >= (PP ? 2 : 1) * L::ONE && L::CS >= 3 * n * L::PS, "cell slot holds its tiles");
  static __device__ __forceinline__ void tile_sync()
  {
    if constexpr (WAVE_LOCAL) team_sync<1>();
    else {
      // explicit drain (see lds_drain()): with alternating tiles a tile is rewritten one barrier after its reads -- reads and writes of
      // every wave must have left the LDS queue when the wave arrives (the fence of __syncthreads() asks for the same wait)
      if constexpr (PP) lds_drain();
      __syncthreads();
    }
  }
  // the barrier behind the READS of a tile (write-after-read): not needed between alternating tiles
  static __device__ __forceinline__ void tile_sync_r()
  {
    if constexpr (!PP) tile_sync();
  }

  // issue index / position / metric loads of the cell named by r.ent
  static __device__ __forceinline__ void issue_loads(const ApplyArgs &a, const BlockPlan &bp, R &r, int abm, bool lane_ok, bool exists)
  {
    r.active = lane_ok && exists && !(r.ent >> 31);
    const uint64_t cell = r.ent & 0x7fffffffu;
    if constexpr (LATT) r.cpos = bp.cell_pos[cell];
    else if constexpr (PACK) load_pencil_idx<n, uint16_t>(bp.packed + cell * n3, abm, r.ps); // decoded by decode_and_gather
    else {
      load_pencil_idx<n, uint32_t>(bp.gidx + cell * n3, abm, r.idx);
      load_pencil_idx<n, uint16_t>(bp.pos + cell * n3, abm, r.ps);
    }
    r.round = bp.cell_round[cell];
    if constexpr (HANG) r.mask = a.hang_mask[cell];
    if constexpr (!SINGLE) issue_metric(a, r, abm);
  }
  // ROLL: the whole metric of the cell named by `ent` (first pass of a workgroup only)
  static __device__ __forceinline__ void issue_metric_roll(const ApplyArgs &a, uint32_t ent, int abm, double (&S)[NPL][n])
  {
    const double *cf = a.coef + (uint64_t)(ent & 0x7fffffffu) * a.cell_stride;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      if constexpr (ABL & 2) {
#pragma unroll
        for (int i = 0; i < n; ++i) S[pl][i] = 1.0 + pl + i;
      } else
        load_pencil<n, (ABL & 32768) != 0>(cf + pl * a.plane_stride, abm, S[pl]);
    }
  }
  static __device__ __forceinline__ void issue_metric(const ApplyArgs &a, R &r, int abm)
  {
    if constexpr (ROLL) return;
    const uint64_t cell = r.ent & 0x7fffffffu;
    const double *cf = a.coef + cell * a.cell_stride; // cell base (pair layout, load_pencil)
    if constexpr (AFFINE) {
      load_pencil<n>(cf, abm, r.S[0]);
#pragma unroll
      for (int pl = 0; pl < 6; ++pl) r.Gc[pl] = a.gcell[(uint64_t)pl * a.n_cells_total + cell];
    } else {
#ifdef BP5_TIMING_BUILDS
      // timing-only probes of the metric stream's SHAPE (wrong values by construction; profiles/r4): what would a layout be worth
      //   33554432   in which every wave-instruction reads 1 KB of whole, aligned lines (all 64 lanes load 16 B at consecutive addresses: 12 instead of 18
      //              instructions per pass, no partial line, no 8-byte tail loads)
      //   134217728  in which the unpaired last entries of two planes share one 16-byte load (15 instead of 18 instructions, same lines)
      if constexpr ((ABL & 33554432) != 0 && (n & 1)) {
        const uint64_t cell0 = (uint64_t)__builtin_amdgcn_readfirstlane((int)cell);
        const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          const double *base = a.coef + pl * a.plane_stride + ((cell0 * a.cell_stride) & ~(uint64_t)15);
#pragma unroll
          for (int m = 0; m < n / 2; ++m) {
            const bp5_d2u v = *reinterpret_cast<const bp5_d2u *>(base + (m * 64 + lane) * 2);
            r.S[pl][2 * m] = v.x;
            r.S[pl][2 * m + 1] = v.y;
          }
          r.S[pl][n - 1] = r.S[pl][0] + 1.0;
        }
        return;
      }
      if constexpr ((ABL & 134217728) != 0 && (n & 1) && NPL == 6) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
          for (int m = 0; m < n / 2; ++m) load_pencil_pair<n>(cf + pl * a.plane_stride, abm, m, r.S[pl][2 * m], r.S[pl][2 * m + 1]);
        }
#pragma unroll
        for (int pl = 0; pl < NPL; pl += 2) load_pencil_pair<n>(cf + pl * a.plane_stride, abm, n / 2, r.S[pl][n - 1], r.S[pl + 1][n - 1]); // (reads into the next cell's first pairs: same lines)
        return;
      }
#endif
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        if constexpr (ABL & 2) {
#pragma unroll
          for (int i = 0; i < n; ++i) r.S[pl][i] = 1.0 + pl + i;
        } else
          load_pencil<n, (ABL & 32768) != 0>(cf + pl * a.plane_stride, abm, r.S[pl]);
      }
    }
  }
  // PACK: r.ps holds the packed entries of the pass; turn them into list slots (kept in r.ps for the accumulation) and
  // DoF indices through the run table `rt` of the pass's block, and start the gather
  static __device__ __forceinline__ void decode_and_gather(const ApplyArgs &a, R &r, const uint32_t *rt, const double *staged = nullptr,
                                                           const uint32_t *lt = nullptr, int a_ = 0, int b_ = 0)
  {
    if constexpr (LATT) { // lattice blocks: slots and DoFs in closed form (detect_lattice_blocks verified every entry)
      const uint32_t hdr = lt[54];
      const int NXl = (int)(hdr & 255u) * P, NYl = (int)((hdr >> 8) & 255u) * P, NZl = (int)((hdr >> 16) & 255u) * P;
      const int I = (int)(r.cpos & 15u) * P + a_, J = (int)((r.cpos >> 4) & 15u) * P + b_, K0 = (int)(r.cpos >> 8) * P;
      const int eI = I == 0 ? 0 : I == NXl ? 2 : 1, eJ = J == 0 ? 0 : J == NYl ? 2 : 1;
      const uint32_t LX = eI == 1 ? (uint32_t)(NXl - 1) : 1u, LXY = LX * (eJ == 1 ? (uint32_t)(NYl - 1) : 1u);
      const uint32_t offIJ = (eI == 1 ? (uint32_t)(I - 1) : 0u) + LX * (eJ == 1 ? (uint32_t)(J - 1) : 0u);
      const int entIJ = eI + 3 * eJ;
#pragma unroll
      for (int k = 0; k < n; ++k) {
        const int K = K0 + k, eK = K == 0 ? 0 : K == NZl ? 2 : 1;
        const uint32_t off = offIJ + (eK == 1 ? LXY * (uint32_t)(K - 1) : 0u);
        const int ent = entIJ + 9 * eK;
        r.ps[k] = (uint16_t)(lt[ent] + off);
        if constexpr (STAGE) r.u[k] = staged[r.ps[k]];
        else {
          const uint32_t dof = lt[27 + ent] + off;
          r.u[k] = (ABL & 4) ? 1e-9 * dof : a.src[dof];
        }
      }
      return;
    }
    (void)lt; (void)a_; (void)b_;
#pragma unroll
    for (int k = 0; k < n; ++k) {
      const uint32_t e = r.ps[k], run = e >> BLOCK_PACK_OFF_BITS, off = e & ((1u << BLOCK_PACK_OFF_BITS) - 1u);
      r.ps[k] = (uint16_t)(rt[run] + off);
      if constexpr (STAGE) r.u[k] = staged[r.ps[k]];
      else {
        const uint32_t dof = (rt[BLOCK_MAX_RUNS + run] & BLOCK_DOF_MASK) + off;
        r.u[k] = (ABL & 4) ? 1e-9 * dof : a.src[dof];
      }
    }
  }
  static __device__ __forceinline__ void issue_gather(const ApplyArgs &a, R &r)
  {
#pragma unroll
    for (int k = 0; k < n; ++k) r.u[k] = (ABL & 4) ? 1e-9 * r.idx[k] : a.src[r.idx[k]];
  }

  // one pass: compute with `cur`, keep the loads of `nxt` in flight.  Returns nothing; all
  // block bookkeeping is done by the caller.
  static __device__ __forceinline__ void run(const ApplyArgs &a, const ShapeArg<n> &sh, R &cur, R &nxt, double *T, double *acc, int a_, int b_,
                                             int n_rounds, int abm, unsigned long long (&ph)[8], unsigned long long &tprev, double &energy,
                                             double (&Sr)[ROLL ? NPL : 1][ROLL ? n : 1])
  {
    if constexpr (SINGLE && !ROLL) issue_metric(a, cur, abm);
    BP5_STAMP(0) // issue of this pass's loads
    if constexpr (SEQ) {
#define T1(k, j, i) T[(k) * L::PS + (j) * L::RS + (i)]
#define T2(k, j, i) T[(PP ? TILE_ONE : 0) + (k) * L::PS + (j) * L::RS + (i)]
      const bool act = cur.active;
      double(&uu)[n] = cur.u;
      bool hang_any = false;
      if constexpr (HANG) {
        const bool flagged = act && (cur.mask & BP5_HANG_ANY) != 0;
        if constexpr (WAVE_LOCAL) hang_any = __builtin_amdgcn_ballot_w64(flagged) != 0ull; // the cells of a wave exchange through their own tiles
        else hang_any = __syncthreads_or(flagged) != 0;
        if (hang_any) pencil_hang_resolve<n, WAVE_LOCAL ? 1 : 4, false, L>(cur.mask, a.hang_I, uu, T, a_, b_, act);
      }
      double q0[n], q1[n], q2[n];
      double um[HELM ? n : 1]; // HELM: u at this lane's quadrature points, then a JxW u
      if constexpr (!COLL) {
        double aN[n], aD[n], vN[n], vD[n];
        MV_N(sh.N, uu, aN);
        MV_D(sh.D, uu, aD);
        if (act) {
#pragma unroll
          for (int k = 0; k < n; ++k) T1(k, b_, a_) = aN[k];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) vN[j] = T1(b_, j, a_);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int k = 0; k < n; ++k) T2(k, b_, a_) = aD[k];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) vD[j] = T2(b_, j, a_);
        tile_sync_r();
        double c1[n], c2[n], c3[n], r1[n], r2[n], r3[n];
        MV_N(sh.N, vN, c1);
        MV_D(sh.D, vN, c2);
        MV_N(sh.N, vD, c3);
        if (act) {
#pragma unroll
          for (int j = 0; j < n; ++j) T1(b_, j, a_) = c1[j];
        }
        tile_sync();
#pragma unroll
        for (int i = 0; i < n; ++i) r1[i] = T1(b_, a_, i);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int j = 0; j < n; ++j) T2(b_, j, a_) = c2[j];
        }
        tile_sync();
#pragma unroll
        for (int i = 0; i < n; ++i) r2[i] = T2(b_, a_, i);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int j = 0; j < n; ++j) T1(b_, j, a_) = c3[j];
        }
        tile_sync();
#pragma unroll
        for (int i = 0; i < n; ++i) r3[i] = T1(b_, a_, i);
        tile_sync_r();
        MV_D(sh.D, r1, q0);
        MV_N(sh.N, r2, q1);
        MV_N(sh.N, r3, q2);
        if constexpr (HELM) MV_N(sh.N, r1, um);
      } else {
        double gz[n], vN[n], c2[n], r1[n];
        MV_D(sh.D, uu, gz);
        if (act) {
#pragma unroll
          for (int k = 0; k < n; ++k) T1(k, b_, a_) = uu[k];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) vN[j] = T1(b_, j, a_);
#pragma unroll
        for (int i = 0; i < n; ++i) r1[i] = T1(b_, a_, i);
        tile_sync_r();
        MV_D(sh.D, vN, c2);
        if (act) {
#pragma unroll
          for (int j = 0; j < n; ++j) T2(b_, j, a_) = c2[j];
        }
        tile_sync();
#pragma unroll
        for (int i = 0; i < n; ++i) q1[i] = T2(b_, a_, i);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int k = 0; k < n; ++k) T1(k, b_, a_) = gz[k];
        }
        tile_sync();
#pragma unroll
        for (int i = 0; i < n; ++i) q2[i] = T1(b_, a_, i);
        tile_sync_r();
        MV_D(sh.D, r1, q0);
        if constexpr (HELM) {
#pragma unroll
          for (int i = 0; i < n; ++i) um[i] = r1[i];
        }
      }
      BP5_STAMP(1)
      if constexpr (!PACK) issue_gather(a, nxt); // PACK: after the pass, once the next block's run table is parked
      BP5_STAMP(2)
      auto qpoint = [&](int i, auto &&SM) { // SM(plane): the metric entry of this lane's quadrature point i
        const double x0 = q0[i], x1 = q1[i], x2 = q2[i];
        if constexpr (AFFINE) {
          const double sc = SM(0);
          q0[i] = sc * (cur.Gc[0] * x0 + cur.Gc[3] * x1 + cur.Gc[4] * x2);
          q1[i] = sc * (cur.Gc[3] * x0 + cur.Gc[1] * x1 + cur.Gc[5] * x2);
          q2[i] = sc * (cur.Gc[4] * x0 + cur.Gc[5] * x1 + cur.Gc[2] * x2);
        } else {
          q0[i] = SM(0) * x0 + SM(3) * x1 + SM(4) * x2;
          q1[i] = SM(3) * x0 + SM(1) * x1 + SM(5) * x2;
          q2[i] = SM(4) * x0 + SM(5) * x1 + SM(2) * x2;
        }
        // fused CG: src . (A src) is the sum over cells and quadrature points of ghat^T S ghat -- everything is in
        // registers here, the dot product costs no memory traffic at all
        if constexpr ((ABL & 1048576) != 0) { if (act) energy += x0 * q0[i] + x1 * q1[i] + x2 * q2[i]; }
        if constexpr (HELM) {
          const double uq = um[i];
          um[i] = SM(6) * uq; // submit_value(coef * get_value(q)), JxW folded into the plane
          if constexpr ((ABL & 1048576) != 0) { if (act) energy += uq * um[i]; }
        }
      };
      if constexpr (ROLL) {
        // consume the metric pair by pair and refill each pair with the NEXT pass's values right away (same registers)
        const double *cfn = a.coef + (uint64_t)(nxt.ent & 0x7fffffffu) * a.cell_stride;
        constexpr bool NTM = (ABL & 32768) != 0;
#pragma unroll
        for (int m = 0; m < n / 2; ++m) {
          qpoint(2 * m, [&](int pl) { return Sr[pl][2 * m]; });
          qpoint(2 * m + 1, [&](int pl) { return Sr[pl][2 * m + 1]; });
          __builtin_amdgcn_sched_barrier(0); // the refill must not be hoisted above the last use of the registers it overwrites (it would need new ones)
          if constexpr (!(ABL & 2)) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) load_pencil_pair<n, NTM>(cfn + pl * a.plane_stride, abm, m, Sr[pl][2 * m], Sr[pl][2 * m + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n & 1) {
          qpoint(n - 1, [&](int pl) { return Sr[pl][n - 1]; });
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!(ABL & 2)) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) Sr[pl][n - 1] = load_pencil_tail<n, NTM>(cfn + pl * a.plane_stride, abm);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < n; ++i) qpoint(i, [&](int pl) { return cur.S[pl][i]; });
      }
      BP5_STAMP(3)
      double yy[n];
      if constexpr (!COLL) {
        double e1[n], e2[n], e3[n], w1[n], w2[n], w3[n];
        MV_DT(sh.D, q0, e1);
        if constexpr (HELM) MV_NT_ADD(sh.N, um, e1);
        MV_NT(sh.N, q1, e2);
        MV_NT(sh.N, q2, e3);
        if (act) {
#pragma unroll
          for (int i = 0; i < n; ++i) T2(b_, a_, i) = e1[i];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) w1[j] = T2(b_, j, a_);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int i = 0; i < n; ++i) T1(b_, a_, i) = e2[i];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) w2[j] = T1(b_, j, a_);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int i = 0; i < n; ++i) T2(b_, a_, i) = e3[i];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) w3[j] = T2(b_, j, a_);
        tile_sync_r();
        double f1[n], f2[n], z1[n], z2[n];
        MV_NT(sh.N, w1, f1);
        MV_DT_ADD(sh.D, w2, f1);
        MV_NT(sh.N, w3, f2);
        if (act) {
#pragma unroll
          for (int j = 0; j < n; ++j) T1(b_, j, a_) = f1[j];
        }
        tile_sync();
#pragma unroll
        for (int k = 0; k < n; ++k) z1[k] = T1(k, b_, a_);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int j = 0; j < n; ++j) T2(b_, j, a_) = f2[j];
        }
        tile_sync();
#pragma unroll
        for (int k = 0; k < n; ++k) z2[k] = T2(k, b_, a_);
        tile_sync_r();
        MV_NT(sh.N, z1, yy);
        MV_DT_ADD(sh.D, z2, yy);
      } else {
        double e1[n], w1[n], w2[n], z2[n];
        MV_DT(sh.D, q0, e1);
        if constexpr (HELM) {
#pragma unroll
          for (int i = 0; i < n; ++i) e1[i] += um[i];
        }
        // y-direction: w1 = e1 + D^T q1 (both re-oriented x-owner -> y-owner)
        if (act) {
#pragma unroll
          for (int i = 0; i < n; ++i) T2(b_, a_, i) = e1[i];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) w1[j] = T2(b_, j, a_);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int i = 0; i < n; ++i) T1(b_, a_, i) = q1[i];
        }
        tile_sync();
#pragma unroll
        for (int j = 0; j < n; ++j) w2[j] = T1(b_, j, a_);
        tile_sync_r();
        MV_DT_ADD(sh.D, w2, w1);
        if (act) {
#pragma unroll
          for (int j = 0; j < n; ++j) T2(b_, j, a_) = w1[j];
        }
        tile_sync();
#pragma unroll
        for (int k = 0; k < n; ++k) yy[k] = T2(k, b_, a_);
        tile_sync_r();
        if (act) {
#pragma unroll
          for (int i = 0; i < n; ++i) T1(b_, a_, i) = q2[i];
        }
        tile_sync();
#pragma unroll
        for (int k = 0; k < n; ++k) z2[k] = T1(k, b_, a_);
        tile_sync_r();
        MV_DT_ADD(sh.D, z2, yy);
      }
      if constexpr (HANG) { if (hang_any) pencil_hang_resolve<n, WAVE_LOCAL ? 1 : 4, true, L>(cur.mask, a.hang_I, yy, T, a_, b_, act); } // adjoint
      BP5_STAMP(4)
      for (int rd = 0; rd < n_rounds; ++rd) {
        lds_drain(); // see lds_drain(): the barrier is a loop header
        __syncthreads();
        if (act && cur.round == rd) {
#pragma unroll
          for (int k = 0; k < n; ++k) {
            if constexpr ((ABL & 4194304) != 0) lds_add_f64(acc + cur.ps[k], yy[k]); // ds_add_f64, no return value: one LDS op per entry
            else acc[cur.ps[k]] += yy[k];
          }
        }
      }
      BP5_STAMP(5)
#undef T1
#undef T2
      return;
    }
#define TL(f, k, j, i) T[(f) * (n * L::PS) + (k) * L::PS + (j) * L::RS + (i)]
    const bool active = cur.active;
    double(&u)[n] = cur.u;
    double g0[n], g1[n], g2[n];
    if constexpr (ABL & 8) {
#pragma unroll
      for (int i = 0; i < n; ++i) { g0[i] = u[i]; g1[i] = 2.0 * u[i]; g2[i] = 3.0 * u[i]; }
    } else if constexpr (!COLL) {
      double aN[n], aD[n];
      MV_N(sh.N, u, aN);
      MV_D(sh.D, u, aD);
      if (active) {
#pragma unroll
        for (int k = 0; k < n; ++k) { TL(0, k, b_, a_) = aN[k]; TL(1, k, b_, a_) = aD[k]; }
      }
      tile_sync();
      double vN[n], vD[n];
#pragma unroll
      for (int j = 0; j < n; ++j) { vN[j] = TL(0, b_, j, a_); vD[j] = TL(1, b_, j, a_); }
      double c1[n], c2[n], c3[n];
      MV_N(sh.N, vN, c1);
      MV_D(sh.D, vN, c2);
      MV_N(sh.N, vD, c3);
      tile_sync();
      if (active) {
#pragma unroll
        for (int j = 0; j < n; ++j) { TL(0, b_, j, a_) = c1[j]; TL(1, b_, j, a_) = c2[j]; TL(2, b_, j, a_) = c3[j]; }
      }
      tile_sync();
      double r1[n], r2[n], r3[n];
#pragma unroll
      for (int i = 0; i < n; ++i) { r1[i] = TL(0, b_, a_, i); r2[i] = TL(1, b_, a_, i); r3[i] = TL(2, b_, a_, i); }
      MV_D(sh.D, r1, g0);
      MV_N(sh.N, r2, g1);
      MV_N(sh.N, r3, g2);
    } else {
      double gz[n];
      MV_D(sh.D, u, gz);
      if (active) {
#pragma unroll
        for (int k = 0; k < n; ++k) { TL(0, k, b_, a_) = u[k]; TL(2, k, b_, a_) = gz[k]; }
      }
      tile_sync();
      double vN[n], c2[n];
#pragma unroll
      for (int j = 0; j < n; ++j) vN[j] = TL(0, b_, j, a_);
      MV_D(sh.D, vN, c2);
      if (active) {
#pragma unroll
        for (int j = 0; j < n; ++j) TL(1, b_, j, a_) = c2[j];
      }
      tile_sync();
      double r1[n];
#pragma unroll
      for (int i = 0; i < n; ++i) { r1[i] = TL(0, b_, a_, i); g1[i] = TL(1, b_, a_, i); g2[i] = TL(2, b_, a_, i); }
      MV_D(sh.D, r1, g0);
    }

    BP5_STAMP(1) // evaluate (z/y/x contractions; its first use of u waits for the gather)
    // the index loads of the next pass have landed by now: start its src gather
    if constexpr (!PACK) issue_gather(a, nxt);
    BP5_STAMP(2) // wait for the next pass's indices + issue of its gather
    BP5_PRIO_PHASE(BP5_PRIO_Q)

#pragma unroll
    for (int i = 0; i < n; ++i) {
      const double x0 = g0[i], x1 = g1[i], x2 = g2[i];
      if constexpr (AFFINE) {
        const double sc = cur.S[0][i];
        g0[i] = sc * (cur.Gc[0] * x0 + cur.Gc[3] * x1 + cur.Gc[4] * x2);
        g1[i] = sc * (cur.Gc[3] * x0 + cur.Gc[1] * x1 + cur.Gc[5] * x2);
        g2[i] = sc * (cur.Gc[4] * x0 + cur.Gc[5] * x1 + cur.Gc[2] * x2);
      } else {
        g0[i] = cur.S[0][i] * x0 + cur.S[3][i] * x1 + cur.S[4][i] * x2;
        g1[i] = cur.S[3][i] * x0 + cur.S[1][i] * x1 + cur.S[5][i] * x2;
        g2[i] = cur.S[4][i] * x0 + cur.S[5][i] * x1 + cur.S[2][i] * x2;
      }
    }

    BP5_STAMP(3) // quadrature-point operation (first use of the metric: waits for its loads)
    BP5_PRIO_PHASE(BP5_PRIO_I)
    double y[n];
    if constexpr (ABL & 8) {
#pragma unroll
      for (int i = 0; i < n; ++i) y[i] = g0[i] + g1[i] + g2[i];
    } else if constexpr (!COLL) {
      double e1[n], e2[n], e3[n];
      MV_DT(sh.D, g0, e1);
      MV_NT(sh.N, g1, e2);
      MV_NT(sh.N, g2, e3);
      tile_sync();
      if (active) {
#pragma unroll
        for (int i = 0; i < n; ++i) { TL(0, b_, a_, i) = e1[i]; TL(1, b_, a_, i) = e2[i]; TL(2, b_, a_, i) = e3[i]; }
      }
      tile_sync();
      double w1[n], w2[n], w3[n];
#pragma unroll
      for (int j = 0; j < n; ++j) { w1[j] = TL(0, b_, j, a_); w2[j] = TL(1, b_, j, a_); w3[j] = TL(2, b_, j, a_); }
      double f1[n], f2[n];
      MV_NT(sh.N, w1, f1);
      MV_DT_ADD(sh.D, w2, f1);
      MV_NT(sh.N, w3, f2);
      tile_sync();
      if (active) {
#pragma unroll
        for (int j = 0; j < n; ++j) { TL(0, b_, j, a_) = f1[j]; TL(1, b_, j, a_) = f2[j]; }
      }
      tile_sync();
      double z1[n], z2[n];
#pragma unroll
      for (int k = 0; k < n; ++k) { z1[k] = TL(0, k, b_, a_); z2[k] = TL(1, k, b_, a_); }
      MV_NT(sh.N, z1, y);
      MV_DT_ADD(sh.D, z2, y);
    } else {
      double e1[n];
      MV_DT(sh.D, g0, e1);
      tile_sync();
      if (active) {
#pragma unroll
        for (int i = 0; i < n; ++i) { TL(0, b_, a_, i) = e1[i]; TL(1, b_, a_, i) = g1[i]; TL(2, b_, a_, i) = g2[i]; }
      }
      tile_sync();
      double w1[n], w2[n];
#pragma unroll
      for (int j = 0; j < n; ++j) { w1[j] = TL(0, b_, j, a_); w2[j] = TL(1, b_, j, a_); }
      MV_DT_ADD(sh.D, w2, w1);
      if (active) {
#pragma unroll
        for (int j = 0; j < n; ++j) TL(0, b_, j, a_) = w1[j];
      }
      tile_sync();
      double z2[n];
#pragma unroll
      for (int k = 0; k < n; ++k) { y[k] = TL(0, k, b_, a_); z2[k] = TL(2, k, b_, a_); }
      MV_DT_ADD(sh.D, z2, y);
    }

    BP5_STAMP(4) // integrate
    BP5_PRIO_PHASE(BP5_PRIO_A)
    // accumulate into the block's LDS vector; cells of one round share no DoF (normally the
    // whole pass is one round: host packing, bp5_host.cpp)
    // One block barrier per round, BEFORE the update: it orders this pass's accumulation after the previous
    // pass's (cells of different passes may share DoFs and run on different waves).
    if constexpr (!(ABL & 1)) {
      for (int rd = 0; rd < n_rounds; ++rd) {
        lds_drain(); // see lds_drain(): the barrier is a loop header
        __syncthreads();
        if (active && cur.round == rd) {
#pragma unroll
          for (int k = 0; k < n; ++k) acc[cur.ps[k]] += y[k];
        }
      }
    } else {
      double sacc = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) sacc += y[k];
      if (sacc == 1.2345e300) acc[cur.ps[0]] = sacc;
      __syncthreads();
    }
    BP5_STAMP(5) // accumulate
    BP5_PRIO_PHASE(BP5_PRIO_END)
#undef TL
  }
};

// the rolling-prefetch builds (BlockPass::ROLL; libbp5_timing.so only) take 210-221 VGPRs: two workgroups per CU (capped at three waves per SIMD they spill
// 100-150 registers and run at half the speed)
#define BP5_ROLL_TWO(ABL) (((ABL) & 67108864) != 0)
// workgroups per CU (= waves per SIMD: 256-thread workgroups) the block kernel is compiled for and launched with: three for the single-buffered builds
// up to p = 4 (168 VGPRs), two otherwise (p >= 5: registers; Helmholtz; the rolling-prefetch probe).  p = 1 (round 4): FOUR -- the kernel needs 101 VGPRs
// and 15 KB of LDS, and its waves wait two thirds of their life (profiles/r4 g_*): a pass is 64 cells with 6 loads per lane, all latency
#ifndef BP5_WG_PER_CU_P1
#define BP5_WG_PER_CU_P1 4
#endif
#ifndef BP5_WG_PER_CU_P2
#define BP5_WG_PER_CU_P2 4
#endif
template <int P, int ABL>
constexpr int block_wg_per_cu()
{
  if (!(ABL & 2048) || (ABL & 8388608) || BP5_ROLL_TWO(ABL) || P > 4) return 2;
  if (P == 1 && !(ABL & 2097152) && !(ABL & 1024)) return BP5_WG_PER_CU_P1;
  if (P == 2 && BP5_WAVE_PACK != 0 && !(ABL & 2097152) && !(ABL & 1024)) return BP5_WG_PER_CU_P2; // (wave-packed cells: one tile per slot, 36 KB of LDS on 8x8x4 bricks, 126 VGPRs)
  return 3;
}
// wave priority (round 4): the instructions that ISSUE a pass's loads (metric planes, indices, gather) and the write-out of a block run at s_setprio 3, the
// arithmetic at 0 -- a wave about to put memory requests in flight goes ahead of the waves of the other workgroups on its SIMD that are computing.  Rotating
// A/B of two builds, one box per comparison (profiles/r4/x_wave_priority_ab.txt): fused p = 4 kernel 2.649 -> 2.572, 2.770 -> 2.738, 2.768 -> 2.721, 2.619 -> 2.562 ms -- and
// 2.548 -> 2.586 on one box of six; either bracket alone gains nothing, raising the priority towards the accumulation barrier as well (BP5_PRIO_Q / _I / _A
// below) nothing more.  Compile-time (-DBP5_SETPRIO=0 -DBP5_SETPRIO_WO=0 rebuilds the
// old code): a run-time switch around s_setprio costs the headline build 15 spilled registers.
#ifndef BP5_SETPRIO
#define BP5_SETPRIO 3
#endif
#ifndef BP5_SETPRIO_WO
#define BP5_SETPRIO_WO 3
#endif
// (lattice and Helmholtz builds: measured there; the packed-index build of p = 4 sits at its 168 registers and would spill two)
#define BP5_PRIO_ON (BP::LATT || (ABL & 8388608) != 0)
#define BP5_PRIO_HI if constexpr (BP5_SETPRIO != 0 && BP5_PRIO_ON) __builtin_amdgcn_s_setprio(BP5_SETPRIO);
#define BP5_PRIO_LO if constexpr (BP5_SETPRIO != 0 && BP5_PRIO_ON) __builtin_amdgcn_s_setprio(0);
#define BP5_PRIO_WO_HI if constexpr (BP5_SETPRIO_WO != 0 && BP5_PRIO_ON) __builtin_amdgcn_s_setprio(BP5_SETPRIO_WO);
#define BP5_PRIO_WO_LO if constexpr (BP5_SETPRIO_WO != 0 && BP5_PRIO_ON) __builtin_amdgcn_s_setprio(0);
template <int P, bool COLL, int LPC, int SCATTER, int ABL = 0>
__global__ void __launch_bounds__(256, (block_wg_per_cu<P, ABL>())) apply_block_kernel(ApplyArgs a, BlockPlan bp, ShapeArg<P + 1> sh)
{
  using BP = BlockPass<P, COLL, LPC, SCATTER, ABL>;
  constexpr int n = P + 1, n2 = n * n;
  constexpr int TEAM = 256;
  constexpr int CPT = TEAM / LPC;
  static_assert(LPC >= n2 && CPT >= 1, "lanes per cell");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *acc = lds + CPT * BP::TILE_CS; // accumulator behind the transpose tiles

  const int t = threadIdx.x;
  // thread -> (cell slot c, lane ab of the cell): consecutive threads, or (BlockPass::WPACK) whole cells wave by wave, the wave's last lanes idle
  constexpr int CW = 64 / LPC; // whole cells per wave
  const int c = BP::WPACK ? (((t & 63) < CW * LPC) ? (t >> 6) * CW + (t & 63) / LPC : CPT) : t / LPC;
  const int ab = BP::WPACK ? (t & 63) % LPC : t - c * LPC;
  // persistent workgroup w owns the contiguous block range [b0,b1); workgroups that share an XCD
  // (blockIdx % 8, speed only) own neighbouring ranges
  const uint32_t w = (blockIdx.x & 7u) * (bp.n_wg >> 3) + (blockIdx.x >> 3);
  // DOTS: the write-out also accumulates the merged CG's v-dependent dot products over the DoFs it stores (and writes src
  // instead of the sum on Dirichlet DoFs); brick-surface DoFs are handled the same way by combine_runs_kernel<.., true>
  constexpr bool DOTS = (ABL & 1048576) != 0;
  static_assert(!DOTS || (((ABL & 16384) != 0) && ((ABL & 8192) != 0) && SCATTER == SC_OWNER_SET), "fused dot products: run-length write-out, sequential tiles, overwrite mode");
  double ds[4] = {0.0, 0.0, 0.0, 0.0};
  // this workgroup is through with its ghost-touching bricks: release its stores (every wave's -- the barrier orders them before
  // thread 0's device-scope fence) and count it in
  auto signal_part_done = [&](bool stores_pending) {
    if (stores_pending) { lds_drain(); __syncthreads(); }
    if (t == 0) {
      // release only (L2 write-back, no invalidate: the brick-local re-reads of src keep their L2 lines); whoever consumes the rows is a
      // later kernel launch on the communication stream, which acquires at its start
      if (stores_pending) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(bp.signal, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if constexpr (DOTS) {
    if (bp.cg_state[0]) { // the solve has stopped: a no-op launch -- whoever waits for the ghost rows must still be released
      if (bp.signal) signal_part_done(false);
      return;
    }
  }
  // The workgroup's blocks: [ba0, ba1) (part 0) then [bb0, bb1) (part 1; empty in one-part launches), walked as ONE sequence of
  // virtual block indices vb = 0 .. nb - 1 and virtual pass indices gp = 0 .. gp_end - 1; only the look-ups translate (scalar
  // arithmetic: everything here is wave-uniform)
  const uint32_t ba0 = bp.wg_block[w], ba1 = bp.wg_block[w + 1];
  const uint32_t bb0 = bp.n_parts > 1 ? bp.wg_block[bp.n_wg + 1 + w] : 0u, bb1 = bp.n_parts > 1 ? bp.wg_block[bp.n_wg + 1 + w + 1] : 0u;
  const uint32_t nba = ba1 - ba0, b1 = nba + (bb1 - bb0);          // blocks of part 0 / of both parts
  if (b1 == 0) {
    if constexpr (DOTS) { if (t < 7) bp.dot_partials[t * PARTIAL_STRIDE + bp.dot_col0 + blockIdx.x] = 0.0; } // an idle workgroup still owns a column
    if (bp.signal) signal_part_done(false);
    return;
  }
  if (bp.signal && nba == 0) signal_part_done(false);               // no ghost-touching brick in this workgroup's share
  const uint32_t pa0 = bp.pass_off[ba0], npa = bp.pass_off[ba1] - pa0, pb0 = bp.pass_off[bb0]; // physical first passes, passes of part 0
  auto blk = [&](uint32_t vb) -> uint32_t { return vb < nba ? ba0 + vb : bb0 + (vb - nba); };      // physical block of a virtual one
  uint32_t b = 0;                                                   // virtual block index
  uint32_t pb = blk(0);                                             // ... and its physical id
  uint32_t gp = 0;
  const uint32_t gp_end = npa + (bp.pass_off[bb1] - pb0);
  uint32_t boundary = bp.pass_off[pb + 1] - bp.pass_off[pb];        // virtual pass index behind the current block
  uint32_t o0 = bp.off[pb];
  int m = (int)(bp.off[pb + 1] - o0);
  int n_rounds = bp.blk_rounds[pb];
  // write-out by runs: the block's run table is fetched into registers at the top of its last pass, parked in one of
  // two LDS tables (block parity: a wave may still be writing out block b while another one enters b + 1) right
  // before the write-out barrier, and every thread walks it forward for its slots -- no list loads in the write-out
  constexpr bool RUNS = (ABL & 16384) != 0;
  constexpr bool STAGE = BP::STAGE;
  double *const staged = acc + bp.max_list; // STAGE: src values of the current brick, indexed like acc
  uint32_t *const run_tab = reinterpret_cast<uint32_t *>(acc + (STAGE ? 2 : 1) * (size_t)bp.max_list);
  // lattice tables of the current / next block behind the two run tables (block parity, parked like them)
  uint32_t *const lat_tab = run_tab + 4 * BLOCK_MAX_RUNS;
  constexpr bool use_lattice = BP::LATT;
  uint32_t lat_word = 0;
  // face carry: two buffers (block parity: the write-out of block b reads the one block b - 1 filled and fills the other)
  constexpr bool CARRY = (ABL & 268435456) != 0;
  static_assert(!CARRY || (BP::LATT && RUNS), "face carry: lattice blocks with run tables");
  double *const carry_buf = reinterpret_cast<double *>(lat_tab + 2 * BLOCK_LATTICE_WORDS);
  bool c_carried = false;             // the previous block of this workgroup handed its face on
  uint32_t c_in_w = 0, c_out_w = 0;   // write-out of the current block: DoFs << 16 | first slot of the face taken over / handed on (0: none)
  auto c_in_at = [&](int i) { return (uint32_t)(i - (int)(c_in_w & 0xffffu)) < (c_in_w >> 16); };
  auto c_out_at = [&](int i) { return (uint32_t)(i - (int)(c_out_w & 0xffffu)) < (c_out_w >> 16); };
  uint32_t r0 = RUNS ? bp.run_off[pb] : 0u;
  int n_runs = RUNS ? (int)(bp.run_off[pb + 1] - r0) : 0;
  uint32_t run_slot = 0, run_dof = 0;
  const bool lane_ok = (ab < n2) && (c < CPT);
  const int abm = ab < n2 ? ab : ab % n2;
  const int slot = c < CPT ? c : 0;
  const int a_ = abm % n, b_ = abm / n;
  double *T = lds + slot * BP::TILE_CS;

  // the whole accumulator, not only this workgroup's first block: a later block of the range may be longer (a
  // partial brick first, full bricks after it), and write-outs re-arm only the slots they read
  for (int i = t; i < (int)bp.max_list; i += TEAM) acc[i] = 0.0;
  __syncthreads();

  auto entry = [&](uint32_t pass) {
    const uint32_t v = pass < gp_end ? pass : gp_end - 1;           // virtual pass -> physical pass
    return bp.pass_cell[(uint64_t)(v < npa ? pa0 + v : pb0 + (v - npa)) * CPT + slot];
  };
  // STAGE: fill staged[0, m_) with the src values of the brick whose run table rt_ (n_runs_ runs) is in LDS: thread t takes the
  // slot pairs (2t, 2t+1) + 2 TEAM j and walks the table forward, consecutive lanes read consecutive DoFs of a run
  auto enter_block = [&](const uint32_t *rt_, int n_runs_, int m_) {
    int r = 0;
    for (int i = 2 * t; i < m_; i += 2 * TEAM) {
      while (r + 1 < n_runs_ && (int)rt_[r + 1] <= i) ++r;
      const uint32_t g = (rt_[BLOCK_MAX_RUNS + r] & BLOCK_DOF_MASK) + (uint32_t)(i - (int)rt_[r]);
      const bool run_ends = r + 1 < n_runs_ && (int)rt_[r + 1] == i + 1;
      if (i + 1 < m_ && !run_ends) {
        const bp5_d2u v = *reinterpret_cast<const bp5_d2u *>(a.src + g);
        staged[i] = v.x;
        staged[i + 1] = v.y;
      } else {
        staged[i] = a.src[g];
        if (i + 1 < m_) staged[i + 1] = a.src[rt_[BLOCK_MAX_RUNS + r + 1] & BLOCK_DOF_MASK]; // first slot of the next run
      }
    }
  };

  // two register sets, used alternately (loop unrolled by two): the loads of pass q+1 are issued
  // at the top of pass q and first waited for inside pass q+1
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  if constexpr (ABL & 4096) tprev = stamp_now();
  typename BP::R A, B;
  double Sroll[BP::ROLL ? BP::NPL : 1][BP::ROLL ? n : 1]; // ROLL: the one metric register set, refilled piece by piece (BlockPass::run)
  A.ent = entry(gp);
  B.ent = entry(gp + 1);
  BP::issue_loads(a, bp, A, abm, lane_ok, true);
  if constexpr (BP::ROLL) BP::issue_metric_roll(a, A.ent, abm, Sroll);
  if constexpr (BP::PACK) {
    // the first block's run table (and lattice table) must be in LDS before the first decode
    uint32_t *const rt0 = run_tab + (b & 1u) * (2 * BLOCK_MAX_RUNS);
    if (t < n_runs) {
      rt0[t] = bp.runs[2 * (r0 + t)];
      rt0[BLOCK_MAX_RUNS + t] = bp.runs[2 * (r0 + t) + 1];
    }
    if constexpr (use_lattice) { if (t < BLOCK_LATTICE_WORDS) lat_tab[(b & 1u) * BLOCK_LATTICE_WORDS + t] = bp.lattice[(uint64_t)pb * BLOCK_LATTICE_WORDS + t]; }
    __syncthreads();
    if constexpr (STAGE) {
      enter_block(rt0, n_runs, m);
      __syncthreads();
    }
    BP::decode_and_gather(a, A, rt0, staged, use_lattice ? lat_tab + (b & 1u) * BLOCK_LATTICE_WORDS : nullptr, a_, b_);
  } else
    BP::issue_gather(a, A);

  // end-of-pass bookkeeping: when a block is finished, write it out and re-arm the accumulator.  The block's
  // DoF list is prefetched into registers at the top of its last pass (prefetch_list), so the write-out does
  // not wait on dependent loads; entries beyond MAXW per thread (very large blocks) take the slow path.
  constexpr int MAXW = (ABL & 2048) ? 1 : 6; // registers are scarce (the single-buffered build is capped at 168)
  uint32_t gl[MAXW];
  // bookkeeping scalars of the NEXT block, fetched at the top of this block's last pass so that the block switch
  // does not wait for them
  uint32_t nx_boundary = 0, nx_o0 = 0, nx_o1 = 0, nx_r0 = 0, nx_r1 = 0;
  int nx_rounds = 1;
  auto prefetch_list = [&]() {
    if (gp + 1 == boundary && b + 1 < b1) {
      const uint32_t p1 = blk(b + 1);
      nx_boundary = boundary + (bp.pass_off[p1 + 1] - bp.pass_off[p1]);
      nx_o0 = bp.off[p1];
      nx_o1 = bp.off[p1 + 1];
      nx_rounds = bp.blk_rounds[p1];
      if constexpr (RUNS) {
        nx_r0 = bp.run_off[p1];
        nx_r1 = bp.run_off[p1 + 1];
      }
      if constexpr (use_lattice) { if (t < BLOCK_LATTICE_WORDS) lat_word = bp.lattice[(uint64_t)p1 * BLOCK_LATTICE_WORDS + t]; }
    }
    if (gp + 1 == boundary) {
      if constexpr (BP::PACK) {
        // table of the NEXT block (this block's own table has been in LDS since the end of the previous block)
        if (b + 1 < b1) {
          if (t < (int)(nx_r1 - nx_r0)) {
            run_slot = bp.runs[2 * (nx_r0 + t)];
            run_dof = bp.runs[2 * (nx_r0 + t) + 1];
          }
        }
      } else if constexpr (RUNS) {
        if (t < n_runs) {
          run_slot = bp.runs[2 * (r0 + t)];
          run_dof = bp.runs[2 * (r0 + t) + 1];
        }
      } else {
#pragma unroll
        for (int r = 0; r < MAXW; ++r) gl[r] = (t + r * TEAM < m) ? bp.dofs[o0 + t + r * TEAM] : 0u;
      }
    }
  };
  // face carry: a slot of the face taken over from the previous block is an owner slot here (carry_mark sets bit 31 of its run word in the
  // walks below) and its value is this block's sum + the carried one; a slot of the face handed on goes to the carry buffer, not the slab
  auto carry_mark = [&](int i, uint32_t g) -> uint32_t {
    if constexpr (CARRY) { if (c_in_at(i)) return g | 0x80000000u; }
    return g;
  };
  auto emit = [&](int i, uint32_t g, double ri = 0.0) { // ri: DOTS builds, r[g] loaded ahead of the stores
    double v = acc[i];
    acc[i] = 0.0; // re-arm for the next block (invariant: the accumulator is all zero between blocks)
    if constexpr (CARRY) {
      if (c_in_at(i)) v += carry_buf[((b + 1) & 1u) * BLOCK_CARRY_MAX + (i - (int)(c_in_w & 0xffffu))];
      if (c_out_at(i)) { carry_buf[(b & 1u) * BLOCK_CARRY_MAX + (i - (int)(c_out_w & 0xffffu))] = v; return; }
    }
    if (g & 0x80000000u) {
      const uint32_t gi = g & BLOCK_DOF_MASK;
      if constexpr (DOTS) {
        // p.v comes from the cells (BlockPass::run: energy); a Dirichlet row stores v = p instead of the assembled sum, which
        // changes that product by p (p - sum)   (copy_constrained_values, bp5/step-64.cu:275)
        double vi = v;
        if (g & BLOCK_DOF_CONSTRAINED) {
          vi = a.src[gi];
          if (gi < bp.n_owned) ds[0] = __builtin_fma(vi, vi - v, ds[0]);
        }
        __builtin_nontemporal_store(vi, a.dst + gi);
        // (explicit fused multiply-adds, here and in emit2: left to the compiler's contraction, `s += a a + b b` rounds differently from one
        // build of this kernel to the next -- the packed and the lattice build must give the same bits)
        if (gi < bp.n_owned) { ds[1] = __builtin_fma(vi, vi, ds[1]); ds[2] = __builtin_fma(ri, vi, ds[2]); ds[3] = __builtin_fma(ri, ri, ds[3]); }
      } else if constexpr (ABL & 16) { if (v == 1.2345e300) a.dst[gi] = v; }
      else if constexpr (SCATTER == SC_OWNER_SET || SCATTER == SC_OWNER_SET_ATOMIC) {
        if constexpr (ABL & 65536) a.dst[gi] = v;
        else __builtin_nontemporal_store(v, a.dst + gi);
      }
      else a.dst[gi] += v;
    } else {
      if constexpr (ABL & 16) { if (v == 1.2345e300) bp.partial[o0 + i] = v; }
      else if constexpr (SCATTER == SC_OWNER_SET_ATOMIC || SCATTER == SC_OWNER_ADD_ATOMIC) atomic_add_f64(a.dst + (g & BLOCK_DOF_MASK), v); // (a shared run may carry the Dirichlet flag)
      else if constexpr (ABL & 65536) bp.partial[o0 + i] = v;
      else __builtin_nontemporal_store(v, bp.partial + o0 + i);
    }
  };
  // two consecutive list slots of one run: 16-byte LDS read, 16-byte global store (half the store instructions)
  auto emit2 = [&](int i, uint32_t g, bp5_d2u rv = bp5_d2u{0.0, 0.0}) {
    double v0 = acc[i], v1 = acc[i + 1];
    acc[i] = 0.0;
    acc[i + 1] = 0.0;
    if constexpr (CARRY) { // (a carried face is exactly one run: both slots of the pair or none)
      if (c_in_at(i)) {
        const double *cb = carry_buf + ((b + 1) & 1u) * BLOCK_CARRY_MAX + (i - (int)(c_in_w & 0xffffu));
        v0 += cb[0]; v1 += cb[1];
      }
      if (c_out_at(i)) {
        double *cb = carry_buf + (b & 1u) * BLOCK_CARRY_MAX + (i - (int)(c_out_w & 0xffffu));
        cb[0] = v0; cb[1] = v1;
        return;
      }
    }
    if (g & 0x80000000u) {
      const uint32_t gi = g & BLOCK_DOF_MASK;
      double *d = a.dst + gi;
      if constexpr (DOTS) {
        double w0 = v0, w1 = v1;
        if (g & BLOCK_DOF_CONSTRAINED) { // see emit(): Dirichlet rows
          const bp5_d2u pv = *reinterpret_cast<const bp5_d2u *>(a.src + gi);
          w0 = pv.x; w1 = pv.y;
          if (gi < bp.n_owned) ds[0] = __builtin_fma(w0, w0 - v0, ds[0]);
          if (gi + 1 < bp.n_owned) ds[0] = __builtin_fma(w1, w1 - v1, ds[0]);
        }
        __builtin_nontemporal_store(bp5_d2u{w0, w1}, reinterpret_cast<bp5_d2u *>(d));
        if (gi < bp.n_owned) { ds[1] = __builtin_fma(w0, w0, ds[1]); ds[2] = __builtin_fma(rv.x, w0, ds[2]); ds[3] = __builtin_fma(rv.x, rv.x, ds[3]); }
        if (gi + 1 < bp.n_owned) { ds[1] = __builtin_fma(w1, w1, ds[1]); ds[2] = __builtin_fma(rv.y, w1, ds[2]); ds[3] = __builtin_fma(rv.y, rv.y, ds[3]); } // (a pair may straddle the end of the owned range)
      } else if constexpr (SCATTER == SC_OWNER_SET || SCATTER == SC_OWNER_SET_ATOMIC) {
        if constexpr (ABL & 65536) { d[0] = v0; d[1] = v1; }
        else __builtin_nontemporal_store(bp5_d2u{v0, v1}, reinterpret_cast<bp5_d2u *>(d));
      } else {
        const bp5_d2u o = *reinterpret_cast<const bp5_d2u *>(d);
        *reinterpret_cast<bp5_d2u *>(d) = bp5_d2u{o.x + v0, o.y + v1};
      }
    } else {
      if constexpr (SCATTER == SC_OWNER_SET_ATOMIC || SCATTER == SC_OWNER_ADD_ATOMIC) {
        atomic_add_f64(a.dst + (g & BLOCK_DOF_MASK), v0);
        atomic_add_f64(a.dst + (g & BLOCK_DOF_MASK) + 1, v1);
      } else if constexpr (ABL & 65536) { bp.partial[o0 + i] = v0; bp.partial[o0 + i + 1] = v1; }
      else __builtin_nontemporal_store(bp5_d2u{v0, v1}, reinterpret_cast<bp5_d2u *>(bp.partial + o0 + i));
    }
  };
  auto finish_pass = [&]() {
    if (gp + 1 == boundary) {
      uint32_t *const rt = run_tab + (b & 1u) * (2 * BLOCK_MAX_RUNS);
      if constexpr (BP::PACK) {
        // park the next block's table in the other buffer: its last readers (block b - 1) finished before this
        // block's first accumulation barrier
        if (b + 1 < b1) {
          uint32_t *const rn = run_tab + ((b + 1) & 1u) * (2 * BLOCK_MAX_RUNS);
          if (t < (int)(nx_r1 - nx_r0)) {
            rn[t] = run_slot;
            rn[BLOCK_MAX_RUNS + t] = run_dof;
          }
          if constexpr (use_lattice) { if (t < BLOCK_LATTICE_WORDS) lat_tab[((b + 1) & 1u) * BLOCK_LATTICE_WORDS + t] = lat_word; }
        }
      } else if constexpr (RUNS) {
        if (t < n_runs) {
          rt[t] = run_slot;
          rt[BLOCK_MAX_RUNS + t] = run_dof;
        }
      }
      __syncthreads(); // every wave has added its last contributions of this block
      BP5_PRIO_WO_HI
      if constexpr (CARRY) {
        const uint32_t *lt = lat_tab + (b & 1u) * BLOCK_LATTICE_WORDS;
        uint32_t wo = 0u, wi = 0u;
        if (bp.carry) { wo = lt[55]; wi = lt[56]; }
        // the successor must be the next block of the SAME part of this workgroup's range (the host applies the same rule to the combine tables)
        c_out_w = (b + 1 < b1 && b + 1 != nba) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)wo) : 0u;
        c_in_w = c_carried ? (uint32_t)__builtin_amdgcn_readfirstlane((int)wi) : 0u;
      }
      if constexpr (DOTS) {
        // as below, in batches of NW pairs per thread: first the run walk and ALL the batch's loads of r (they are
        // independent: one memory latency per batch instead of one per pair), then the stores and the dot products
        constexpr int NW = 5;
        const bool has_r = bp.cg_r != nullptr; // NULL: only p.v is wanted (standard CG), r is not read
        int r = 0;
        for (int base = 2 * t; base < m; base += NW * 2 * TEAM) {
          uint32_t g0[NW], g1[NW];
          int kind[NW]; // 0 nothing, 1 pair inside one run, 2 one slot, 3 one slot + the first slot of the next run
          bp5_d2u rv[NW];
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            const int i = base + j * 2 * TEAM;
            g0[j] = g1[j] = 0u;
            kind[j] = 0;
            rv[j] = bp5_d2u{0.0, 0.0};
            if (i < m) {
              while (r + 1 < n_runs && (int)rt[r + 1] <= i) ++r;
              g0[j] = carry_mark(i, rt[BLOCK_MAX_RUNS + r] + (uint32_t)(i - (int)rt[r]));
              const bool run_ends = r + 1 < n_runs && (int)rt[r + 1] == i + 1;
              const uint32_t gi = g0[j] & BLOCK_DOF_MASK; // (every DoF of a list is < n_local: the vectors hold owned + ghost entries)
              if (i + 1 < m && !run_ends) {
                kind[j] = 1;
                if (has_r && (g0[j] & 0x80000000u)) rv[j] = *reinterpret_cast<const bp5_d2u *>(bp.cg_r + gi);
              } else {
                kind[j] = 2;
                if (has_r && (g0[j] & 0x80000000u)) rv[j].x = bp.cg_r[gi];
                if (i + 1 < m) {
                  kind[j] = 3;
                  g1[j] = carry_mark(i + 1, rt[BLOCK_MAX_RUNS + r + 1]);
                  if (has_r && (g1[j] & 0x80000000u)) rv[j].y = bp.cg_r[g1[j] & BLOCK_DOF_MASK];
                }
              }
            }
          }
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            const int i = base + j * 2 * TEAM;
            if (kind[j] == 1) emit2(i, g0[j], rv[j]);
            else if (kind[j]) {
              emit(i, g0[j], rv[j].x);
              if (kind[j] == 3) emit(i + 1, g1[j], rv[j].y);
            }
          }
        }
      } else if constexpr (RUNS && !(ABL & 1)) {
        // thread t takes the slot pairs (2t, 2t+1) + 2 TEAM j: a pair inside one run goes out as 16 bytes
        int r = 0;
        for (int i = 2 * t; i < m; i += 2 * TEAM) {
          while (r + 1 < n_runs && (int)rt[r + 1] <= i) ++r;
          const uint32_t g = carry_mark(i, rt[BLOCK_MAX_RUNS + r] + (uint32_t)(i - (int)rt[r]));
          const bool run_ends = r + 1 < n_runs && (int)rt[r + 1] == i + 1;
          if (i + 1 < m && !run_ends) emit2(i, g);
          else {
            emit(i, g);
            if (i + 1 < m) emit(i + 1, carry_mark(i + 1, rt[BLOCK_MAX_RUNS + r + 1])); // first slot of the next run
          }
        }
      }
      if constexpr (!RUNS && !(ABL & 1)) {
#pragma unroll
        for (int r = 0; r < MAXW; ++r) {
          const int i = t + r * TEAM;
          if (i < m) emit(i, gl[r]);
        }
        for (int base = t + MAXW * TEAM; base < m; base += 8 * TEAM) { // remainder: batches of eight loads
          uint32_t g8[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) g8[r] = (base + r * TEAM < m) ? bp.dofs[o0 + base + r * TEAM] : 0u;
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (base + r * TEAM < m) emit(base + r * TEAM, g8[r]);
        }
      }
      if constexpr (CARRY) c_carried = c_out_w != 0u;
      BP5_PRIO_WO_LO
      ++b;
      if (bp.signal && b == nba) signal_part_done(true);            // the last ghost-touching brick of this workgroup is written out
      if (b < b1) {
        boundary = nx_boundary;
        o0 = nx_o0;
        m = (int)(nx_o1 - nx_o0);
        n_rounds = nx_rounds;
        if constexpr (RUNS) {
          r0 = nx_r0;
          n_runs = (int)(nx_r1 - nx_r0);
        }
        if constexpr (STAGE) {
          // the next brick's table was parked before the write-out barrier; every pass of the finished brick has read its
          // staged values (same barrier), so the array can be refilled now; the cells of the next pass read it after the barrier
          enter_block(run_tab + (b & 1u) * (2 * BLOCK_MAX_RUNS), n_runs, m);
          __syncthreads();
        }
      }
      // no barrier here (unstaged builds): the next pass starts with tile work and reaches the accumulation barrier before it
      // touches the accumulator again
    }
    ++gp;
  };

  while (gp < gp_end) {
    prefetch_list();
    BP5_PRIO_HI BP::issue_loads(a, bp, B, abm, lane_ok, gp + 1 < gp_end); BP5_PRIO_LO
    const uint32_t entA2 = entry(gp + 2);
    BP::run(a, sh, A, B, T, acc, a_, b_, n_rounds, abm, ph, tprev, ds[0], Sroll);
    finish_pass();
    BP5_STAMP(6) // block boundary: write-out + re-arm (zero in passes that do not end a block)
    if (gp >= gp_end) break;
    BP5_PRIO_HI if constexpr (BP::PACK) BP::decode_and_gather(a, B, run_tab + (b & 1u) * (2 * BLOCK_MAX_RUNS), staged, use_lattice ? lat_tab + (b & 1u) * BLOCK_LATTICE_WORDS : nullptr, a_, b_); // b: block of the next pass BP5_PRIO_LO
    A.ent = entA2;
    prefetch_list();
    BP5_PRIO_HI BP::issue_loads(a, bp, A, abm, lane_ok, gp + 1 < gp_end); BP5_PRIO_LO
    const uint32_t entB2 = entry(gp + 2);
    BP::run(a, sh, B, A, T, acc, a_, b_, n_rounds, abm, ph, tprev, ds[0], Sroll);
    finish_pass();
    BP5_STAMP(6)
    BP5_PRIO_HI if constexpr (BP::PACK) { if (gp < gp_end) BP::decode_and_gather(a, A, run_tab + (b & 1u) * (2 * BLOCK_MAX_RUNS), staged, use_lattice ? lat_tab + (b & 1u) * BLOCK_LATTICE_WORDS : nullptr, a_, b_); } BP5_PRIO_LO
    B.ent = entB2;
  }
  if constexpr (DOTS) {
    // block reduction in a fixed order (wave shuffles, then the four waves through LDS): bitwise reproducible for a fixed grid
    __syncthreads(); // the tiles are free: every pass of this workgroup is done
    double *red = lds;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = ds[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if ((t & 63) == 0) red[k * 4 + (t >> 6)] = v;
    }
    __syncthreads();
    if (t < 7) {
      const int k = t < 4 ? t : (t == 4 ? 2 : t == 5 ? 1 : 3); // D == 1: r.Dv = r.v, v.Dv = v.v, r.Dr = r.r
      bp.dot_partials[t * PARTIAL_STRIDE + bp.dot_col0 + blockIdx.x] = (red[k * 4] + red[k * 4 + 1]) + (red[k * 4 + 2] + red[k * 4 + 3]);
    }
  }
  if constexpr (ABL & 4096) {
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) bp.stamps[(uint64_t)w * 16 + k] = ph[k];
      bp.stamps[(uint64_t)w * 16 + 8] = gp_end;
    }
  }
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

// hanging-node fix-up in the generic layout (one thread per (i,j,k), v in LDS); the whole block calls it
template <int n, bool TR>
__device__ void hang_resolve3(uint32_t m, const double *I, double *v, double *t, int i, int j, int k)
{
  if (!(m & BP5_HANG_ANY)) return; // block-uniform
  const int idx[3] = {i, j, k}, q = i + n * (j + n * k);
  for (int d = 0; d < 3; ++d) {
    if (!hang_dir_any(m, d)) continue; // block-uniform
    const double *M = I + ((m >> (6 + d)) & 1u) * n * n;
    double acc = v[q];
    if (hang_on_line(m, d, i, j, k, n - 1)) {
      acc = 0.0;
      for (int r = 0; r < n; ++r) {
        int e[3] = {i, j, k};
        e[d] = r;
        acc += (TR ? M[r * n + idx[d]] : M[idx[d] * n + r]) * v[e[0] + n * (e[1] + n * e[2])];
      }
    }
    t[q] = acc;
    __syncthreads();
    v[q] = t[q];
    __syncthreads();
  }
}

struct GeomOut {
  double *coef;          // permuted merged metric or NULL
  uint64_t plane_stride, cell_stride; // layout of the six planes (see ApplyArgs)
  double *inv_jac;       // 9 planes of n_cells*pad, or NULL
  double *JxW;           // n_cells*pad or NULL
  double *q_points;      // 3 planes of n_cells*pad or NULL
  uint32_t pad;
  uint64_t geo_plane;    // n_cells * pad
  // affine mode: scalar plane kappa*JxW (permuted layout), per-cell K K^T (6 planes of n_cells), and a
  // per-cell deviation measure max_q |K K^T(q) - K K^T(q0)| / |K K^T(q0)|
  double *scalar, *gcell, *deviation;
  uint32_t n_cells;
  const uint32_t *hang_mask; // NULL: conforming mesh
  const double *hang_I;
  // Helmholtz operator (step-64/step-64.cu:99-118,154-160): planes 0-5 = JxW K K^T (Laplace part, coefficient 1), plane 6 = a(x_q) JxW
  // (VaryingCoefficientFunctor's coef with JxW folded in); coef then holds SEVEN planes
  int helmholtz;
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
    if (o.hang_mask) // the coordinate field is an FE function: the constrained face takes its nodes from the coarse face's
      for (int e = 0; e < 3; ++e) hang_resolve3<n, false>(o.hang_mask[cell], o.hang_I, X + e * n3, t1, i, j, k);
    double J[3][3], K[3][3], xq[3];
    cell_jacobian<n>(tab, X, t1, t2, i, j, k, J, xq);
    const double det = invert3(J, K);
    const double *w = tab + 2 * n2;
    const double jxw = fabs(det) * w[i] * w[j] * w[k];
    if (o.coef) {
      const double kap = kappa_eval(kappa_mode, xq[0], xq[1], xq[2]);
      const double s = o.helmholtz ? jxw : jxw * kap;
      double *c = o.coef + cell * o.cell_stride + coef_off<n>(i, j + n * k); // pair layout (coef_off)
      if (o.helmholtz) c[6 * o.plane_stride] = jxw * kap;
      c[0 * o.plane_stride] = s * (K[0][0] * K[0][0] + K[0][1] * K[0][1] + K[0][2] * K[0][2]);
      c[1 * o.plane_stride] = s * (K[1][0] * K[1][0] + K[1][1] * K[1][1] + K[1][2] * K[1][2]);
      c[2 * o.plane_stride] = s * (K[2][0] * K[2][0] + K[2][1] * K[2][1] + K[2][2] * K[2][2]);
      c[3 * o.plane_stride] = s * (K[0][0] * K[1][0] + K[0][1] * K[1][1] + K[0][2] * K[1][2]);
      c[4 * o.plane_stride] = s * (K[0][0] * K[2][0] + K[0][1] * K[2][1] + K[0][2] * K[2][2]);
      c[5 * o.plane_stride] = s * (K[1][0] * K[2][0] + K[1][1] * K[2][1] + K[1][2] * K[2][2]);
    }
    if (o.scalar) {
      double Gq[6] = {K[0][0] * K[0][0] + K[0][1] * K[0][1] + K[0][2] * K[0][2], K[1][0] * K[1][0] + K[1][1] * K[1][1] + K[1][2] * K[1][2],
                      K[2][0] * K[2][0] + K[2][1] * K[2][1] + K[2][2] * K[2][2], K[0][0] * K[1][0] + K[0][1] * K[1][1] + K[0][2] * K[1][2],
                      K[0][0] * K[2][0] + K[0][1] * K[2][1] + K[0][2] * K[2][2], K[1][0] * K[2][0] + K[1][1] * K[2][1] + K[1][2] * K[2][2]};
      o.scalar[cell * n3 + coef_off<n>(i, j + n * k)] = jxw * kappa_eval(kappa_mode, xq[0], xq[1], xq[2]);
      // reference values of the cell: q-point 0
      for (int c6 = 0; c6 < 6; ++c6) {
        if (q == 0) { t1[c6] = Gq[c6]; o.gcell[(uint64_t)c6 * o.n_cells + cell] = Gq[c6]; }
      }
      __syncthreads();
      double dev = 0.0, nrm = 0.0;
      for (int c6 = 0; c6 < 6; ++c6) { dev = fmax(dev, fabs(Gq[c6] - t1[c6])); nrm = fmax(nrm, fabs(t1[c6])); }
      t2[q] = dev / nrm;
      __syncthreads();
      if (q == 0) {
        double mx = 0.0;
        for (int m = 0; m < n3; ++m) mx = fmax(mx, t2[m]);
        o.deviation[cell] = mx;
      }
      __syncthreads();
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
__global__ void metric_permute_kernel(const double *in, double *out, uint64_t total /*6*n_cells*n3*/, uint64_t n_cells, uint64_t plane_stride,
                                      uint64_t cell_stride)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  for (uint64_t o = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; o < total; o += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t cellplane = o / n3, plane = cellplane / n_cells, cell = cellplane - plane * n_cells; // reference order [c][cell][q]
    const int q = (int)(o - cellplane * n3); // reference index qi + n(qj + n qk)
    const int qi = q % n, rest = q / n;      // rest = qj + n qk
    out[o] = in[plane * plane_stride + cell * cell_stride + coef_off<n>(qi, rest)];
  }
}

// b_i = sum_q phi_i(x_q) JxW(q), Gauss tables (assemble_rhs, bp5/step-64.cu:372-418)
template <int n>
__global__ void __launch_bounds__(n *n *n) rhs_kernel(const uint32_t *l2g, const double *coords, const double *tab_gauss,
                                                     uint32_t n_cells, double *b, const uint32_t *hang_mask, const double *hang_I)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  __shared__ double X[3 * n3], t1[n3], t2[n3], v[n3];
  const int i = threadIdx.x, j = threadIdx.y, k = threadIdx.z;
  const int q = i + n * (j + n * k);
  for (uint64_t cell = blockIdx.x; cell < n_cells; cell += gridDim.x) {
    const uint32_t g = l2g[cell * n3 + q];
    for (int e = 0; e < 3; ++e) X[e * n3 + q] = coords[3 * (uint64_t)g + e];
    __syncthreads();
    const uint32_t hm = hang_mask ? hang_mask[cell] : 0u;
    for (int e = 0; e < 3; ++e) hang_resolve3<n, false>(hm, hang_I, X + e * n3, t1, i, j, k);
    double J[3][3], K[3][3], xq[3];
    cell_jacobian<n>(tab_gauss, X, t1, t2, i, j, k, J, xq);
    const double det = invert3(J, K);
    const double *w = tab_gauss + 2 * n2;
    v[q] = fabs(det) * w[i] * w[j] * w[k];
    __syncthreads();
    const double *N = tab_gauss;
    double y = Cell3<n>::template tensor3<true>(N, N, N, v, t1, t2, i, j, k);
    if (hm & BP5_HANG_ANY) { // adjoint of the hanging-node interpolation before the scatter
      v[q] = y;
      __syncthreads();
      hang_resolve3<n, true>(hm, hang_I, v, t1, i, j, k);
      y = v[q];
    }
    atomic_add_f64(b + g, y);
    __syncthreads();
  }
}

// diag(A) of the cell loop: diag_ijk += sum_abc S_de(a,b,c) X_de[a,i] Y_de[b,j] Z_de[c,k] with the entrywise products
// N*N, D*D, N*D as 1-D factors (six transposed tensor contractions per cell).  Setup-time kernel: one block per cell,
// one thread per local DoF.  coef: the handle's six planes (device layout, q index = a*n*n + b + n*c), or, in affine
// mode (gcell != NULL), the scalar plane times the cell's constant K K^T.
template <int n>
__global__ void __launch_bounds__(n *n *n) diagonal_kernel(const uint32_t *l2g, const double *coef, uint64_t plane_stride, uint64_t cell_stride, const double *gcell,
                                                          const double *tab, uint32_t n_cells, double *diag, const uint32_t *hang_mask, int n_planes = 6)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  __shared__ double S[n3], t1[n3], t2[n3], NN[n2], DD[n2], ND[n2];
  const int i = threadIdx.x, j = threadIdx.y, k = threadIdx.z;
  const int q = i + n * (j + n * k);
  if (q < n2) {
    const double a = tab[q], b = tab[n2 + q];
    NN[q] = a * a;
    DD[q] = b * b;
    ND[q] = a * b;
  }
  __syncthreads();
  for (uint64_t cell = blockIdx.x; cell < n_cells; cell += gridDim.x) {
    double acc = 0.0;
    for (int c = 0; c < n_planes; ++c) { // (n_planes == 7: the Helmholtz operator's mass plane a JxW with the factors N*N in every direction)
      // this thread's q-point (a,b,c) = (i,j,k) in the device (pair) layout
      const uint64_t at = cell * (gcell ? (uint64_t)n3 : cell_stride) + coef_off<n>(i, j + n * k);
      S[q] = gcell ? coef[at] * gcell[(uint64_t)c * n_cells + cell] : coef[(uint64_t)c * plane_stride + at];
      __syncthreads();
      const double *X = (c == 0) ? DD : (c == 3 || c == 4) ? ND : NN;
      const double *Y = (c == 1) ? DD : (c == 3 || c == 5) ? ND : NN;
      const double *Z = (c == 2) ? DD : (c == 4 || c == 5) ? ND : NN;
      const double y = Cell3<n>::template tensor3<true>(X, Y, Z, S, t1, t2, i, j, k);
      acc += (c < 3 || c == 6) ? y : 2.0 * y;
    }
    // entries on constrained faces / edges stand for coarse DoFs: diagonal_hanging_kernel computes theirs
    const uint32_t hm = hang_mask ? hang_mask[cell] : 0u;
    const bool coarse_entry = (hm & BP5_HANG_ANY) && (hang_on_line(hm, 0, i, j, k, n - 1) || hang_on_line(hm, 1, i, j, k, n - 1) || hang_on_line(hm, 2, i, j, k, n - 1));
    if (!coarse_entry) atomic_add_f64(diag + l2g[cell * n3 + q], acc);
  }
}

// Diagonal entries of the coarse DoFs a cell with hanging nodes refers to: (R^T A_e R)[s][s] with R the cell's hanging-node
// interpolation -- R e_s is not a tensor product in general (a DoF on the edge shared by two constrained faces spreads into
// both), so each such entry takes one application of the cell operator to R e_s.  Setup-time kernel, flagged cells only.
template <int n>
__global__ void __launch_bounds__(n *n *n) diagonal_hanging_kernel(const uint32_t *l2g, const double *coef, uint64_t plane_stride, uint64_t cell_stride, const double *gcell,
                                                                  const double *tab, uint32_t n_cells, double *diag, const uint32_t *hang_mask,
                                                                  const double *hang_I)
{
  constexpr int n2 = n * n, n3 = n2 * n;
  __shared__ double v[n3], w[n3], t1[n3], t2[n3];
  const int i = threadIdx.x, j = threadIdx.y, k = threadIdx.z;
  const int q = i + n * (j + n * k);
  const double *N = tab, *D = tab + n2;
  for (uint64_t cell = blockIdx.x; cell < n_cells; cell += gridDim.x) {
    const uint32_t hm = hang_mask[cell];
    if (!(hm & BP5_HANG_ANY)) continue; // block-uniform
    double S[6];
    const uint64_t at = cell * cell_stride + coef_off<n>(i, j + n * k);
#pragma unroll
    for (int c = 0; c < 6; ++c) S[c] = gcell ? coef[at] * gcell[(uint64_t)c * n_cells + cell] : coef[(uint64_t)c * plane_stride + at]; // (affine mode: scalar plane x the cell's K K^T)
    for (int s = 0; s < n3; ++s) {
      const int si = s % n, sj = (s / n) % n, sk = s / n2;
      if (!(hang_on_line(hm, 0, si, sj, sk, n - 1) || hang_on_line(hm, 1, si, sj, sk, n - 1) || hang_on_line(hm, 2, si, sj, sk, n - 1))) continue; // block-uniform
      v[q] = q == s ? 1.0 : 0.0;
      __syncthreads();
      hang_resolve3<n, false>(hm, hang_I, v, t1, i, j, k); // v = R e_s
      const double g0 = Cell3<n>::template tensor3<false>(D, N, N, v, t1, t2, i, j, k);
      const double g1 = Cell3<n>::template tensor3<false>(N, D, N, v, t1, t2, i, j, k);
      const double g2 = Cell3<n>::template tensor3<false>(N, N, D, v, t1, t2, i, j, k);
      const double h0 = S[0] * g0 + S[3] * g1 + S[4] * g2, h1 = S[3] * g0 + S[1] * g1 + S[5] * g2, h2 = S[4] * g0 + S[5] * g1 + S[2] * g2;
      w[q] = h0;
      __syncthreads();
      double y = Cell3<n>::template tensor3<true>(D, N, N, w, t1, t2, i, j, k);
      w[q] = h1;
      __syncthreads();
      y += Cell3<n>::template tensor3<true>(N, D, N, w, t1, t2, i, j, k);
      w[q] = h2;
      __syncthreads();
      y += Cell3<n>::template tensor3<true>(N, N, D, w, t1, t2, i, j, k);
      v[q] = y;
      __syncthreads();
      hang_resolve3<n, true>(hm, hang_I, v, t1, i, j, k); // R^T A_e R e_s
      if (q == s) atomic_add_f64(diag + l2g[cell * n3 + s], v[s]);
      __syncthreads();
    }
  }
}
static __global__ void reciprocal_kernel(double *v, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = 1.0 / v[i];
}

// sum_cells sum_q (u_h(x_q))^2 JxW  -> *out (atomic)
template <int n>
__global__ void __launch_bounds__(n *n *n) l2norm_kernel(const uint32_t *l2g, const double *coords, const double *tab_gauss,
                                                        uint32_t n_cells, const double *u, double *out, const uint32_t *hang_mask,
                                                        const double *hang_I)
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
    if (hang_mask) {
      const uint32_t hm = hang_mask[cell];
      for (int e = 0; e < 3; ++e) hang_resolve3<n, false>(hm, hang_I, X + e * n3, t1, i, j, k);
      hang_resolve3<n, false>(hm, hang_I, v, t1, i, j, k);
    }
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

// shared DoFs of the OWNER scatter modes: dst[g] (+)= sum of the teams' partials, fixed order
template <bool ADD>
__global__ void __launch_bounds__(256) combine_kernel(const uint32_t *sh_dof, const uint32_t *sh_off, const uint32_t *sh_slot,
                                                     const double *partial, double *dst, uint32_t n_shared)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_shared) return;
  const uint32_t b = sh_off[i], e = sh_off[i + 1];
  double s = partial[sh_slot[b]];
  for (uint32_t j = b + 1; j < e; ++j) s += partial[sh_slot[j]];
  const uint32_t g = sh_dof[i];
  if (ADD) dst[g] += s; else dst[g] = s;
}

// The same sums from a run-length form of the CSR: consecutive shared DoFs whose partial slots are consecutive in every
// contributing group (a brick face is one run of 225 DoFs in both bricks' lists) share one record
//   run r: shared-DoF ordinals [start[r], start[r+1]), first DoF dof0[r], first slot per contributing group slots[soff[r] + q]
// Tile T (256 ordinals) touches the runs tile_run[T] .. tile_run[T+1]; their records are staged in LDS and every thread
// finds its run there.  Same summation order as combine_kernel (groups in ascending order): bitwise identical results,
// about half the bytes (no per-DoF index arrays).
struct CombineRuns {
  const uint32_t *start, *dof0, *soff, *slots, *tile_run; // dof0 bit 31: the run's DoFs are Dirichlet DoFs
  uint32_t n_shared;
  // window of this launch: tiles [tile0, tile0 + n_tiles) and, inside them, the DoFs dof_lo <= g < dof_hi only.  The boundary-first
  // schedule of the halo exchange completes the GHOST rows first (they are sent to their owners while the interior bricks run) and
  // the owned rows after the last brick; every row is written by exactly one of the two launches, in the same summation order
  uint32_t tile0, dof_lo, dof_hi;
  // DOTS builds (fused CG dot products, see apply_block_kernel): the launch is a fixed grid walking the tiles
  const double *cg_p, *cg_r;
  double *dot_partials;    // [7][PARTIAL_STRIDE]; this launch writes the columns [dot_col0, dot_col0 + gridDim.x)
  uint32_t dot_col0, n_owned, n_tiles;
  const int *cg_state;
  // DOTS builds, exchange under the owned-row combine in ONE launch: the first ghost_blocks workgroups complete the GHOST rows (tile
  // ghost_tile0 + blockIdx.x, DoFs >= n_owned, the plain sum: ghost rows never enter the dot products and keep no Dirichlet copy), release
  // their stores and count themselves in at *signal -- the communication stream waits for the count (hipStreamWaitValue64) and sends the
  // rows while the remaining workgroups walk the owned rows exactly as a launch of their own would (same tiles, same columns: same bits)
  uint32_t ghost_blocks, ghost_tile0;
  unsigned long long *signal;
};
constexpr int COMBINE_TILE = 512; // shared-DoF ordinals per tile: 256 threads x two ordinals
// PAIR: a thread takes two CONSECUTIVE ordinals (16-byte slab loads when both lie in one run): 1 % of a CG iteration at 1e8 DoFs;
// else ordinals tid and tid + 256 of the tile, one after the other (fewer instructions on the dependent-load chain: 1-2 % of an
// iteration at 1e7 DoFs, where the pass is latency-bound).  Same sums, bit for bit; the launcher picks by n_shared.
template <bool ADD, bool DOTS = false, bool PAIR = true>
__global__ void __launch_bounds__(256) combine_runs_kernel(CombineRuns cr, const double *partial, double *dst)
{
  __shared__ uint32_t s_start[COMBINE_TILE + 2], s_dof0[COMBINE_TILE + 1], s_soff[COMBINE_TILE + 2];
  __shared__ double s_red[4][4];
  double ds[4] = {0.0, 0.0, 0.0, 0.0};
  const bool ghost_job = DOTS && blockIdx.x < cr.ghost_blocks;                       // (workgroup-uniform)
  const uint32_t bid = blockIdx.x - (DOTS ? cr.ghost_blocks : 0u), nblk = gridDim.x - (DOTS ? cr.ghost_blocks : 0u);
  auto ghost_rows_done = [&](bool stores_pending) { // as apply_block_kernel's signal_part_done
    if (stores_pending) __syncthreads();
    if (threadIdx.x == 0) {
      if (stores_pending) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(cr.signal, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if constexpr (DOTS) {
    if (cr.cg_state[0]) { // the solve has stopped: a no-op launch -- whoever waits for the ghost rows must still be released
      if (ghost_job) ghost_rows_done(false);
      return;
    }
  }
  const uint32_t dof_lo = ghost_job ? cr.n_owned : cr.dof_lo, dof_hi = ghost_job ? 0xffffffffu : cr.dof_hi;
  // one value: sum of its partials in ascending group order; Dirichlet rows of the fused build store p instead; dot products
  auto finish = [&](uint32_t g, bool dirichlet, double s) {
    if (g < dof_lo || g >= dof_hi) return; // another launch's (or job's) row
    if (ghost_job) { dst[g] = s; return; }  // (what combine_runs_kernel<false, false> stores: a ghost row)
    if constexpr (DOTS) {
      double vi = s;
      if (dirichlet) { // v = p (copy_constrained_values, bp5/step-64.cu:275); p.v correction as in the block kernel
        vi = cr.cg_p[g];
        if (g < cr.n_owned) ds[0] += vi * (vi - s);
      }
      dst[g] = vi;
      if (g < cr.n_owned) {
        const double ri = cr.cg_r ? cr.cg_r[g] : 0.0; // (NULL: only p.v is wanted)
        ds[1] += vi * vi; ds[2] += ri * vi; ds[3] += ri * ri;
      }
    } else if (ADD) dst[g] += s;
    else dst[g] = s;
  };
  // the partials of one ordinal (offset j in its run), summed in ascending group order: the first four slot bases are loaded together, then
  // the four partials together -- two memory latencies instead of two per contributing group (faces have 2 groups, edges 4, corners 8)
  auto sum1 = [&](uint32_t b, uint32_t e, uint32_t j) -> double {
    if (b >= e) return 0.0; // (a DoF no cell touches has no slot: its sum is zero)
    const uint32_t nq = e - b;
    const uint32_t s0 = cr.slots[b], s1 = nq > 1 ? cr.slots[b + 1] : 0u, s2 = nq > 2 ? cr.slots[b + 2] : 0u, s3 = nq > 3 ? cr.slots[b + 3] : 0u;
    const double p0 = partial[s0 + j], p1 = nq > 1 ? partial[s1 + j] : 0.0, p2 = nq > 2 ? partial[s2 + j] : 0.0, p3 = nq > 3 ? partial[s3 + j] : 0.0;
    double sum = p0;
    if (nq > 1) sum += p1;
    if (nq > 2) sum += p2;
    if (nq > 3) sum += p3;
    for (uint32_t q = b + 4; q < e; ++q) sum += partial[cr.slots[q] + j];
    return sum;
  };
  auto sum2 = [&](uint32_t b, uint32_t e, uint32_t j) -> bp5_d2u { // two consecutive ordinals of one run: 16-byte loads
    if (b >= e) return bp5_d2u{0.0, 0.0};
    const uint32_t nq = e - b;
    const uint32_t s0 = cr.slots[b], s1 = nq > 1 ? cr.slots[b + 1] : 0u, s2 = nq > 2 ? cr.slots[b + 2] : 0u, s3 = nq > 3 ? cr.slots[b + 3] : 0u;
    const bp5_d2u z{0.0, 0.0};
    const bp5_d2u p0 = *reinterpret_cast<const bp5_d2u *>(partial + s0 + j), p1 = nq > 1 ? *reinterpret_cast<const bp5_d2u *>(partial + s1 + j) : z,
                  p2 = nq > 2 ? *reinterpret_cast<const bp5_d2u *>(partial + s2 + j) : z, p3 = nq > 3 ? *reinterpret_cast<const bp5_d2u *>(partial + s3 + j) : z;
    bp5_d2u sum = p0;
    if (nq > 1) { sum.x += p1.x; sum.y += p1.y; }
    if (nq > 2) { sum.x += p2.x; sum.y += p2.y; }
    if (nq > 3) { sum.x += p3.x; sum.y += p3.y; }
    for (uint32_t q = b + 4; q < e; ++q) {
      const bp5_d2u t2 = *reinterpret_cast<const bp5_d2u *>(partial + cr.slots[q] + j);
      sum.x += t2.x; sum.y += t2.y;
    }
    return sum;
  };
  const uint32_t t_begin = ghost_job ? cr.ghost_tile0 + blockIdx.x : cr.tile0 + bid;
  const uint32_t t_end = ghost_job ? t_begin + 1 : (DOTS ? cr.tile0 + cr.n_tiles : cr.tile0 + blockIdx.x + 1);
  if constexpr (DOTS && PAIR) {
    // Long passes (round 4): the pass is a chain of dependent memory latencies per tile (run records -> slot bases -> partials), at full occupancy
    // -- so a workgroup takes its tiles TWO at a time: the records of both are staged behind one barrier, the slot bases of both are loaded
    // together, then the partials (and r) of both.  Same tiles per workgroup, same order of the sums and of the dot-product terms: same bits.
    if (!ghost_job) {
      __shared__ uint32_t s2_start[COMBINE_TILE + 2], s2_dof0[COMBINE_TILE + 1], s2_soff[COMBINE_TILE + 2];
      const uint32_t step = nblk ? nblk : 1u;
      for (uint32_t tile = t_begin; tile < t_end; tile += 2u * step) {
        const uint32_t tl[2] = {tile, tile + step};
        const bool have[2] = {true, tile + step < t_end};
        uint32_t cntv[2] = {0u, 0u};
        __syncthreads(); // the staging arrays of the previous batch are no longer read
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          if (!have[x]) continue;
          uint32_t *sst = x ? s2_start : s_start, *sso = x ? s2_soff : s_soff, *sd0 = x ? s2_dof0 : s_dof0;
          const uint32_t r_lo = cr.tile_run[tl[x]], r_hi = cr.tile_run[tl[x] + 1];
          cntv[x] = r_hi - r_lo + 1;
          for (uint32_t j = threadIdx.x; j <= cntv[x]; j += 256) {
            sst[j] = cr.start[r_lo + j];
            sso[j] = cr.soff[r_lo + j];
            if (j < cntv[x]) sd0[j] = cr.dof0[r_lo + j];
          }
        }
        __syncthreads();
        // per tile: this thread's two consecutive ordinals
        uint32_t lov[2], jv[2], bv[2], ev[2], gv[2], iv[2];
        bool dirv[2], pairv[2], livev[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const uint32_t *sst = x ? s2_start : s_start, *sso = x ? s2_soff : s_soff, *sd0 = x ? s2_dof0 : s_dof0;
          iv[x] = tl[x] * (uint32_t)COMBINE_TILE + 2u * threadIdx.x;
          livev[x] = have[x] && iv[x] < cr.n_shared;
          lov[x] = jv[x] = bv[x] = ev[x] = gv[x] = 0u; dirv[x] = pairv[x] = false;
          if (!livev[x]) continue;
          uint32_t lo = 0, hi = cntv[x]; // invariant: start[lo] <= i < start[hi]
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sst[mid] <= iv[x]) lo = mid; else hi = mid;
          }
          lov[x] = lo; jv[x] = iv[x] - sst[lo]; bv[x] = sso[lo]; ev[x] = sso[lo + 1];
          gv[x] = (sd0[lo] & 0x7fffffffu) + jv[x];
          dirv[x] = (sd0[lo] & 0x80000000u) != 0;
          pairv[x] = iv[x] + 1 < cr.n_shared && iv[x] + 1 < sst[lo + 1];
        }
        // slot bases of both tiles, then the partials and r of both (pairs inside one run; the rare pair that straddles two runs below)
        uint32_t sl[2][4];
        bp5_d2u pq[2][4], rr[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const uint32_t nq = ev[x] - bv[x];
#pragma unroll
          for (int q = 0; q < 4; ++q) sl[x][q] = (livev[x] && pairv[x] && (uint32_t)q < nq) ? cr.slots[bv[x] + q] : 0u;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const uint32_t nq = ev[x] - bv[x];
          const bool on = livev[x] && pairv[x];
#pragma unroll
          for (int q = 0; q < 4; ++q) pq[x][q] = (on && (uint32_t)q < nq) ? *reinterpret_cast<const bp5_d2u *>(partial + sl[x][q] + jv[x]) : bp5_d2u{0.0, 0.0};
          rr[x] = bp5_d2u{0.0, 0.0};
          if (on && cr.cg_r && gv[x] >= dof_lo && gv[x] < dof_hi) { // (the rows of this launch; r is read at owned rows only)
            if (gv[x] + 1 < cr.n_owned) rr[x] = *reinterpret_cast<const bp5_d2u *>(cr.cg_r + gv[x]);
            else if (gv[x] < cr.n_owned) rr[x].x = cr.cg_r[gv[x]];
          }
        }
        auto finish_r = [&](uint32_t g, bool dirichlet, double sv, double ri) { // finish() with r[g] loaded ahead (same arithmetic, same order)
          if (g < dof_lo || g >= dof_hi) return;
          double vi = sv;
          if (dirichlet) {
            vi = cr.cg_p[g];
            if (g < cr.n_owned) ds[0] += vi * (vi - sv);
          }
          dst[g] = vi;
          if (g < cr.n_owned) { ds[1] += vi * vi; ds[2] += ri * vi; ds[3] += ri * ri; }
        };
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          if (!livev[x]) continue;
          const uint32_t *sso = x ? s2_soff : s_soff, *sd0 = x ? s2_dof0 : s_dof0;
          if (pairv[x]) {
            const uint32_t nq = ev[x] - bv[x];
            bp5_d2u sum{0.0, 0.0};
            if (nq > 0) {
              sum = pq[x][0];
              if (nq > 1) { sum.x += pq[x][1].x; sum.y += pq[x][1].y; }
              if (nq > 2) { sum.x += pq[x][2].x; sum.y += pq[x][2].y; }
              if (nq > 3) { sum.x += pq[x][3].x; sum.y += pq[x][3].y; }
              for (uint32_t q = bv[x] + 4; q < ev[x]; ++q) {
                const bp5_d2u t2 = *reinterpret_cast<const bp5_d2u *>(partial + cr.slots[q] + jv[x]);
                sum.x += t2.x; sum.y += t2.y;
              }
            }
            finish_r(gv[x], dirv[x], sum.x, rr[x].x);
            finish_r(gv[x] + 1, dirv[x], sum.y, rr[x].y);
          } else {
            finish(gv[x], dirv[x], sum1(bv[x], ev[x], jv[x]));
            if (iv[x] + 1 < cr.n_shared) { // first ordinal of the next run
              const uint32_t l2 = lov[x] + 1, b2 = sso[l2], e2 = sso[l2 + 1];
              finish(sd0[l2] & 0x7fffffffu, (sd0[l2] & 0x80000000u) != 0, sum1(b2, e2, 0u));
            }
          }
        }
      }
    }
  }
  for (uint32_t tile = t_begin; tile < ((DOTS && PAIR && !ghost_job) ? t_begin : t_end); tile += (nblk ? nblk : 1u)) {
    if constexpr (DOTS) __syncthreads(); // the staging arrays of the previous tile are no longer read
    const uint32_t r_lo = cr.tile_run[tile], r_hi = cr.tile_run[tile + 1]; // inclusive range, r_hi - r_lo <= COMBINE_TILE
    const uint32_t cnt = r_hi - r_lo + 1;
    for (uint32_t j = threadIdx.x; j <= cnt; j += 256) {
      s_start[j] = cr.start[r_lo + j];
      s_soff[j] = cr.soff[r_lo + j];
      if (j < cnt) s_dof0[j] = cr.dof0[r_lo + j];
    }
    __syncthreads();
    if constexpr (!PAIR) {
#pragma unroll
      for (uint32_t half = 0; half < 2; ++half) {
        const uint32_t i = tile * (uint32_t)COMBINE_TILE + half * 256u + threadIdx.x;
        if (i >= cr.n_shared) break;
        uint32_t lo = 0, hi = cnt; // invariant: s_start[lo] <= i < s_start[hi]
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_start[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t j = i - s_start[lo], b = s_soff[lo], e = s_soff[lo + 1];
        finish((s_dof0[lo] & 0x7fffffffu) + j, (s_dof0[lo] & 0x80000000u) != 0, sum1(b, e, j));
      }
      continue;
    }
    const uint32_t i = tile * (uint32_t)COMBINE_TILE + 2u * threadIdx.x;
    if (i >= cr.n_shared) continue;
    uint32_t lo = 0, hi = cnt; // invariant: s_start[lo] <= i < s_start[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_start[mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t j = i - s_start[lo], b = s_soff[lo], e = s_soff[lo + 1];
    const uint32_t g = (s_dof0[lo] & 0x7fffffffu) + j;
    const bool dir = (s_dof0[lo] & 0x80000000u) != 0;
    if (i + 1 < cr.n_shared && i + 1 < s_start[lo + 1]) {
      // both ordinals in one run: their partials are neighbours in every contributing group's slab range -> 16-byte loads
      const bp5_d2u s2 = sum2(b, e, j);
      finish(g, dir, s2.x);
      finish(g + 1, dir, s2.y);
    } else {
      finish(g, dir, sum1(b, e, j));
      if (i + 1 < cr.n_shared) { // first ordinal of the next run
        const uint32_t l2 = lo + 1, b2 = s_soff[l2], e2 = s_soff[l2 + 1];
        finish(s_dof0[l2] & 0x7fffffffu, (s_dof0[l2] & 0x80000000u) != 0, sum1(b2, e2, 0u));
      }
    }
  }
  if constexpr (DOTS) {
    if (ghost_job) { ghost_rows_done(true); return; } // (no column: ghost rows never enter the dot products)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double v = ds[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) s_red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
      const int t = threadIdx.x, k = t < 4 ? t : (t == 4 ? 2 : t == 5 ? 1 : 3); // D == 1
      cr.dot_partials[t * PARTIAL_STRIDE + cr.dot_col0 + bid] = (s_red[k][0] + s_red[k][1]) + (s_red[k][2] + s_red[k][3]);
    }
  }
}

static __global__ void __launch_bounds__(256) zero_indexed_kernel(const uint32_t *idx, uint32_t n, double *dst)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[idx[i]] = 0.0;
}

// ------------------------------------------------------------------------------------ small vector kernels
static __global__ void copy_constrained_kernel(const uint32_t *cdofs, uint32_t n, const double *src, double *dst)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t c = cdofs[i]; dst[c] = src[c]; }
}
static __global__ void set_constrained_kernel(const uint32_t *cdofs, uint32_t n, double val, double *dst)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[cdofs[i]] = val;
}
static __global__ void pack_kernel(const uint32_t *idx, uint32_t n, const double *v, double *buf)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = v[idx[i]];
}
static __global__ void unpack_add_kernel(const uint32_t *idx, uint32_t n, const double *buf, double *v)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[idx[i]] += buf[i]; // indices of one neighbour are distinct
}

// the same for a solve with fused dot products (SolverCGFullMerge on the block kernel): the operator's write-out has already
// counted the LOCAL sums of these owner DoFs in v.v and r.v; adding the neighbour's contribution c changes them by
// (v + c)^2 - v^2 and r c -- accumulated here into columns [col0, col0 + gridDim.x) of the partial-sum rows (rows 0 and 3,
// p.v and r.r, do not change: p.v is a sum over cells, r.r does not involve v)
// A Dirichlet DoF among them keeps the value the write-out gave it (v = p): whatever the neighbour's cells contributed to that
// row is discarded, as copy_constrained_values would do afterwards (bp5/step-64.cu:275).
static __global__ void __launch_bounds__(256) unpack_add_dots_kernel(const uint32_t *idx, const uint8_t *dirichlet, uint32_t n, const double *buf, double *v,
                                                                     const double *r, double *partials, uint32_t col0, const int *state,
                                                                     double *ghosts_a, double *ghosts_b, uint32_t n_ghost)
{
  if (state[0]) return;
  // the ghost ranges of v (just sent to the owners) and of src (read by this application's cells) are zeroed here instead of by
  // two more launches
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_ghost; i += gridDim.x * 256u) { ghosts_a[i] = 0.0; ghosts_b[i] = 0.0; }
  __shared__ double red[2][4];
  double dvv = 0.0, drv = 0.0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    if (dirichlet[i]) continue;
    const uint32_t g = idx[i];
    const double c = buf[i], vo = v[g], vn = vo + c;
    v[g] = vn;
    dvv += vn * vn - vo * vo;
    if (r) drv += r[g] * c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { dvv += __shfl_down(dvv, off, 64); drv += __shfl_down(drv, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dvv; red[1][threadIdx.x >> 6] = drv; }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int t = threadIdx.x;
    const double svv = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), srv = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    partials[t * PARTIAL_STRIDE + col0 + blockIdx.x] = (t == 1 || t == 5) ? svv : (t == 2 || t == 4) ? srv : 0.0; // D == 1: rows 5, 4 mirror 1, 2
  }
}

constexpr int VB = 256;      // threads per block of streaming kernels
constexpr int MAXBLK = 2048; // grid cap of the streaming kernels (<= PARTIAL_STRIDE)

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
__device__ __forceinline__ void block_reduce_store(double (&acc)[K], double *partials /*[K][PARTIAL_STRIDE]*/)
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
    partials[threadIdx.x * PARTIAL_STRIDE + blockIdx.x] = s;
  }
}

// sums partials[K][nblk] in a fixed tree order into out[K]; one workgroup per value when launched with K blocks
template <int K>
__global__ void __launch_bounds__(VB) finalize_kernel(const double *partials, int nblk, double *out, const int *state)
{
  if (state && state[0]) return;
  __shared__ double red[VB];
  for (int k = blockIdx.x; k < K; k += gridDim.x) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += VB) s += partials[k * PARTIAL_STRIDE + i];
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

static __global__ void __launch_bounds__(VB) dot_kernel(const double *x, const double *y, size_t n, double *partials)
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

// d.h of the standard solver with copy_constrained_values applied on the fly (h = d on the Dirichlet DoFs of the bitmap, stored)
static __global__ void __launch_bounds__(VB) cg_dh_kernel(const double *d, double *h, size_t n, double *partials, const uint32_t *cbits)
{
  double acc[1] = {0.0};
  const size_t stride = (size_t)gridDim.x * VB * 2;
  for (size_t i = ((size_t)blockIdx.x * VB + threadIdx.x) * 2; i < n; i += stride) {
    const uint32_t cb = (cbits[i >> 5] >> (i & 31)) & 3u;
    if (i + 1 < n) {
      const double2 a = *reinterpret_cast<const double2 *>(d + i);
      double2 b = *reinterpret_cast<const double2 *>(h + i);
      if (cb) {
        if (cb & 1u) b.x = a.x;
        if (cb & 2u) b.y = a.y;
        *reinterpret_cast<double2 *>(h + i) = b;
      }
      acc[0] += a.x * b.x + a.y * b.y;
    } else {
      double b = h[i];
      if (cb & 1u) { b = d[i]; h[i] = b; }
      acc[0] += d[i] * b;
    }
  }
  block_reduce_store<1>(acc, partials);
}

// number of nonzero (or NaN) entries, as a double so that it travels through the same reduction / all-reduce
static __global__ void __launch_bounds__(VB) count_nonzero_kernel(const double *x, size_t n, double *partials)
{
  double acc[1] = {0.0};
  const size_t stride = (size_t)gridDim.x * VB;
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += stride) acc[0] += (x[i] == 0.0) ? 0.0 : 1.0;
  block_reduce_store<1>(acc, partials);
}

// ------------------------------------------------------------------------------------ CG kernels
// Device-resident solver state.  sc[] doubles, st[] ints.
enum { SC_GH = 0, SC_DH, SC_GG, SC_GDG, SC_ALPHA, SC_BETA, SC_ALPHA_OLD, SC_BETA_OLD, SC_RES, SC_RES0, SC_TOL, SC_R0 /* 7 merged dots R0..R6 */,
       SC_COUNT = SC_R0 + 7 };
enum { ST_DONE = 0, ST_ITER, ST_PENDING, ST_MAXIT, ST_BREAKDOWN, ST_COUNT };

// ---- plain CG (deal.II SolverCG, Appendix A.5)
// init: g = -b ; d = b (= -D g with D = 1) ; x = 0 ; partial sums of g.g and g.Dg
static __global__ void __launch_bounds__(VB) cg_init_kernel(const double *b, const double *diag, double *x, double *g, double *d,
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
static __global__ void cg_init_control_kernel(double *sc, int *st)
{
  sc[SC_RES0] = sc[SC_RES] = sqrt(sc[SC_GG]);
  sc[SC_GH] = sc[SC_GDG];
  st[ST_ITER] = 0; st[ST_PENDING] = 0; st[ST_BREAKDOWN] = 0;
  st[ST_DONE] = (sc[SC_RES] <= sc[SC_TOL]) ? 1 : 0;
}
// x += alpha d ; g += alpha h ; partial sums g.g, g.Dg   with alpha = gh / dh
// zero_h: h is consumed here for the last time before the next operator application; an operator that scatters with atomics needs
// it zeroed, so this kernel stores the zeros (see cgm_update_kernel<.., ZV>)
static __global__ void __launch_bounds__(VB) cg_update_kernel(double *x, double *g, const double *d, double *h, const double *diag,
                                                      size_t n, const double *sc, const int *st, double *partials, bool zero_h)
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
      if (zero_h) *reinterpret_cast<double2 *>(h + i) = double2{0.0, 0.0};
      const double z0 = diag ? diag[i] * gv.x : gv.x, z1 = diag ? diag[i + 1] * gv.y : gv.y;
      acc[0] += gv.x * gv.x + gv.y * gv.y;
      acc[1] += gv.x * z0 + gv.y * z1;
    } else {
      const double xi = x[i] + alpha * d[i], gi = g[i] + alpha * h[i];
      x[i] = xi; g[i] = gi;
      if (zero_h) h[i] = 0.0;
      acc[0] += gi * gi; acc[1] += gi * (diag ? diag[i] * gi : gi);
    }
  }
  block_reduce_store<2>(acc, partials);
}
// res = sqrt(gg); ++it; stop test; beta = gDg / gh ; gh = gDg
static __global__ void cg_control_kernel(double *sc, int *st)
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
static __global__ void __launch_bounds__(VB) cg_direction_kernel(double *d, const double *g, const double *diag, size_t n, const double *sc,
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
// MODE 0: update_a0 (solver.h:48-72)    p = -D r
// MODE 1: update_a<false> (:74-104)     r += alpha v ; p = beta p - D r
// (the reference also zeroes v here for its atomic scatter; our operator overwrites v instead)
// MODE 2: update_a1 (:106-140)          x += (alpha + ao/bo) p + (ao/bo) D r_old ; then as MODE 1
// If the solve finished in the previous iteration with an x update pending, the kernel performs
// the epilogue (solver.h:510-526) instead: odd it: x += alpha p ; even it: update_c (:315-336).
template <int MODE>
__device__ __forceinline__ void cgm_update_one(double &p, double &r, const double v, double &x, const double di, const bool done,
                                               const bool epi_odd, const double alpha, const double beta, const double aob)
{
  // explicit fma / mul: cgm_update_kernel and cgm_pack_updated_kernel (the early ghost gather) inline this into different contexts and
  // must round identically -- a ghost copy of p is then bitwise its owner's p, whatever the compiler would have contracted
  if (done) {
    if (epi_odd) x = fma(alpha, p, x);
    else x = fma(aob, di * r, fma(alpha + aob, p, x));
    return;
  }
  if (MODE == 0) {
    p = -(di * r);
  } else {
    if (MODE == 2) x = fma(aob, di * r, fma(alpha + aob, p, x));
    const double rn = fma(alpha, v, r);
    r = rn;
    p = fma(beta, p, -(di * rn));
  }
}
// ZV: v is consumed here for the last time before the next operator application overwrites it -- an operator kernel that accumulates
// with atomics needs it zeroed, and this kernel holds v's values in registers anyway: it stores the zeros itself (one zero-fill
// launch -- two fill kernels in the runtime -- less per iteration)
// NTX: v (consumed here for the last time) and x (touched by nothing else) go past the caches with non-temporal accesses, so that on a mesh whose
// vectors fit the 256 MB memory-side cache p and r are still there when the operator kernel gathers them (bp5_mf_set_streaming)
template <int MODE, int U = 2, bool ZV = false, bool NTX = false>
__global__ void __launch_bounds__(VB) cgm_update_kernel(double *p, double *r, typename std::conditional<ZV, double, const double>::type *v, double *x,
                                                       const double *diag, size_t n, const double *sc, const int *st)
{
  const bool done = st[ST_DONE];
  if (done && !st[ST_PENDING]) return;
  const double alpha = sc[SC_ALPHA], beta = sc[SC_BETA];
  const double aob = (MODE == 2 || done) ? sc[SC_ALPHA_OLD] / sc[SC_BETA_OLD] : 0.0;
  const bool epi_odd = done && (st[ST_ITER] & 1);
  const bool touch_x = done || MODE == 2, touch_rp = !done;
  // U chunks of VB pairs per loop trip: all loads of a trip are issued before its first store (3-4 streams x U 16-byte loads in flight)
  const size_t stride = (size_t)gridDim.x * VB * 2 * U;
  for (size_t base = ((size_t)blockIdx.x * VB * U + threadIdx.x) * 2; base < n; base += stride) {
    double2 pv[U], rv[U], vv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * VB * 2;
      pv[u] = rv[u] = vv[u] = xv[u] = double2{0, 0};
      if (i + 1 < n) {
        pv[u] = *reinterpret_cast<double2 *>(p + i);
        rv[u] = *reinterpret_cast<double2 *>(r + i);
        if (MODE != 0 && !done) {
          if constexpr (NTX) { const bp5_d2u t = __builtin_nontemporal_load(reinterpret_cast<const bp5_d2u *>(v + i)); vv[u] = double2{t.x, t.y}; }
          else vv[u] = *reinterpret_cast<const double2 *>(v + i);
        }
        if (touch_x) {
          if constexpr (NTX) { const bp5_d2u t = __builtin_nontemporal_load(reinterpret_cast<const bp5_d2u *>(x + i)); xv[u] = double2{t.x, t.y}; }
          else xv[u] = *reinterpret_cast<double2 *>(x + i);
        }
      } else if (i < n) {
        pv[u].x = p[i]; rv[u].x = r[i];
        if (MODE != 0 && !done) vv[u].x = v[i];
        if (touch_x) xv[u].x = x[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * VB * 2;
      if (i >= n) continue;
      const bool pair = i + 1 < n;
      const double d0 = diag ? diag[i] : 1.0, d1 = (diag && pair) ? diag[i + 1] : 1.0;
      cgm_update_one<MODE>(pv[u].x, rv[u].x, vv[u].x, xv[u].x, d0, done, epi_odd, alpha, beta, aob);
      if (pair) cgm_update_one<MODE>(pv[u].y, rv[u].y, vv[u].y, xv[u].y, d1, done, epi_odd, alpha, beta, aob);
      if (pair) {
        if (touch_rp) {
          *reinterpret_cast<double2 *>(p + i) = pv[u];
          if (MODE != 0) *reinterpret_cast<double2 *>(r + i) = rv[u];
        }
        if (touch_x) {
          if constexpr (NTX) __builtin_nontemporal_store(bp5_d2u{xv[u].x, xv[u].y}, reinterpret_cast<bp5_d2u *>(x + i));
          else *reinterpret_cast<double2 *>(x + i) = xv[u];
        }
        if constexpr (ZV) { if (MODE != 0 && !done) *reinterpret_cast<double2 *>(v + i) = double2{0.0, 0.0}; }
      } else {
        if (touch_rp) { p[i] = pv[u].x; if (MODE != 0) r[i] = rv[u].x; }
        if (touch_x) x[i] = xv[u].x;
        if constexpr (ZV) { if (MODE != 0 && !done) v[i] = 0.0; }
      }
    }
  }
}
// The NEW search direction at the interface DoFs a rank has to send, written straight into the send buffer BEFORE the update kernel
// runs (same arithmetic, cgm_update_one): the ghost exchange of p then travels under the update kernel instead of after it.
template <int MODE>
__global__ void __launch_bounds__(256) cgm_pack_updated_kernel(const uint32_t *idx, uint32_t n, const double *p, const double *r, const double *v,
                                                             const double *diag, const double *sc, const int *st, double *buf)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = idx[i];
  double pi = p[g];
  if (!st[ST_DONE]) { // (a stopped solve leaves p alone)
    double ri = r[g], xi = 0.0;
    const double vi = MODE != 0 ? v[g] : 0.0;
    cgm_update_one<MODE>(pi, ri, vi, xi, diag ? diag[g] : 1.0, false, false, sc[SC_ALPHA], sc[SC_BETA], 0.0);
  }
  buf[i] = pi;
}

// update_b (solver.h:142-311): [p.v, v.v, r.v, r.r, r.Dv, v.Dv, r.Dr]
// cbits != NULL: bit i of the bitmap marks a Dirichlet DoF; the operator left its row unset and this kernel applies
// copy_constrained_values (v = p there, bp5/step-64.cu:275) on the fly, stores it, and uses it in the sums (one launch less per iteration)
static __global__ void __launch_bounds__(VB) cgm_dots_kernel(const double *p, const double *r, double *v, const double *diag, size_t n,
                                                     const int *st, double *partials, const uint32_t *cbits)
{
  if (st[ST_DONE]) return;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  auto one = [&](double pi, double ri, double vi, double di) {
    acc[0] += pi * vi; acc[1] += vi * vi; acc[2] += ri * vi; acc[3] += ri * ri;
    acc[4] += ri * di * vi; acc[5] += vi * di * vi; acc[6] += ri * di * ri;
  };
  const size_t stride = (size_t)gridDim.x * VB * 2;
  for (size_t i = ((size_t)blockIdx.x * VB + threadIdx.x) * 2; i < n; i += stride) {
    const uint32_t cb = cbits ? (cbits[i >> 5] >> (i & 31)) & 3u : 0u; // (i is even: both bits of the pair sit in one word)
    if (i + 1 < n) {
      const double2 pv = *reinterpret_cast<const double2 *>(p + i), rv = *reinterpret_cast<const double2 *>(r + i);
      double2 vv = *reinterpret_cast<const double2 *>(v + i);
      if (cb) {
        if (cb & 1u) vv.x = pv.x;
        if (cb & 2u) vv.y = pv.y;
        *reinterpret_cast<double2 *>(v + i) = vv;
      }
      one(pv.x, rv.x, vv.x, diag ? diag[i] : 1.0);
      one(pv.y, rv.y, vv.y, diag ? diag[i + 1] : 1.0);
    } else {
      double vi = v[i];
      if (cb & 1u) { vi = p[i]; v[i] = vi; }
      one(p[i], r[i], vi, diag ? diag[i] : 1.0);
    }
  }
  block_reduce_store<7>(acc, partials);
}
// scalars of one merged iteration (solver.h:496-506,533)
static __global__ void cgm_control_kernel(double *sc, int *st);
// Fused iteration (D == 1: rows 4-6 of the partial sums mirror rows 2, 1, 3): ONE workgroup sums the four distinct rows in a fixed order
// and -- on one rank, where no all-reduce sits in between -- takes the scalar step right away: one launch instead of two (the earlier
// one-launch probe used seven workgroups and a device-scope fence, which cost more than it saved)
__device__ __forceinline__ void cgm_control_step(double *sc, int *st)
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
constexpr int FIN4_THREADS = 1024; // sixteen waves: four per row
template <bool CONTROL>
__global__ void __launch_bounds__(FIN4_THREADS) cgm_finalize4_kernel(const double *partials, int nblk, double *sc, int *st)
{
  if (st[ST_DONE]) { // frozen solve: nothing to sum; the scalar step still clears the pending flag
    if (CONTROL && threadIdx.x == 0) cgm_control_step(sc, st);
    return;
  }
  // 256 lanes per row, four independent loads per lane and trip (a lane that sums its columns one dependent load after the other pays a memory
  // latency per column: 22-27 us per launch on the 54^3 problem, more than the two launches this kernel replaced); fixed order: lane-strided
  // partial sums, shuffle tree per wave, the four waves of a row in ascending order
  __shared__ double red[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = wave >> 2, l = ((wave & 3) << 6) | lane; // row, lane within the row (0..255)
  const double *row = partials + k * PARTIAL_STRIDE;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int i = l; i < nblk; i += 1024) {
    const double a0 = row[i], a1 = i + 256 < nblk ? row[i + 256] : 0.0, a2 = i + 512 < nblk ? row[i + 512] : 0.0, a3 = i + 768 < nblk ? row[i + 768] : 0.0;
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  double s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  double *out = sc + SC_R0;
  if (threadIdx.x < 4) {
    const int r = threadIdx.x;
    const double t = (red[4 * r] + red[4 * r + 1]) + (red[4 * r + 2] + red[4 * r + 3]);
    out[r] = t;
    if (r == 1) out[5] = t; // v.Dv = v.v
    if (r == 2) out[4] = t; // r.Dv = r.v
    if (r == 3) out[6] = t; // r.Dr = r.r
  }
  if constexpr (CONTROL) {
    __syncthreads(); // (workgroup scope: the four sums are visible to thread 0)
    if (threadIdx.x == 0) cgm_control_step(sc, st);
  }
}

static __global__ void cgm_control_kernel(double *sc, int *st) { cgm_control_step(sc, st); }

// merged init: r = -b, x = 0, partial r.r
// x = 0, r = -b, and the first search direction p = -D r right away (update_a0, solver.h:48-72: the same expression as cgm_update_one<0>, so
// the first iteration needs no update launch); v zeroed only for operator kernels that accumulate into it (zero_v)
static __global__ void __launch_bounds__(VB) cgm_init_kernel(const double *b, double *x, double *r, double *p, double *v, const double *diag, size_t n,
                                                     double *partials, bool zero_v)
{
  double acc[2] = {0.0, 0.0};
  const size_t stride = (size_t)gridDim.x * VB;
  for (size_t i = (size_t)blockIdx.x * VB + threadIdx.x; i < n; i += stride) {
    const double ri = -b[i], di = diag ? diag[i] : 1.0;
    x[i] = 0.0; r[i] = ri; p[i] = -(di * ri);
    if (zero_v) v[i] = 0.0;
    acc[0] += ri * ri;
  }
  acc[1] = acc[0];
  block_reduce_store<2>(acc, partials);
}
static __global__ void cgm_init_control_kernel(double *sc, int *st)
{
  sc[SC_RES0] = sc[SC_RES] = sqrt(sc[SC_GG]);
  sc[SC_ALPHA] = sc[SC_BETA] = sc[SC_ALPHA_OLD] = sc[SC_BETA_OLD] = 0.0;
  st[ST_ITER] = 0; st[ST_PENDING] = 0; st[ST_BREAKDOWN] = 0;
  st[ST_DONE] = (sc[SC_RES] <= sc[SC_TOL]) ? 1 : 0;
}

// ---- self-check of the in-launch stream wait-value schedules (bp5_device.hip: halo_streams).  The producer counts itself in exactly like
// apply_block_kernel's signal_part_done and then stays alive until the consumer -- enqueued on the waiting stream behind
// hipStreamWaitValue64 -- has run, or 2 ms have passed: *result = 1 says the wait was released WHILE the producing kernel was running
static __global__ void wait_value_probe_producer(unsigned long long *signal, unsigned long long *consumer_ran, unsigned long long *result, long long max_ticks)
{
  if (threadIdx.x != 0) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __hip_atomic_fetch_add(signal, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long t0 = wall_clock64();
  unsigned long long ok = 0;
  while ((long long)(wall_clock64() - t0) < max_ticks) { // bounded: the kernel ends whatever the runtime does
    if (__hip_atomic_load(consumer_ran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { ok = 1; break; }
    __builtin_amdgcn_s_sleep(32);
  }
  __hip_atomic_store(result, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __global__ void wait_value_probe_consumer(unsigned long long *consumer_ran)
{
  if (threadIdx.x == 0) __hip_atomic_store(consumer_ran, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

} // namespace bp5
