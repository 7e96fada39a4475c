// Host-only parts of the BP5 library: error strings, 1-D shape tables, structured hex mesh
// generator with z-slab partition and halo plan.  No HIP here; usable without a GPU.
//
// Replaces (reference call sites): the shape tables MatrixFree::reinit builds
// (bp5/step-64.cu:243-248), GridGenerator::subdivided_hyper_rectangle + refine_global +
// distribute_dofs + boundary constraints (bp5/step-64.cu:341-358,629-663) and the p4est
// partition / Partitioner (bp5/step-64.cu:310,347-349).
#include "bp5_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

namespace bp5 {
thread_local std::string g_last_error;
int fail(int status, const std::string &msg)
{
  g_last_error = msg;
  return status;
}

// ---------------------------------------------------------------------------------- tables
static void legendre(int n, long double x, long double &P, long double &dP)
{
  if (n == 0) { P = 1; dP = 0; return; }
  long double p0 = 1, p1 = x, d0 = 0, d1 = 1;
  for (int k = 2; k <= n; ++k) {
    long double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    long double d2 = d0 + (2 * k - 1) * p1;
    p0 = p1; p1 = p2; d0 = d1; d1 = d2;
  }
  P = p1; dP = d1;
}

static void symmetrise(std::vector<long double> &x, std::vector<long double> *w)
{
  const int n = (int)x.size();
  for (int i = 0; i < n / 2; ++i) {
    long double a = (x[i] - x[n - 1 - i]) / 2;
    x[i] = a; x[n - 1 - i] = -a;
    if (w) { long double b = ((*w)[i] + (*w)[n - 1 - i]) / 2; (*w)[i] = b; (*w)[n - 1 - i] = b; }
  }
  if (n % 2) x[n / 2] = 0;
}

static void gauss(int n, std::vector<long double> &x, std::vector<long double> &w)
{
  const long double pi = acosl(-1.0L);
  x.resize(n); w.resize(n);
  for (int i = 0; i < n; ++i) {
    long double z = -cosl(pi * (i + 0.75L) / (n + 0.5L)), P, dP;
    for (int it = 0; it < 100; ++it) {
      legendre(n, z, P, dP);
      long double dz = P / dP; z -= dz;
      if (fabsl(dz) < 1e-19L) break;
    }
    legendre(n, z, P, dP);
    x[i] = z; w[i] = 2 / ((1 - z * z) * dP * dP);
  }
  symmetrise(x, &w);
}

static void gll(int n, std::vector<long double> &x, std::vector<long double> &w)
{
  const long double pi = acosl(-1.0L);
  const int m = n - 1;
  x.assign(n, 0); w.assign(n, 0);
  x[0] = -1; x[n - 1] = 1;
  for (int i = 1; i < n - 1; ++i) {
    long double z = -cosl(pi * i / m);
    for (int it = 0; it < 100; ++it) {
      long double P, dP; legendre(m, z, P, dP);
      long double ddP = (2 * z * dP - m * (m + 1.0L) * P) / (1 - z * z);
      long double dz = dP / ddP; z -= dz;
      if (fabsl(dz) < 1e-19L) break;
    }
    x[i] = z;
  }
  symmetrise(x, nullptr);
  for (int i = 0; i < n; ++i) { long double P, dP; legendre(m, x[i], P, dP); w[i] = 2 / (m * (m + 1.0L) * P * P); }
}

int shape_tables(int degree, int quadrature, Tables &t)
{
  if (degree < 1 || degree > BP5_MAX_DEGREE) return fail(BP5_ERR_INVALID, "degree must be in 1..8");
  if (quadrature != BP5_QUAD_GAUSS && quadrature != BP5_QUAD_GLL) return fail(BP5_ERR_INVALID, "unknown quadrature");
  const int n = degree + 1;
  t.n = n;
  std::vector<long double> xn, wn, xq, wq;
  gll(n, xn, wn);
  if (quadrature == BP5_QUAD_GLL) gll(n, xq, wq); else gauss(n, xq, wq);
  for (auto &v : xn) v = (v + 1) / 2;
  for (auto &v : xq) v = (v + 1) / 2;
  for (auto &v : wq) v /= 2;
  for (int i = 0; i < n; ++i) { t.nodes[i] = (double)xn[i]; t.pts[i] = (double)xq[i]; t.w[i] = (double)wq[i]; }
  for (int q = 0; q < n; ++q)
    for (int i = 0; i < n; ++i) {
      long double den = 1, num = 1, s = 0;
      for (int m = 0; m < n; ++m) if (m != i) { den *= xn[i] - xn[m]; num *= xq[q] - xn[m]; }
      for (int l = 0; l < n; ++l) {
        if (l == i) continue;
        long double pr = 1;
        for (int m = 0; m < n; ++m) if (m != i && m != l) pr *= xq[q] - xn[m];
        s += pr;
      }
      t.N[q * n + i] = (double)(num / den);
      t.D[q * n + i] = (double)(s / den);
    }
  if (quadrature == BP5_QUAD_GLL)
    for (int q = 0; q < n; ++q) for (int i = 0; i < n; ++i) t.N[q * n + i] = (q == i) ? 1.0 : 0.0;
  // bitwise (anti)symmetry under x -> 1-x: the device kernels read only half of each table
  for (int q = 0; q < n; ++q)
    for (int i = 0; i < n; ++i) {
      const int f = q * n + i, g = (n - 1 - q) * n + (n - 1 - i);
      if (f < g) {
        const double a = 0.5 * (t.N[f] + t.N[g]), b = 0.5 * (t.D[f] - t.D[g]);
        t.N[f] = a; t.N[g] = a; t.D[f] = b; t.D[g] = -b;
      } else if (f == g) t.D[f] = 0.0;
    }
  return BP5_OK;
}
} // namespace bp5

using namespace bp5;

extern "C" const char *bp5_strerror(int s)
{
  switch (s) {
    case BP5_OK: return "ok";
    case BP5_ERR_INVALID: return "invalid argument";
    case BP5_ERR_HIP: return "HIP runtime error";
    case BP5_ERR_NO_DEVICE: return "no HIP device (there is no CPU fallback)";
    case BP5_ERR_RCCL: return "RCCL error";
    case BP5_ERR_UNSUPPORTED: return "unsupported";
    case BP5_ERR_BREAKDOWN: return "CG breakdown";
    case BP5_ERR_NO_CONVERGENCE: return "no convergence";
  }
  return "unknown status";
}
extern "C" const char *bp5_last_error(void) { return g_last_error.c_str(); }

extern "C" int bp5_shape_tables(int degree, int quadrature, double *nodes, double *pts, double *w, double *N, double *D)
{
  Tables t;
  int st = shape_tables(degree, quadrature, t);
  if (st) return st;
  const int n = t.n;
  if (nodes) memcpy(nodes, t.nodes, n * sizeof(double));
  if (pts) memcpy(pts, t.pts, n * sizeof(double));
  if (w) memcpy(w, t.w, n * sizeof(double));
  if (N) memcpy(N, t.N, n * n * sizeof(double));
  if (D) memcpy(D, t.D, n * n * sizeof(double));
  return BP5_OK;
}

// ---------------------------------------------------------------------------------- mesh
struct bp5_mesh {
  bp5_mesh_desc desc;
  uint32_t n_cells = 0, n_interior = 0, n_owned = 0, n_ghost = 0;
  uint64_t n_global = 0;
  uint32_t ND[3] = {0, 0, 0};
  std::vector<uint32_t> l2g, constrained, send_offsets, send_indices, recv_offsets;
  std::vector<double> coords;
  std::vector<uint64_t> gids;
  std::vector<int> neighbors;
};

extern "C" int bp5_mesh_create_brick(const bp5_mesh_desc *d, bp5_mesh **out)
{
  if (!d || !out) return fail(BP5_ERR_INVALID, "null argument");
  const int p = d->degree, n = p + 1;
  if (p < 1 || p > BP5_MAX_DEGREE) return fail(BP5_ERR_INVALID, "degree must be in 1..8");
  if (d->n_ranks < 1 || d->rank < 0 || d->rank >= d->n_ranks) return fail(BP5_ERR_INVALID, "bad rank");
  const uint64_t n0 = d->cells[0], n1 = d->cells[1], n2 = d->cells[2];
  if (!n0 || !n1 || !n2 || n2 < (uint64_t)d->n_ranks) return fail(BP5_ERR_INVALID, "need >= 1 cell layer per rank");
  const uint64_t NX = p * n0 + 1, NY = p * n1 + 1, NZ = p * n2 + 1;
  const int r = d->rank, R = d->n_ranks;
  const uint64_t z0 = n2 * r / R, z1 = n2 * (r + 1) / R; // owned cell layers [z0,z1)
  const uint64_t Kbot = p * z0, Ktop = p * z1;           // DoF planes touched
  const uint64_t Kown0 = Kbot + (r > 0 ? 1 : 0);         // interface plane owned by the lower rank
  const uint64_t plane = NX * NY;
  const uint64_t n_owned = (Ktop - Kown0 + 1) * plane, n_ghost = (r > 0 ? plane : 0);
  const uint64_t n_cells = n0 * n1 * (z1 - z0);
  if (n_owned + n_ghost >= (1ull << 32) || n_cells * n * n * n >= (1ull << 40))
    return fail(BP5_ERR_INVALID, "local problem exceeds 32-bit DoF indices");
  bp5_mesh *m = new (std::nothrow) bp5_mesh;
  if (!m) return fail(BP5_ERR_INVALID, "out of memory");
  m->desc = *d;
  m->n_cells = (uint32_t)n_cells; m->n_owned = (uint32_t)n_owned; m->n_ghost = (uint32_t)n_ghost;
  m->n_global = NX * NY * NZ;
  m->ND[0] = (uint32_t)NX; m->ND[1] = (uint32_t)NY; m->ND[2] = (uint32_t)NZ;
  Tables t;
  shape_tables(p, BP5_QUAD_GLL, t);

  auto local_of = [&](uint64_t I, uint64_t J, uint64_t K) -> uint32_t {
    if (K >= Kown0) return (uint32_t)(I + NX * (J + NY * (K - Kown0)));
    return (uint32_t)(n_owned + I + NX * J); // ghost plane K == Kbot, owned by rank-1
  };

  // cells: layers not touching ghosts first, the ghost-touching bottom layer (r>0) last
  const size_t nl = (size_t)n * n * n;
  m->l2g.resize(n_cells * nl);
  std::vector<uint64_t> layers;
  for (uint64_t z = z0 + (r > 0 ? 1 : 0); z < z1; ++z) layers.push_back(z);
  m->n_interior = (uint32_t)(layers.size() * n0 * n1);
  if (r > 0) layers.push_back(z0);
  size_t c = 0;
  for (uint64_t z : layers)
    for (uint64_t y = 0; y < n1; ++y)
      for (uint64_t x = 0; x < n0; ++x, ++c) {
        uint32_t *dst = &m->l2g[c * nl];
        for (int k = 0; k < n; ++k)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) dst[i + n * (j + n * k)] = local_of(p * x + i, p * y + j, p * z + k);
      }

  // coordinates, global ids, constraints
  const uint64_t nloc = n_owned + n_ghost;
  m->coords.resize(nloc * 3);
  m->gids.resize(nloc);
  const double L[3] = {n0 * d->h, n1 * d->h, n2 * d->h};
  auto coord1 = [&](uint64_t G, uint64_t ncell) -> double {
    if (G == (uint64_t)p * ncell) return ncell * d->h;
    return ((double)(G / p) + t.nodes[G % p]) * d->h;
  };
  const double twopi = 2.0 * 3.14159265358979323846;
  auto put = [&](uint32_t loc, uint64_t I, uint64_t J, uint64_t K) {
    double X[3] = {coord1(I, n0), coord1(J, n1), coord1(K, n2)};
    if (d->deform_amp != 0.0) {
      const double s = sin(twopi * X[0] / L[0]) * sin(twopi * X[1] / L[1]) * sin(twopi * X[2] / L[2]);
      const double sc[3] = {1.0, -0.8, 0.6};
      for (int e = 0; e < 3; ++e) X[e] += d->deform_amp * sc[e] * L[e] * s;
    }
    for (int e = 0; e < 3; ++e) m->coords[3 * (size_t)loc + e] = X[e];
    m->gids[loc] = I + NX * (J + NY * K);
    if (I == 0 || I == NX - 1 || J == 0 || J == NY - 1 || K == 0 || K == NZ - 1) m->constrained.push_back(loc);
  };
  for (uint64_t K = Kown0; K <= Ktop; ++K)
    for (uint64_t J = 0; J < NY; ++J)
      for (uint64_t I = 0; I < NX; ++I) put(local_of(I, J, K), I, J, K);
  if (r > 0)
    for (uint64_t J = 0; J < NY; ++J)
      for (uint64_t I = 0; I < NX; ++I) put(local_of(I, J, Kbot), I, J, Kbot);
  std::sort(m->constrained.begin(), m->constrained.end());

  // halo plan: ghosts come from rank-1 (its top plane); our top plane goes to rank+1
  m->send_offsets.push_back(0);
  m->recv_offsets.push_back(0);
  if (r > 0) { // neighbour r-1: we receive the ghost plane, send nothing
    m->neighbors.push_back(r - 1);
    m->send_offsets.push_back((uint32_t)m->send_indices.size());
    m->recv_offsets.push_back((uint32_t)plane);
  }
  if (r < R - 1) { // neighbour r+1: we send our top plane, receive nothing
    m->neighbors.push_back(r + 1);
    for (uint64_t q = 0; q < plane; ++q) m->send_indices.push_back((uint32_t)(n_owned - plane + q));
    m->send_offsets.push_back((uint32_t)m->send_indices.size());
    m->recv_offsets.push_back(m->recv_offsets.back());
  }
  *out = m;
  return BP5_OK;
}

extern "C" int bp5_mesh_view_get(const bp5_mesh *m, bp5_mesh_view *v)
{
  if (!m || !v) return fail(BP5_ERR_INVALID, "null argument");
  v->degree = m->desc.degree;
  v->n_cells = m->n_cells; v->n_interior_cells = m->n_interior;
  v->n_owned = m->n_owned; v->n_ghost = m->n_ghost; v->n_global_dofs = m->n_global;
  for (int e = 0; e < 3; ++e) v->global_dofs_per_dir[e] = m->ND[e];
  v->local_to_global_host = m->l2g.data();
  v->node_coords_host = m->coords.data();
  v->global_ids_host = m->gids.data();
  v->constrained_host = m->constrained.data();
  v->n_constrained = (uint32_t)m->constrained.size();
  v->n_neighbors = (int)m->neighbors.size();
  v->neighbor_rank_host = m->neighbors.data();
  v->send_offsets_host = m->send_offsets.data();
  v->send_indices_host = m->send_indices.data();
  v->recv_offsets_host = m->recv_offsets.data();
  return BP5_OK;
}

extern "C" void bp5_mesh_destroy(bp5_mesh *m) { delete m; }
