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
#include <cstdlib>
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

// ---------------------------------------------------------------------------------- team plan
int build_team_plan(const uint32_t *l2g, uint32_t n_cells, int n3, size_t n_local, int cpt, TeamPlanHost &out,
                    const uint32_t *blk_off, uint32_t n_blocks, int cells_per_pass)
{
  // groups: either uniform teams of `cpt` consecutive cells, or the explicit blocks blk_off[]
  std::vector<uint32_t> goff;
  if (blk_off) goff.assign(blk_off, blk_off + n_blocks + 1);
  else for (size_t c0 = 0;; c0 += cpt) { goff.push_back((uint32_t)std::min<size_t>(c0, n_cells)); if (c0 >= n_cells) break; }
  out.group_cell_off = goff;
  if (n_local >= (1ull << 31)) return fail(BP5_ERR_UNSUPPORTED, "team plan needs fewer than 2^31 local DoFs");
  const size_t n_teams = goff.size() - 1;

  out.off.assign(n_teams + 1, 0);
  out.pos.assign((size_t)n_cells * n3, 0);
  std::vector<uint32_t> count(n_teams, 0);
  // pass 1: distinct count per team, positions
  std::vector<std::vector<uint32_t>> lists(n_teams);
  int too_long = 0, too_many_rounds = 0;
#pragma omp parallel
  {
    std::vector<uint32_t> tmp;
#pragma omp for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t)n_teams; ++t) {
      const size_t c0 = goff[t], c1 = goff[t + 1];
      tmp.assign(l2g + c0 * n3, l2g + c1 * n3);
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      if (tmp.size() > 65536) { // positions are 16 bits: refuse before they are truncated
#pragma omp atomic write
        too_long = 1;
        continue;
      }
      for (size_t s = c0 * n3; s < c1 * n3; ++s)
        out.pos[s] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), l2g[s]) - tmp.begin());
      lists[t] = tmp;
    }
  }
  if (too_long) return fail(BP5_ERR_UNSUPPORTED, "cell group touches more than 65536 DoFs (16-bit positions)");
  for (size_t t = 0; t < n_teams; ++t) {
    out.off[t + 1] = out.off[t] + (uint32_t)lists[t].size();
  }
  out.dofs.resize(out.off[n_teams]);
  std::vector<uint8_t> mult(n_local, 0);
  for (size_t t = 0; t < n_teams; ++t)
    for (uint32_t g : lists[t]) if (mult[g] < 2) ++mult[g];
  out.covers_all = true;
  for (size_t g = 0; g < n_local; ++g) if (!mult[g]) { out.covers_all = false; break; }
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < (int64_t)n_teams; ++t) {
    uint32_t *d = out.dofs.data() + out.off[t];
    for (size_t i = 0; i < lists[t].size(); ++i) d[i] = lists[t][i] | (mult[lists[t][i]] == 1 ? 0x80000000u : 0u);
  }
  // accumulation order.  Team kernel (cells_per_pass == 0): one pass per group, greedy colouring of
  // its cells into rounds.  Block kernel: the cells of a group are packed into passes of
  // cells_per_pass slots such that cells of one pass share no DoF whenever possible (for a brick
  // of hexes these are the 2x2x2 parity classes); only if a pass cannot be filled that way are
  // conflicting cells admitted, in a later round of that pass.
  out.cell_round.assign(n_cells, 0);
  out.team_rounds.assign(n_teams, 1);
  if (cells_per_pass > 0) out.pass_off.assign(n_teams + 1, 0);
  std::vector<std::vector<uint32_t>> group_passes(cells_per_pass > 0 ? n_teams : 0);
#pragma omp parallel
  {
    std::vector<uint64_t> pos_mask; // per list entry: bitmask of passes (or rounds) already using it
    std::vector<int> fill;
#pragma omp for schedule(dynamic, 64)
    for (int64_t t = 0; t < (int64_t)n_teams; ++t) {
      const size_t c0 = goff[t], c1 = goff[t + 1];
      pos_mask.assign(lists[t].size(), 0ull);
      if (cells_per_pass <= 0) {
        uint8_t nr = 1;
        for (size_t c = c0; c < c1; ++c) {
          uint64_t used = 0;
          for (int i = 0; i < n3; ++i) used |= pos_mask[out.pos[c * n3 + i]];
          uint8_t r = 0;
          while (r < 64 && (used & (1ull << r))) ++r;
          if (r >= 64) { // more than 64 mutually conflicting cells in one group: the 64-bit round masks are exhausted
#pragma omp atomic write
            too_many_rounds = 1;
            r = 63;
          }
          out.cell_round[c] = r;
          nr = std::max<uint8_t>(nr, r + 1);
          for (int i = 0; i < n3; ++i) pos_mask[out.pos[c * n3 + i]] |= 1ull << r;
        }
        out.team_rounds[t] = nr;
        continue;
      }
      // conflict-free packing into at most 64 passes
      fill.clear();
      std::vector<uint32_t> &passes = group_passes[t];
      passes.clear();
      std::vector<uint32_t> leftover;
      for (size_t c = c0; c < c1; ++c) {
        uint64_t used = 0;
        for (int i = 0; i < n3; ++i) used |= pos_mask[out.pos[c * n3 + i]];
        int pick = -1;
        for (int q = 0; q < (int)fill.size(); ++q)
          if (!(used & (1ull << q)) && fill[q] < cells_per_pass) { pick = q; break; }
        if (pick < 0 && fill.size() < 64) { pick = (int)fill.size(); fill.push_back(0); passes.resize(passes.size() + cells_per_pass, 0xffffffffu); }
        if (pick < 0) { leftover.push_back((uint32_t)c); continue; }
        passes[(size_t)pick * cells_per_pass + fill[pick]++] = (uint32_t)c;
        for (int i = 0; i < n3; ++i) pos_mask[out.pos[c * n3 + i]] |= 1ull << pick;
      }
      // cells that found no conflict-free pass: append to the emptiest passes, later rounds
      uint8_t nr = 1;
      for (uint32_t c : leftover) {
        int pick = -1;
        for (int q = 0; q < (int)fill.size(); ++q) if (fill[q] < cells_per_pass && (pick < 0 || fill[q] < fill[pick])) pick = q;
        if (pick < 0) { pick = (int)fill.size(); fill.push_back(0); passes.resize(passes.size() + cells_per_pass, 0xffffffffu); }
        // round = 1 + highest round among conflicting cells already in that pass (conservative: next round)
        int r = 0;
        for (int k = 0; k < fill[pick]; ++k) r = std::max<int>(r, out.cell_round[passes[(size_t)pick * cells_per_pass + k]] + 1);
        if (r > 254) { // rounds are 8 bits: a wrapped round would put conflicting cells in the same round
#pragma omp atomic write
          too_many_rounds = 1;
          r = 254;
        }
        out.cell_round[c] = (uint8_t)r;
        nr = std::max<uint8_t>(nr, r + 1);
        passes[(size_t)pick * cells_per_pass + fill[pick]++] = c;
      }
      // merge under-full passes pairwise: the cells of the absorbed pass run in a later round of the
      // absorbing pass (e.g. 4x4x2 bricks: 8 parity classes of 4 cells -> 4 passes of 8 cells, 2 rounds)
      for (bool merged = true; merged;) {
        merged = false;
        for (int qa = 0; qa < (int)fill.size() && !merged; ++qa)
          for (int qb = (int)fill.size() - 1; qb > qa; --qb)
            if (fill[qa] > 0 && fill[qb] > 0 && fill[qa] + fill[qb] <= cells_per_pass) {
              uint8_t ra = 0;
              for (int k = 0; k < fill[qa]; ++k) ra = std::max<uint8_t>(ra, out.cell_round[passes[(size_t)qa * cells_per_pass + k]]);
              for (int k = 0; k < fill[qb]; ++k) {
                const uint32_t cb = passes[(size_t)qb * cells_per_pass + k];
                if ((int)out.cell_round[cb] + ra + 1 > 254) {
#pragma omp atomic write
                  too_many_rounds = 1;
                }
                out.cell_round[cb] = (uint8_t)std::min<int>(254, out.cell_round[cb] + ra + 1);
                nr = std::max<uint8_t>(nr, out.cell_round[cb] + 1);
                passes[(size_t)qa * cells_per_pass + fill[qa]++] = cb;
              }
              // drop pass qb
              passes.erase(passes.begin() + (size_t)qb * cells_per_pass, passes.begin() + (size_t)(qb + 1) * cells_per_pass);
              fill.erase(fill.begin() + qb);
              merged = true;
              break;
            }
      }
      // dissolve the emptiest pass into the free slots of the others as long as that saves a pass (e.g. p = 5, 6x4x2 bricks, 7 cells per pass:
      // 8 parity classes of 6 cells -> 7 passes, the eighth class spread one cell per pass, in a second round)
      for (;;) {
        const int np = (int)fill.size();
        int total = 0, qs = -1;
        for (int q = 0; q < np; ++q) { total += fill[q]; if (fill[q] > 0 && (qs < 0 || fill[q] <= fill[qs])) qs = q; }
        if (np < 2 || qs < 0 || total > (np - 1) * cells_per_pass) break;
        std::vector<uint8_t> top(np, 0); // highest round of every pass BEFORE it absorbs anything: the absorbed cells keep their mutual order
        for (int q = 0; q < np; ++q)
          for (int k = 0; k < fill[q]; ++k) top[q] = std::max<uint8_t>(top[q], out.cell_round[passes[(size_t)q * cells_per_pass + k]]);
        for (int k = 0; k < fill[qs]; ++k) {
          const uint32_t cb = passes[(size_t)qs * cells_per_pass + k];
          int qa = -1;
          for (int q = 0; q < np; ++q) if (q != qs && fill[q] < cells_per_pass && (qa < 0 || fill[q] < fill[qa])) qa = q;
          if ((int)out.cell_round[cb] + top[qa] + 1 > 254) {
#pragma omp atomic write
            too_many_rounds = 1;
          }
          out.cell_round[cb] = (uint8_t)std::min<int>(254, out.cell_round[cb] + top[qa] + 1);
          nr = std::max<uint8_t>(nr, out.cell_round[cb] + 1);
          passes[(size_t)qa * cells_per_pass + fill[qa]++] = cb;
        }
        passes.erase(passes.begin() + (size_t)qs * cells_per_pass, passes.begin() + (size_t)(qs + 1) * cells_per_pass);
        fill.erase(fill.begin() + qs);
      }
      out.team_rounds[t] = nr;
      // idle slots repeat the pass's first cell, flagged by bit 31
      for (size_t q = 0; q < fill.size(); ++q)
        for (int k = fill[q]; k < cells_per_pass; ++k) passes[q * cells_per_pass + k] = passes[q * cells_per_pass] | 0x80000000u;
    }
  }
  if (too_many_rounds) return fail(BP5_ERR_UNSUPPORTED, "cell groups need too many accumulation rounds (highly irregular cell order): use the atomic kernel");
  if (cells_per_pass > 0) {
    for (size_t t = 0; t < n_teams; ++t) out.pass_off[t + 1] = out.pass_off[t] + (uint32_t)(group_passes[t].size() / cells_per_pass);
    out.pass_cell.resize((size_t)out.pass_off[n_teams] * cells_per_pass);
    for (size_t t = 0; t < n_teams; ++t)
      std::copy(group_passes[t].begin(), group_passes[t].end(), out.pass_cell.begin() + (size_t)out.pass_off[t] * cells_per_pass);
  }
  // shared DoFs: CSR over partial-slab slots
  {
    std::vector<uint32_t> cnt(n_local, 0);
    for (size_t e = 0; e < out.dofs.size(); ++e)
      if (!(out.dofs[e] & 0x80000000u)) ++cnt[out.dofs[e]];
    std::vector<uint32_t> sh_index(n_local, 0xffffffffu);
    for (size_t g = 0; g < n_local; ++g)
      if (cnt[g]) { sh_index[g] = (uint32_t)out.sh_dof.size(); out.sh_dof.push_back((uint32_t)g); }
    out.sh_off.assign(out.sh_dof.size() + 1, 0);
    for (size_t k = 0; k < out.sh_dof.size(); ++k) out.sh_off[k + 1] = out.sh_off[k] + cnt[out.sh_dof[k]];
    out.sh_slot.resize(out.sh_off.back());
    std::vector<uint32_t> cur(out.sh_off.begin(), out.sh_off.end() - 1);
    for (size_t e = 0; e < out.dofs.size(); ++e) // e ascends with the team index -> fixed summation order
      if (!(out.dofs[e] & 0x80000000u)) out.sh_slot[cur[sh_index[out.dofs[e]]]++] = (uint32_t)e;
  }
  return BP5_OK;
}

// ---------------------------------------------------------------------------------- z-marching plan
int build_march_plan(const uint32_t *l2g, uint32_t n_cells, int n, int cpt, int max_steps, MarchPlanHost &out)
{
  const int n2 = n * n, n3 = n2 * n;
  // successor of a cell: the cell whose bottom layer (k = 0) has exactly the DoFs of this cell's top layer.
  // key = (first, last) DoF of the layer, verified by full comparison
  std::vector<std::pair<uint64_t, uint32_t>> bottoms(n_cells);
  for (uint32_t c = 0; c < n_cells; ++c) bottoms[c] = {((uint64_t)l2g[(size_t)c * n3] << 32) | l2g[(size_t)c * n3 + n2 - 1], c};
  std::sort(bottoms.begin(), bottoms.end());
  std::vector<uint32_t> succ(n_cells, 0xffffffffu);
  std::vector<uint8_t> has_pred(n_cells, 0);
  for (uint32_t c = 0; c < n_cells; ++c) {
    const uint32_t *top = l2g + (size_t)c * n3 + (size_t)(n - 1) * n2;
    const uint64_t key = ((uint64_t)top[0] << 32) | top[n2 - 1];
    auto it = std::lower_bound(bottoms.begin(), bottoms.end(), std::make_pair(key, 0u));
    for (; it != bottoms.end() && it->first == key; ++it) {
      const uint32_t d = it->second;
      if (d != c && !has_pred[d] && std::equal(top, top + n2, l2g + (size_t)d * n3)) { succ[c] = d; has_pred[d] = 1; break; }
    }
  }
  // chains, cut into segments of at most max_steps cells; segment heads in cell order
  std::vector<std::vector<uint32_t>> segs;
  for (uint32_t c = 0; c < n_cells; ++c) {
    if (has_pred[c]) continue;
    std::vector<uint32_t> cur;
    for (uint32_t d = c; d != 0xffffffffu; d = succ[d]) {
      cur.push_back(d);
      if ((int)cur.size() == max_steps) { segs.push_back(cur); cur.clear(); }
    }
    if (!cur.empty()) segs.push_back(cur);
  }
  // sort segments by their head cell so that teams take neighbouring columns
  std::sort(segs.begin(), segs.end(), [](const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) { return a[0] < b[0]; });
  const size_t n_teams = (segs.size() + cpt - 1) / cpt;
  out.team_off.assign(n_teams + 1, 0);
  for (size_t t = 0; t < n_teams; ++t) {
    size_t steps = 0;
    for (size_t k = t * cpt; k < std::min(segs.size(), (t + 1) * (size_t)cpt); ++k) steps = std::max(steps, segs[k].size());
    out.team_off[t + 1] = out.team_off[t] + (uint32_t)steps;
  }
  out.entries.assign((size_t)out.team_off[n_teams] * cpt, 0x80000000u);
  for (size_t t = 0; t < n_teams; ++t) {
    const size_t steps = out.team_off[t + 1] - out.team_off[t];
    for (int sl = 0; sl < cpt; ++sl) {
      const size_t k = t * cpt + sl;
      for (size_t st = 0; st < steps; ++st) {
        uint32_t &e = out.entries[((size_t)out.team_off[t] + st) * cpt + sl];
        if (k < segs.size() && st < segs[k].size()) e = segs[k][st] | (st > 0 ? 0x40000000u : 0u);
        else e = (k < segs.size() ? segs[k][0] : segs[t * cpt][0]) | 0x80000000u; // idle: any valid cell id
      }
    }
  }
  if (n_cells >= 0x40000000u) return fail(BP5_ERR_UNSUPPORTED, "march plan needs fewer than 2^30 cells");
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
  std::vector<uint32_t> l2g, constrained, send_offsets, send_indices, recv_offsets, block_off;
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

  // ---- local DoF numbering of the owned range [0,NX) x [0,NY) x [Kown0,Ktop]
  const uint64_t zi0_ = z0 + (r > 0 ? 1 : 0);
  const bool blocked_ = d->cell_block[0] && d->cell_block[1] && d->cell_block[2];
  if (d->dof_numbering < 0 || d->dof_numbering > 2) { delete m; return fail(BP5_ERR_INVALID, "dof_numbering must be 0, 1 or 2"); }
  if (d->dof_numbering == 1 && !blocked_) { delete m; return fail(BP5_ERR_INVALID, "block-major numbering needs cell_block"); }
  const bool block_major = d->dof_numbering == 1 || (d->dof_numbering == 2 && blocked_); // (2: cell interiors first, the rest block-major where there are blocks)
  if (d->cell_block_order != 0 && (d->cell_block_order != 1 || !blocked_)) { delete m; return fail(BP5_ERR_INVALID, "cell_block_order must be 0, or 1 with cell_block"); }
  // per direction: slots are alternately block-boundary planes (length 1) and the runs between them
  struct Dir {
    std::vector<uint8_t> kind;      // per coordinate (relative to lo): 1 = plane, 0 = run
    std::vector<uint32_t> slot, in; // slot index among its kind, offset inside the run
    std::vector<uint64_t> L[2], Pre[2]; // per kind: lengths and exclusive prefix sums
    uint64_t S[2] = {0, 0};
  } dir[3];
  auto build_dir = [&](Dir &D, uint64_t lo, uint64_t hi, const std::vector<uint64_t> &planes) {
    const uint64_t N = hi - lo + 1;
    D.kind.assign(N, 0); D.slot.assign(N, 0); D.in.assign(N, 0);
    for (uint64_t q : planes) if (q >= lo && q <= hi) D.kind[q - lo] = 1;
    uint32_t nrun = 0, nplane = 0;
    uint64_t runlen = 0;
    for (uint64_t x = 0; x < N; ++x) {
      if (D.kind[x]) {
        if (runlen) { D.L[0].push_back(runlen); ++nrun; runlen = 0; }
        D.slot[x] = nplane++; D.L[1].push_back(1);
      } else { D.slot[x] = nrun; D.in[x] = (uint32_t)runlen++; }
    }
    if (runlen) D.L[0].push_back(runlen);
    for (int k = 0; k < 2; ++k) {
      D.Pre[k].assign(D.L[k].size() + 1, 0);
      for (size_t i = 0; i < D.L[k].size(); ++i) D.Pre[k][i + 1] = D.Pre[k][i] + D.L[k][i];
      D.S[k] = D.Pre[k].back();
    }
  };
  uint64_t class_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (block_major) {
    std::vector<uint64_t> px, py, pz;
    for (uint64_t X = 0; X < n0; X += d->cell_block[0]) px.push_back(p * X);
    px.push_back(p * n0);
    for (uint64_t Y = 0; Y < n1; Y += d->cell_block[1]) py.push_back(p * Y);
    py.push_back(p * n1);
    for (uint64_t Z = zi0_; Z < z1; Z += d->cell_block[2]) pz.push_back(p * Z);
    pz.push_back(p * z1);
    if (r > 0) pz.push_back(p * z0); // ghost plane: outside the owned range, ignored by build_dir
    build_dir(dir[0], 0, NX - 1, px);
    build_dir(dir[1], 0, NY - 1, py);
    build_dir(dir[2], Kown0, Ktop, pz);
    // classes ordered: interior, x-/y-/z-faces, edges, vertices; class id = kx + 2 ky + 4 kz
    const int order[8] = {0, 1, 2, 4, 3, 5, 6, 7};
    uint64_t base = 0;
    for (int o = 0; o < 8; ++o) {
      const int cl = order[o];
      class_base[cl] = base;
      base += dir[0].S[cl & 1] * dir[1].S[(cl >> 1) & 1] * dir[2].S[(cl >> 2) & 1];
    }
  }
  // position of (I, J) inside a z-plane that is exchanged between ranks (the ghost plane K == Kbot of rank > 0 = the top plane of the rank
  // below).  Lexicographic, or (block-major numbering) the 2-D analogue of the owned numbering: footprint interiors of the bricks first,
  // then the lines between them, then their crossings -- every entity contiguous, x fastest -- so that the bricks of the ghost-touching
  // layer are lattice blocks like all others (bp5_device.hip: detect_lattice_blocks).  Sender and receiver use the same function.
  uint64_t plane_class_base[4] = {0, 0, 0, 0};
  if (block_major) {
    uint64_t base2 = 0;
    for (int cl = 0; cl < 4; ++cl) { // (kx, ky) = (0,0), (1,0), (0,1), (1,1)
      plane_class_base[cl] = base2;
      base2 += dir[0].S[cl & 1] * dir[1].S[(cl >> 1) & 1];
    }
  }
  auto plane_pos = [&](uint64_t I, uint64_t J) -> uint64_t {
    if (!block_major) return I + NX * J;
    const uint64_t c2[2] = {I, J};
    int k[2]; uint64_t sl[2], in[2], Ls[2], Pr[2], Ss[2];
    for (int e = 0; e < 2; ++e) {
      k[e] = dir[e].kind[c2[e]]; sl[e] = dir[e].slot[c2[e]]; in[e] = dir[e].in[c2[e]];
      Ls[e] = dir[e].L[k[e]][sl[e]]; Pr[e] = dir[e].Pre[k[e]][sl[e]]; Ss[e] = dir[e].S[k[e]];
    }
    return plane_class_base[k[0] + 2 * k[1]] + Pr[1] * Ss[0] + Ls[1] * Pr[0] + in[0] + Ls[0] * in[1];
  };
  auto local_of = [&](uint64_t I, uint64_t J, uint64_t K) -> uint32_t {
    if (K < Kown0) return (uint32_t)(n_owned + plane_pos(I, J)); // ghost plane K == Kbot, owned by rank-1
    if (!block_major) return (uint32_t)(I + NX * (J + NY * (K - Kown0)));
    const uint64_t c3[3] = {I, J, K - Kown0};
    int k[3]; uint64_t sl[3], in[3], Ls[3], Pr[3], Ss[3];
    for (int e = 0; e < 3; ++e) {
      k[e] = dir[e].kind[c3[e]]; sl[e] = dir[e].slot[c3[e]]; in[e] = dir[e].in[c3[e]];
      Ls[e] = dir[e].L[k[e]][sl[e]]; Pr[e] = dir[e].Pre[k[e]][sl[e]]; Ss[e] = dir[e].S[k[e]];
    }
    const int cl = k[0] + 2 * k[1] + 4 * k[2];
    // entities nested z outer, y, x inner; inside an entity x fastest
    const uint64_t ent = Pr[2] * (Ss[1] * Ss[0]) + Ls[2] * (Pr[1] * Ss[0] + Ls[1] * Pr[0]);
    return (uint32_t)(class_base[cl] + ent + in[0] + Ls[0] * (in[1] + Ls[1] * in[2]));
  };

  // cells: layers not touching ghosts first, the ghost-touching bottom layer (r>0) last; inside
  // each of the two regions cells are emitted brick by brick (desc.cell_block) so that consecutive
  // cells form compact groups for the block-assembled operator kernel
  const size_t nl = (size_t)n * n * n;
  m->l2g.resize(n_cells * nl);
  const uint64_t zi0 = z0 + (r > 0 ? 1 : 0); // interior layers [zi0, z1)
  m->n_interior = (uint32_t)((z1 - zi0) * n0 * n1);
  const bool blocked = d->cell_block[0] && d->cell_block[1] && d->cell_block[2];
  const uint64_t bx = blocked ? d->cell_block[0] : n0, by = blocked ? d->cell_block[1] : 1, bz = blocked ? d->cell_block[2] : 1;
  size_t c = 0;
  m->block_off.push_back(0);
  const bool class_major = blocked && d->cell_block_order == 1;
  auto emit_region = [&](uint64_t za, uint64_t zb) {
    for (uint64_t Z = za; Z < zb; Z += bz)
      for (uint64_t Y = 0; Y < n1; Y += by)
        for (uint64_t X = 0; X < n0; X += bx) {
          // inside a brick: lexicographic, or parity class by parity class so that the cells of one conflict-free
          // pass of the block kernel are consecutive in memory
          const int n_cls = class_major ? 8 : 1;
          for (int cls = 0; cls < n_cls; ++cls)
            for (uint64_t z = Z; z < std::min(Z + bz, zb); ++z)
              for (uint64_t y = Y; y < std::min(Y + by, n1); ++y)
                for (uint64_t x = X; x < std::min(X + bx, n0); ++x) {
                  if (class_major && (int)(((x - X) & 1) | (((y - Y) & 1) << 1) | (((z - Z) & 1) << 2)) != cls) continue;
                  uint32_t *dst = &m->l2g[c * nl];
                  for (int k = 0; k < n; ++k)
                    for (int j = 0; j < n; ++j)
                      for (int i = 0; i < n; ++i) dst[i + n * (j + n * k)] = local_of(p * x + i, p * y + j, p * z + k);
                  ++c;
                }
          if (blocked) m->block_off.push_back((uint32_t)c);
        }
  };
  emit_region(zi0, z1);
  if (r > 0) emit_region(z0, z0 + 1);
  if (!blocked) m->block_off.clear();

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
  if (r < R - 1) { // neighbour r+1: we send our top plane (in the order of ITS ghost range), receive nothing
    m->neighbors.push_back(r + 1);
    std::vector<uint32_t> by_pos(plane);
    for (uint64_t J = 0; J < NY; ++J)
      for (uint64_t I = 0; I < NX; ++I) by_pos[plane_pos(I, J)] = local_of(I, J, Ktop);
    m->send_indices.insert(m->send_indices.end(), by_pos.begin(), by_pos.end());
    m->send_offsets.push_back((uint32_t)m->send_indices.size());
    m->recv_offsets.push_back(m->recv_offsets.back());
  }
  // dof_numbering = 2: the owned DoFs strictly inside a cell first, cell after cell in the order the cells are handed over ((p - 1)^3 consecutive
  // DoFs per cell, x fastest), then all other owned DoFs in the order of numbering 0 (or 1 with cell_block); ghosts keep their places.  A renumbering
  // of the finished mesh: every array that names a local DoF is mapped through it.
  if (d->dof_numbering == 2 && p >= 2) {
    std::vector<uint32_t> newid(n_owned, 0xffffffffu);
    uint32_t next = 0;
    for (size_t cc = 0; cc < n_cells; ++cc)
      for (int k = 1; k < p; ++k)
        for (int j = 1; j < p; ++j)
          for (int i = 1; i < p; ++i) {
            const uint32_t old = m->l2g[cc * nl + i + n * (j + n * k)];
            if (old < n_owned) newid[old] = next++; // (always: only the bottom PLANE of a rank's cells is ghost)
          }
    for (uint64_t o = 0; o < n_owned; ++o)
      if (newid[o] == 0xffffffffu) newid[o] = next++;
    auto map = [&](uint32_t o) -> uint32_t { return o < n_owned ? newid[o] : o; };
    for (uint32_t &v : m->l2g) v = map(v);
    for (uint32_t &v : m->constrained) v = map(v);
    std::sort(m->constrained.begin(), m->constrained.end());
    for (uint32_t &v : m->send_indices) v = map(v);
    std::vector<double> co(m->coords.size());
    std::vector<uint64_t> gi(m->gids.size());
    for (uint64_t o = 0; o < nloc; ++o) {
      const uint32_t q = map((uint32_t)o);
      for (int e = 0; e < 3; ++e) co[3 * (size_t)q + e] = m->coords[3 * (size_t)o + e];
      gi[q] = m->gids[o];
    }
    m->coords.swap(co);
    m->gids.swap(gi);
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
  v->n_cell_blocks = m->block_off.empty() ? 0 : (uint32_t)m->block_off.size() - 1;
  v->cell_block_offsets_host = m->block_off.empty() ? nullptr : m->block_off.data();
  return BP5_OK;
}

extern "C" void bp5_mesh_destroy(bp5_mesh *m) { delete m; }
