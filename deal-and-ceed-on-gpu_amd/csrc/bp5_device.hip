// C-ABI implementation: MatrixFree handle, operator launches, BLAS-1, CG drivers, RCCL halo.
// Every entry point cites the reference interface it replaces in include/bp5.h.
#include "bp5_internal.hpp"
#include "bp5_kernels.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

using namespace bp5;

#define HIP_TRY(expr)                                                                                              \
  do {                                                                                                             \
    hipError_t e_ = (expr);                                                                                        \
    if (e_ != hipSuccess)                                                                                          \
      return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? BP5_ERR_NO_DEVICE : BP5_ERR_HIP,         \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                              \
  } while (0)
#define NCCL_TRY(expr)                                                                                             \
  do {                                                                                                             \
    ncclResult_t r_ = (expr);                                                                                      \
    if (r_ != ncclSuccess) return fail(BP5_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(r_));         \
  } while (0)
#define BP5_TRY(expr)                                                                                              \
  do {                                                                                                             \
    int s_ = (expr);                                                                                               \
    if (s_ != BP5_OK) return s_;                                                                                   \
  } while (0)
#define KERNEL_CHECK() HIP_TRY(hipGetLastError())

struct bp5_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
};
struct bp5_event {
  hipEvent_t ev = nullptr;
};

struct bp5_mf {
  int degree = 0, quadrature = 0, coefficient = 0, n = 0, n3 = 0, device = 0;
  uint32_t n_cells = 0, n_interior = 0, n_owned = 0, n_ghost = 0, n_constrained = 0;
  int apply_variant = 0, n_cus = 0, geometry_mode = 0, march_max_steps = 32;
  uint32_t blk_b0 = 0, blk_b1 = 0; // block range of the next block-kernel launch (0,0 = all blocks)
  bool combine_csr = false; // A/B: per-DoF CSR combine kernel instead of the run-length one
  int block_max_wg = 0; // 0: persistent grid sized from the CU count; > 0: cap (tests force several blocks per workgroup)
  int auto_block = -1; // -1 not decided; 1: the caller's cell blocks fit three block-kernel workgroups per CU
  double *d_scalar_plane = nullptr, *d_gcell = nullptr;
  bool force_atomic_scatter = false, block_shared_atomic = false;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Tables tab, tab_gauss;
  // device arrays
  uint32_t *d_l2g = nullptr, *d_constrained = nullptr, *d_send_idx = nullptr;
  double *d_coords = nullptr, *d_tab = nullptr, *d_tab_gauss = nullptr;
  // Data mirror (lazy)
  uint32_t *d_l2g_padded = nullptr, *d_constraint_mask = nullptr;
  double *d_inv_jac = nullptr, *d_JxW = nullptr, *d_qpoints = nullptr;
  uint32_t pad = 0;
  // halo plan
  std::vector<int> neighbors;
  std::vector<uint32_t> send_off, recv_off;
  double *d_sendbuf = nullptr, *d_recvbuf = nullptr;
  bp5_comm *comm = nullptr;
  // solver workspace
  double *d_partials = nullptr, *d_sc = nullptr, *d_scalar = nullptr;
  int *d_st = nullptr;
  double *ws_g = nullptr, *ws_d = nullptr, *ws_h = nullptr, *d_evec = nullptr;
  char *ws_base = nullptr;
  unsigned long long *d_stamps = nullptr;
  double *h_sc = nullptr; // pinned
  int *h_st = nullptr;    // pinned
  std::vector<hipEvent_t> ev_pool;
  hipEvent_t prof_mark = nullptr; // profiling: recorded once before the combine pass (= end of the dominant kernel)
  // team plans of the team-assembled kernel, keyed by cells per team
  std::vector<uint32_t> h_l2g;
  struct DevPlan {
    uint32_t *off = nullptr, *dofs = nullptr, *sh_dof = nullptr, *sh_off = nullptr, *sh_slot = nullptr;
    uint16_t *pos = nullptr;
    uint8_t *cell_round = nullptr, *team_rounds = nullptr;
    double *partial = nullptr;
    uint32_t *cell_off = nullptr, *pass_cell = nullptr, *pass_off = nullptr, *run_off = nullptr, *runs = nullptr, *gidx = nullptr;
    uint16_t *packed = nullptr;
    std::vector<double> h_cost;                       // [n_groups+1] prefix sum of the estimated cost of the blocks (pass units)
    uint32_t *wg_block = nullptr;                     // cached ranges for (wg_n, wg_b0, wg_b1)
    uint32_t wg_n = 0, wg_b0 = 0, wg_b1 = 0;
    uint32_t *cr_start = nullptr, *cr_dof0 = nullptr, *cr_soff = nullptr, *cr_slots = nullptr, *cr_tile = nullptr; // run-length combine
    uint32_t n_shared = 0, n_groups = 0, max_list = 0, max_runs = 0;
    bool covers_all = false;
  };
  std::vector<uint32_t> h_block_off; // caller-provided cell blocks (may be empty)
  struct DevMarch { uint32_t *team_off = nullptr, *entries = nullptr; uint32_t n_teams = 0; };
  std::map<int, DevMarch> march_plans; // keyed by cells per team
  std::map<int, DevPlan> plans;
  size_t n_local() const { return (size_t)n_owned + n_ghost; }
};

// ------------------------------------------------------------------------------------ device / vectors
extern "C" int bp5_device_count(int *count)
{
  if (!count) return fail(BP5_ERR_INVALID, "null argument");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *count = 0; return fail(BP5_ERR_NO_DEVICE, hipGetErrorString(e)); }
  *count = c;
  return BP5_OK;
}
extern "C" int bp5_vec_alloc(size_t n, double **out)
{
  if (!out) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipMalloc((void **)out, std::max<size_t>(n, 1) * sizeof(double)));
  HIP_TRY(hipMemset(*out, 0, std::max<size_t>(n, 1) * sizeof(double)));
  return BP5_OK;
}
extern "C" int bp5_vec_free(double *v) { HIP_TRY(hipFree(v)); return BP5_OK; }
extern "C" int bp5_copy_h2d(void *dst, const void *src, size_t bytes) { HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return BP5_OK; }
extern "C" int bp5_copy_d2h(void *dst, const void *src, size_t bytes) { HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return BP5_OK; }

// ------------------------------------------------------------------------------------ create / destroy
template <typename T>
static int upload(T **dptr, const T *host, size_t count)
{
  HIP_TRY(hipMalloc((void **)dptr, std::max<size_t>(count, 1) * sizeof(T)));
  if (count) HIP_TRY(hipMemcpy(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
  return BP5_OK;
}
static void pack_tab(const Tables &t, std::vector<double> &v)
{
  const int n = t.n;
  v.assign(2 * n * n + n, 0.0);
  memcpy(v.data(), t.N, n * n * sizeof(double));
  memcpy(v.data() + n * n, t.D, n * n * sizeof(double));
  memcpy(v.data() + 2 * n * n, t.w, n * sizeof(double));
}

extern "C" int bp5_mf_create(const bp5_mf_desc *d, bp5_mf **out)
{
  if (!d || !out) return fail(BP5_ERR_INVALID, "null argument");
  if (d->dim != 3) return fail(BP5_ERR_UNSUPPORTED, "only dim == 3");
  if (!d->local_to_global_host || !d->node_coords_host) return fail(BP5_ERR_INVALID, "mesh arrays missing");
  if (d->n_interior_cells > d->n_cells) return fail(BP5_ERR_INVALID, "n_interior_cells > n_cells");
  Tables tab, tabg;
  BP5_TRY(shape_tables(d->degree, d->quadrature, tab));
  BP5_TRY(shape_tables(d->degree, BP5_QUAD_GAUSS, tabg));
  int ndev = 0;
  BP5_TRY(bp5_device_count(&ndev));
  if (ndev <= 0) return fail(BP5_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
  if (d->device < 0 || d->device >= ndev) return fail(BP5_ERR_INVALID, "bad device ordinal");
  HIP_TRY(hipSetDevice(d->device));
  bp5_mf *mf = new bp5_mf;
  mf->degree = d->degree; mf->quadrature = d->quadrature; mf->coefficient = d->coefficient;
  mf->n = d->degree + 1; mf->n3 = mf->n * mf->n * mf->n; mf->device = d->device;
  mf->n_cells = d->n_cells; mf->n_interior = d->n_interior_cells; mf->n_owned = d->n_owned; mf->n_ghost = d->n_ghost;
  mf->n_constrained = d->n_constrained;
  mf->tab = tab; mf->tab_gauss = tabg;
  // validate indices on the host: a bad index would fault on the GPU
  const size_t nl = (size_t)d->n_cells * mf->n3, nloc = mf->n_local();
  for (size_t s = 0; s < nl; ++s)
    if (d->local_to_global_host[s] >= nloc) { delete mf; return fail(BP5_ERR_INVALID, "local_to_global entry out of range"); }
  for (uint32_t s = 0; s < d->n_constrained; ++s)
    if (d->constrained_host[s] >= nloc) { delete mf; return fail(BP5_ERR_INVALID, "constrained index out of range"); }
  mf->stream = (hipStream_t)d->stream; // NULL == the HIP default stream (ordered with the host's other default-stream work)
  BP5_TRY(upload(&mf->d_l2g, d->local_to_global_host, nl));
  mf->h_l2g.assign(d->local_to_global_host, d->local_to_global_host + nl);
  if (d->n_cell_blocks && d->cell_block_offsets_host) {
    const uint32_t *o = d->cell_block_offsets_host;
    bool ok = o[0] == 0 && o[d->n_cell_blocks] == d->n_cells;
    for (uint32_t b = 0; ok && b < d->n_cell_blocks; ++b) ok = o[b] < o[b + 1];
    if (!ok) { delete mf; return fail(BP5_ERR_INVALID, "cell_block_offsets must ascend from 0 to n_cells"); }
    mf->h_block_off.assign(o, o + d->n_cell_blocks + 1);
  }
  BP5_TRY(upload(&mf->d_coords, d->node_coords_host, nloc * 3));
  BP5_TRY(upload(&mf->d_constrained, d->constrained_host, d->n_constrained));
  std::vector<double> tv;
  pack_tab(tab, tv);  BP5_TRY(upload(&mf->d_tab, tv.data(), tv.size()));
  pack_tab(tabg, tv); BP5_TRY(upload(&mf->d_tab_gauss, tv.data(), tv.size()));
  // halo plan
  if (d->n_neighbors > 0) {
    if (!d->neighbor_rank_host || !d->send_offsets_host || !d->recv_offsets_host) { delete mf; return fail(BP5_ERR_INVALID, "halo plan arrays missing"); }
    mf->neighbors.assign(d->neighbor_rank_host, d->neighbor_rank_host + d->n_neighbors);
    mf->send_off.assign(d->send_offsets_host, d->send_offsets_host + d->n_neighbors + 1);
    mf->recv_off.assign(d->recv_offsets_host, d->recv_offsets_host + d->n_neighbors + 1);
    const uint32_t ns = mf->send_off.back();
    if (mf->recv_off.back() != d->n_ghost) { delete mf; return fail(BP5_ERR_INVALID, "recv ranges must cover the ghost range"); }
    for (uint32_t s = 0; s < ns; ++s)
      if (d->send_indices_host[s] >= d->n_owned) { delete mf; return fail(BP5_ERR_INVALID, "send index out of owned range"); }
    BP5_TRY(upload(&mf->d_send_idx, d->send_indices_host, ns));
    HIP_TRY(hipMalloc((void **)&mf->d_sendbuf, std::max<size_t>(ns, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&mf->d_recvbuf, std::max<size_t>(ns, 1) * sizeof(double)));
  } else if (d->n_ghost) { delete mf; return fail(BP5_ERR_INVALID, "ghosts without a halo plan"); }
  // solver workspace
  HIP_TRY(hipMalloc((void **)&mf->d_partials, 8 * MAXBLK * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&mf->d_sc, SC_COUNT * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&mf->d_scalar, 8 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&mf->d_st, ST_COUNT * sizeof(int)));
  HIP_TRY(hipMemset(mf->d_sc, 0, SC_COUNT * sizeof(double)));
  HIP_TRY(hipMemset(mf->d_st, 0, ST_COUNT * sizeof(int)));
  HIP_TRY(hipHostMalloc((void **)&mf->h_sc, SC_COUNT * sizeof(double)));
  HIP_TRY(hipHostMalloc((void **)&mf->h_st, ST_COUNT * sizeof(int)));
  *out = mf;
  return BP5_OK;
}

extern "C" int bp5_mf_destroy(bp5_mf *mf)
{
  if (!mf) return BP5_OK;
  hipSetDevice(mf->device);
  hipStreamSynchronize(mf->stream);
  void *ptrs[] = {mf->d_l2g, mf->d_constrained, mf->d_send_idx, mf->d_coords, mf->d_tab, mf->d_tab_gauss, mf->d_l2g_padded,
                  mf->d_constraint_mask, mf->d_inv_jac, mf->d_JxW, mf->d_qpoints, mf->d_sendbuf, mf->d_recvbuf, mf->d_partials,
                  mf->d_sc, mf->d_scalar, mf->d_st, mf->ws_base, mf->d_stamps, mf->d_evec, mf->d_scalar_plane, mf->d_gcell};
  for (void *p : ptrs) if (p) hipFree(p);
  if (mf->h_sc) hipHostFree(mf->h_sc);
  if (mf->h_st) hipHostFree(mf->h_st);
  for (hipEvent_t e : mf->ev_pool) hipEventDestroy(e);
  for (auto &kv : mf->march_plans) { hipFree(kv.second.team_off); hipFree(kv.second.entries); }
  for (auto &kv : mf->plans) {
    auto &q = kv.second;
    void *pp[] = {q.off, q.dofs, q.sh_dof, q.sh_off, q.sh_slot, q.pos, q.cell_round, q.team_rounds, q.partial, q.cell_off, q.pass_cell, q.pass_off, q.run_off, q.runs, q.gidx, q.packed, q.cr_start, q.cr_dof0, q.cr_soff, q.cr_slots, q.cr_tile, q.wg_block};
    for (void *x : pp) if (x) hipFree(x);
  }
  if (mf->own_stream) hipStreamDestroy(mf->stream);
  delete mf;
  return BP5_OK;
}
extern "C" int bp5_mf_set_stream(bp5_mf *mf, void *s)
{
  if (!mf) return fail(BP5_ERR_INVALID, "null handle");
  if (mf->own_stream) { hipStreamSynchronize(mf->stream); hipStreamDestroy(mf->stream); mf->own_stream = false; }
  mf->stream = (hipStream_t)s;
  return BP5_OK;
}
extern "C" int bp5_mf_sync(bp5_mf *mf)
{
  if (!mf) return fail(BP5_ERR_INVALID, "null handle");
  HIP_TRY(hipStreamSynchronize(mf->stream));
  return BP5_OK;
}
extern "C" int bp5_mf_coef_size(const bp5_mf *mf, size_t *n)
{
  if (!mf || !n) return fail(BP5_ERR_INVALID, "null argument");
  *n = (size_t)6 * mf->n_cells * mf->n3;
  return BP5_OK;
}
extern "C" int bp5_mf_set_apply_variant(bp5_mf *mf, int v)
{
  if (!mf) return fail(BP5_ERR_INVALID, "null handle");
  mf->apply_variant = v;
  return BP5_OK;
}

static int effective_variant(bp5_mf *mf, uint32_t c0, uint32_t c1);
extern "C" int bp5_mf_set_block_workgroups(bp5_mf *mf, int max_workgroups)
{
  if (!mf || max_workgroups < 0) return fail(BP5_ERR_INVALID, "bad argument");
  mf->block_max_wg = max_workgroups;
  return BP5_OK;
}
static int get_plan_raw(bp5_mf *mf, int key, bp5_mf::DevPlan **dpo, int default_block = 64);
extern "C" int bp5_mf_block_plan_info(bp5_mf *mf, uint32_t *n_blocks, uint32_t *max_runs, int *packed_indices)
{
  if (!mf || !n_blocks || !max_runs || !packed_indices) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan_raw(mf, -8, &dp, 64));
  *n_blocks = dp->n_groups;
  *max_runs = dp->max_runs;
  *packed_indices = dp->packed != nullptr;
  return BP5_OK;
}
extern "C" int bp5_mf_get_apply_variant(bp5_mf *mf, int *effective)
{
  if (!mf || !effective) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  *effective = effective_variant(mf, 0, mf->n_cells);
  return BP5_OK;
}

// ------------------------------------------------------------------------------------ geometry
template <int n>
static int launch_geometry(bp5_mf *mf, GeomOut o)
{
  const uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(mf->n_cells, 1), 65535u * 16);
  hipLaunchKernelGGL(geometry_kernel<n>, dim3(grid), dim3(n, n, n), 0, mf->stream, mf->d_l2g, mf->d_coords, mf->d_tab, mf->coefficient,
                     mf->n_cells, o);
  KERNEL_CHECK();
  return BP5_OK;
}
#define DISPATCH_N(fn, ...)                                                                                        \
  switch (mf->n) {                                                                                                 \
    case 2: return fn<2>(__VA_ARGS__);                                                                             \
    case 3: return fn<3>(__VA_ARGS__);                                                                             \
    case 4: return fn<4>(__VA_ARGS__);                                                                             \
    case 5: return fn<5>(__VA_ARGS__);                                                                             \
    case 6: return fn<6>(__VA_ARGS__);                                                                             \
    case 7: return fn<7>(__VA_ARGS__);                                                                             \
    case 8: return fn<8>(__VA_ARGS__);                                                                             \
    case 9: return fn<9>(__VA_ARGS__);                                                                             \
  }                                                                                                                \
  return fail(BP5_ERR_INVALID, "unsupported degree")

static int geometry_affine(bp5_mf *mf, GeomOut o) { DISPATCH_N(launch_geometry, mf, o); }
extern "C" int bp5_mf_set_geometry_mode(bp5_mf *mf, int mode)
{
  if (!mf) return fail(BP5_ERR_INVALID, "null handle");
  if (mode != BP5_GEOM_MERGED6 && mode != BP5_GEOM_AFFINE) return fail(BP5_ERR_INVALID, "unknown geometry mode");
  HIP_TRY(hipSetDevice(mf->device));
  if (mode == BP5_GEOM_AFFINE && !mf->d_scalar_plane) {
    double *sp = nullptr, *gc = nullptr, *dev = nullptr;
    const size_t nq = (size_t)mf->n_cells * mf->n3;
    HIP_TRY(hipMalloc((void **)&sp, std::max<size_t>(nq, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&gc, std::max<size_t>(6 * (size_t)mf->n_cells, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&dev, std::max<size_t>(mf->n_cells, 1) * sizeof(double)));
    GeomOut o{};
    o.scalar = sp; o.gcell = gc; o.deviation = dev; o.n_cells = mf->n_cells;
    int st = geometry_affine(mf, o);
    std::vector<double> h(mf->n_cells);
    if (st == BP5_OK && hipMemcpyAsync(h.data(), dev, mf->n_cells * sizeof(double), hipMemcpyDeviceToHost, mf->stream) != hipSuccess) st = BP5_ERR_HIP;
    if (st == BP5_OK && hipStreamSynchronize(mf->stream) != hipSuccess) st = BP5_ERR_HIP;
    hipFree(dev);
    double worst = 0.0;
    for (double v : h) worst = std::max(worst, v);
    if (st != BP5_OK || !(worst <= 1e-10)) { // rounding noise of the Jacobian is ~eps*|x|/h; real curvature is orders larger
      hipFree(sp); hipFree(gc);
      if (st != BP5_OK) return fail(st, "affine geometry setup failed");
      return fail(BP5_ERR_UNSUPPORTED, "mesh is not affine (K K^T varies inside a cell): use BP5_GEOM_MERGED6");
    }
    mf->d_scalar_plane = sp; mf->d_gcell = gc;
  }
  mf->geometry_mode = mode;
  return BP5_OK;
}

extern "C" int bp5_mf_compute_merged_metric(bp5_mf *mf, double *coef)
{
  if (!mf || !coef) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  GeomOut o{};
  o.coef = coef;
  o.plane_stride = (uint64_t)mf->n_cells * mf->n3;
  DISPATCH_N(launch_geometry, mf, o);
}

template <int n>
static int launch_permute(bp5_mf *mf, const double *in, double *out)
{
  const uint64_t total = (uint64_t)6 * mf->n_cells * mf->n3;
  hipLaunchKernelGGL(metric_permute_kernel<n>, dim3(2048), dim3(256), 0, mf->stream, in, out, total);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_mf_metric_to_reference_layout(bp5_mf *mf, const double *coef, double *coef_ref)
{
  if (!mf || !coef || !coef_ref) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  DISPATCH_N(launch_permute, mf, coef, coef_ref);
}

static uint32_t padding_length(int n)
{ // deal.II: 2^ceil(dim*log2(n)) [upstream], SURVEY 8(a3)
  uint32_t p = 1;
  while (p < (uint32_t)(n * n * n)) p <<= 1;
  return p;
}
static int geometry_data(bp5_mf *mf, GeomOut o) { DISPATCH_N(launch_geometry, mf, o); }

extern "C" int bp5_mf_get_data(bp5_mf *mf, int color, bp5_mf_data *out)
{
  if (!mf || !out) return fail(BP5_ERR_INVALID, "null argument");
  if (color != 0) return fail(BP5_ERR_INVALID, "this build keeps all cells in one colour");
  HIP_TRY(hipSetDevice(mf->device));
  if (!mf->d_inv_jac) {
    mf->pad = padding_length(mf->n);
    const size_t gp = (size_t)mf->n_cells * mf->pad;
    HIP_TRY(hipMalloc((void **)&mf->d_inv_jac, std::max<size_t>(9 * gp, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&mf->d_JxW, std::max<size_t>(gp, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&mf->d_qpoints, std::max<size_t>(3 * gp, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&mf->d_l2g_padded, std::max<size_t>(gp, 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&mf->d_constraint_mask, std::max<size_t>(mf->n_cells, 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(mf->d_inv_jac, 0, 9 * gp * sizeof(double), mf->stream));
    HIP_TRY(hipMemsetAsync(mf->d_JxW, 0, gp * sizeof(double), mf->stream));
    HIP_TRY(hipMemsetAsync(mf->d_qpoints, 0, 3 * gp * sizeof(double), mf->stream));
    HIP_TRY(hipMemsetAsync(mf->d_l2g_padded, 0, gp * sizeof(uint32_t), mf->stream));
    HIP_TRY(hipMemsetAsync(mf->d_constraint_mask, 0, mf->n_cells * sizeof(uint32_t), mf->stream));
    if (mf->n_cells)
      HIP_TRY(hipMemcpy2DAsync(mf->d_l2g_padded, mf->pad * sizeof(uint32_t), mf->d_l2g, mf->n3 * sizeof(uint32_t),
                               mf->n3 * sizeof(uint32_t), mf->n_cells, hipMemcpyDeviceToDevice, mf->stream));
    GeomOut o{};
    o.inv_jac = mf->d_inv_jac; o.JxW = mf->d_JxW; o.q_points = mf->d_qpoints; o.pad = mf->pad; o.geo_plane = gp;
    BP5_TRY(geometry_data(mf, o));
    HIP_TRY(hipStreamSynchronize(mf->stream));
  }
  out->local_to_global = mf->d_l2g_padded; out->inv_jacobian = mf->d_inv_jac; out->JxW = mf->d_JxW; out->q_points = mf->d_qpoints;
  out->constraint_mask = mf->d_constraint_mask; out->n_cells = mf->n_cells; out->padding_length = mf->pad; out->row_start = 0;
  out->use_coloring = 0;
  return BP5_OK;
}

// ------------------------------------------------------------------------------------ operator
template <int P, bool COLL, int TW, int LPC, int TPB, bool PF, int ABL = 0>
static int launch_apply_t(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1)
{
  constexpr int n = P + 1;
  constexpr int CPT = 64 * TW / LPC;
  using L = LdsLayout<n, LPC>;
  ApplyArgs a{};
  a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst;
  a.plane_stride = (uint64_t)mf->n_cells * mf->n3;
  a.cell_begin = c0; a.cell_end = c1;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  a.n_teams = (c1 - c0 + CPT - 1) / CPT;
  const uint32_t nblk = (a.n_teams + TPB - 1) / TPB;
  a.teams_per_xcd = (nblk + 7) / 8;
  ShapeArg<n> sh;
  memcpy(sh.N, mf->tab.N, sizeof(sh.N));
  memcpy(sh.D, mf->tab.D, sizeof(sh.D));
  const size_t lds = (size_t)TPB * CPT * L::CS * sizeof(double);
  hipLaunchKernelGGL((apply_pencil_kernel<P, COLL, TW, LPC, TPB, PF, ABL>), dim3(a.teams_per_xcd * 8), dim3(64 * TW * TPB), lds, mf->stream, a,
                     sh);
  KERNEL_CHECK();
  return BP5_OK;
}

// key > 0: uniform teams of `key` cells (team kernel); key < 0: cell blocks walked in passes of
// -key cells (block kernel) -- the caller's blocks if given, else groups of `default_block` cells
static int get_plan_raw(bp5_mf *mf, int key, bp5_mf::DevPlan **dpo, int default_block)
{
  auto it = mf->plans.find(key);
  if (it == mf->plans.end()) {
    TeamPlanHost h;
    if (key > 0) BP5_TRY(build_team_plan(mf->h_l2g.data(), mf->n_cells, mf->n3, mf->n_local(), key, h));
    else if (!mf->h_block_off.empty())
      BP5_TRY(build_team_plan(mf->h_l2g.data(), mf->n_cells, mf->n3, mf->n_local(), 0, h, mf->h_block_off.data(),
                              (uint32_t)mf->h_block_off.size() - 1, -key));
    else BP5_TRY(build_team_plan(mf->h_l2g.data(), mf->n_cells, mf->n3, mf->n_local(), default_block, h, nullptr, 0, -key));
    bp5_mf::DevPlan dp;
    BP5_TRY(upload(&dp.off, h.off.data(), h.off.size()));
    BP5_TRY(upload(&dp.dofs, h.dofs.data(), h.dofs.size()));
    std::vector<uint32_t> run_off, runs;
    if (key < 0) {
      // run-length form of the sorted block lists: consecutive DoFs with equal ownership flag, cut at 1024 entries so
      // that (run, offset) packs into 6 + 10 bits
      run_off.assign(h.off.size(), 0);
      for (size_t g = 0; g + 1 < h.off.size(); ++g) {
        uint32_t start = h.off[g];
        for (uint32_t i = h.off[g]; i < h.off[g + 1]; ++i)
          if (i == h.off[g] || h.dofs[i] != h.dofs[i - 1] + 1 || i - start == (1u << BLOCK_PACK_OFF_BITS)) {
            start = i;
            runs.push_back(i - h.off[g]);
            runs.push_back(h.dofs[i]);
          }
        run_off[g + 1] = (uint32_t)(runs.size() / 2);
        dp.max_runs = std::max(dp.max_runs, run_off[g + 1] - run_off[g]);
      }
      // block kernel: per-cell index arrays in the pair layout of the z-pencils (two entries per load, see coef_off)
      const int n = mf->degree + 1, n2 = n * n;
      auto off = [&](int k, int ab) { return k < 2 * (n / 2) ? (k / 2) * (2 * n2) + 2 * ab + (k & 1) : (n / 2) * (2 * n2) + ab; };
      std::vector<uint16_t> pos2(h.pos.size()), packed(dp.max_runs <= (uint32_t)BLOCK_PACK_MAX_RUNS ? h.pos.size() : 0);
      std::vector<uint32_t> gidx(h.pos.size());
#pragma omp parallel
      {
        std::vector<uint16_t> run_of_slot;
#pragma omp for schedule(dynamic, 16)
        for (int64_t g = 0; g < (int64_t)h.group_cell_off.size() - 1; ++g) {
          const uint32_t nr = run_off[g + 1] - run_off[g], m = h.off[g + 1] - h.off[g];
          if (!packed.empty()) {
            run_of_slot.assign(m, 0);
            for (uint32_t r = 0; r < nr; ++r) {
              const uint32_t s0 = runs[2 * (run_off[g] + r)], s1 = r + 1 < nr ? runs[2 * (run_off[g] + r + 1)] : m;
              for (uint32_t sl = s0; sl < s1; ++sl) run_of_slot[sl] = (uint16_t)r;
            }
          }
          for (size_t c = h.group_cell_off[g]; c < h.group_cell_off[g + 1]; ++c)
            for (int k = 0; k < n; ++k)
              for (int ab = 0; ab < n2; ++ab) {
                const size_t from = c * mf->n3 + (size_t)k * n2 + ab, to = c * mf->n3 + off(k, ab);
                pos2[to] = h.pos[from];
                gidx[to] = mf->h_l2g[from];
                if (!packed.empty()) {
                  const uint32_t sl = h.pos[from], r = run_of_slot[sl];
                  packed[to] = (uint16_t)(r << BLOCK_PACK_OFF_BITS | (sl - runs[2 * (run_off[g] + r)]));
                }
              }
        }
      }
      BP5_TRY(upload(&dp.pos, pos2.data(), pos2.size()));
      BP5_TRY(upload(&dp.gidx, gidx.data(), gidx.size()));
      if (!packed.empty()) BP5_TRY(upload(&dp.packed, packed.data(), packed.size()));
    } else
      BP5_TRY(upload(&dp.pos, h.pos.data(), h.pos.size()));
    BP5_TRY(upload(&dp.cell_round, h.cell_round.data(), h.cell_round.size()));
    BP5_TRY(upload(&dp.team_rounds, h.team_rounds.data(), h.team_rounds.size()));
    BP5_TRY(upload(&dp.sh_dof, h.sh_dof.data(), h.sh_dof.size()));
    BP5_TRY(upload(&dp.sh_off, h.sh_off.data(), h.sh_off.size()));
    BP5_TRY(upload(&dp.sh_slot, h.sh_slot.data(), h.sh_slot.size()));
    HIP_TRY(hipMalloc((void **)&dp.partial, std::max<size_t>(h.dofs.size(), 1) * sizeof(double)));
    HIP_TRY(hipMemset(dp.partial, 0, std::max<size_t>(h.dofs.size(), 1) * sizeof(double)));
    BP5_TRY(upload(&dp.cell_off, h.group_cell_off.data(), h.group_cell_off.size()));
    if (key < 0) {
      BP5_TRY(upload(&dp.pass_cell, h.pass_cell.data(), h.pass_cell.size()));
      BP5_TRY(upload(&dp.pass_off, h.pass_off.data(), h.pass_off.size()));
      // cost model of a block in units of one pass, calibrated on the slab mesh of a rank > 0 (profiles/r1 k_*: thin
      // boundary bricks next to full ones): every accumulation round after the first +0.18, the write-out 1.5 per 4913
      // list slots, ONE pass-equivalent fixed per block (barrier, block switch, table hand-over)
      dp.h_cost.assign(h.pass_off.size(), 0.0);
      for (size_t g = 0; g + 1 < h.pass_off.size(); ++g) {
        const double passes = h.pass_off[g + 1] - h.pass_off[g], rounds = h.team_rounds[g], m = h.off[g + 1] - h.off[g];
        dp.h_cost[g + 1] = dp.h_cost[g] + passes * (1.0 + 0.18 * (rounds - 1.0)) + 1.5 * m / 4913.0 + 1.0;
      }
      runs.push_back(0); runs.push_back(0);
      BP5_TRY(upload(&dp.run_off, run_off.data(), run_off.size()));
      BP5_TRY(upload(&dp.runs, runs.data(), runs.size()));
    }
    dp.n_shared = (uint32_t)h.sh_dof.size();
    if (dp.n_shared) { // run-length form of the shared-DoF CSR for combine_runs_kernel
      std::vector<uint32_t> start, dof0, soff, slots, tile;
      const size_t ns = h.sh_dof.size();
      for (size_t i = 0; i < ns; ++i) {
        const uint32_t b = h.sh_off[i], e = h.sh_off[i + 1];
        bool cont = i > 0 && h.sh_dof[i] == h.sh_dof[i - 1] + 1 && (e - b) == (h.sh_off[i] - h.sh_off[i - 1]);
        for (uint32_t q = 0; cont && q < e - b; ++q) cont = h.sh_slot[b + q] == h.sh_slot[h.sh_off[i - 1] + q] + 1;
        if (!cont) {
          start.push_back((uint32_t)i);
          dof0.push_back(h.sh_dof[i]);
          soff.push_back((uint32_t)slots.size());
          slots.insert(slots.end(), h.sh_slot.begin() + b, h.sh_slot.begin() + e);
        }
      }
      start.push_back((uint32_t)ns);
      soff.push_back((uint32_t)slots.size());
      const size_t n_tiles = (ns + 255) / 256;
      tile.resize(n_tiles + 1);
      size_t r = 0;
      for (size_t t = 0; t <= n_tiles; ++t) { // run containing ordinal min(256 t, ns - 1)
        const size_t i = std::min(t * 256, ns - 1);
        while (start[r + 1] <= i) ++r;
        tile[t] = (uint32_t)r;
      }
      start.push_back((uint32_t)ns); // one entry of slack for the staging loop (reads r_hi + 1)
      soff.push_back((uint32_t)slots.size());
      BP5_TRY(upload(&dp.cr_start, start.data(), start.size()));
      BP5_TRY(upload(&dp.cr_dof0, dof0.data(), dof0.size()));
      BP5_TRY(upload(&dp.cr_soff, soff.data(), soff.size()));
      BP5_TRY(upload(&dp.cr_slots, slots.data(), slots.size()));
      BP5_TRY(upload(&dp.cr_tile, tile.data(), tile.size()));
    }
    dp.covers_all = h.covers_all;
    dp.n_groups = (uint32_t)h.group_cell_off.size() - 1;
    for (uint32_t g = 0; g < dp.n_groups; ++g) dp.max_list = std::max(dp.max_list, h.off[g + 1] - h.off[g]);
    it = mf->plans.emplace(key, dp).first;
  }
  *dpo = &it->second;
  return BP5_OK;
}
static int get_plan(bp5_mf *mf, int cpt, TeamPlan &tp, bp5_mf::DevPlan **dpo)
{
  bp5_mf::DevPlan *q = nullptr;
  BP5_TRY(get_plan_raw(mf, cpt, &q));
  tp.off = q->off; tp.dofs = q->dofs; tp.pos = q->pos; tp.cell_round = q->cell_round; tp.team_rounds = q->team_rounds; tp.partial = q->partial;
  if (dpo) *dpo = q;
  return BP5_OK;
}

static int launch_combine(bp5_mf *mf, bp5_mf::DevPlan *dp, double *dst, bool set)
{
  if (!dp->n_shared) return BP5_OK;
  if (mf->prof_mark) { HIP_TRY(hipEventRecord(mf->prof_mark, mf->stream)); mf->prof_mark = nullptr; }
  const dim3 cg((dp->n_shared + 255) / 256);
  if (dp->cr_tile && !mf->combine_csr) {
    const CombineRuns cr{dp->cr_start, dp->cr_dof0, dp->cr_soff, dp->cr_slots, dp->cr_tile, dp->n_shared};
    if (set) hipLaunchKernelGGL(combine_runs_kernel<false>, cg, dim3(256), 0, mf->stream, cr, dp->partial, dst);
    else hipLaunchKernelGGL(combine_runs_kernel<true>, cg, dim3(256), 0, mf->stream, cr, dp->partial, dst);
    KERNEL_CHECK();
    return BP5_OK;
  }
  if (set) hipLaunchKernelGGL(combine_kernel<false>, cg, dim3(256), 0, mf->stream, dp->sh_dof, dp->sh_off, dp->sh_slot, dp->partial, dst, dp->n_shared);
  else hipLaunchKernelGGL(combine_kernel<true>, cg, dim3(256), 0, mf->stream, dp->sh_dof, dp->sh_off, dp->sh_slot, dp->partial, dst, dp->n_shared);
  KERNEL_CHECK();
  return BP5_OK;
}

// block-assembled kernel; falls back to the team kernel path when the range is partial
template <int P, bool COLL, int LPC, int ABL = 0>
static int launch_block_t(bp5_mf *mf, const double *coef, const double *src, double *dst, bool overwrite)
{
  constexpr int n = P + 1;
  constexpr int CPT = 256 / LPC;
  using L = LdsLayout<n, LPC>;
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan_raw(mf, -CPT, &dp));
  const size_t tile_cs = (ABL & 8192) ? (size_t)(n * L::PS + 3) : (size_t)L::CS;
  const size_t lds = ((size_t)CPT * tile_cs + dp->max_list) * sizeof(double) + ((ABL & 16384) ? 4 * BLOCK_MAX_RUNS * sizeof(uint32_t) : 0);
  if ((ABL & 16384) && dp->max_runs > (uint32_t)BLOCK_MAX_RUNS) return fail(BP5_ERR_UNSUPPORTED, "too many runs per block for the run-length write-out");
  if ((ABL & 262144) && !dp->packed) return fail(BP5_ERR_UNSUPPORTED, "more than 64 runs per block: packed indices unavailable");
  if (lds > 160 * 1024) return fail(BP5_ERR_UNSUPPORTED, "cell block does not fit in LDS; pass smaller cell blocks");
  BlockPlan bp{}; // value-initialised: a field this launcher forgets is null, not garbage
  bp.pass_cell = dp->pass_cell; bp.pass_off = dp->pass_off; bp.off = dp->off; bp.dofs = dp->dofs; bp.pos = dp->pos; bp.gidx = dp->gidx;
  bp.packed = dp->packed;
  bp.cell_round = dp->cell_round; bp.blk_rounds = dp->team_rounds; bp.partial = dp->partial; bp.n_blocks = dp->n_groups;
  bp.run_off = dp->run_off; bp.runs = dp->runs; bp.max_list = dp->max_list;
  // a block-aligned cell range: only these blocks run, accumulate mode; DoFs shared with other blocks go to dst by
  // atomics (the partial slab + combine pass needs every block of the plan in the launch)
  const bool sub_range = mf->blk_b1 > mf->blk_b0 && (mf->blk_b0 != 0 || mf->blk_b1 != dp->n_groups);
  bp.blk_begin = sub_range ? mf->blk_b0 : 0;
  if (sub_range) bp.n_blocks = mf->blk_b1 - mf->blk_b0;
  if (sub_range && overwrite) return fail(BP5_ERR_INVALID, "a cell range cannot overwrite dst");
  // persistent grid: two workgroups per CU (LDS budget), a multiple of 8 for the XCD mapping
  if (!mf->n_cus) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, mf->device));
    mf->n_cus = prop.multiProcessorCount;
  }
  const int wg_per_cu = ((ABL & 2048) && lds * 3 <= 160 * 1024) ? 3 : lds * 2 <= 160 * 1024 ? 2 : 1;
  uint32_t n_wg = (uint32_t)(mf->n_cus * wg_per_cu);
  if (mf->block_max_wg > 0) n_wg = std::min<uint32_t>(n_wg, (uint32_t)mf->block_max_wg);
  n_wg = std::max<uint32_t>(8, std::min<uint32_t>(n_wg, (bp.n_blocks + 7) / 8 * 8) / 8 * 8);
  bp.n_wg = n_wg;
  { // block ranges of the persistent workgroups: equal shares of the estimated COST (thin or partial bricks are cheaper per
    // block but dearer per cell than full ones), cached
    const uint32_t B0 = bp.blk_begin, B1 = bp.blk_begin + bp.n_blocks;
    if (!dp->wg_block || dp->wg_n != n_wg || dp->wg_b0 != B0 || dp->wg_b1 != B1) {
      std::vector<uint32_t> wb(n_wg + 1);
      const std::vector<double> &pc = dp->h_cost;
      const double c0 = pc[B0], total = pc[B1] - c0;
      for (uint32_t w = 0; w <= n_wg; ++w)
        wb[w] = (uint32_t)(std::lower_bound(pc.begin() + B0, pc.begin() + B1 + 1, c0 + total * w / n_wg - 1e-9) - pc.begin());
      wb[0] = B0; wb[n_wg] = B1;
      if (dp->wg_block) { HIP_TRY(hipStreamSynchronize(mf->stream)); HIP_TRY(hipFree(dp->wg_block)); dp->wg_block = nullptr; }
      BP5_TRY(upload(&dp->wg_block, wb.data(), wb.size()));
      dp->wg_n = n_wg; dp->wg_b0 = B0; dp->wg_b1 = B1;
    }
    bp.wg_block = dp->wg_block;
  }
  bp.stamps = nullptr;
  if (ABL & 4096) {
    if (!mf->d_stamps) HIP_TRY(hipMalloc((void **)&mf->d_stamps, 4096 * 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(mf->d_stamps, 0, 4096 * 16 * sizeof(unsigned long long), mf->stream));
    bp.stamps = mf->d_stamps;
  }
  ApplyArgs a{};
  a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst;
  a.plane_stride = (uint64_t)mf->n_cells * mf->n3;
  a.cell_begin = 0; a.cell_end = mf->n_cells; a.n_teams = dp->n_groups; a.teams_per_xcd = 0;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  ShapeArg<n> sh;
  memcpy(sh.N, mf->tab.N, sizeof(sh.N));
  memcpy(sh.D, mf->tab.D, sizeof(sh.D));
  const bool set = overwrite && dp->covers_all;
  if (overwrite && !set) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
  const dim3 grid(n_wg), block(256);
  if (mf->block_shared_atomic || sub_range) {
    // brick-surface DoFs by atomics: zero exactly those first (SET mode), no partial slab / combine
    if (set && dp->n_shared) {
      hipLaunchKernelGGL(zero_indexed_kernel, dim3((dp->n_shared + 255) / 256), dim3(256), 0, mf->stream, dp->sh_dof, dp->n_shared, dst);
      KERNEL_CHECK();
    }
    if (set) {
      auto kern = apply_block_kernel<P, COLL, LPC, SC_OWNER_SET_ATOMIC, ABL>;
      HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, grid, block, lds, mf->stream, a, bp, sh);
    } else {
      auto kern = apply_block_kernel<P, COLL, LPC, SC_OWNER_ADD_ATOMIC, ABL>;
      HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, grid, block, lds, mf->stream, a, bp, sh);
    }
    KERNEL_CHECK();
    return BP5_OK;
  }
  if (set) {
    auto kern = apply_block_kernel<P, COLL, LPC, SC_OWNER_SET, ABL>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, mf->stream, a, bp, sh);
  } else {
    auto kern = apply_block_kernel<P, COLL, LPC, SC_OWNER_ADD, ABL>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, mf->stream, a, bp, sh);
  }
  KERNEL_CHECK();
  if (ABL & 4096) { // diagnostic build: print the per-phase cycle shares (never quote its run time)
    HIP_TRY(hipStreamSynchronize(mf->stream));
    std::vector<unsigned long long> hs((size_t)n_wg * 16);
    HIP_TRY(hipMemcpy(hs.data(), mf->d_stamps, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double tot[9] = {0};
    for (uint32_t w = 0; w < n_wg; ++w) for (int k = 0; k < 9; ++k) tot[k] += (double)hs[(size_t)w * 16 + k];
    double all = 0; for (int k = 0; k < 7; ++k) all += tot[k];
    static const char *nm[7] = {"issue loads", "evaluate (+wait u)", "wait idx + issue gather", "q-op (+wait metric)", "integrate", "accumulate", "block boundary"};
    fprintf(stderr, "[bp5 stamps] passes/wg %.1f, cycles/pass %.0f\n", tot[8] / n_wg, all / tot[8]);
    for (int k = 0; k < 7; ++k) fprintf(stderr, "[bp5 stamps]   %-26s %5.1f %%  %8.0f cycles/pass\n", nm[k], 100.0 * tot[k] / all, tot[k] / tot[8]);
  }
  if (ABL & 1023) return BP5_OK; // (1024 and above are real modes) timing-only ablation builds skip the combine pass (1024/2048/8192 are real modes)
  return launch_combine(mf, dp, dst, set);
}

// overwrite == true: dst need not be zeroed by the caller, the launch defines every entry
template <int P, bool COLL, int TW, int LPC, bool PF, int OPT = 0>
static int launch_team_t(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1, bool overwrite)
{
  constexpr int n = P + 1;
  constexpr int CPT = 64 * TW / LPC;
  using L = LdsLayout<n, LPC>;
  TeamPlan tp{};
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan(mf, CPT, tp, &dp));
  ApplyArgs a{};
  a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst;
  a.plane_stride = (uint64_t)mf->n_cells * mf->n3;
  a.cell_begin = c0; a.cell_end = c1;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  a.n_teams = (c1 + CPT - 1) / CPT - c0 / CPT;
  a.teams_per_xcd = (a.n_teams + 7) / 8;
  ShapeArg<n> sh;
  memcpy(sh.N, mf->tab.N, sizeof(sh.N));
  memcpy(sh.D, mf->tab.D, sizeof(sh.D));
  const size_t lds = (size_t)CPT * L::CS * sizeof(double);
  const dim3 grid(a.teams_per_xcd * 8), block(64 * TW);
  const bool whole = (c0 == 0 && c1 == mf->n_cells);
  if (!whole || mf->force_atomic_scatter) {
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    hipLaunchKernelGGL((apply_team_kernel<P, COLL, TW, LPC, PF, SC_ATOMIC, OPT>), grid, block, lds, mf->stream, a, tp, sh);
  } else {
    const bool set = overwrite && dp->covers_all;
    if (overwrite && !set) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    if (set) hipLaunchKernelGGL((apply_team_kernel<P, COLL, TW, LPC, PF, SC_OWNER_SET, OPT>), grid, block, lds, mf->stream, a, tp, sh);
    else hipLaunchKernelGGL((apply_team_kernel<P, COLL, TW, LPC, PF, SC_OWNER_ADD, OPT>), grid, block, lds, mf->stream, a, tp, sh);
    KERNEL_CHECK();
    return launch_combine(mf, dp, dst, set);
  }
  KERNEL_CHECK();
  return BP5_OK;
}
#define TEAM_CASE(P, V, TW, LPC, PF)                                                                               \
  case (P)*100 + (V):                                                                                              \
    return coll ? launch_team_t<P, true, TW, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite)                       \
                : launch_team_t<P, false, TW, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite)

// z-marching kernel (whole cell range only; partial ranges take the plain pencil kernel)
template <int P, bool COLL, int TW, int LPC, bool PF, int ABL = 0>
static int launch_march_t(bp5_mf *mf, const double *coef, const double *src, double *dst)
{
  constexpr int n = P + 1;
  constexpr int CPT = 64 * TW / LPC;
  using L = LdsLayout<n, LPC>;
  auto it = mf->march_plans.find(CPT);
  if (it == mf->march_plans.end()) {
    MarchPlanHost h;
    BP5_TRY(build_march_plan(mf->h_l2g.data(), mf->n_cells, n, CPT, mf->march_max_steps, h));
    bp5_mf::DevMarch dm;
    BP5_TRY(upload(&dm.team_off, h.team_off.data(), h.team_off.size()));
    BP5_TRY(upload(&dm.entries, h.entries.data(), h.entries.size()));
    dm.n_teams = (uint32_t)h.team_off.size() - 1;
    it = mf->march_plans.emplace(CPT, dm).first;
  }
  MarchPlan mp{};
  mp.team_off = it->second.team_off; mp.entries = it->second.entries; mp.n_teams = it->second.n_teams;
  mp.teams_per_xcd = (mp.n_teams + 7) / 8;
  ApplyArgs a{};
  a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst;
  a.plane_stride = (uint64_t)mf->n_cells * mf->n3;
  a.cell_begin = 0; a.cell_end = mf->n_cells; a.n_teams = mp.n_teams; a.teams_per_xcd = mp.teams_per_xcd;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  ShapeArg<n> sh;
  memcpy(sh.N, mf->tab.N, sizeof(sh.N));
  memcpy(sh.D, mf->tab.D, sizeof(sh.D));
  const size_t lds = (size_t)CPT * L::CS * sizeof(double);
  hipLaunchKernelGGL((apply_march_kernel<P, COLL, TW, LPC, PF, ABL>), dim3(mp.teams_per_xcd * 8), dim3(64 * TW), lds, mf->stream, a, mp, sh);
  KERNEL_CHECK();
  return BP5_OK;
}

// variant table: (degree, variant) -> (TW, LPC, TPB, PF); variant 0 = default for the degree
#define APPLY_CASE(P, V, TW, LPC, TPB, PF)                                                                         \
  case (P)*100 + (V):                                                                                              \
    if (overwrite && hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream) != hipSuccess)             \
      return fail(BP5_ERR_HIP, "hipMemsetAsync");                                                                  \
    return coll ? launch_apply_t<P, true, TW, LPC, TPB, PF>(mf, coef, src, dst, c0, c1)                            \
                : launch_apply_t<P, false, TW, LPC, TPB, PF>(mf, coef, src, dst, c0, c1)

// overwrite: the launch must leave dst = A src (no prior zeroing by the caller); otherwise dst += A src
template <int P>
static int launch_affine(bp5_mf *mf, const double *src, double *dst, uint32_t c0, uint32_t c1)
{ // TW = 4 teams when n^2 lanes per cell pack well into 256 threads, as for the 6-plane default
  constexpr int n2 = (P + 1) * (P + 1);
  constexpr int LPC = n2;
  constexpr bool PF = true;
  return mf->quadrature == BP5_QUAD_GLL ? launch_apply_t<P, true, 4, LPC, 1, PF, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1)
                                        : launch_apply_t<P, false, 4, LPC, 1, PF, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1);
}
// [c0,c1) == union of whole cell blocks [b0,b1) of the caller's blocking?
static bool block_aligned(const bp5_mf *mf, uint32_t c0, uint32_t c1, uint32_t *b0, uint32_t *b1)
{
  const auto &o = mf->h_block_off;
  if (o.empty() || c1 <= c0) return false;
  const auto i0 = std::lower_bound(o.begin(), o.end(), c0), i1 = std::lower_bound(o.begin(), o.end(), c1);
  if (i0 == o.end() || i1 == o.end() || *i0 != c0 || *i1 != c1) return false;
  *b0 = (uint32_t)(i0 - o.begin());
  *b1 = (uint32_t)(i1 - o.begin());
  return true;
}
// Variant 0 = library default.  The measured choices (profiles/r1): p = 1, 3 x-row team kernel; p = 4 on a mesh
// handed over in cell blocks that fit three workgroups per CU: block-assembled kernel (no atomics, no zero-fill,
// bitwise reproducible), whole cell range only; p = 4 affine geometry: team kernel; everything else: pencil kernel.
static int effective_variant(bp5_mf *mf, uint32_t c0, uint32_t c1)
{
  const int v = mf->apply_variant;
  if (v != 0) return v;
  if (mf->degree == 1 || mf->degree == 3) return mf->geometry_mode == BP5_GEOM_AFFINE ? 0 : 10;
  if (mf->degree != 4) return 0;
  const int fallback = mf->geometry_mode == BP5_GEOM_AFFINE ? 10 : 0; // affine: team kernel, else pencil kernel
  uint32_t b0_, b1_;
  if (mf->h_block_off.empty() || !block_aligned(mf, c0, c1, &b0_, &b1_)) return fallback;
  if (mf->auto_block < 0) {
    bp5_mf::DevPlan *dp = nullptr;
    mf->auto_block = 0;
    if (get_plan_raw(mf, -8, &dp, 64) == BP5_OK) {
      const size_t lds = ((size_t)8 * (5 * LdsLayout<5, 32>::PS + 3) + dp->max_list) * sizeof(double) + 4 * BLOCK_MAX_RUNS * sizeof(uint32_t);
      mf->auto_block = lds * 3 <= 160 * 1024;
      // persistent workgroups need enough bricks each to balance (measured: 3.6 bricks per workgroup at 54^3 cells
      // loses 4 % against the pencil kernel, 32 per workgroup at 116^3 wins): at least 10 per workgroup
      if (!mf->n_cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, mf->device) == hipSuccess) mf->n_cus = prop.multiProcessorCount;
      }
      if (dp->n_groups < 30u * (uint32_t)std::max(mf->n_cus, 1)) mf->auto_block = 0;
    }
  }
  // sub-ranges: worth it only while the range still feeds the persistent grid (else the pencil kernel)
  if (mf->auto_block && (c0 != 0 || c1 != mf->n_cells) && (b1_ - b0_) < 30u * (uint32_t)std::max(mf->n_cus, 1)) return fallback;
  if (mf->auto_block && mf->geometry_mode == BP5_GEOM_AFFINE) { // the affine build needs the packed indices
    bp5_mf::DevPlan *dp = nullptr;
    if (get_plan_raw(mf, -8, &dp, 64) != BP5_OK || !dp->packed) return fallback;
  }
  return mf->auto_block ? 56 : fallback;
}
// kernels that define every entry of dst themselves (owner stores + combine pass) need no zero-fill
static bool variant_overwrites(const bp5_mf *mf, int ev)
{
  const int v = ev % 100;
  return ev < 100 && ((v >= 10 && v <= 14) || (v >= 48 && v <= 59)) && !(mf->geometry_mode == BP5_GEOM_AFFINE && mf->degree != 4);
}

static int launch_apply_impl(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1, bool overwrite)
{
  if (mf->geometry_mode == BP5_GEOM_AFFINE && c1 > c0) {
    const bool coll_ = mf->quadrature == BP5_QUAD_GLL;
    const bool whole = c0 == 0 && c1 == mf->n_cells;
    if (mf->degree == 4 && mf->apply_variant % 100 == 10) {
      mf->force_atomic_scatter = mf->apply_variant >= 100;
      return coll_ ? launch_team_t<4, true, 4, 25, true, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1, overwrite)
                   : launch_team_t<4, false, 4, 25, true, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1, overwrite);
    }
    if (mf->degree == 4 && mf->apply_variant == 56) { // the default block-kernel shape on the scalar plane + per-cell K K^T
      if (!block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) return fail(BP5_ERR_INVALID, "variant 56 needs a cell range aligned with the cell blocks");
      struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; } } reset{mf};
      return coll_ ? launch_block_t<4, true, 32, 1024 + 2048 + 8192 + 16384 + 262144>(mf, mf->d_scalar_plane, src, dst, overwrite)
                   : launch_block_t<4, false, 32, 1024 + 2048 + 8192 + 16384 + 262144>(mf, mf->d_scalar_plane, src, dst, overwrite);
    }
    if (mf->degree == 4 && whole && (mf->apply_variant == 54 || mf->apply_variant == 55)) {
      mf->block_shared_atomic = true;
      const int st_ = mf->apply_variant == 54 ? (coll_ ? launch_block_t<4, true, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                                       : launch_block_t<4, false, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite))
                                              : (coll_ ? launch_block_t<4, true, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                                       : launch_block_t<4, false, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite));
      mf->block_shared_atomic = false;
      return st_;
    }
    if (mf->degree == 4 && whole && (mf->apply_variant == 50 || mf->apply_variant == 51))
      return mf->apply_variant == 50 ? (coll_ ? launch_block_t<4, true, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                              : launch_block_t<4, false, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite))
                                     : (coll_ ? launch_block_t<4, true, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                              : launch_block_t<4, false, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite));
    if (mf->degree == 4 && mf->apply_variant == 85) // timing only: affine, no scatter atomics -> compute/latency floor of the pencil kernel
      return launch_apply_t<4, false, 4, 25, 1, true, 1025>(mf, mf->d_scalar_plane, src, dst, c0, c1);
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    switch (mf->degree) {
      case 1: return launch_affine<1>(mf, src, dst, c0, c1);
      case 2: return launch_affine<2>(mf, src, dst, c0, c1);
      case 3: return launch_affine<3>(mf, src, dst, c0, c1);
      case 4: return launch_affine<4>(mf, src, dst, c0, c1);
      case 5: return launch_affine<5>(mf, src, dst, c0, c1);
      case 6: return launch_affine<6>(mf, src, dst, c0, c1);
      case 7: return launch_affine<7>(mf, src, dst, c0, c1);
      case 8: return launch_affine<8>(mf, src, dst, c0, c1);
    }
  }
  if (c1 <= c0) {
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    return BP5_OK;
  }
  const bool coll = mf->quadrature == BP5_QUAD_GLL;
  // variants >= 100: the team kernel of (variant - 100) with the global-atomic scatter (A/B tests)
  mf->force_atomic_scatter = mf->apply_variant >= 100;
  int variant = mf->apply_variant % 100;
  switch (mf->degree * 100 + variant) {
    APPLY_CASE(1, 0, 1, 4, 4, true);
    APPLY_CASE(1, 1, 1, 4, 4, true);
    APPLY_CASE(2, 0, 1, 9, 4, true);
    APPLY_CASE(3, 0, 1, 16, 4, true);
    APPLY_CASE(3, 1, 1, 16, 4, true);
    APPLY_CASE(4, 0, 4, 25, 1, true);
    APPLY_CASE(4, 6, 1, 25, 4, true);
    APPLY_CASE(4, 1, 1, 32, 4, true);
    APPLY_CASE(4, 2, 2, 25, 1, true);
    APPLY_CASE(4, 3, 4, 25, 1, true);
    APPLY_CASE(4, 4, 1, 25, 1, true);
    APPLY_CASE(4, 5, 1, 25, 4, false);
    APPLY_CASE(5, 0, 4, 36, 1, true);
    APPLY_CASE(5, 1, 1, 36, 4, true);
    APPLY_CASE(5, 2, 4, 36, 1, false);
    APPLY_CASE(5, 3, 2, 36, 1, true);
    APPLY_CASE(6, 0, 4, 49, 1, true);   // defaults for p >= 6 from the high-degree sweep: prefetch all planes
    APPLY_CASE(6, 5, 4, 49, 1, false);
    APPLY_CASE(6, 1, 1, 49, 4, false);
    APPLY_CASE(6, 2, 4, 49, 1, true);
    APPLY_CASE(6, 3, 1, 49, 1, true);
    APPLY_CASE(6, 4, 2, 49, 1, false);
    APPLY_CASE(7, 0, 4, 64, 1, true);
    APPLY_CASE(7, 5, 1, 64, 4, false);
    APPLY_CASE(7, 1, 4, 64, 1, false);
    APPLY_CASE(7, 2, 4, 64, 1, true);
    APPLY_CASE(7, 3, 1, 64, 1, true);
    APPLY_CASE(8, 0, 4, 81, 1, true);
    APPLY_CASE(8, 5, 4, 81, 1, false);
    APPLY_CASE(8, 1, 2, 81, 1, false);
    APPLY_CASE(8, 2, 4, 81, 1, true);
    APPLY_CASE(8, 3, 2, 81, 1, true);
    // timing-only ablations of variant 3 (results are wrong by construction): 20 + ABL mask
#define ABL_CASE(M) case 400 + 20 + (M): return launch_apply_t<4, false, 4, 25, 1, true, M>(mf, coef, src, dst, c0, c1)
    case 407: return coll ? launch_apply_t<4, true, 4, 25, 1, true, 256>(mf, coef, src, dst, c0, c1) : launch_apply_t<4, false, 4, 25, 1, true, 256>(mf, coef, src, dst, c0, c1);
    case 408: return coll ? launch_apply_t<4, true, 4, 25, 1, true, 512>(mf, coef, src, dst, c0, c1) : launch_apply_t<4, false, 4, 25, 1, true, 512>(mf, coef, src, dst, c0, c1);
    case 409: return coll ? launch_apply_t<4, true, 2, 25, 1, true, 512>(mf, coef, src, dst, c0, c1) : launch_apply_t<4, false, 2, 25, 1, true, 512>(mf, coef, src, dst, c0, c1);
    case 482: return launch_apply_t<4, false, 4, 25, 1, true, 257>(mf, coef, src, dst, c0, c1);
    case 483: return launch_apply_t<4, false, 4, 25, 1, true, 4096>(mf, coef, src, dst, c0, c1);
    case 484: return launch_apply_t<4, false, 4, 25, 1, true, 8192>(mf, coef, src, dst, c0, c1);
    case 480: return launch_apply_t<4, false, 4, 25, 1, true, 64>(mf, coef, src, dst, c0, c1);
    case 481: { // E-vector stores need a big scratch target
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 128>(mf, coef, src, mf->d_evec, c0, c1); }
    case 490: {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, ((size_t)mf->n_cells * mf->n3 + 4096 * 5) * sizeof(double) * 2));
      return launch_apply_t<4, false, 4, 25, 1, true, 262144>(mf, coef, src, mf->d_evec, c0, c1); }
    case 488: {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 128 + 65536>(mf, coef, src, mf->d_evec, c0, c1); }
    case 489: {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 1 + 131072>(mf, coef, src, mf->d_evec, c0, c1); }
    case 486: {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 128 + 16384>(mf, coef, src, mf->d_evec, c0, c1); }
    ABL_CASE(1); ABL_CASE(2); ABL_CASE(3); ABL_CASE(4); ABL_CASE(5); ABL_CASE(7); ABL_CASE(8); ABL_CASE(9); ABL_CASE(15); ABL_CASE(14); ABL_CASE(13); ABL_CASE(11);
    // z-marching kernel, variants 70+ (atomic scatter: dst must be zero-filled like for the pencil kernel)
#define MARCH_CASE(P, V, TW, LPC, PF)                                                                              \
  case (P)*100 + (V):                                                                                              \
    if (overwrite && hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream) != hipSuccess)             \
      return fail(BP5_ERR_HIP, "hipMemsetAsync");                                                                  \
    if (c0 != 0 || c1 != mf->n_cells)                                                                              \
      return coll ? launch_apply_t<P, true, TW, LPC, 1, PF>(mf, coef, src, dst, c0, c1)                            \
                  : launch_apply_t<P, false, TW, LPC, 1, PF>(mf, coef, src, dst, c0, c1);                          \
    return coll ? launch_march_t<P, true, TW, LPC, PF>(mf, coef, src, dst) : launch_march_t<P, false, TW, LPC, PF>(mf, coef, src, dst)
    case 473: return launch_march_t<4, false, 4, 25, true, 1>(mf, coef, src, dst); // timing only: march, no scatter
    MARCH_CASE(1, 70, 4, 4, true);
    MARCH_CASE(2, 70, 4, 9, true);
    MARCH_CASE(3, 70, 4, 16, true);
    MARCH_CASE(4, 70, 4, 25, true);
    MARCH_CASE(4, 71, 2, 25, true);
    MARCH_CASE(4, 72, 1, 25, true);
    MARCH_CASE(5, 70, 4, 36, true);
    MARCH_CASE(6, 70, 4, 49, true);
    MARCH_CASE(7, 70, 4, 64, true);
    MARCH_CASE(8, 70, 4, 81, true);
    // block-assembled kernel (compact cell blocks, LDS accumulator, no atomics), variants 50+;
    // a partial cell range cannot use the owner scatter and takes the atomic team kernel instead
#define BLOCK_CASE(P, V, LPC, TW_FALLBACK, PF)                                                                     \
  case (P)*100 + (V):                                                                                              \
    if (c0 != 0 || c1 != mf->n_cells)                                                                              \
      return coll ? launch_team_t<P, true, TW_FALLBACK, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite)            \
                  : launch_team_t<P, false, TW_FALLBACK, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite);          \
    return coll ? launch_block_t<P, true, LPC>(mf, coef, src, dst, overwrite) : launch_block_t<P, false, LPC>(mf, coef, src, dst, overwrite)
    BLOCK_CASE(1, 50, 4, 4, true);
    BLOCK_CASE(2, 50, 9, 4, true);
    BLOCK_CASE(3, 50, 16, 4, true);
    BLOCK_CASE(4, 50, 25, 4, true);
    BLOCK_CASE(4, 51, 32, 4, true);
    case 454: case 455:
      if (c0 == 0 && c1 == mf->n_cells) {
        mf->block_shared_atomic = true;
        const int st_ = variant == 454 - 400 ? (coll ? launch_block_t<4, true, 32>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32>(mf, coef, src, dst, overwrite))
                                             : (coll ? launch_block_t<4, true, 25>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 25>(mf, coef, src, dst, overwrite));
        mf->block_shared_atomic = false;
        return st_;
      }
      return coll ? launch_team_t<4, true, 4, 25, true>(mf, coef, src, dst, c0, c1, overwrite) : launch_team_t<4, false, 4, 25, true>(mf, coef, src, dst, c0, c1, overwrite);
    case 448: // = 56 with the per-DoF CSR combine kernel instead of the run-length one (A/B)
    case 449: // = 56 with run-length write-out but without packed indices (A/B)
    case 456: if (block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) {
        struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; m->combine_csr = false; } } reset{mf};
        mf->combine_csr = variant == 48;
        bp5_mf::DevPlan *dp_ = nullptr;
        BP5_TRY(get_plan_raw(mf, -8, &dp_));
        if (dp_->packed && variant != 49) // few long runs (block-major numbering): one packed u16 per cell-local DoF, no local_to_global stream
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, overwrite)
                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, overwrite);
        if (dp_->max_runs <= (uint32_t)BLOCK_MAX_RUNS) // write-out without list loads
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192 + 16384>(mf, coef, src, dst, overwrite);
        return coll ? launch_block_t<4, true, 32, 2048 + 8192>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192>(mf, coef, src, dst, overwrite);
      }
      return fail(BP5_ERR_INVALID, "variant 56 needs a cell range aligned with the cell blocks");
    case 459: if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 2048 + 8192>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 59 needs the whole cell range");
    case 457: if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 8192>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 8192>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 57 needs the whole cell range");
    case 458: if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 32768>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192 + 32768>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 58 needs the whole cell range");
    case 499: return launch_block_t<4, false, 32, 4096 + 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, true);   // stamps of the default shape (sequential tiles, 3 WG/CU, run write-out, packed indices)
    case 452: if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 2048>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 52 needs the whole cell range");
    case 453: if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 25, 2048>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 25, 2048>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 53 needs the whole cell range");
    case 487: return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 65536>(mf, coef, src, dst, true);  // variant 56 with plain (not non-temporal) stores
    case 491: return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 1>(mf, coef, src, dst, true);  // variant 56 without write-out (and combine)
    case 493: return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 2>(mf, coef, src, dst, true);  // ... without metric loads
    case 495: return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 4>(mf, coef, src, dst, true);  // ... without gather
    case 497: return launch_block_t<4, false, 32, 4096>(mf, coef, src, dst, true);          // stamps, double-buffered
    case 498: return launch_block_t<4, false, 32, 4096 + 2048>(mf, coef, src, dst, true);   // stamps, single-buffered
    case 492: return launch_block_t<4, false, 32, 2049>(mf, coef, src, dst, true);
    case 496: return launch_block_t<4, false, 32, 2053>(mf, coef, src, dst, true);
    BLOCK_CASE(5, 50, 36, 4, true);
    BLOCK_CASE(6, 50, 49, 4, false);
    BLOCK_CASE(7, 50, 64, 4, false);
    BLOCK_CASE(8, 50, 81, 4, false);
#define BABL_CASE(M) case 400 + 60 + (M): return launch_block_t<4, false, 25, M>(mf, coef, src, dst, true)
    BABL_CASE(16); BABL_CASE(1); BABL_CASE(2); BABL_CASE(3); BABL_CASE(4); BABL_CASE(5); BABL_CASE(7); BABL_CASE(8); BABL_CASE(9); BABL_CASE(15);
    // timing-only ablations of the team kernel (SET mode): 40 + mask (1: no scatter stage, 4: no gather stage)
#define TABL_CASE(M)                                                                                               \
  case 400 + 40 + (M): {                                                                                           \
    TeamPlan tp; bp5_mf::DevPlan *dp = nullptr;                                                                    \
    BP5_TRY(get_plan(mf, 10, tp, &dp));                                                                            \
    ApplyArgs a; a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst; a.plane_stride = (uint64_t)mf->n_cells * mf->n3; \
    a.cell_begin = c0; a.cell_end = c1; a.n_teams = (c1 + 9) / 10 - c0 / 10; a.teams_per_xcd = (a.n_teams + 7) / 8;  \
    ShapeArg<5> sh; memcpy(sh.N, mf->tab.N, sizeof(sh.N)); memcpy(sh.D, mf->tab.D, sizeof(sh.D));                  \
    hipLaunchKernelGGL((apply_team_kernel<4, false, 4, 25, true, SC_OWNER_SET, M>), dim3(a.teams_per_xcd * 8), dim3(256), \
                       (10 * LdsLayout<5, 25>::CS * sizeof(double)), mf->stream, a, tp, sh);                          \
    KERNEL_CHECK(); return BP5_OK; }
    TABL_CASE(0); TABL_CASE(1); TABL_CASE(4); TABL_CASE(5);
    // team-assembled kernel (LDS-staged gather + scatter), variants 10+
    TEAM_CASE(1, 10, 4, 4, true);
    TEAM_CASE(2, 10, 4, 9, true);
    TEAM_CASE(3, 10, 4, 16, true);
    TEAM_CASE(4, 10, 4, 25, true);
    TEAM_CASE(4, 11, 8, 25, true);
    TEAM_CASE(4, 12, 4, 25, false);
    TEAM_CASE(4, 13, 2, 25, true);
    case 414: return coll ? launch_team_t<4, true, 4, 25, true, 32>(mf, coef, src, dst, c0, c1, overwrite)
                          : launch_team_t<4, false, 4, 25, true, 32>(mf, coef, src, dst, c0, c1, overwrite);
    TEAM_CASE(5, 10, 4, 36, true);
    TEAM_CASE(6, 10, 4, 49, false);
    TEAM_CASE(7, 10, 4, 64, false);
    TEAM_CASE(8, 10, 4, 81, false);
  }
  return fail(BP5_ERR_INVALID, "unknown (degree, apply variant)");
}
static int launch_apply(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1, bool overwrite = false)
{
  const int user = mf->apply_variant;
  mf->apply_variant = effective_variant(mf, c0, c1);
  const int st = launch_apply_impl(mf, coef, src, dst, c0, c1, overwrite);
  mf->apply_variant = user;
  return st;
}

extern "C" int bp5_apply_cells(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1)
{
  if (!mf || (!coef && mf->geometry_mode != BP5_GEOM_AFFINE) || !src || !dst) return fail(BP5_ERR_INVALID, "null argument");
  if (c1 > mf->n_cells || c0 > c1) return fail(BP5_ERR_INVALID, "cell range out of bounds");
  if (src == dst) return fail(BP5_ERR_INVALID, "src and dst must differ");
  HIP_TRY(hipSetDevice(mf->device));
  return launch_apply(mf, coef, src, dst, c0, c1);
}
extern "C" int bp5_copy_constrained(bp5_mf *mf, const double *src, double *dst)
{
  if (!mf || !src || !dst) return fail(BP5_ERR_INVALID, "null argument");
  if (!mf->n_constrained) return BP5_OK;
  hipLaunchKernelGGL(copy_constrained_kernel, dim3((mf->n_constrained + 255) / 256), dim3(256), 0, mf->stream, mf->d_constrained,
                     mf->n_constrained, src, dst);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_set_constrained(bp5_mf *mf, double value, double *dst)
{
  if (!mf || !dst) return fail(BP5_ERR_INVALID, "null argument");
  if (!mf->n_constrained) return BP5_OK;
  hipLaunchKernelGGL(set_constrained_kernel, dim3((mf->n_constrained + 255) / 256), dim3(256), 0, mf->stream, mf->d_constrained,
                     mf->n_constrained, value, dst);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_apply(bp5_mf *mf, const double *coef, const double *src, double *dst, int zero_dst)
{
  if (!mf || (!coef && mf->geometry_mode != BP5_GEOM_AFFINE) || !src || !dst) return fail(BP5_ERR_INVALID, "null argument");
  if (src == dst) return fail(BP5_ERR_INVALID, "src and dst must differ");
  HIP_TRY(hipSetDevice(mf->device));
  BP5_TRY(launch_apply(mf, coef, src, dst, 0, mf->n_cells, zero_dst != 0));
  return bp5_copy_constrained(mf, src, dst);
}

// ------------------------------------------------------------------------------------ rhs / norms
template <int n>
static int launch_rhs(bp5_mf *mf, double *b)
{
  const uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(mf->n_cells, 1), 65535u * 16);
  hipLaunchKernelGGL(rhs_kernel<n>, dim3(grid), dim3(n, n, n), 0, mf->stream, mf->d_l2g, mf->d_coords, mf->d_tab_gauss, mf->n_cells, b);
  KERNEL_CHECK();
  return BP5_OK;
}
static int rhs_dispatch(bp5_mf *mf, double *b) { DISPATCH_N(launch_rhs, mf, b); }
extern "C" int bp5_assemble_rhs(bp5_mf *mf, double *b)
{
  if (!mf || !b) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  HIP_TRY(hipMemsetAsync(b, 0, mf->n_local() * sizeof(double), mf->stream));
  BP5_TRY(rhs_dispatch(mf, b));
  if (mf->comm && !mf->neighbors.empty()) BP5_TRY(bp5_halo_scatter_add(mf, b));
  return bp5_set_constrained(mf, 0.0, b);
}
template <int n>
static int launch_diagonal(bp5_mf *mf, const double *coef, double *diag)
{
  const uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(mf->n_cells, 1), 65536u);
  const bool affine = mf->geometry_mode == BP5_GEOM_AFFINE;
  hipLaunchKernelGGL(diagonal_kernel<n>, dim3(grid), dim3(n, n, n), 0, mf->stream, mf->d_l2g, affine ? mf->d_scalar_plane : coef,
                     (uint64_t)mf->n_cells * mf->n3, affine ? mf->d_gcell : (const double *)nullptr, mf->d_tab, mf->n_cells, diag);
  KERNEL_CHECK();
  return BP5_OK;
}
static int diagonal_dispatch(bp5_mf *mf, const double *coef, double *diag) { DISPATCH_N(launch_diagonal, mf, coef, diag); }
extern "C" int bp5_compute_diagonal(bp5_mf *mf, const double *coef, double *diag, int invert)
{
  if (!mf || (!coef && mf->geometry_mode != BP5_GEOM_AFFINE) || !diag) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  HIP_TRY(hipMemsetAsync(diag, 0, mf->n_local() * sizeof(double), mf->stream));
  if (mf->n_cells) BP5_TRY(diagonal_dispatch(mf, coef, diag));
  if (mf->comm && !mf->neighbors.empty()) { // ghost contributions to their owners
    BP5_TRY(bp5_halo_scatter_add(mf, diag));
    BP5_TRY(bp5_halo_zero_ghosts(mf, diag));
  }
  BP5_TRY(bp5_set_constrained(mf, 1.0, diag));                                     // A_eff = P A P + (I - P)
  if (invert && mf->n_owned) {
    hipLaunchKernelGGL(reciprocal_kernel, dim3((mf->n_owned + 255) / 256), dim3(256), 0, mf->stream, diag, (size_t)mf->n_owned);
    KERNEL_CHECK();
  }
  return BP5_OK;
}
template <int n>
static int launch_l2(bp5_mf *mf, const double *u, double *out)
{
  const uint32_t grid = std::min<uint32_t>(std::max<uint32_t>(mf->n_cells, 1), 4096u);
  hipLaunchKernelGGL(l2norm_kernel<n>, dim3(grid), dim3(n, n, n), 0, mf->stream, mf->d_l2g, mf->d_coords, mf->d_tab_gauss, mf->n_cells, u,
                     out);
  KERNEL_CHECK();
  return BP5_OK;
}
static int l2_dispatch(bp5_mf *mf, const double *u, double *out) { DISPATCH_N(launch_l2, mf, u, out); }
extern "C" int bp5_l2_norm_solution(bp5_mf *mf, const double *u, double *result)
{
  if (!mf || !u || !result) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  HIP_TRY(hipMemsetAsync(mf->d_scalar, 0, sizeof(double), mf->stream));
  BP5_TRY(l2_dispatch(mf, u, mf->d_scalar));
  if (mf->comm) BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_scalar, 1));
  double s = 0.0;
  HIP_TRY(hipMemcpyAsync(&s, mf->d_scalar, sizeof(double), hipMemcpyDeviceToHost, mf->stream));
  HIP_TRY(hipStreamSynchronize(mf->stream));
  *result = std::sqrt(s);
  return BP5_OK;
}

// ------------------------------------------------------------------------------------ BLAS-1
static inline int stream_grid(size_t n, int per_thread)
{
  size_t b = (n + (size_t)VB * per_thread - 1) / ((size_t)VB * per_thread);
  return (int)std::min<size_t>(std::max<size_t>(b, 1), MAXBLK);
}
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

extern "C" int bp5_vec_fill(bp5_mf *mf, double *v, double value, size_t n)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(v)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<0>, dim3(stream_grid(n, 2)), dim3(VB), 0, mf->stream, v, (const double *)nullptr, value, 0.0, n);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_vec_axpy(bp5_mf *mf, double *y, double a, const double *x, size_t n)
{
  if (!mf || !y || !x) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(y) || !aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<1>, dim3(stream_grid(n, 2)), dim3(VB), 0, mf->stream, y, x, 0.0, a, n);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_vec_equ(bp5_mf *mf, double *y, double a, const double *x, size_t n)
{
  if (!mf || !y || !x) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(y) || !aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<2>, dim3(stream_grid(n, 2)), dim3(VB), 0, mf->stream, y, x, 0.0, a, n);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_vec_sadd(bp5_mf *mf, double *y, double s, double a, const double *x, size_t n)
{
  if (!mf || !y || !x) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(y) || !aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<3>, dim3(stream_grid(n, 2)), dim3(VB), 0, mf->stream, y, x, s, a, n);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_vec_dot(bp5_mf *mf, const double *x, const double *y, size_t n, double *result)
{
  if (!mf || !y || !x || !result) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(y) || !aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  const int g = stream_grid(n, 2);
  hipLaunchKernelGGL(dot_kernel, dim3(g), dim3(VB), 0, mf->stream, x, y, n, mf->d_partials);
  hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(VB), 0, mf->stream, mf->d_partials, g, mf->d_scalar, (const int *)nullptr);
  KERNEL_CHECK();
  HIP_TRY(hipMemcpyAsync(result, mf->d_scalar, sizeof(double), hipMemcpyDeviceToHost, mf->stream));
  HIP_TRY(hipStreamSynchronize(mf->stream));
  return BP5_OK;
}

// global reductions of the distributed vector (owned entries of every rank): one on-stream all-reduce when a
// communicator is attached
static int reduce_to_host(bp5_mf *mf, int grid, double *result)
{
  hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(VB), 0, mf->stream, mf->d_partials, grid, mf->d_scalar, (const int *)nullptr);
  KERNEL_CHECK();
  if (mf->comm) NCCL_TRY(ncclAllReduce(mf->d_scalar, mf->d_scalar, 1, ncclDouble, ncclSum, mf->comm->comm, mf->stream));
  HIP_TRY(hipMemcpyAsync(result, mf->d_scalar, sizeof(double), hipMemcpyDeviceToHost, mf->stream));
  HIP_TRY(hipStreamSynchronize(mf->stream));
  return BP5_OK;
}
extern "C" int bp5_vec_l2_norm(bp5_mf *mf, const double *x, size_t n, double *result)
{
  if (!mf || !x || !result) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  HIP_TRY(hipSetDevice(mf->device));
  const int g = stream_grid(n, 2);
  hipLaunchKernelGGL(dot_kernel, dim3(g), dim3(VB), 0, mf->stream, x, x, n, mf->d_partials);
  double s = 0.0;
  BP5_TRY(reduce_to_host(mf, g, &s));
  *result = sqrt(s);
  return BP5_OK;
}
extern "C" int bp5_vec_all_zero(bp5_mf *mf, const double *x, size_t n, int *result)
{
  if (!mf || !x || !result) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  const int g = stream_grid(n, 1);
  hipLaunchKernelGGL(count_nonzero_kernel, dim3(g), dim3(VB), 0, mf->stream, x, n, mf->d_partials);
  double s = 0.0;
  BP5_TRY(reduce_to_host(mf, g, &s));
  *result = s == 0.0;
  return BP5_OK;
}

// ------------------------------------------------------------------------------------ RCCL
extern "C" int bp5_comm_unique_id(char *id)
{
  if (!id) return fail(BP5_ERR_INVALID, "null argument");
  static_assert(sizeof(ncclUniqueId) <= BP5_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId uid;
  NCCL_TRY(ncclGetUniqueId(&uid));
  memset(id, 0, BP5_UNIQUE_ID_BYTES);
  memcpy(id, &uid, sizeof(uid));
  return BP5_OK;
}
extern "C" int bp5_comm_create(const char *id, int rank, int n_ranks, bp5_comm **out)
{
  if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(BP5_ERR_INVALID, "bad argument");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  bp5_comm *c = new bp5_comm;
  c->rank = rank; c->n_ranks = n_ranks;
  ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, uid, rank);
  if (r != ncclSuccess) { delete c; return fail(BP5_ERR_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
  *out = c;
  return BP5_OK;
}
extern "C" int bp5_comm_destroy(bp5_comm *c)
{
  if (!c) return BP5_OK;
  if (c->comm) ncclCommDestroy(c->comm);
  delete c;
  return BP5_OK;
}
extern "C" int bp5_mf_set_comm(bp5_mf *mf, bp5_comm *comm)
{
  if (!mf) return fail(BP5_ERR_INVALID, "null handle");
  mf->comm = comm;
  return BP5_OK;
}
extern "C" int bp5_comm_allreduce_sum(bp5_mf *mf, double *buf, size_t n)
{
  if (!mf || !buf) return fail(BP5_ERR_INVALID, "null argument");
  if (!mf->comm) return BP5_OK; // (a one-rank communicator still goes through RCCL: the single-GPU tests exercise the call)
  NCCL_TRY(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, mf->comm->comm, mf->stream));
  return BP5_OK;
}
// ghost gather: owners send their interface values (packed through send_indices), ghosts are
// received straight into the vector's ghost range (contiguous per neighbour)
extern "C" int bp5_halo_gather(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->neighbors.empty()) return BP5_OK;
  if (!mf->comm) return fail(BP5_ERR_INVALID, "halo exchange needs bp5_mf_set_comm");
  const uint32_t ns = mf->send_off.back();
  if (ns) {
    hipLaunchKernelGGL(pack_kernel, dim3((ns + 255) / 256), dim3(256), 0, mf->stream, mf->d_send_idx, ns, v, mf->d_sendbuf);
    KERNEL_CHECK();
  }
  NCCL_TRY(ncclGroupStart());
  for (size_t k = 0; k < mf->neighbors.size(); ++k) {
    const uint32_t sc = mf->send_off[k + 1] - mf->send_off[k], rc = mf->recv_off[k + 1] - mf->recv_off[k];
    if (sc) NCCL_TRY(ncclSend(mf->d_sendbuf + mf->send_off[k], sc, ncclDouble, mf->neighbors[k], mf->comm->comm, mf->stream));
    if (rc) NCCL_TRY(ncclRecv(v + mf->n_owned + mf->recv_off[k], rc, ncclDouble, mf->neighbors[k], mf->comm->comm, mf->stream));
  }
  NCCL_TRY(ncclGroupEnd());
  return BP5_OK;
}
// compress(add): ghost contributions travel back to the owners and are added; ghosts zeroed
extern "C" int bp5_halo_scatter_add(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->neighbors.empty()) return BP5_OK;
  if (!mf->comm) return fail(BP5_ERR_INVALID, "halo exchange needs bp5_mf_set_comm");
  NCCL_TRY(ncclGroupStart());
  for (size_t k = 0; k < mf->neighbors.size(); ++k) {
    const uint32_t sc = mf->send_off[k + 1] - mf->send_off[k], rc = mf->recv_off[k + 1] - mf->recv_off[k];
    if (rc) NCCL_TRY(ncclSend(v + mf->n_owned + mf->recv_off[k], rc, ncclDouble, mf->neighbors[k], mf->comm->comm, mf->stream));
    if (sc) NCCL_TRY(ncclRecv(mf->d_recvbuf + mf->send_off[k], sc, ncclDouble, mf->neighbors[k], mf->comm->comm, mf->stream));
  }
  NCCL_TRY(ncclGroupEnd());
  for (size_t k = 0; k < mf->neighbors.size(); ++k) { // per neighbour: indices distinct -> race-free, fixed order
    const uint32_t sc = mf->send_off[k + 1] - mf->send_off[k];
    if (!sc) continue;
    hipLaunchKernelGGL(unpack_add_kernel, dim3((sc + 255) / 256), dim3(256), 0, mf->stream, mf->d_send_idx + mf->send_off[k], sc,
                       mf->d_recvbuf + mf->send_off[k], v);
    KERNEL_CHECK();
  }
  return bp5_halo_zero_ghosts(mf, v);
}
extern "C" int bp5_halo_zero_ghosts(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->n_ghost) HIP_TRY(hipMemsetAsync(v + mf->n_owned, 0, (size_t)mf->n_ghost * sizeof(double), mf->stream));
  return BP5_OK;
}
extern "C" int bp5_apply_distributed(bp5_mf *mf, const double *coef, double *src, double *dst, int zero_dst)
{
  if (!mf || !coef || !src || !dst) return fail(BP5_ERR_INVALID, "null argument");
  if (src == dst) return fail(BP5_ERR_INVALID, "src and dst must differ");
  HIP_TRY(hipSetDevice(mf->device));
  // stream order: ghost gather, all cells, scatter-add.  (Interior cells [0,n_interior) do not
  // read ghosts; the overlapped 3-phase schedule of SURVEY 3.2 is a later optimisation.)
  BP5_TRY(bp5_halo_gather(mf, src));
  BP5_TRY(launch_apply(mf, coef, src, dst, 0, mf->n_cells, zero_dst != 0));
  BP5_TRY(bp5_halo_scatter_add(mf, dst));
  BP5_TRY(bp5_halo_zero_ghosts(mf, src));
  return bp5_copy_constrained(mf, src, dst);
}

// ------------------------------------------------------------------------------------ events
extern "C" int bp5_event_create(bp5_event **out)
{
  if (!out) return fail(BP5_ERR_INVALID, "null argument");
  bp5_event *e = new bp5_event;
  hipError_t r = hipEventCreate(&e->ev);
  if (r != hipSuccess) { delete e; return fail(BP5_ERR_HIP, hipGetErrorString(r)); }
  *out = e;
  return BP5_OK;
}
extern "C" int bp5_event_record(bp5_mf *mf, bp5_event *ev)
{
  if (!mf || !ev) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipEventRecord(ev->ev, mf->stream));
  return BP5_OK;
}
extern "C" int bp5_event_elapsed_ms(bp5_event *a, bp5_event *b, double *ms)
{
  if (!a || !b || !ms) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipEventSynchronize(b->ev));
  float f = 0.f;
  HIP_TRY(hipEventElapsedTime(&f, a->ev, b->ev));
  *ms = f;
  return BP5_OK;
}
extern "C" int bp5_event_destroy(bp5_event *e)
{
  if (!e) return BP5_OK;
  hipEventDestroy(e->ev);
  delete e;
  return BP5_OK;
}

// ------------------------------------------------------------------------------------ CG
static int ensure_ws(bp5_mf *mf)
{
  if (mf->ws_g) return BP5_OK;
  // one allocation for the three work vectors (staggering their bases by 256 B ... 1 MiB was measured
  // to make no difference to the BLAS-1 kernels: profiles/r1/README.md)
  const size_t stagger = 0;
  const size_t nb = (std::max<size_t>(mf->n_local(), 2) * sizeof(double) + 4095) / 4096 * 4096;
  char *base = nullptr;
  HIP_TRY(hipMalloc((void **)&base, 3 * nb + 2 * stagger + 4096));
  HIP_TRY(hipMemset(base, 0, 3 * nb + 2 * stagger + 4096));
  mf->ws_base = base;
  mf->ws_g = (double *)base;
  mf->ws_d = (double *)(base + nb + stagger);
  mf->ws_h = (double *)(base + 2 * nb + 2 * stagger);
  return BP5_OK;
}

// four events per profiled application: [0] before the zero-fill, [1] before the cell kernel, [2] after the cell kernel
// (before the combine pass of the owner-scatter kernels), [3] after everything launch_apply enqueued
struct ApplyProfile {
  bp5_mf *mf;
  bool on;
  int used = 0;
  int mark(int k)
  {
    if (!on) return BP5_OK;
    while ((size_t)used + 4 > mf->ev_pool.size()) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); mf->ev_pool.push_back(e); }
    HIP_TRY(hipEventRecord(mf->ev_pool[used + k], mf->stream));
    return BP5_OK;
  }
};

// A.vmult(h, d) inside the solvers: dst already zero on entry when zeroed == true
static int solver_vmult(bp5_mf *mf, const double *coef, double *src, double *dst, bool zero, ApplyProfile &prof)
{
  const bool dist = mf->comm && !mf->neighbors.empty(); // halo exchange: whenever there are neighbours (tests: a self neighbour)
  if (dist) BP5_TRY(bp5_halo_gather(mf, src));
  // kernels that accumulate with atomics need a zeroed target (owner-scatter kernels define every entry themselves)
  const bool owner_scatter = variant_overwrites(mf, effective_variant(mf, 0, mf->n_cells));
  BP5_TRY(prof.mark(0));
  if (zero && !owner_scatter) {
    HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    zero = false;
  }
  BP5_TRY(prof.mark(1));
  if (prof.on) mf->prof_mark = mf->ev_pool[prof.used + 2];
  const int st = launch_apply(mf, coef, src, dst, 0, mf->n_cells, zero);
  const bool marked = prof.on && mf->prof_mark == nullptr;
  mf->prof_mark = nullptr;
  BP5_TRY(st);
  if (!marked) BP5_TRY(prof.mark(2));
  BP5_TRY(prof.mark(3));
  if (prof.on) prof.used += 4;
  if (dist) { BP5_TRY(bp5_halo_scatter_add(mf, dst)); BP5_TRY(bp5_halo_zero_ghosts(mf, src)); }
  return bp5_copy_constrained(mf, src, dst);
}

static int poll_state(bp5_mf *mf)
{
  HIP_TRY(hipMemcpyAsync(mf->h_st, mf->d_st, ST_COUNT * sizeof(int), hipMemcpyDeviceToHost, mf->stream));
  HIP_TRY(hipMemcpyAsync(mf->h_sc, mf->d_sc, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, mf->stream));
  HIP_TRY(hipStreamSynchronize(mf->stream));
  return BP5_OK;
}

extern "C" int bp5_cg_solve(bp5_mf *mf, const double *coef, const double *diag, const double *b, double *x, const bp5_cg_params *prm,
                            bp5_cg_result *res)
{
  if (!mf || (!coef && mf->geometry_mode != BP5_GEOM_AFFINE) || !b || !x || !prm || !res) return fail(BP5_ERR_INVALID, "null argument");
  if (prm->max_iter < 0) return fail(BP5_ERR_INVALID, "max_iter < 0");
  if (prm->variant != BP5_CG_PLAIN && prm->variant != BP5_CG_MERGED) return fail(BP5_ERR_INVALID, "unknown CG variant");
  if (!aligned16(b) || !aligned16(x) || (diag && !aligned16(diag))) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  HIP_TRY(hipSetDevice(mf->device));
  BP5_TRY(ensure_ws(mf));
  const size_t n = mf->n_owned;
  const int grid2 = stream_grid(n, 2), grid1 = stream_grid(n, 1);
  hipStream_t s = mf->stream;
  double *g = mf->ws_g, *d = mf->ws_d, *h = mf->ws_h;
  ApplyProfile prof{mf, prm->profile != 0};
  hipEvent_t ev0, ev1;
  HIP_TRY(hipEventCreate(&ev0));
  HIP_TRY(hipEventCreate(&ev1));
  // scalars: tolerance + iteration cap
  mf->h_sc[SC_TOL] = prm->abs_tol;
  HIP_TRY(hipMemcpyAsync(mf->d_sc + SC_TOL, mf->h_sc + SC_TOL, sizeof(double), hipMemcpyHostToDevice, s));
  mf->h_st[ST_MAXIT] = prm->max_iter;
  HIP_TRY(hipMemcpyAsync(mf->d_st + ST_MAXIT, mf->h_st + ST_MAXIT, sizeof(int), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s)); // pinned staging words are reused below
  HIP_TRY(hipEventRecord(ev0, s));
  const bool plain = prm->variant == BP5_CG_PLAIN;
  const int check = prm->check_every;
  int status = BP5_OK;

  if (plain) {
    // g = -b, d = -D g, x = 0   (x0 = 0 short-circuit, bp5/solver.h:375-381)
    hipLaunchKernelGGL(cg_init_kernel, dim3(grid1), dim3(VB), 0, s, b, diag, x, g, d, n, mf->d_partials);
    hipLaunchKernelGGL(finalize_kernel<2>, dim3(2), dim3(VB), 0, s, mf->d_partials, grid1, mf->d_sc + SC_GG, (const int *)nullptr);
    KERNEL_CHECK();
    BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_GG, 2));
    hipLaunchKernelGGL(cg_init_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
    KERNEL_CHECK();
    for (int it = 1; it <= prm->max_iter; ++it) {
      BP5_TRY(solver_vmult(mf, coef, d, h, true, prof));
      hipLaunchKernelGGL(dot_kernel, dim3(grid2), dim3(VB), 0, s, d, h, n, mf->d_partials);
      hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(VB), 0, s, mf->d_partials, grid2, mf->d_sc + SC_DH, mf->d_st);
      KERNEL_CHECK();
      BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_DH, 1));
      hipLaunchKernelGGL(cg_update_kernel, dim3(grid2), dim3(VB), 0, s, x, g, d, h, diag, n, mf->d_sc, mf->d_st, mf->d_partials);
      hipLaunchKernelGGL(finalize_kernel<2>, dim3(2), dim3(VB), 0, s, mf->d_partials, grid2, mf->d_sc + SC_GG, mf->d_st);
      KERNEL_CHECK();
      BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_GG, 2));
      hipLaunchKernelGGL(cg_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
      hipLaunchKernelGGL(cg_direction_kernel, dim3(grid2), dim3(VB), 0, s, d, g, diag, n, mf->d_sc, mf->d_st);
      KERNEL_CHECK();
      if (check > 0 && it % check == 0 && it < prm->max_iter) {
        BP5_TRY(poll_state(mf));
        if (mf->h_st[ST_DONE]) break;
      }
    }
  } else {
    // SolverCGFullMerge: g == r, d == p, h == v
    hipLaunchKernelGGL(cgm_init_kernel, dim3(grid1), dim3(VB), 0, s, b, x, g, d, h, n, mf->d_partials);
    hipLaunchKernelGGL(finalize_kernel<2>, dim3(2), dim3(VB), 0, s, mf->d_partials, grid1, mf->d_sc + SC_GG, (const int *)nullptr);
    KERNEL_CHECK();
    BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_GG, 2));
    hipLaunchKernelGGL(cgm_init_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
    KERNEL_CHECK();
    int it = 1;
    for (; it <= prm->max_iter; ++it) {
      if (it == 1) hipLaunchKernelGGL(cgm_update_kernel<0>, dim3(grid2), dim3(VB), 0, s, d, g, h, x, diag, n, mf->d_sc, mf->d_st);
      else if (it % 2 == 0) hipLaunchKernelGGL(cgm_update_kernel<1>, dim3(grid2), dim3(VB), 0, s, d, g, h, x, diag, n, mf->d_sc, mf->d_st);
      else hipLaunchKernelGGL(cgm_update_kernel<2>, dim3(grid2), dim3(VB), 0, s, d, g, h, x, diag, n, mf->d_sc, mf->d_st);
      KERNEL_CHECK();
      BP5_TRY(solver_vmult(mf, coef, d, h, true, prof)); // overwrite mode: v needs no zeroing (the reference zeroes it in update_a*)
      hipLaunchKernelGGL(cgm_dots_kernel, dim3(grid2), dim3(VB), 0, s, d, g, h, diag, n, mf->d_st, mf->d_partials);
      hipLaunchKernelGGL(finalize_kernel<7>, dim3(7), dim3(VB), 0, s, mf->d_partials, grid2, mf->d_sc + SC_R0, mf->d_st);
      KERNEL_CHECK();
      BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_R0, 7));
      hipLaunchKernelGGL(cgm_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
      KERNEL_CHECK();
      if (check > 0 && it % check == 0 && it < prm->max_iter) {
        BP5_TRY(poll_state(mf));
        if (mf->h_st[ST_DONE]) break;
      }
    }
    // epilogue x update (solver.h:510-526) runs inside the next update kernel; with max_iter == 0 no
    // iteration has been done and nothing is pending
    if (prm->max_iter > 0) {
      hipLaunchKernelGGL(cgm_update_kernel<1>, dim3(grid2), dim3(VB), 0, s, d, g, h, x, diag, n, mf->d_sc, mf->d_st);
      hipLaunchKernelGGL(cgm_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
      KERNEL_CHECK();
    }
  }
  HIP_TRY(hipEventRecord(ev1, s));
  BP5_TRY(poll_state(mf));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  hipEventDestroy(ev0);
  hipEventDestroy(ev1);
  res->iterations = mf->h_st[ST_ITER];
  res->residual = mf->h_sc[SC_RES];
  res->initial_residual = mf->h_sc[SC_RES0];
  res->solve_ms = ms;
  res->apply_ms_avg = res->operator_ms_avg = 0.0;
  res->apply_launches = prof.used / 4;
  if (prof.on && prof.used) {
    double tot = 0.0, tot_op = 0.0;
    for (int k = 0; k < prof.used; k += 4) {
      float t = 0.f;
      HIP_TRY(hipEventElapsedTime(&t, mf->ev_pool[k + 1], mf->ev_pool[k + 2]));
      tot += t;
      HIP_TRY(hipEventElapsedTime(&t, mf->ev_pool[k], mf->ev_pool[k + 3]));
      tot_op += t;
    }
    res->apply_ms_avg = tot / (prof.used / 4);
    res->operator_ms_avg = tot_op / (prof.used / 4);
  }
  if (mf->h_st[ST_BREAKDOWN]) status = fail(BP5_ERR_BREAKDOWN, "CG breakdown: p.Ap is zero or NaN");
  return status;
}
