// C-ABI implementation: MatrixFree handle, operator launches, BLAS-1, CG drivers, RCCL halo.
// Every entry point cites the reference interface it replaces in include/bp5.h.
#include "bp5_device.hpp"


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
  HIP_TRY(hipStreamSynchronize(nullptr)); // the fill runs on the null stream; callers use it on (possibly non-blocking) streams
  return BP5_OK;
}
extern "C" int bp5_vec_free(double *v) { HIP_TRY(hipFree(v)); return BP5_OK; }
extern "C" int bp5_copy_h2d(void *dst, const void *src, size_t bytes) { HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return BP5_OK; }
extern "C" int bp5_copy_d2h(void *dst, const void *src, size_t bytes) { HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return BP5_OK; }

// ------------------------------------------------------------------------------------ create / destroy
static void tuning_from_environment(bp5_mf *mf);
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
  struct Guard { // every early return below releases the handle and whatever has been uploaded so far
    bp5_mf *m;
    ~Guard() { if (m) bp5_mf_destroy(m); }
  } guard{mf};
  mf->degree = d->degree; mf->quadrature = d->quadrature; mf->coefficient = d->coefficient;
  mf->n = d->degree + 1; mf->n3 = mf->n * mf->n * mf->n; mf->device = d->device;
  mf->n_cells = d->n_cells; mf->n_interior = d->n_interior_cells; mf->n_owned = d->n_owned; mf->n_ghost = d->n_ghost;
  mf->n_constrained = d->n_constrained;
  mf->tab = tab; mf->tab_gauss = tabg;
  tuning_from_environment(mf);
  { // metric layout (A/B knob: BP5_COEF_LAYOUT = plane | cell)
    const char *e = getenv("BP5_COEF_LAYOUT");
    const bool cell_major = e && !strcmp(e, "cell");
    mf->coef_plane_stride = cell_major ? (uint64_t)mf->n3 : (uint64_t)mf->n_cells * mf->n3;
    mf->coef_cell_stride = cell_major ? (uint64_t)6 * mf->n3 : (uint64_t)mf->n3;
  }
  // validate indices on the host: a bad index would fault on the GPU
  const size_t nl = (size_t)d->n_cells * mf->n3, nloc = mf->n_local();
  for (size_t s = 0; s < nl; ++s)
    if (d->local_to_global_host[s] >= nloc) { return fail(BP5_ERR_INVALID, "local_to_global entry out of range"); }
  for (uint32_t s = 0; s < d->n_constrained; ++s)
    if (d->constrained_host[s] >= nloc) { return fail(BP5_ERR_INVALID, "constrained index out of range"); }
  { // are the DoFs strictly inside a cell numbered ahead of all others, cell after cell, x fastest?  (bp5_mesh_desc.dof_numbering = 2, or any mesh numbered so)
    const int p = d->degree, n = p + 1;
    const uint32_t per = (uint32_t)((p - 1) * (p - 1) * (p - 1));
    bool ok = p >= 2 && d->n_cells > 0 && (uint64_t)d->n_cells * per <= d->n_owned;
    for (uint32_t c = 0; ok && c < d->n_cells; ++c) {
      const uint32_t *l = d->local_to_global_host + (size_t)c * mf->n3;
      uint32_t e = c * per;
      for (int k = 1; ok && k < p; ++k)
        for (int j = 1; j < p; ++j)
          for (int i = 1; i < p; ++i, ++e)
            if (l[i + n * (j + n * k)] != e) ok = false;
    }
    mf->cell_interiors_first = ok;
  }
  mf->stream = (hipStream_t)d->stream; // NULL == the HIP default stream (ordered with the host's other default-stream work)
  BP5_TRY(upload(&mf->d_l2g, d->local_to_global_host, nl));
  mf->h_l2g.assign(d->local_to_global_host, d->local_to_global_host + nl);
  if (d->n_cell_blocks && d->cell_block_offsets_host) {
    const uint32_t *o = d->cell_block_offsets_host;
    bool ok = o[0] == 0 && o[d->n_cell_blocks] == d->n_cells;
    for (uint32_t b = 0; ok && b < d->n_cell_blocks; ++b) ok = o[b] < o[b + 1];
    if (!ok) { return fail(BP5_ERR_INVALID, "cell_block_offsets must ascend from 0 to n_cells"); }
    mf->h_block_off.assign(o, o + d->n_cell_blocks + 1);
  }
  BP5_TRY(upload(&mf->d_coords, d->node_coords_host, nloc * 3));
  BP5_TRY(upload(&mf->d_constrained, d->constrained_host, d->n_constrained));
  mf->h_constrained.assign(nloc, false);
  for (uint32_t s = 0; s < d->n_constrained; ++s) mf->h_constrained[d->constrained_host[s]] = true;
  {
    std::vector<uint32_t> bits((size_t)d->n_owned / 32 + 2, 0u);
    for (uint32_t s = 0; s < d->n_constrained; ++s)
      if (d->constrained_host[s] < d->n_owned) bits[d->constrained_host[s] >> 5] |= 1u << (d->constrained_host[s] & 31);
    BP5_TRY(upload(&mf->d_constrained_bits, bits.data(), bits.size()));
  }
  std::vector<double> tv;
  pack_tab(tab, tv);  BP5_TRY(upload(&mf->d_tab, tv.data(), tv.size()));
  pack_tab(tabg, tv); BP5_TRY(upload(&mf->d_tab_gauss, tv.data(), tv.size()));
  // hanging nodes: validate the masks on the host, upload them and the two 1-D interpolation matrices
  if (d->constraint_mask_host) {
    std::vector<uint32_t> hm(d->constraint_mask_host, d->constraint_mask_host + d->n_cells);
    for (uint32_t m : hm) {
      if (!m) continue;
      if (m >> 12) return fail(BP5_ERR_INVALID, "constraint_mask: unknown bits");
      if (!(m & 0xe07u)) return fail(BP5_ERR_INVALID, "constraint_mask: position bits without a constrained face or edge");
      for (int e = 0; e < 3; ++e) // one position per direction: it locates faces / edges AND selects the half of the coarse entity
        if (((m >> (3 + e)) & 1u) != ((m >> (6 + e)) & 1u)) {
          // planar round-2 masks name only what they use (SIDE of the face's normal, HALF of its two tangential directions)
          const int e1 = e == 0 ? 1 : 0, e2 = e == 2 ? 1 : 2;
          const bool side_used = ((m >> e) & 1u) || ((m >> (9 + e1)) & 1u) || ((m >> (9 + e2)) & 1u);
          const bool half_used = ((m >> e1) & 1u) || ((m >> e2) & 1u) || ((m >> (9 + e)) & 1u);
          if (side_used && half_used) return fail(BP5_ERR_INVALID, "constraint_mask: BP5_HANG_SIDE_d and BP5_HANG_HALF_d disagree");
        }
      mf->has_hanging = true;
    }
    if (mf->has_hanging) {
      BP5_TRY(upload(&mf->d_hang_mask, hm.data(), hm.size()));
      const int n = mf->n;
      std::vector<double> I(2 * n * n);
      for (int h = 0; h < 2; ++h)
        for (int a = 0; a < n; ++a) {
          const long double x = 0.5L * tab.nodes[a] + 0.5L * h;
          for (int b = 0; b < n; ++b) { // Lagrange polynomial of node b at x
            long double num = 1, den = 1;
            for (int c = 0; c < n; ++c) if (c != b) { num *= x - (long double)tab.nodes[c]; den *= (long double)tab.nodes[b] - (long double)tab.nodes[c]; }
            I[(h * n + a) * n + b] = (double)(num / den);
          }
        }
      BP5_TRY(upload(&mf->d_hang_I, I.data(), I.size()));
    }
  }
  // halo plan
  if (d->n_neighbors > 0) {
    if (!d->neighbor_rank_host || !d->send_offsets_host || !d->recv_offsets_host) { return fail(BP5_ERR_INVALID, "halo plan arrays missing"); }
    mf->neighbors.assign(d->neighbor_rank_host, d->neighbor_rank_host + d->n_neighbors);
    mf->send_off.assign(d->send_offsets_host, d->send_offsets_host + d->n_neighbors + 1);
    mf->recv_off.assign(d->recv_offsets_host, d->recv_offsets_host + d->n_neighbors + 1);
    const uint32_t ns = mf->send_off.back();
    if (mf->recv_off.back() != d->n_ghost) { return fail(BP5_ERR_INVALID, "recv ranges must cover the ghost range"); }
    for (uint32_t s = 0; s < ns; ++s)
      if (d->send_indices_host[s] >= d->n_owned) { return fail(BP5_ERR_INVALID, "send index out of owned range"); }
    BP5_TRY(upload(&mf->d_send_idx, d->send_indices_host, ns));
    std::vector<uint8_t> sd(ns);
    for (uint32_t s = 0; s < ns; ++s) sd[s] = mf->h_constrained[d->send_indices_host[s]] ? 1 : 0;
    BP5_TRY(upload(&mf->d_send_dirichlet, sd.data(), sd.size()));
    HIP_TRY(hipMalloc((void **)&mf->d_sendbuf, std::max<size_t>(ns, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&mf->d_recvbuf, std::max<size_t>(ns, 1) * sizeof(double)));
  } else if (d->n_ghost) { return fail(BP5_ERR_INVALID, "ghosts without a halo plan"); }
  // solver workspace
  HIP_TRY(hipMalloc((void **)&mf->d_partials, 8 * PARTIAL_STRIDE * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&mf->d_sc, SC_COUNT * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&mf->d_scalar, 8 * sizeof(double)));
  HIP_TRY(hipMalloc((void **)&mf->d_st, ST_COUNT * sizeof(int)));
  HIP_TRY(hipMemset(mf->d_sc, 0, SC_COUNT * sizeof(double)));
  HIP_TRY(hipMemset(mf->d_st, 0, ST_COUNT * sizeof(int)));
  HIP_TRY(hipStreamSynchronize(nullptr)); // null-stream fills must not race with work on a non-blocking handle stream
  HIP_TRY(hipHostMalloc((void **)&mf->h_sc, SC_COUNT * sizeof(double)));
  HIP_TRY(hipHostMalloc((void **)&mf->h_st, ST_COUNT * sizeof(int)));
  HIP_TRY(hipEventCreate(&mf->ev_solve[0]));
  HIP_TRY(hipEventCreate(&mf->ev_solve[1]));
  guard.m = nullptr;
  *out = mf;
  return BP5_OK;
}

extern "C" int bp5_mf_destroy(bp5_mf *mf)
{
  if (!mf) return BP5_OK;
  hipSetDevice(mf->device);
  hipStreamSynchronize(mf->stream);
  void *ptrs[] = {mf->d_constrained_bits, mf->d_l2g, mf->d_constrained, mf->d_send_idx, mf->d_coords, mf->d_tab, mf->d_tab_gauss, mf->d_l2g_padded,
                  mf->d_constraint_mask, mf->d_inv_jac, mf->d_JxW, mf->d_qpoints, mf->d_sendbuf, mf->d_recvbuf, mf->d_partials,
                  mf->d_sc, mf->d_scalar, mf->d_st, mf->ws_base, mf->d_stamps, mf->d_evec, mf->d_scalar_plane, mf->d_gcell, mf->d_hang_mask, mf->d_hang_I, mf->d_send_dirichlet, mf->d_signal};
  for (void *p : ptrs) if (p) hipFree(p);
  if (mf->h_sc) hipHostFree(mf->h_sc);
  if (mf->h_st) hipHostFree(mf->h_st);
  for (hipEvent_t e : mf->ev_pool) hipEventDestroy(e);
  for (hipEvent_t e : mf->phase.ev) hipEventDestroy(e);
  for (hipEvent_t e : mf->ev_solve) if (e) hipEventDestroy(e);
  for (hipEvent_t e : mf->ev_halo) if (e) hipEventDestroy(e);
  if (mf->comm_stream) { hipStreamSynchronize(mf->comm_stream); hipStreamDestroy(mf->comm_stream); }
  for (auto &kv : mf->march_plans) { hipFree(kv.second.team_off); hipFree(kv.second.entries); }
  for (auto &kv : mf->plans) {
    auto &q = kv.second;
    void *pp[] = {q.off, q.dofs, q.sh_dof, q.sh_off, q.sh_slot, q.pos, q.cell_round, q.team_rounds, q.partial, q.cell_off, q.pass_cell, q.pass_off, q.run_off, q.runs, q.gidx, q.packed, q.lattice, q.cell_pos, q.cr_start, q.cr_dof0, q.cr_soff, q.cr_slots, q.cr_tile};
    for (void *x : pp) if (x) hipFree(x);
    if (q.wg_blocks) { for (auto &w : *q.wg_blocks) hipFree(w.second); delete q.wg_blocks; }
    if (q.cr_carry) {
      for (auto &c : *q.cr_carry) { void *cc[] = {c.second.start, c.second.dof0, c.second.soff, c.second.slots, c.second.tile}; for (void *x : cc) if (x) hipFree(x); }
      delete q.cr_carry;
    }
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
  *n = (size_t)mf->n_planes() * mf->n_cells * mf->n3;
  mf->coef_planes_committed = mf->n_planes();
  return BP5_OK;
}
extern "C" int bp5_mf_set_operator(bp5_mf *mf, int op)
{
  if (!mf || (op != BP5_OP_POISSON && op != BP5_OP_HELMHOLTZ)) return fail(BP5_ERR_INVALID, "unknown operator");
  if (op == BP5_OP_HELMHOLTZ && (mf->has_hanging || mf->geometry_mode == BP5_GEOM_AFFINE))
    return fail(BP5_ERR_UNSUPPORTED, "the Helmholtz operator needs a conforming mesh and the six-plane geometry (hanging nodes: the facade's FEEvaluation)");
  if (op == BP5_OP_HELMHOLTZ && mf->apply_variant != 0 && mf->apply_variant != 56) return fail(BP5_ERR_UNSUPPORTED, "the Helmholtz operator runs apply variants 0 and 56");
  if (mf->coef_planes_committed && mf->coef_planes_committed != (op == BP5_OP_HELMHOLTZ ? 7 : 6))
    return fail(BP5_ERR_INVALID, "the metric array of this handle has been sized or filled for another plane count: set the operator before bp5_mf_coef_size / bp5_mf_compute_merged_metric");
  mf->operator_kind = op;
  mf->auto_block = -1; // (decided per operator: the Helmholtz build of the block kernel runs two workgroups per CU)
  return BP5_OK;
}
// the variants of the product library: every one of them computes the operator (they differ in launch shape, staging and
// scatter strategy).  The timing-only ablation builds (wrong results by construction) exist only in libbp5_timing.so
// (make timing, -DBP5_TIMING_BUILDS; used by tools/bench_apply.py) and are refused here.
static bool product_variant(int degree, int v)
{
  if (v == 0) return true;
  if (v >= 100) return v < 200 && product_variant(degree, v - 100) && v - 100 >= 10 && v - 100 <= 14; // team kernel, atomic scatter
  if (v == 10 || v == 50 || v == 70) return true;
  if (v == 56 && block_lpc(degree) != 0) return true;
  switch (degree) {
    case 1: case 3: return v == 1;
    case 4: return (v >= 1 && v <= 6) || (v >= 11 && v <= 14) || (v >= 48 && v <= 62) || v == 71 || v == 72;
    case 5: return v >= 1 && v <= 3;
    case 6: return v >= 1 && v <= 5;
    case 7: case 8: return (v >= 1 && v <= 3) || v == 5;
  }
  return false;
}
extern "C" int bp5_mf_set_apply_variant(bp5_mf *mf, int v)
{
  if (!mf) return fail(BP5_ERR_INVALID, "null handle");
  if (mf->has_hanging) { // 0: the library decides; 56: block kernel (deterministic; needs cell blocks); 90: pencil kernel with atomics (any mesh)
    if (v != 0 && v != 90 && !(v == 56 && block_lpc(mf->degree) != 0)) return fail(BP5_ERR_UNSUPPORTED, "meshes with hanging nodes run apply variants 90 (pencil kernel) and 56 (block kernel)");
    if (v == 56 && mf->geometry_mode == BP5_GEOM_AFFINE) return fail(BP5_ERR_UNSUPPORTED, "hanging nodes in the affine geometry mode run the pencil kernel: apply variants 0 and 90");
    mf->apply_variant = v;
    return BP5_OK;
  }
  if (v == 90) return fail(BP5_ERR_INVALID, "apply variant 90 is the hanging-node kernel: the mesh has no constraint masks");
#ifndef BP5_TIMING_BUILDS
  if (mf->operator_kind == BP5_OP_HELMHOLTZ && v != 0 && !(v == 56 && block_lpc(mf->degree) != 0)) return fail(BP5_ERR_UNSUPPORTED, "the Helmholtz operator runs apply variants 0 (pencil kernel) and 56 (block kernel)");
#endif
#ifndef BP5_TIMING_BUILDS
  if (!product_variant(mf->degree, v)) return fail(BP5_ERR_INVALID, "unknown (degree, apply variant): timing-only builds live in libbp5_timing.so");
#endif
  mf->apply_variant = v;
  return BP5_OK;
}

static int effective_variant(bp5_mf *mf, uint32_t c0, uint32_t c1);
extern "C" int bp5_mf_set_cg_fusion(bp5_mf *mf, int on)
{
  if (!mf) return fail(BP5_ERR_INVALID, "null handle");
  mf->cg_fusion = on != 0;
  return BP5_OK;
}
extern "C" int bp5_mf_set_block_workgroups(bp5_mf *mf, int max_workgroups)
{
  if (!mf || max_workgroups < 0) return fail(BP5_ERR_INVALID, "bad argument");
  mf->block_max_wg = max_workgroups;
  return BP5_OK;
}
extern "C" int bp5_mf_set_streaming(bp5_mf *mf, int policy)
{
  if (!mf || policy < -1 || policy > 1) return fail(BP5_ERR_INVALID, "bad argument");
  mf->streaming = policy;
  return BP5_OK;
}
extern "C" int bp5_mf_block_plan_info(bp5_mf *mf, uint32_t *n_blocks, uint32_t *max_runs, int *packed_indices)
{
  if (!mf || !n_blocks || !max_runs || !packed_indices) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan_raw(mf, -block_cpt(mf), &dp, 64));
  *n_blocks = dp->n_groups;
  *max_runs = dp->max_runs;
  *packed_indices = dp->packed != nullptr;
  return BP5_OK;
}
extern "C" int bp5_mf_block_plan_lattice(bp5_mf *mf, uint32_t *n_lattice_blocks)
{
  if (!mf || !n_lattice_blocks) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan_raw(mf, -block_cpt(mf), &dp, 64));
  *n_lattice_blocks = dp->n_lattice_blocks;
  return BP5_OK;
}
extern "C" int bp5_mf_block_plan_carry(bp5_mf *mf, uint32_t *n_faces, uint32_t *n_shared, uint32_t *n_shared_last_launch)
{
  if (!mf || !n_faces || !n_shared || !n_shared_last_launch) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan_raw(mf, -block_cpt(mf), &dp, 64));
  *n_faces = 0;
  for (uint32_t len : dp->h_carry_len) *n_faces += len != 0;
  *n_shared = dp->n_shared;
  *n_shared_last_launch = dp->cr_active ? dp->cr_active->n_shared : dp->n_shared;
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
  o.hang_mask = mf->has_hanging ? mf->d_hang_mask : nullptr;
  o.hang_I = mf->d_hang_I;
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
  if (mode == BP5_GEOM_AFFINE && mf->operator_kind == BP5_OP_HELMHOLTZ) return fail(BP5_ERR_UNSUPPORTED, "the Helmholtz operator needs the six-plane geometry");
  if (mode == BP5_GEOM_AFFINE && mf->has_hanging && mf->apply_variant == 56) return fail(BP5_ERR_UNSUPPORTED, "hanging nodes in the affine geometry mode run the pencil kernel: set apply variant 0 or 90 first");
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
  o.plane_stride = mf->coef_plane_stride; o.cell_stride = mf->coef_cell_stride;
  o.helmholtz = mf->operator_kind == BP5_OP_HELMHOLTZ;
  mf->coef_planes_committed = mf->n_planes();
  if (o.helmholtz && mf->coef_cell_stride != (uint64_t)mf->n3) return fail(BP5_ERR_UNSUPPORTED, "the Helmholtz operator needs the plane-major metric layout");
  DISPATCH_N(launch_geometry, mf, o);
}

template <int n>
static int launch_permute(bp5_mf *mf, const double *in, double *out)
{
  const uint64_t total = (uint64_t)mf->n_planes() * mf->n_cells * mf->n3;
  hipLaunchKernelGGL(metric_permute_kernel<n>, dim3(2048), dim3(256), 0, mf->stream, in, out, total, (uint64_t)mf->n_cells, mf->coef_plane_stride,
                     mf->coef_cell_stride);
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
    if (mf->has_hanging) HIP_TRY(hipMemcpyAsync(mf->d_constraint_mask, mf->d_hang_mask, mf->n_cells * sizeof(uint32_t), hipMemcpyDeviceToDevice, mf->stream));
    else HIP_TRY(hipMemsetAsync(mf->d_constraint_mask, 0, mf->n_cells * sizeof(uint32_t), mf->stream));
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

// Structured cell blocks ("lattice" blocks).  A block is a lattice block when (1) its cells form a full box of bx x by x bz cells that are
// face neighbours with parallel local axes, and (2) its DoFs are numbered entity by entity: the DoFs of each of the 27 lattice entities of
// the box (interior, 6 faces, 12 edges, 8 corners of its (bx p + 1) x (by p + 1) x (bz p + 1) node lattice) are consecutive, x fastest --
// what a brick-major numbering (bp5_mesh_create_brick: dof_numbering = 1) produces.  The list slot and the DoF of every cell-local entry
// then follow in closed form from the cell's position (cx, cy, cz) and the entry's (i, j, k):  I = cx p + i, ..., entity = class(I) +
// 3 class(J) + 9 class(K) (class: 0 at the low plane, 2 at the high plane, 1 between), offset = interior coordinates in mixed radix --
// the kernels compute both instead of reading 2 bytes per entry (2 r bytes per DoF of HBM traffic).  Recognition is topological (shared
// face corners), the numbering is verified entry by entry; any block that fails keeps the packed stream.
// out `lat`: [n_blocks][BLOCK_LATTICE_WORDS]: slots of the 27 entities, their first DoFs, then bx | by << 8 | bz << 16 | 1 << 31 (0: no lattice).
static uint32_t detect_lattice_blocks(const uint32_t *l2g, int n, const std::vector<uint32_t> &cell_off, const std::vector<uint32_t> &list_off,
                                      const std::vector<uint32_t> &list_dofs, std::vector<uint32_t> &lat, std::vector<uint16_t> &cpos)
{
  const int p = n - 1, n2 = n * n, n3 = n2 * n;
  const size_t n_blocks = cell_off.size() - 1;
  lat.assign(n_blocks * BLOCK_LATTICE_WORDS, 0u);
  cpos.assign(cell_off.back(), 0);
  uint32_t n_ok = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : n_ok)
  for (int64_t g = 0; g < (int64_t)n_blocks; ++g) {
    const uint32_t c0 = cell_off[g], nc = cell_off[g + 1] - c0;
    if (nc == 0 || nc > 4096) continue;
    auto L = [&](uint32_t c, int i, int j, int k) { return l2g[(size_t)(c0 + c) * n3 + i + n * (j + n * k)]; };
    // faces by their four corner DoFs: face f = 2 d + side of cell c
    auto face_key = [&](uint32_t c, int d, int side, uint32_t (&key)[4]) {
      int q = 0;
      for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
          int ijk[3];
          ijk[d] = side * p;
          ijk[(d + 1) % 3] = a * p;
          ijk[(d + 2) % 3] = b * p;
          key[q++] = L(c, ijk[0], ijk[1], ijk[2]);
        }
    };
    // neighbours: sort the 6 nc faces by their keys; a +side face of one cell and the -side face of another with the same corners meet
    struct Face { uint32_t k[4]; uint32_t cell; int d, side; };
    std::vector<Face> faces;
    faces.reserve(6 * (size_t)nc);
    for (uint32_t c = 0; c < nc; ++c)
      for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
          Face f;
          face_key(c, d, side, f.k);
          f.cell = c; f.d = d; f.side = side;
          faces.push_back(f);
        }
    std::sort(faces.begin(), faces.end(), [](const Face &a, const Face &b) { return std::lexicographical_compare(a.k, a.k + 4, b.k, b.k + 4); });
    std::vector<uint32_t> nbr(6 * (size_t)nc, UINT32_MAX); // [cell][2 d + side]
    for (size_t f = 0; f + 1 < faces.size(); ++f) {
      const Face &a = faces[f], &b = faces[f + 1];
      if (std::equal(a.k, a.k + 4, b.k) && a.d == b.d && a.side != b.side && a.cell != b.cell) {
        nbr[6 * (size_t)a.cell + 2 * a.d + a.side] = b.cell;
        nbr[6 * (size_t)b.cell + 2 * b.d + b.side] = a.cell;
      }
    }
    std::vector<int> pos(3 * (size_t)nc, INT32_MIN);
    std::vector<uint32_t> queue{0};
    pos[0] = pos[1] = pos[2] = 0;
    bool ok = true;
    for (size_t qh = 0; qh < queue.size() && ok; ++qh) { // breadth-first over face neighbours
      const uint32_t c = queue[qh];
      for (int d = 0; d < 3 && ok; ++d)
        for (int side = 0; side < 2 && ok; ++side) {
          const uint32_t o = nbr[6 * (size_t)c + 2 * d + side];
          if (o == UINT32_MAX) continue;
          int want[3] = {pos[3 * c], pos[3 * c + 1], pos[3 * c + 2]};
          want[d] += side ? 1 : -1;
          if (pos[3 * o] == INT32_MIN) {
            pos[3 * o] = want[0]; pos[3 * o + 1] = want[1]; pos[3 * o + 2] = want[2];
            queue.push_back(o);
          } else if (pos[3 * o] != want[0] || pos[3 * o + 1] != want[1] || pos[3 * o + 2] != want[2])
            ok = false;
        }
    }
    if (!ok || queue.size() != nc) continue;
    int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (uint32_t c = 0; c < nc; ++c)
      for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], pos[3 * c + d]); hi[d] = std::max(hi[d], pos[3 * c + d]); }
    const int bx = hi[0] - lo[0] + 1, by = hi[1] - lo[1] + 1, bz = hi[2] - lo[2] + 1;
    if ((int64_t)bx * by * bz != (int64_t)nc || bx > 15 || by > 15 || bz > 15) continue;
    std::vector<uint32_t> cell_at((size_t)nc, UINT32_MAX);
    for (uint32_t c = 0; c < nc && ok; ++c) {
      const size_t at = (pos[3 * c] - lo[0]) + (size_t)bx * ((pos[3 * c + 1] - lo[1]) + (size_t)by * (pos[3 * c + 2] - lo[2]));
      if (cell_at[at] != UINT32_MAX) ok = false;
      cell_at[at] = c;
    }
    if (!ok) continue;
    const int N[3] = {bx * p, by * p, bz * p};
    auto cls = [&](int X, int d) { return X == 0 ? 0 : X == N[d] ? 2 : 1; };
    auto dof_at = [&](int I, int J, int K) { // the DoF at a lattice point, through any cell that holds it
      const int cx = std::min(I / p, bx - 1), cy = std::min(J / p, by - 1), cz = std::min(K / p, bz - 1);
      return L(cell_at[cx + (size_t)bx * (cy + (size_t)by * cz)], I - cx * p, J - cy * p, K - cz * p);
    };
    uint32_t ent_dof[27];
    for (int e = 0; e < 27; ++e) {
      const int eI = e % 3, eJ = (e / 3) % 3, eK = e / 9;
      // an entity that does not exist (a direction without interior points: N == 1 has none between the planes) is never referenced
      const int I = eI == 0 ? 0 : eI == 2 ? N[0] : 1, J = eJ == 0 ? 0 : eJ == 2 ? N[1] : 1, K = eK == 0 ? 0 : eK == 2 ? N[2] : 1;
      ent_dof[e] = (I <= N[0] && J <= N[1] && K <= N[2] && !(eI == 1 && N[0] < 2) && !(eJ == 1 && N[1] < 2) && !(eK == 1 && N[2] < 2)) ? dof_at(I, J, K) : 0u;
    }
    for (uint32_t c = 0; c < nc && ok; ++c) { // the numbering, entry by entry
      const int cx = pos[3 * c] - lo[0], cy = pos[3 * c + 1] - lo[1], cz = pos[3 * c + 2] - lo[2];
      for (int k = 0; k < n && ok; ++k)
        for (int j = 0; j < n && ok; ++j)
          for (int i = 0; i < n; ++i) {
            const int I = cx * p + i, J = cy * p + j, K = cz * p + k;
            const int eI = cls(I, 0), eJ = cls(J, 1), eK = cls(K, 2);
            const uint32_t LX = eI == 1 ? N[0] - 1 : 1, LY = eJ == 1 ? N[1] - 1 : 1;
            const uint32_t off = (eI == 1 ? I - 1 : 0) + LX * ((eJ == 1 ? J - 1 : 0) + LY * (eK == 1 ? K - 1 : 0));
            if (L(c, i, j, k) != ent_dof[eI + 3 * eJ + 9 * eK] + off) { ok = false; break; }
          }
    }
    if (!ok) continue;
    uint32_t *row = lat.data() + (size_t)g * BLOCK_LATTICE_WORDS;
    const uint32_t *lb = list_dofs.data() + list_off[g], *le = list_dofs.data() + list_off[g + 1];
    for (int e = 0; e < 27 && ok; ++e) {
      const uint32_t *it = std::lower_bound(lb, le, ent_dof[e], [](uint32_t a, uint32_t b) { return (a & 0x7fffffffu) < b; });
      const int eI = e % 3, eJ = (e / 3) % 3, eK = e / 9;
      const bool exists = !(eI == 1 && N[0] < 2) && !(eJ == 1 && N[1] < 2) && !(eK == 1 && N[2] < 2);
      if (exists && (it == le || (*it & 0x7fffffffu) != ent_dof[e])) ok = false;
      row[e] = exists ? (uint32_t)(it - lb) : 0u;
      row[27 + e] = ent_dof[e];
    }
    if (!ok) { std::fill(row, row + BLOCK_LATTICE_WORDS, 0u); continue; }
    row[54] = (uint32_t)bx | (uint32_t)by << 8 | (uint32_t)bz << 16 | 0x80000000u;
    for (uint32_t c = 0; c < nc; ++c)
      cpos[c0 + c] = (uint16_t)((pos[3 * c] - lo[0]) | (pos[3 * c + 1] - lo[1]) << 4 | (pos[3 * c + 2] - lo[2]) << 8);
    ++n_ok;
  }
  return n_ok;
}

// run-length combine tables (runs of consecutive shared DoFs whose contributions sit in consecutive slab slots): start[r] = first ordinal of run r
// (start[n_runs] = number of shared DoFs), dof0[r] (bit 31: Dirichlet run), slots[soff[r] ...] = the slab slots of the run's first DoF.  Adds the
// tile table (run containing ordinal COMBINE_TILE t) and one entry of slack, uploads.
static int upload_combine_tables(std::vector<uint32_t> start, const std::vector<uint32_t> &dof0, std::vector<uint32_t> soff, const std::vector<uint32_t> &slots,
                                 bp5_mf::DevPlan::CombineTables *ct)
{
  const size_t ns = start.back();
  ct->n_shared = (uint32_t)ns;
  if (!ns) return BP5_OK;
  const size_t n_tiles = (ns + COMBINE_TILE - 1) / COMBINE_TILE;
  std::vector<uint32_t> tile(n_tiles + 1);
  size_t r = 0;
  for (size_t t = 0; t <= n_tiles; ++t) { // run containing ordinal min(COMBINE_TILE t, ns - 1)
    const size_t i = std::min(t * (size_t)COMBINE_TILE, ns - 1);
    while (start[r + 1] <= i) ++r;
    tile[t] = (uint32_t)r;
  }
  start.push_back((uint32_t)ns); // one entry of slack for the staging loop (reads r_hi + 1)
  soff.push_back((uint32_t)slots.size());
  BP5_TRY(upload(&ct->start, start.data(), start.size()));
  BP5_TRY(upload(&ct->dof0, dof0.data(), dof0.size()));
  BP5_TRY(upload(&ct->soff, soff.data(), soff.size()));
  BP5_TRY(upload(&ct->slots, slots.data(), slots.size()));
  BP5_TRY(upload(&ct->tile, tile.data(), tile.size()));
  return BP5_OK;
}

// Face carry (bp5_kernels.hpp: BLOCK_CARRY_MAX).  Block g can hand a face to block g + 1 when the two are lattice blocks that meet in a face whose
// interior DoFs (a) are ONE shared run in both blocks' lists, (b) are owned, unconstrained and (c) have exactly the two contributions of these
// blocks, at the slab slots the lists imply.  Two steps around the construction of the run tables: find_carry_faces (a), (b), (c) without the run
// condition -- the run tables are then CUT at the faces' first and last slots (a run of shared DoFs may span several entities) --, and
// mark_carry_faces, which checks (a) on the finished tables and writes the carry words of the lattice table (lat) and dp.h_carry_*.
struct CarryFace { uint32_t dof, len, s_out, s_in; };
static std::vector<CarryFace> find_carry_faces(const bp5_mf *mf, int p, const TeamPlanHost &h, const std::vector<uint32_t> &lat)
{
  const size_t ng = h.off.size() - 1;
  std::vector<CarryFace> faces(ng, CarryFace{0u, 0u, 0u, 0u});
  static const int E_OUT[3] = {14, 16, 22}, E_IN[3] = {12, 10, 4}; // entity = eI + 3 eJ + 9 eK, e* in {low face, interior, high face}
  for (size_t g = 0; g + 1 < ng; ++g) {
    const uint32_t *ra = lat.data() + g * BLOCK_LATTICE_WORDS, *rb = ra + BLOCK_LATTICE_WORDS;
    if (!(ra[54] >> 31) || !(rb[54] >> 31)) continue;
    const uint32_t da[3] = {ra[54] & 0xffu, (ra[54] >> 8) & 0xffu, (ra[54] >> 16) & 0xffu}, db[3] = {rb[54] & 0xffu, (rb[54] >> 8) & 0xffu, (rb[54] >> 16) & 0xffu};
    for (int d = 0; d < 3; ++d) {
      const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
      if (da[d1] != db[d1] || da[d2] != db[d2]) continue;
      const uint32_t len = (da[d1] * p - 1) * (da[d2] * p - 1), dof = ra[27 + E_OUT[d]];
      if (len == 0 || len > (uint32_t)BLOCK_CARRY_MAX || dof != rb[27 + E_IN[d]] || (uint64_t)dof + len > mf->n_owned) continue;
      const uint32_t s_out = ra[E_OUT[d]], s_in = rb[E_IN[d]];
      if (s_out > 0xffffu || s_in > 0xffffu) continue;
      const auto it = std::lower_bound(h.sh_dof.begin(), h.sh_dof.end(), dof);
      const size_t o = it - h.sh_dof.begin();
      bool ok = o + len <= h.sh_dof.size();
      for (uint32_t k = 0; ok && k < len; ++k) {
        const uint32_t b = h.sh_off[o + k], want_a = h.off[g] + s_out + k, want_b = h.off[g + 1] + s_in + k;
        ok = h.sh_dof[o + k] == dof + k && h.sh_off[o + k + 1] - b == 2 && !mf->h_constrained[dof + k] &&
             ((h.sh_slot[b] == want_a && h.sh_slot[b + 1] == want_b) || (h.sh_slot[b] == want_b && h.sh_slot[b + 1] == want_a));
      }
      if (!ok) continue;
      faces[g] = CarryFace{dof, len, s_out, s_in};
      break;
    }
  }
  return faces;
}
static size_t mark_carry_faces(const TeamPlanHost &h, const std::vector<CarryFace> &faces, const std::vector<uint32_t> &run_off, const std::vector<uint32_t> &runs,
                               std::vector<uint32_t> &lat, bp5_mf::DevPlan &dp)
{
  const size_t ng = h.off.size() - 1;
  dp.h_carry_dof.assign(ng, 0u);
  dp.h_carry_len.assign(ng, 0u);
  auto one_shared_run = [&](size_t g, uint32_t slot, uint32_t dof, uint32_t len) {
    const uint32_t nr = run_off[g + 1] - run_off[g], m = h.off[g + 1] - h.off[g];
    for (uint32_t r = 0; r < nr; ++r) {
      const uint32_t s0 = runs[2 * (run_off[g] + r)];
      if (s0 != slot) continue;
      const uint32_t s1 = r + 1 < nr ? runs[2 * (run_off[g] + r + 1)] : m;
      return runs[2 * (run_off[g] + r) + 1] == dof && s1 == slot + len; // (no ownership bit, no Dirichlet bit)
    }
    return false;
  };
  size_t n_faces = 0;
  for (size_t g = 0; g + 1 < ng && g < faces.size(); ++g) {
    const CarryFace &f = faces[g];
    if (!f.len || !one_shared_run(g, f.s_out, f.dof, f.len) || !one_shared_run(g + 1, f.s_in, f.dof, f.len)) continue;
    uint32_t *ra = lat.data() + g * BLOCK_LATTICE_WORDS, *rb = ra + BLOCK_LATTICE_WORDS;
    ra[55] = f.len << 16 | f.s_out; // (list slots < 2^16: the packed indices need that already)
    rb[56] = f.len << 16 | f.s_in;
    dp.h_carry_dof[g] = f.dof; dp.h_carry_len[g] = f.len;
    ++n_faces;
  }
  if (!n_faces) { dp.h_carry_dof.clear(); dp.h_carry_len.clear(); }
  return n_faces;
}

int build_carry_tables(bp5_mf *mf, bp5_mf::DevPlan *dp, const std::vector<uint32_t> &wb, uint32_t n_wg, bool two_parts, bp5_mf::DevPlan::CombineTables *out)
{
  // the faces this partition carries: consecutive blocks inside one part of one workgroup's range (the kernel's rule, apply_block_kernel: c_out)
  std::vector<std::pair<uint32_t, uint32_t>> cut; // (first DoF, count)
  for (int part = 0; part < (two_parts ? 2 : 1); ++part) {
    const uint32_t *w = wb.data() + (size_t)part * (n_wg + 1);
    for (uint32_t i = 0; i < n_wg; ++i)
      for (uint32_t g = w[i]; g + 1 < w[i + 1]; ++g)
        if (dp->h_carry_len[g]) cut.emplace_back(dp->h_carry_dof[g], dp->h_carry_len[g]);
  }
  std::sort(cut.begin(), cut.end());
  const std::vector<uint32_t> &st = dp->h_cr_start, &d0 = dp->h_cr_dof0, &so = dp->h_cr_soff, &sl = dp->h_cr_slots;
  const size_t nr = d0.size();
  std::vector<uint32_t> start, dof0, soff, slots;
  uint32_t ord = 0, owned = 0;
  size_t ci = 0;
  auto emit = [&](size_t r, uint32_t first, uint32_t count) { // DoFs [first, first + count) of run r stay
    if (!count) return;
    const uint32_t base = d0[r] & 0x7fffffffu;
    start.push_back(ord);
    dof0.push_back(first | (d0[r] & 0x80000000u));
    soff.push_back((uint32_t)slots.size());
    for (uint32_t q = so[r]; q < so[r + 1]; ++q) slots.push_back(sl[q] + (first - base));
    ord += count;
    if (first < mf->n_owned) owned += std::min(count, mf->n_owned - first);
  };
  for (size_t r = 0; r < nr; ++r) {
    uint32_t first = d0[r] & 0x7fffffffu;
    const uint32_t end = first + (st[r + 1] - st[r]);
    while (ci < cut.size() && cut[ci].first + cut[ci].second <= first) ++ci;
    size_t c = ci;
    while (c < cut.size() && cut[c].first < end) { // (faces are disjoint; each lies inside one run of shared DoFs)
      const uint32_t c0 = std::max(cut[c].first, first), c1 = std::min(cut[c].first + cut[c].second, end);
      emit(r, first, c0 - first);
      first = c1;
      ++c;
    }
    emit(r, first, end - first);
  }
  start.push_back(ord);
  soff.push_back((uint32_t)slots.size());
  BP5_TRY(upload_combine_tables(start, dof0, soff, slots, out));
  out->n_shared_owned = owned;
  return BP5_OK;
}

// key > 0: uniform teams of `key` cells (team kernel); key < 0: cell blocks walked in passes of
// -key cells (block kernel) -- the caller's blocks if given, else groups of `default_block` cells
int get_plan_raw(bp5_mf *mf, int key, bp5_mf::DevPlan **dpo, int default_block)
{
  auto it = mf->plans.find(key);
  if (it == mf->plans.end()) {
    TeamPlanHost h;
    if (key < 0 && mf->n_local() >= (1ull << 30)) return fail(BP5_ERR_UNSUPPORTED, "block plan needs fewer than 2^30 local DoFs");
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
      // lattice blocks: recognised and VERIFIED entry by entry here; everything else keeps the packed stream
      std::vector<uint32_t> lat;
      std::vector<uint16_t> cpos;
      const bool lattice_enabled = mf->tune[BP5_TUNE_LATTICE_INDICES] != 0; // (A/B knob, fixed once the plan is built)
      dp.n_lattice_blocks = detect_lattice_blocks(mf->h_l2g.data(), mf->degree + 1, h.group_cell_off, h.off, h.dofs, lat, cpos);
      // ... and the faces consecutive blocks could hand on in LDS (face carry): the run tables are cut at their ends
      std::vector<CarryFace> faces;
      if (dp.n_lattice_blocks && dp.n_lattice_blocks == (uint32_t)(h.off.size() - 1)) faces = find_carry_faces(mf, mf->degree, h, lat);
      auto face_edge = [&](size_t g, uint32_t slot) { // slot = first slot of a carried face of block g, or the first slot behind one
        if (faces.empty()) return false;
        if (faces[g].len && (slot == faces[g].s_out || slot == faces[g].s_out + faces[g].len)) return true;
        return g > 0 && faces[g - 1].len && (slot == faces[g - 1].s_in || slot == faces[g - 1].s_in + faces[g - 1].len);
      };
      // run-length form of the sorted block lists: consecutive DoFs with equal ownership flag, cut at 512 entries (and where the Dirichlet flag changes) so
      // that (run, offset) packs into 7 + 9 bits
      run_off.assign(h.off.size(), 0);
      for (size_t g = 0; g + 1 < h.off.size(); ++g) {
        uint32_t start = h.off[g];
        for (uint32_t i = h.off[g]; i < h.off[g + 1]; ++i) {
          const bool con = mf->h_constrained[h.dofs[i] & 0x7fffffffu];
          if (i == h.off[g] || h.dofs[i] != h.dofs[i - 1] + 1 || i - start == (1u << BLOCK_PACK_OFF_BITS) ||
              con != (bool)mf->h_constrained[h.dofs[i - 1] & 0x7fffffffu] || face_edge(g, i - h.off[g])) {
            start = i;
            runs.push_back(i - h.off[g]);
            runs.push_back(h.dofs[i] | (con ? BLOCK_DOF_CONSTRAINED : 0u)); // bit 31 exclusive (from dofs), bit 30 Dirichlet
          }
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
      if (!packed.empty() && dp.n_lattice_blocks && lattice_enabled) { // (knob off: the same run tables, cuts included -- the two builds give the same bits)
        if (!faces.empty()) mark_carry_faces(h, faces, run_off, runs, lat, dp);
        BP5_TRY(upload(&dp.lattice, lat.data(), lat.size()));
        BP5_TRY(upload(&dp.cell_pos, cpos.data(), cpos.size()));
      } else
        dp.n_lattice_blocks = 0;
    } else
      BP5_TRY(upload(&dp.pos, h.pos.data(), h.pos.size()));
    BP5_TRY(upload(&dp.cell_round, h.cell_round.data(), h.cell_round.size()));
    BP5_TRY(upload(&dp.team_rounds, h.team_rounds.data(), h.team_rounds.size()));
    BP5_TRY(upload(&dp.sh_dof, h.sh_dof.data(), h.sh_dof.size()));
    BP5_TRY(upload(&dp.sh_off, h.sh_off.data(), h.sh_off.size()));
    BP5_TRY(upload(&dp.sh_slot, h.sh_slot.data(), h.sh_slot.size()));
    HIP_TRY(hipMalloc((void **)&dp.partial, std::max<size_t>(h.dofs.size(), 1) * sizeof(double)));
    HIP_TRY(hipMemsetAsync(dp.partial, 0, std::max<size_t>(h.dofs.size(), 1) * sizeof(double), mf->stream)); // on the handle's stream (bp5.h:14)
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
    dp.n_shared_owned = (uint32_t)(std::lower_bound(h.sh_dof.begin(), h.sh_dof.end(), mf->n_owned) - h.sh_dof.begin());
    if (dp.n_shared) { // run-length form of the shared-DoF CSR for combine_runs_kernel
      std::vector<uint32_t> start, dof0, soff, slots, tile;
      const size_t ns = h.sh_dof.size();
      for (size_t i = 0; i < ns; ++i) {
        const uint32_t b = h.sh_off[i], e = h.sh_off[i + 1];
        bool cont = i > 0 && h.sh_dof[i] == h.sh_dof[i - 1] + 1 && (e - b) == (h.sh_off[i] - h.sh_off[i - 1]) &&
                    mf->h_constrained[h.sh_dof[i]] == mf->h_constrained[h.sh_dof[i - 1]];
        for (uint32_t q = 0; cont && q < e - b; ++q) cont = h.sh_slot[b + q] == h.sh_slot[h.sh_off[i - 1] + q] + 1;
        if (!cont) {
          start.push_back((uint32_t)i);
          dof0.push_back(h.sh_dof[i] | (mf->h_constrained[h.sh_dof[i]] ? 0x80000000u : 0u)); // bit 31: Dirichlet run
          soff.push_back((uint32_t)slots.size());
          slots.insert(slots.end(), h.sh_slot.begin() + b, h.sh_slot.begin() + e);
        }
      }
      start.push_back((uint32_t)ns);
      soff.push_back((uint32_t)slots.size());
      bp5_mf::DevPlan::CombineTables ct;
      BP5_TRY(upload_combine_tables(start, dof0, soff, slots, &ct));
      dp.cr_start = ct.start; dp.cr_dof0 = ct.dof0; dp.cr_soff = ct.soff; dp.cr_slots = ct.slots; dp.cr_tile = ct.tile;
      if (!dp.h_carry_len.empty()) { dp.h_cr_start = start; dp.h_cr_dof0 = dof0; dp.h_cr_soff = soff; dp.h_cr_slots = slots; } // (the partitions' tables are cut from these)
    }
    dp.covers_all = h.covers_all;
    dp.n_groups = (uint32_t)h.group_cell_off.size() - 1;
    for (uint32_t g = 0; g < dp.n_groups; ++g) dp.max_list = std::max(dp.max_list, h.off[g + 1] - h.off[g]);
    it = mf->plans.emplace(key, dp).first;
  }
  *dpo = &it->second;
  return BP5_OK;
}
int get_plan(bp5_mf *mf, int cpt, TeamPlan &tp, bp5_mf::DevPlan **dpo)
{
  bp5_mf::DevPlan *q = nullptr;
  BP5_TRY(get_plan_raw(mf, cpt, &q));
  tp.off = q->off; tp.dofs = q->dofs; tp.pos = q->pos; tp.cell_round = q->cell_round; tp.team_rounds = q->team_rounds; tp.partial = q->partial;
  if (dpo) *dpo = q;
  return BP5_OK;
}

// fixed grid of the fused solver's combine pass: k workgroups per CU (BP5_TUNE_COMBINE_WG_PER_CU, default 16: profiles/r4 h_*; each
// workgroup walks its tiles two at a time), never more than tiles or free dot-product columns
static uint32_t combine_grid(bp5_mf *mf, uint32_t tiles)
{
  const uint32_t cols = (uint32_t)PARTIAL_STRIDE - mf->fuse.n_cols - 1024u;
  uint32_t grid = std::min<uint32_t>(tiles, cols);
  const int k = mf->tune[BP5_TUNE_COMBINE_WG_PER_CU];
  if (k > 0) {
    if (!mf->n_cus) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, mf->device) == hipSuccess) mf->n_cus = prop.multiProcessorCount;
    }
    grid = std::min<uint32_t>(grid, (uint32_t)k * (uint32_t)std::max(mf->n_cus, 1));
  }
  return std::max<uint32_t>(grid, 1u);
}
int launch_combine(bp5_mf *mf, bp5_mf::DevPlan *dp, double *dst, bool set, int window)
{
  if (!dp->n_shared) return BP5_OK;
  // the tables of the block launch this pass follows: the plan's own, or (face carry) its partition's without the carried faces
  bp5_mf::DevPlan::CombineTables own;
  own.start = dp->cr_start; own.dof0 = dp->cr_dof0; own.soff = dp->cr_soff; own.slots = dp->cr_slots; own.tile = dp->cr_tile;
  own.n_shared = dp->n_shared; own.n_shared_owned = dp->n_shared_owned;
  const bp5_mf::DevPlan::CombineTables &ct = (dp->cr_active && dp->cr_tile && !mf->combine_csr) ? *dp->cr_active : own;
  if (!ct.n_shared) return BP5_OK;
  if (mf->fuse.on && !(set && dp->cr_tile && !mf->combine_csr)) return fail(BP5_ERR_INVALID, "fused dot products need the run-length combine pass in overwrite mode");
  if (window != COMBINE_ALL && !(dp->cr_tile && !mf->combine_csr)) return fail(BP5_ERR_INVALID, "combine windows need the run-length combine pass");
  if (mf->prof_mark && window != COMBINE_GHOST) { HIP_TRY(hipEventRecord(mf->prof_mark, mf->stream)); mf->prof_mark = nullptr; }
  const dim3 cg((dp->n_shared + 255) / 256); // CSR kernel
  if (dp->cr_tile && !mf->combine_csr) {
    CombineRuns cr{};
    cr.start = ct.start; cr.dof0 = ct.dof0; cr.soff = ct.soff; cr.slots = ct.slots; cr.tile_run = ct.tile;
    cr.n_shared = ct.n_shared;
    // tiles of the window: the shared DoFs are listed in ascending order, owned ones first
    const uint32_t all_tiles = (ct.n_shared + COMBINE_TILE - 1) / COMBINE_TILE;
    cr.tile0 = window == COMBINE_GHOST ? ct.n_shared_owned / COMBINE_TILE : 0u;
    const uint32_t tile1 = window == COMBINE_OWNED ? (ct.n_shared_owned + COMBINE_TILE - 1) / COMBINE_TILE : all_tiles;
    cr.dof_lo = window == COMBINE_GHOST ? mf->n_owned : 0u;
    cr.dof_hi = window == COMBINE_OWNED ? mf->n_owned : 0xffffffffu;
    if (window == COMBINE_GHOST_THEN_OWNED) {
      // owned rows exactly as COMBINE_OWNED (tiles, columns), preceded in the SAME launch by one workgroup per ghost tile that signals
      if (!(mf->fuse.on && set && mf->d_signal)) return fail(BP5_ERR_INVALID, "ghost-rows-first combine launch: fused overwrite launches with a signal word only");
      cr.tile0 = 0u;
      cr.dof_lo = 0u; cr.dof_hi = mf->n_owned;
      cr.ghost_tile0 = ct.n_shared_owned / COMBINE_TILE;
      cr.ghost_blocks = all_tiles - cr.ghost_tile0;
      cr.signal = mf->d_signal;
      const uint32_t owned_tiles = (ct.n_shared_owned + COMBINE_TILE - 1) / COMBINE_TILE;
      cr.cg_p = mf->fuse.p; cr.cg_r = mf->fuse.r; cr.dot_partials = mf->d_partials; cr.dot_col0 = mf->fuse.n_cols;
      cr.n_owned = mf->n_owned; cr.n_tiles = owned_tiles; cr.cg_state = mf->d_st;
      if (mf->fuse.n_cols + 1024u + 8u > (uint32_t)PARTIAL_STRIDE) return fail(BP5_ERR_UNSUPPORTED, "no partial-sum columns left for the combine pass");
      const uint32_t grid = combine_grid(mf, owned_tiles);
      if (ct.n_shared >= (8u << 20)) hipLaunchKernelGGL((combine_runs_kernel<false, true, true>), dim3(grid + cr.ghost_blocks), dim3(256), 0, mf->stream, cr, dp->partial, dst);
      else hipLaunchKernelGGL((combine_runs_kernel<false, true, false>), dim3(grid + cr.ghost_blocks), dim3(256), 0, mf->stream, cr, dp->partial, dst);
      KERNEL_CHECK();
      mf->fuse.n_cols += grid;
      mf->signal_target += cr.ghost_blocks; // every ghost workgroup counts itself in once
      return BP5_OK;
    }
    if (tile1 <= cr.tile0) return BP5_OK; // no row in the window
    const dim3 cgt(tile1 - cr.tile0);
    if (mf->fuse.on && window != COMBINE_GHOST) { // fused CG dot products over the brick-surface DoFs; columns behind the block kernel's workgroups
      // (ghost rows never enter the dot products: their window takes the plain kernel below)
      cr.cg_p = mf->fuse.p; cr.cg_r = mf->fuse.r; cr.dot_partials = mf->d_partials; cr.dot_col0 = mf->fuse.n_cols;
      cr.n_owned = mf->n_owned; cr.n_tiles = cgt.x; cr.cg_state = mf->d_st;
      if (mf->fuse.n_cols + 1024u + 8u > (uint32_t)PARTIAL_STRIDE) return fail(BP5_ERR_UNSUPPORTED, "no partial-sum columns left for the combine pass");
      const uint32_t grid = combine_grid(mf, cgt.x); // (1024 columns stay free for the exchange)
      // pairs of consecutive ordinals pay on long passes; short ones (config 2, the strong-scaling ranks) are latency-bound
      if (ct.n_shared >= (8u << 20)) hipLaunchKernelGGL((combine_runs_kernel<false, true, true>), dim3(grid), dim3(256), 0, mf->stream, cr, dp->partial, dst);
      else hipLaunchKernelGGL((combine_runs_kernel<false, true, false>), dim3(grid), dim3(256), 0, mf->stream, cr, dp->partial, dst);
      KERNEL_CHECK();
      mf->fuse.n_cols += grid;
      return BP5_OK;
    }
    const bool pairs = ct.n_shared >= (8u << 20) && window != COMBINE_GHOST;
    if (set && pairs) hipLaunchKernelGGL((combine_runs_kernel<false, false, true>), cgt, dim3(256), 0, mf->stream, cr, dp->partial, dst);
    else if (set) hipLaunchKernelGGL((combine_runs_kernel<false, false, false>), cgt, dim3(256), 0, mf->stream, cr, dp->partial, dst);
    else if (pairs) hipLaunchKernelGGL((combine_runs_kernel<true, false, true>), cgt, dim3(256), 0, mf->stream, cr, dp->partial, dst);
    else hipLaunchKernelGGL((combine_runs_kernel<true, false, false>), cgt, dim3(256), 0, mf->stream, cr, dp->partial, dst);
    KERNEL_CHECK();
    return BP5_OK;
  }
  if (set) hipLaunchKernelGGL(combine_kernel<false>, cg, dim3(256), 0, mf->stream, dp->sh_dof, dp->sh_off, dp->sh_slot, dp->partial, dst, dp->n_shared);
  else hipLaunchKernelGGL(combine_kernel<true>, cg, dim3(256), 0, mf->stream, dp->sh_dof, dp->sh_off, dp->sh_slot, dp->partial, dst, dp->n_shared);
  KERNEL_CHECK();
  return BP5_OK;
}

// [c0,c1) == union of whole cell blocks [b0,b1) of the caller's blocking?
bool block_aligned(const bp5_mf *mf, uint32_t c0, uint32_t c1, uint32_t *b0, uint32_t *b1)
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
  if (mf->has_hanging && mf->geometry_mode == BP5_GEOM_AFFINE) return 90;
  if (mf->operator_kind == BP5_OP_HELMHOLTZ || mf->has_hanging) { // pencil kernel, or the block kernel under the same conditions as below
    const int pencil = mf->has_hanging ? 90 : 0;
    uint32_t hb0, hb1;
    if (!block_lpc(mf->degree) || mf->h_block_off.empty() || !block_aligned(mf, c0, c1, &hb0, &hb1)) return pencil;
    if (mf->auto_block < 0) {
      bp5_mf::DevPlan *dp = nullptr;
      mf->auto_block = 0;
      if (get_plan_raw(mf, -block_cpt(mf), &dp, 64) == BP5_OK && dp->packed) {
        const size_t lds = block_default_lds_bytes(mf->degree, dp->max_list); // (the launcher's own formula)
        if (!mf->n_cus) {
          hipDeviceProp_t prop;
          if (hipGetDeviceProperties(&prop, mf->device) == hipSuccess) mf->n_cus = prop.multiProcessorCount;
        }
        mf->auto_block = lds * 2 <= 160 * 1024 && dp->n_groups >= 2u * (uint32_t)std::max(mf->n_cus, 1);
      }
    }
    if (mf->auto_block && (c0 != 0 || c1 != mf->n_cells) && (hb1 - hb0) < 30u * (uint32_t)std::max(mf->n_cus, 1)) return pencil;
    return mf->auto_block ? 56 : pencil;
  }
  if ((mf->degree == 1 || mf->degree == 3) && mf->h_block_off.empty()) {
    if (mf->geometry_mode == BP5_GEOM_AFFINE) return 0;
    if (mf->auto_team < 0) { // an irregular cell order can exhaust the team plan's rounds: then the atomic pencil kernel
      bp5_mf::DevPlan *dp = nullptr;
      mf->auto_team = get_plan_raw(mf, mf->degree == 1 ? 64 : 16, &dp) == BP5_OK;
    }
    return mf->auto_team ? 10 : 0;
  }
  if (!block_lpc(mf->degree)) return 0;
  const int fallback = (mf->degree == 4 && mf->geometry_mode == BP5_GEOM_AFFINE) ? 10 : (mf->degree == 3 || mf->degree == 1) ? 10 : 0; // else the pencil kernel
  if (mf->degree != 4 && mf->geometry_mode == BP5_GEOM_AFFINE) return 0; // the affine block build exists at p = 4 only
  uint32_t b0_, b1_;
  if (mf->h_block_off.empty() || !block_aligned(mf, c0, c1, &b0_, &b1_)) return fallback;
  if (mf->auto_block < 0) {
    bp5_mf::DevPlan *dp = nullptr;
    mf->auto_block = 0;
    if (get_plan_raw(mf, -block_cpt(mf), &dp, 64) == BP5_OK) {
      // LDS of the default shape: one transpose tile per cell slot + the brick's accumulator + two run tables; p <= 4 must fit
      // three workgroups per CU, p >= 5 (more registers per lane: two workgroups per CU anyway) two
      const size_t lds = block_default_lds_bytes(mf->degree, dp->max_list); // (the launcher's own formula)
      mf->auto_block = lds * (mf->degree <= 4 ? 3 : 2) <= 160 * 1024 && (mf->degree == 4 || dp->packed);
      // persistent workgroups need enough bricks each to balance: round 1 measured 3.6 bricks per workgroup (54^3 cells)
      // 4 % behind the pencil kernel as a bare operator; with the CG dot products fused into the write-out the block kernel
      // is ahead there too (profiles/r2: 0.439 vs 0.446 ms per iteration), so the bar dropped to 3 bricks per workgroup --
      // and, measured down the reference's mesh family at p = 4 (profiles/r2 "small meshes"), the fused iteration wins from
      // 2 bricks per CU on (1.1e6 DoFs: +15 %; 5.4e5 DoFs: par; below: the pencil kernel)
      if (!mf->n_cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, mf->device) == hipSuccess) mf->n_cus = prop.multiProcessorCount;
      }
      if (dp->n_groups < 2u * (uint32_t)std::max(mf->n_cus, 1)) mf->auto_block = 0; // (p = 3: +34 %, p = 6: +14 %, p = 7: +8 %, p = 1: +4 % at 2-7 bricks per CU)
    }
  }
  // sub-ranges: worth it only while the range still feeds the persistent grid (else the pencil kernel)
  if (mf->auto_block && (c0 != 0 || c1 != mf->n_cells) && (b1_ - b0_) < 30u * (uint32_t)std::max(mf->n_cus, 1)) return fallback;
  if (mf->auto_block && mf->geometry_mode == BP5_GEOM_AFFINE) { // the affine build needs the packed indices
    bp5_mf::DevPlan *dp = nullptr;
    if (get_plan_raw(mf, -block_cpt(mf), &dp, 64) != BP5_OK || !dp->packed) return fallback;
  }
  return mf->auto_block ? 56 : fallback;
}
// kernels that define every entry of dst themselves (owner stores + combine pass) need no zero-fill
static bool variant_overwrites(const bp5_mf *mf, int ev)
{
  const int v = ev % 100;
  return ev < 100 && ((v >= 10 && v <= 14) || (v >= 48 && v <= 63)) && !(mf->geometry_mode == BP5_GEOM_AFFINE && mf->degree != 4);
}

static int launch_apply_impl(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1, bool overwrite)
{
  switch (mf->degree) {
    case 1: return apply_degree_impl<1>(mf, coef, src, dst, c0, c1, overwrite);
    case 2: return apply_degree_impl<2>(mf, coef, src, dst, c0, c1, overwrite);
    case 3: return apply_degree_impl<3>(mf, coef, src, dst, c0, c1, overwrite);
    case 4: return apply_degree_impl<4>(mf, coef, src, dst, c0, c1, overwrite);
    case 5: return apply_degree_impl<5>(mf, coef, src, dst, c0, c1, overwrite);
    case 6: return apply_degree_impl<6>(mf, coef, src, dst, c0, c1, overwrite);
    case 7: return apply_degree_impl<7>(mf, coef, src, dst, c0, c1, overwrite);
    case 8: return apply_degree_impl<8>(mf, coef, src, dst, c0, c1, overwrite);
  }
  return fail(BP5_ERR_INVALID, "unsupported degree");
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
  hipLaunchKernelGGL(rhs_kernel<n>, dim3(grid), dim3(n, n, n), 0, mf->stream, mf->d_l2g, mf->d_coords, mf->d_tab_gauss, mf->n_cells, b,
                     mf->has_hanging ? (const uint32_t *)mf->d_hang_mask : (const uint32_t *)nullptr, (const double *)mf->d_hang_I);
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
                     affine ? (uint64_t)mf->n_cells * mf->n3 : mf->coef_plane_stride, mf->coef_cell_stride, affine ? mf->d_gcell : (const double *)nullptr, mf->d_tab, mf->n_cells, diag,
                     mf->has_hanging ? (const uint32_t *)mf->d_hang_mask : (const uint32_t *)nullptr, mf->n_planes());
  KERNEL_CHECK();
  if (mf->has_hanging) { // the coarse DoFs named on constrained faces / edges: one cell-operator application per such entry
    hipLaunchKernelGGL(diagonal_hanging_kernel<n>, dim3(grid), dim3(n, n, n), 0, mf->stream, mf->d_l2g, affine ? mf->d_scalar_plane : coef,
                       affine ? (uint64_t)mf->n_cells * mf->n3 : mf->coef_plane_stride, affine ? (uint64_t)mf->n3 : mf->coef_cell_stride, affine ? mf->d_gcell : (const double *)nullptr,
                       mf->d_tab, mf->n_cells, diag, (const uint32_t *)mf->d_hang_mask, (const double *)mf->d_hang_I);
    KERNEL_CHECK();
  }
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
                     out, mf->has_hanging ? (const uint32_t *)mf->d_hang_mask : (const uint32_t *)nullptr, (const double *)mf->d_hang_I);
  KERNEL_CHECK();
  return BP5_OK;
}
static int l2_dispatch(bp5_mf *mf, const double *u, double *out) { DISPATCH_N(launch_l2, mf, u, out); }
extern "C" int bp5_l2_norm_solution(bp5_mf *mf, const double *u, double *result)
{
  if (!mf || !u || !result) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  // the cell integrals read ghost DoFs through local_to_global: refresh them first, as the reference does on its ghosted copy
  // (ghost_solution_host, bp5/step-64.cu:602-616; deal.II's update_ghost_values() is const as well), and leave them zeroed
  const bool ghosts = mf->comm && !mf->neighbors.empty();
  if (ghosts) BP5_TRY(bp5_halo_gather(mf, const_cast<double *>(u)));
  HIP_TRY(hipMemsetAsync(mf->d_scalar, 0, sizeof(double), mf->stream));
  BP5_TRY(l2_dispatch(mf, u, mf->d_scalar));
  if (ghosts) BP5_TRY(bp5_halo_zero_ghosts(mf, const_cast<double *>(u)));
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
// Streaming kernels WITHOUT a reduction: one trip per workgroup ("flat" launch).  profiles/r4 hbm_sweep: on a capped grid-stride grid the
// workgroups drift apart and the write stream loses its locality -- fill 4.3 against 6.7 TB/s, the update kernels 4.8-4.9 against 5.6-5.9
static inline int stream_grid_flat(const bp5_mf *mf, size_t n, int per_thread)
{
  if (!mf->tune[BP5_TUNE_UPDATE_FLAT]) return stream_grid(n, per_thread);
  const size_t b = (n + (size_t)VB * per_thread - 1) / ((size_t)VB * per_thread);
  return (int)std::min<size_t>(std::max<size_t>(b, 1), (size_t)1 << 30);
}
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

extern "C" int bp5_vec_fill(bp5_mf *mf, double *v, double value, size_t n)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(v)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<0>, dim3(stream_grid_flat(mf, n, 2)), dim3(VB), 0, mf->stream, v, (const double *)nullptr, value, 0.0, n);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_vec_axpy(bp5_mf *mf, double *y, double a, const double *x, size_t n)
{
  if (!mf || !y || !x) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(y) || !aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<1>, dim3(stream_grid_flat(mf, n, 2)), dim3(VB), 0, mf->stream, y, x, 0.0, a, n);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_vec_equ(bp5_mf *mf, double *y, double a, const double *x, size_t n)
{
  if (!mf || !y || !x) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(y) || !aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<2>, dim3(stream_grid_flat(mf, n, 2)), dim3(VB), 0, mf->stream, y, x, 0.0, a, n);
  KERNEL_CHECK();
  return BP5_OK;
}
extern "C" int bp5_vec_sadd(bp5_mf *mf, double *y, double s, double a, const double *x, size_t n)
{
  if (!mf || !y || !x) return fail(BP5_ERR_INVALID, "null argument");
  if (!aligned16(y) || !aligned16(x)) return fail(BP5_ERR_INVALID, "vectors must be 16-byte aligned");
  hipLaunchKernelGGL(vec_kernel<3>, dim3(stream_grid_flat(mf, n, 2)), dim3(VB), 0, mf->stream, y, x, s, a, n);
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
// Halo exchange.  The RCCL traffic runs on the handle's own communication stream, ordered against the compute stream by
// events, so that cell work enqueued between a *_start and its *_finish overlaps the transfer (the reference:
// update_ghost_values_start/finish, compress_start/finish inside cell_loop with overlap_communication_computation,
// bp5/step-64.cu:241,274).  With overlap switched off (bp5_mf_set_overlap) everything stays on the compute stream.
static int halo_streams(bp5_mf *mf)
{
  if (mf->comm_stream) return BP5_OK;
  // highest priority: the stream then gets a hardware queue of its own (a plain second stream was seen to share the
  // compute stream's queue -- in-order, no overlap at all: profiles/r2 c_*), and the short RCCL kernels are not queued
  // behind the cell kernel's workgroups
  int prio_lo = 0, prio_hi = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  HIP_TRY(hipStreamCreateWithPriority(&mf->comm_stream, hipStreamNonBlocking, prio_hi));
  for (hipEvent_t &e : mf->ev_halo) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); // (hipEventReleaseToDevice was tried: +20 us per iteration, profiles/r4 z_*)
  // boundary-first schedule inside ONE launch: the communication stream waits for a counter the block kernel's workgroups bump once
  // their ghost-touching bricks are written out (hipStreamWaitValue64).  The counter is plain device memory: there the runtime implements
  // the wait by POLLING (profiles/r3/a_wait_value_probe.txt: same timing as a spin kernel; on signal memory the wait released only after
  // the producer kernel had ended), i.e. a mid-kernel release rests on runtime behaviour the API does not promise.  So the capability bit
  // alone is not trusted: a one-off producer / consumer pair checks that the waiting stream really is released while the producing kernel
  // is still running (bounded: the producer gives up after 2 ms); if not, the ghost-touching bricks get a launch of their own
  // (BP5_TUNE_BOUNDARY_FIRST = 0 selects that form by hand).
  int can = 0;
  if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, mf->device) != hipSuccess) can = 0;
  if (can && !mf->d_signal) {
    HIP_TRY(hipMalloc((void **)&mf->d_signal, 4 * sizeof(unsigned long long))); // [0] the counter, [1] consumer-ran flag, [2] self-check result
    HIP_TRY(hipMemset(mf->d_signal, 0, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipStreamSynchronize(nullptr));
    mf->signal_target = 0;
  }
  if (can) {
    mf->signal_target += 1;
    hipLaunchKernelGGL(wait_value_probe_producer, dim3(1), dim3(64), 0, mf->stream, mf->d_signal, mf->d_signal + 1, mf->d_signal + 2, 200000LL /* 2 ms of the 100 MHz clock */);
    KERNEL_CHECK();
    const bool waited = hipStreamWaitValue64(mf->comm_stream, mf->d_signal, mf->signal_target, hipStreamWaitValueGte, ~0ull) == hipSuccess;
    if (waited) { hipLaunchKernelGGL(wait_value_probe_consumer, dim3(1), dim3(64), 0, mf->comm_stream, mf->d_signal + 1); KERNEL_CHECK(); }
    else (void)hipGetLastError();
    HIP_TRY(hipStreamSynchronize(mf->stream));
    HIP_TRY(hipStreamSynchronize(mf->comm_stream));
    unsigned long long released_mid_kernel = 0;
    HIP_TRY(hipMemcpy(&released_mid_kernel, mf->d_signal + 2, sizeof(released_mid_kernel), hipMemcpyDeviceToHost));
    if (!waited || !released_mid_kernel) can = 0;
  }
  mf->wait_value_ok = can;
  mf->can_wait_value = can && mf->tune[BP5_TUNE_BOUNDARY_FIRST] != 0;
  return BP5_OK;
}
extern "C" int bp5_mf_wait_value_available(bp5_mf *mf, int *available)
{
  if (!mf || !available) return fail(BP5_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(mf->device));
  BP5_TRY(halo_streams(mf));
  *available = mf->wait_value_ok == 1;
  return BP5_OK;
}
static int env_int(const char *name, int fallback)
{
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : fallback;
}
// initial values of the handle's knobs: the environment is read here, once per handle, and nowhere else (bp5.h: BP5_TUNE_*)
static void tuning_from_environment(bp5_mf *mf)
{
  mf->tune[BP5_TUNE_LATTICE_INDICES] = env_int("BP5_LATTICE_INDICES", 1) != 0;
  mf->tune[BP5_TUNE_EARLY_GATHER] = env_int("BP5_EARLY_GATHER", 1) != 0;
  mf->tune[BP5_TUNE_COMBINE_SIGNAL] = env_int("BP5_COMBINE_SIGNAL", 0) == 1;
  { const char *e = getenv("BP5_BOUNDARY_FIRST"); mf->tune[BP5_TUNE_BOUNDARY_FIRST] = !(e && !strcmp(e, "launches")); }
  mf->tune[BP5_TUNE_FOLD_SMALL] = env_int("BP5_FOLD_SMALL", 1) != 0;
  { const int u = env_int("BP5_UPDATE_UNROLL", 1); mf->tune[BP5_TUNE_UPDATE_UNROLL] = (u == 2 || u == 4) ? u : 1; }
  mf->tune[BP5_TUNE_UPDATE_FLAT] = env_int("BP5_UPDATE_FLAT", 1) != 0;
  { const int v = env_int("BP5_UPDATE_NT", -1); mf->tune[BP5_TUNE_UPDATE_NT] = v < 0 ? -1 : v != 0; }
  { const int v = env_int("BP5_COMBINE_WG_PER_CU", 16); mf->tune[BP5_TUNE_COMBINE_WG_PER_CU] = (v >= 0 && v <= 32) ? v : 16; }
  mf->tune[BP5_TUNE_INTERIOR_STORES] = env_int("BP5_INTERIOR_STORES", 1) != 0;
  mf->tune[BP5_TUNE_GHOST_COMBINE_ON_COMM] = env_int("BP5_GHOST_COMBINE_ON_COMM", 0) != 0;
  mf->tune[BP5_TUNE_FACE_CARRY] = env_int("BP5_FACE_CARRY", 1) != 0;
}
extern "C" int bp5_mf_set_tuning(bp5_mf *mf, int knob, int value)
{
  if (!mf || knob < 0 || knob >= BP5_TUNE_COUNT) return fail(BP5_ERR_INVALID, "unknown tuning knob");
  switch (knob) {
    case BP5_TUNE_UPDATE_UNROLL:
      if (value != 1 && value != 2 && value != 4) return fail(BP5_ERR_INVALID, "BP5_TUNE_UPDATE_UNROLL: 1, 2 or 4");
      break;
    case BP5_TUNE_UPDATE_NT:
      if (value < -1 || value > 1) return fail(BP5_ERR_INVALID, "BP5_TUNE_UPDATE_NT: -1, 0 or 1");
      break;
    case BP5_TUNE_COMBINE_WG_PER_CU:
      if (value < 0 || value > 32) return fail(BP5_ERR_INVALID, "BP5_TUNE_COMBINE_WG_PER_CU: 0 ... 32");
      break;
    case BP5_TUNE_LATTICE_INDICES:
      if (value != 0 && value != 1) return fail(BP5_ERR_INVALID, "switch: 0 or 1");
      if (value != mf->tune[knob] && mf->plans.count(-block_cpt(mf))) return fail(BP5_ERR_INVALID, "BP5_TUNE_LATTICE_INDICES: the block plan of this handle is already built");
      break;
    default:
      if (value != 0 && value != 1) return fail(BP5_ERR_INVALID, "switch: 0 or 1");
  }
  mf->tune[knob] = value;
  if (knob == BP5_TUNE_BOUNDARY_FIRST && mf->wait_value_ok >= 0) mf->can_wait_value = mf->wait_value_ok == 1 && value != 0;
  return BP5_OK;
}
extern "C" int bp5_mf_get_tuning(const bp5_mf *mf, int knob, int *value)
{
  if (!mf || !value || knob < 0 || knob >= BP5_TUNE_COUNT) return fail(BP5_ERR_INVALID, "unknown tuning knob");
  *value = mf->tune[knob];
  return BP5_OK;
}
extern "C" int bp5_mf_set_overlap(bp5_mf *mf, int mode)
{
  if (!mf || mode < 0 || mode > 2) return fail(BP5_ERR_INVALID, "bad argument");
  mf->overlap = mode;
  return BP5_OK;
}
// auto: the split into interior / boundary / interior launches and its four cross-stream dependencies cost 50-80 us per
// application on one GPU (profiles/r2 p_*), the exchange it hides is one DoF plane each way: worth it on large slabs only
static bool overlap_wanted(const bp5_mf *mf) { return mf->overlap == 1 || (mf->overlap == 2 && mf->n_interior >= 1000000u); }
// ghost gather: owners send their interface values (packed through send_indices), ghosts are
// received straight into the vector's ghost range (contiguous per neighbour)
static int gather_exchange(bp5_mf *mf, double *v, bool on_comm_stream);
extern "C" int bp5_halo_gather_start(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->neighbors.empty()) return BP5_OK;
  if (!mf->comm) return fail(BP5_ERR_INVALID, "halo exchange needs bp5_mf_set_comm");
  HIP_TRY(hipSetDevice(mf->device));
  BP5_TRY(halo_streams(mf));
  const uint32_t ns = mf->send_off.back();
  if (ns) {
    hipLaunchKernelGGL(pack_kernel, dim3((ns + 255) / 256), dim3(256), 0, mf->stream, mf->d_send_idx, ns, v, mf->d_sendbuf);
    KERNEL_CHECK();
  }
  return gather_exchange(mf, v, overlap_wanted(mf));
}
// the RCCL part of the ghost gather: d_sendbuf is packed (on the compute stream); ghosts arrive in v's ghost range
static int gather_exchange(bp5_mf *mf, double *v, bool on_comm_stream)
{
  mf->overlap_now = on_comm_stream;
  hipStream_t cs = mf->overlap_now ? mf->comm_stream : mf->stream;
  if (mf->overlap_now) { // the exchange starts once the values are packed (and everything before them on the compute stream is done)
    HIP_TRY(hipEventRecord(mf->ev_halo[0], mf->stream));
    HIP_TRY(hipStreamWaitEvent(cs, mf->ev_halo[0], 0));
  }
  NCCL_TRY(ncclGroupStart());
  for (size_t k = 0; k < mf->neighbors.size(); ++k) {
    const uint32_t sc = mf->send_off[k + 1] - mf->send_off[k], rc = mf->recv_off[k + 1] - mf->recv_off[k];
    if (sc) NCCL_TRY(ncclSend(mf->d_sendbuf + mf->send_off[k], sc, ncclDouble, mf->neighbors[k], mf->comm->comm, cs));
    if (rc) NCCL_TRY(ncclRecv(v + mf->n_owned + mf->recv_off[k], rc, ncclDouble, mf->neighbors[k], mf->comm->comm, cs));
  }
  NCCL_TRY(ncclGroupEnd());
  if (mf->overlap_now) HIP_TRY(hipEventRecord(mf->ev_halo[1], cs));
  return BP5_OK;
}
extern "C" int bp5_halo_gather_finish(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->neighbors.empty()) return BP5_OK;
  if (!mf->comm || !mf->comm_stream) return fail(BP5_ERR_INVALID, "bp5_halo_gather_finish without bp5_halo_gather_start");
  if (mf->overlap_now) HIP_TRY(hipStreamWaitEvent(mf->stream, mf->ev_halo[1], 0)); // later compute work sees the ghosts
  return BP5_OK;
}
extern "C" int bp5_halo_gather(bp5_mf *mf, double *v)
{
  BP5_TRY(bp5_halo_gather_start(mf, v));
  return bp5_halo_gather_finish(mf, v);
}
// compress(add): ghost contributions travel back to the owners and are added; ghosts zeroed
static int scatter_exchange(bp5_mf *mf, double *v, bool on_comm_stream, bool ordered_by_caller = false)
{
  mf->overlap_now = on_comm_stream;
  hipStream_t cs = mf->overlap_now ? mf->comm_stream : mf->stream;
  if (mf->overlap_now && !ordered_by_caller) { // the ghost entries are complete at this point of the compute stream
    HIP_TRY(hipEventRecord(mf->ev_halo[2], mf->stream));
    HIP_TRY(hipStreamWaitEvent(cs, mf->ev_halo[2], 0));
  }
  NCCL_TRY(ncclGroupStart());
  for (size_t k = 0; k < mf->neighbors.size(); ++k) {
    const uint32_t sc = mf->send_off[k + 1] - mf->send_off[k], rc = mf->recv_off[k + 1] - mf->recv_off[k];
    if (rc) NCCL_TRY(ncclSend(v + mf->n_owned + mf->recv_off[k], rc, ncclDouble, mf->neighbors[k], mf->comm->comm, cs));
    if (sc) NCCL_TRY(ncclRecv(mf->d_recvbuf + mf->send_off[k], sc, ncclDouble, mf->neighbors[k], mf->comm->comm, cs));
  }
  NCCL_TRY(ncclGroupEnd());
  if (mf->overlap_now) HIP_TRY(hipEventRecord(mf->ev_halo[3], cs));
  return BP5_OK;
}
extern "C" int bp5_halo_scatter_add_start(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->neighbors.empty()) return BP5_OK;
  if (!mf->comm) return fail(BP5_ERR_INVALID, "halo exchange needs bp5_mf_set_comm");
  HIP_TRY(hipSetDevice(mf->device));
  BP5_TRY(halo_streams(mf));
  return scatter_exchange(mf, v, overlap_wanted(mf));
}
extern "C" int bp5_halo_scatter_add_finish(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->neighbors.empty()) return BP5_OK;
  if (!mf->comm || !mf->comm_stream) return fail(BP5_ERR_INVALID, "bp5_halo_scatter_add_finish without bp5_halo_scatter_add_start");
  if (mf->overlap_now) HIP_TRY(hipStreamWaitEvent(mf->stream, mf->ev_halo[3], 0));
  for (size_t k = 0; k < mf->neighbors.size(); ++k) { // per neighbour: indices distinct -> race-free, fixed order
    const uint32_t sc = mf->send_off[k + 1] - mf->send_off[k];
    if (!sc) continue;
    if (mf->fuse.on) { // fused CG dot products: correct the sums the write-out formed with the local part of these DoFs
      const uint32_t grid = std::min<uint32_t>((sc + 255) / 256, 1024u / (uint32_t)mf->neighbors.size());
      const uint32_t ng = mf->fuse.ghosts_zeroed ? 0u : mf->n_ghost; // the first of these launches also zeroes both ghost ranges
      hipLaunchKernelGGL(unpack_add_dots_kernel, dim3(grid), dim3(256), 0, mf->stream, mf->d_send_idx + mf->send_off[k],
                         mf->d_send_dirichlet + mf->send_off[k], sc, mf->d_recvbuf + mf->send_off[k], v, mf->fuse.r, mf->d_partials,
                         mf->fuse.n_cols, mf->d_st, v + mf->n_owned, const_cast<double *>(mf->fuse.p) + mf->n_owned, ng);
      mf->fuse.n_cols += grid;
      mf->fuse.ghosts_zeroed = true;
    } else
      hipLaunchKernelGGL(unpack_add_kernel, dim3((sc + 255) / 256), dim3(256), 0, mf->stream, mf->d_send_idx + mf->send_off[k], sc,
                         mf->d_recvbuf + mf->send_off[k], v);
    KERNEL_CHECK();
  }
  if (mf->fuse.on && mf->fuse.ghosts_zeroed) return BP5_OK;
  return bp5_halo_zero_ghosts(mf, v);
}
extern "C" int bp5_halo_scatter_add(bp5_mf *mf, double *v)
{
  BP5_TRY(bp5_halo_scatter_add_start(mf, v));
  return bp5_halo_scatter_add_finish(mf, v);
}
extern "C" int bp5_halo_zero_ghosts(bp5_mf *mf, double *v)
{
  if (!mf || !v) return fail(BP5_ERR_INVALID, "null argument");
  if (mf->n_ghost) HIP_TRY(hipMemsetAsync(v + mf->n_owned, 0, (size_t)mf->n_ghost * sizeof(double), mf->stream));
  return BP5_OK;
}

// The operator application in phases (MatrixFree::cell_loop with overlap_communication_computation, bp5/step-64.cu:241,274;
// SURVEY 3.2 / Appendix C3): ghost gather in flight under the first part of the interior cells, then the cells that touch
// ghosts, then the ghost contributions travel to their owners under the rest of the interior cells.  ONE kernel family is
// chosen for the whole application; the block kernel runs its brick ranges with owner stores + partial slab and a single
// combine pass at the end, so the result is bitwise the one of the unsplit launch.
struct ApplyPhases {
  bool block = false, set = false, overwrite = false;
  bp5_mf::DevPlan *dp = nullptr;
  int user_variant = 0;
};
static int phases_begin(bp5_mf *mf, double *dst, bool overwrite, ApplyPhases &ph)
{
  ph.user_variant = mf->apply_variant;
  const int ev = effective_variant(mf, 0, mf->n_cells);
  ph.block = ev < 100 && (ev % 100 == 56 || ev % 100 == 48 || ev % 100 == 49 || ev % 100 == 60 || ev % 100 == 61 || ev % 100 == 62 || ev % 100 == 63) && block_lpc(mf->degree) != 0 && (mf->degree == 4 || ev % 100 == 56);
  ph.overwrite = false;
  if (ph.block) {
    BP5_TRY(get_plan_raw(mf, -block_cpt(mf), &ph.dp));
    ph.set = overwrite && ph.dp->covers_all;
    ph.overwrite = ph.set;
    if (overwrite && !ph.set) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    mf->defer_combine = true;
    mf->apply_variant = ev; // every range takes the block kernel, however few bricks it holds
  } else if (overwrite) // atomic kernels accumulate: one zero-fill, then every range adds
    HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
  return BP5_OK;
}
static int phases_range(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1, ApplyPhases &ph)
{
  if (c1 <= c0) return BP5_OK;
  return launch_apply(mf, coef, src, dst, c0, c1, ph.overwrite);
}
static int phases_end(bp5_mf *mf, double *dst, ApplyPhases &ph, int status, int window = COMBINE_ALL)
{
  mf->defer_combine = false;
  mf->apply_variant = ph.user_variant;
  BP5_TRY(status);
  return ph.block ? launch_combine(mf, ph.dp, dst, ph.set, window) : BP5_OK;
}
// interior cells [0, split) run under the ghost gather, [split, n_interior) under the scatter-add; split on a brick boundary
static uint32_t interior_split(const bp5_mf *mf)
{
  const uint32_t half = mf->n_interior / 2;
  if (mf->h_block_off.empty()) return half;
  const auto it = std::lower_bound(mf->h_block_off.begin(), mf->h_block_off.end(), half);
  return it == mf->h_block_off.end() || *it > mf->n_interior ? mf->n_interior : *it;
}
static int apply_overlapped(bp5_mf *mf, const double *coef, double *src, double *dst, bool overwrite)
{
  // block-kernel ranges must be unions of whole bricks: the generator emits interior and ghost-touching bricks separately;
  // a mesh whose n_interior_cells cuts through a brick runs unsplit (the exchange is then not overlapped)
  uint32_t b0, b1;
  const bool aligned = mf->h_block_off.empty() || mf->n_interior == 0 || mf->n_interior == mf->n_cells ||
                       block_aligned(mf, 0, mf->n_interior, &b0, &b1);
  ApplyPhases ph;
  BP5_TRY(bp5_halo_gather_start(mf, src));
  int st = phases_begin(mf, dst, overwrite, ph);
  if (st == BP5_OK && (!aligned || !overlap_wanted(mf))) {
    st = bp5_halo_gather_finish(mf, src);
    if (st == BP5_OK) st = phases_range(mf, coef, src, dst, 0, mf->n_cells, ph);
    BP5_TRY(phases_end(mf, dst, ph, st));
    return bp5_halo_scatter_add(mf, dst);
  }
  const uint32_t split = interior_split(mf);
  if (st == BP5_OK) st = phases_range(mf, coef, src, dst, 0, split, ph);                 // under the gather
  if (st == BP5_OK) st = bp5_halo_gather_finish(mf, src);
  if (st == BP5_OK) st = phases_range(mf, coef, src, dst, mf->n_interior, mf->n_cells, ph); // cells that touch ghosts
  // The atomic kernels have completed the ghost entries of dst once the ghost-touching cells are done; the block kernel needs the
  // ghost ROWS of its combine pass on top (ghost DoFs on brick faces go through the partial slab): one small launch over the ghost
  // window.  Either way the ghost contributions travel to their owners under the second part of the interior cells, and the block
  // kernel's owned rows are combined after the last range -- every row once, in the order of the unsplit pass.
  const bool ghost_rows = ph.block && ph.dp->n_shared && mf->n_ghost;                  // ghost rows (may) pass through the partial slab
  if (ghost_rows && !(ph.dp->cr_tile && !mf->combine_csr)) { // (per-DoF CSR combine pass: no windows -- the ghost rows are final after the last range only)
    if (st == BP5_OK) st = phases_range(mf, coef, src, dst, split, mf->n_interior, ph);
    BP5_TRY(phases_end(mf, dst, ph, st));
    return bp5_halo_scatter_add(mf, dst);
  }
  if (st == BP5_OK && ghost_rows) st = launch_combine(mf, ph.dp, dst, ph.set, COMBINE_GHOST);
  if (st == BP5_OK) st = bp5_halo_scatter_add_start(mf, dst);
  if (st == BP5_OK) st = phases_range(mf, coef, src, dst, split, mf->n_interior, ph);
  BP5_TRY(phases_end(mf, dst, ph, st, ghost_rows ? COMBINE_OWNED : COMBINE_ALL));
  return bp5_halo_scatter_add_finish(mf, dst);
}
extern "C" int bp5_apply_distributed(bp5_mf *mf, const double *coef, double *src, double *dst, int zero_dst)
{
  if (!mf || (!coef && mf->geometry_mode != BP5_GEOM_AFFINE) || !src || !dst) return fail(BP5_ERR_INVALID, "null argument");
  if (src == dst) return fail(BP5_ERR_INVALID, "src and dst must differ");
  HIP_TRY(hipSetDevice(mf->device));
  if (mf->comm && !mf->neighbors.empty()) BP5_TRY(apply_overlapped(mf, coef, src, dst, zero_dst != 0));
  else BP5_TRY(launch_apply(mf, coef, src, dst, 0, mf->n_cells, zero_dst != 0));
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
  HIP_TRY(hipMemsetAsync(base, 0, 3 * nb + 2 * stagger + 4096, mf->stream)); // ordered with the solver kernels that follow on this stream
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
  static constexpr int MAX_PROFILED = 512; // applications bracketed per solve; later ones run unbracketed (bounded pool)
  int mark(int k)
  {
    if (!on) return BP5_OK;
    if (used >= 4 * MAX_PROFILED) { on = false; return BP5_OK; }
    while ((size_t)used + 4 > mf->ev_pool.size()) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); mf->ev_pool.push_back(e); }
    HIP_TRY(hipEventRecord(mf->ev_pool[used + k], mf->stream));
    return BP5_OK;
  }
};

// profile == 2: stamp k of the iteration being profiled (bp5_cg_result.phase_ms)
static int phase_mark(bp5_mf *mf, int k)
{
  auto &ph = mf->phase;
  if (!ph.on || ph.it >= bp5_mf::PhaseProfile::MAX_ITERS) return BP5_OK;
  HIP_TRY(hipEventRecord(ph.ev[(size_t)ph.it * bp5_mf::PhaseProfile::MARKS + k], mf->stream));
  ph.recorded[ph.it] |= (uint8_t)(1u << k);
  return BP5_OK;
}

// A.vmult(h, d) inside the solvers: dst already zero on entry when zeroed == true
// n_cols != nullptr (CG on the packed block kernel, one rank's worth of cells): the operator's write-out and
// combine pass also form the v-dependent dot products of update_b (bp5/solver.h:142-311) and apply the Dirichlet copy; the
// partial sums land in d_partials, *n_cols columns of them.  fuse_r: the residual vector of the merged solver (D == 1), or NULL when
// only p.v is wanted (standard CG: rows 2-6 of the sums are then meaningless and r is never read)
static int solver_vmult(bp5_mf *mf, const double *coef, double *src, double *dst, bool zero, ApplyProfile &prof, const double *fuse_r = nullptr,
                        uint32_t *n_cols = nullptr)
{
  const bool dist = mf->comm && !mf->neighbors.empty(); // halo exchange: whenever there are neighbours (tests: a self neighbour)
  const bool fusing = n_cols != nullptr;
  if (dist && fusing) {
    // fused dot products across ranks: gather, fused launch(es) over all cells (p.v is a sum over cells, so it needs no owner
    // bookkeeping; v.v, r.v, r.r run over owned DoFs), then the ghost contributions travel to their owners, whose unpack kernel
    // corrects v.v and r.v for what it adds.
    // Boundary-first schedule (mf->cg_split, the twin of overlap_communication_computation, bp5/step-64.cu:241,274): the bricks that
    // touch ghost DoFs run FIRST, one small combine pass completes the ghost rows, the exchange starts on the communication stream and
    // the interior bricks run underneath it; the owned rows are combined after the last brick.  Same kernels, same per-brick sums and
    // the same combine order as the single launch: v is bitwise the same (the dot products are summed over a different column layout).
    const bool split = mf->cg_split;
    if (mf->fuse.gather_in_flight) { mf->fuse.gather_in_flight = false; BP5_TRY(bp5_halo_gather_finish(mf, src)); } // started under the update kernel
    else BP5_TRY(bp5_halo_gather(mf, src));
    BP5_TRY(phase_mark(mf, 2));
    BP5_TRY(prof.mark(0));
    BP5_TRY(prof.mark(1));
    mf->fuse.on = true; mf->fuse.p = src; mf->fuse.r = fuse_r; mf->fuse.n_cols = 0;
    int st = BP5_OK;
    bool marked = false;
    if (!split && mf->cg_late) {
      // all bricks in one launch; then the GHOST rows of the combine pass (a small launch), the exchange on the communication stream,
      // and the owned rows combined underneath it: nothing runs beside the bandwidth-bound brick kernel (a co-running RCCL kernel
      // crawls there -- 300 us for one DoF plane -- and slows it: profiles/r3), the exchange hides behind the owned-row combine
      bp5_mf::DevPlan *dp = nullptr;
      st = get_plan_raw(mf, -block_cpt(mf), &dp);
      if (st == BP5_OK) st = halo_streams(mf);
      mf->defer_combine = true;
      if (st == BP5_OK) st = launch_apply(mf, coef, src, dst, 0, mf->n_cells, true);
      mf->defer_combine = false;
      if (st == BP5_OK) st = prof.mark(2);
      // BP5_TUNE_COMBINE_SIGNAL (A/B knob of the handle): the two combine launches as ONE -- its first workgroups
      // complete the ghost rows and count themselves in, the communication stream waits for the count (stream wait-value) and starts the
      // exchange while the same launch walks the owned rows; same tiles and columns as the two launches: same bits.  One launch, one gap and
      // one cross-stream event less -- and on ONE GPU 16 us per iteration SLOWER than the two launches (0.547 against 0.531 ms on the slab of
      // rank 3 of 8, profiles/r3 z_*: the RCCL kernel then runs beside the bandwidth-bound combine pass from its first microsecond and both
      // crawl), hence not the default; whether a longer xGMI transfer pays for the earlier start is for a multi-GPU run to say
      const bool combine_signal = mf->tune[BP5_TUNE_COMBINE_SIGNAL] != 0;
      const bool one_combine = combine_signal && mf->can_wait_value == 1 && mf->n_ghost && mf->d_signal && dp && dp->n_shared > dp->n_shared_owned && dp->cr_tile && !mf->combine_csr; // (ghost rows among the shared ones: at least one workgroup signals)
      if (one_combine) {
        if (st == BP5_OK) st = launch_combine(mf, dp, dst, true, COMBINE_GHOST_THEN_OWNED);
        if (st == BP5_OK && hipStreamWaitValue64(mf->comm_stream, mf->d_signal, mf->signal_target, hipStreamWaitValueGte, ~0ull) != hipSuccess)
          st = fail(BP5_ERR_HIP, "hipStreamWaitValue64");
        if (st == BP5_OK) st = scatter_exchange(mf, dst, true, true);
      } else if (mf->tune[BP5_TUNE_GHOST_COMBINE_ON_COMM] && mf->n_ghost) {
        // the ghost rows of the combine pass on the COMMUNICATION stream (behind an event that says the brick kernel is done, in front of the send): the
        // owned rows start right behind the brick kernel on the compute stream -- one small launch and its gap off the critical path (same kernels, same bits)
        if (st == BP5_OK && hipEventRecord(mf->ev_halo[2], mf->stream) != hipSuccess) st = fail(BP5_ERR_HIP, "hipEventRecord");
        if (st == BP5_OK && hipStreamWaitEvent(mf->comm_stream, mf->ev_halo[2], 0) != hipSuccess) st = fail(BP5_ERR_HIP, "hipStreamWaitEvent");
        if (st == BP5_OK) {
          hipStream_t compute = mf->stream;
          mf->stream = mf->comm_stream;
          st = launch_combine(mf, dp, dst, true, COMBINE_GHOST);
          mf->stream = compute;
        }
        if (st == BP5_OK) st = scatter_exchange(mf, dst, true, true);
        if (st == BP5_OK) st = launch_combine(mf, dp, dst, true, COMBINE_OWNED);
      } else {
        if (st == BP5_OK && mf->n_ghost) st = launch_combine(mf, dp, dst, true, COMBINE_GHOST);
        if (st == BP5_OK) st = scatter_exchange(mf, dst, true);
        if (st == BP5_OK) st = launch_combine(mf, dp, dst, true, mf->n_ghost ? COMBINE_OWNED : COMBINE_ALL);
      }
      if (st == BP5_OK) st = prof.mark(3);
      if (prof.on) prof.used += 4;
      if (st == BP5_OK) st = phase_mark(mf, 3);
      if (st == BP5_OK) st = bp5_halo_scatter_add_finish(mf, dst);
    } else if (!split) {
      if (prof.on) mf->prof_mark = mf->ev_pool[prof.used + 2];
      st = launch_apply(mf, coef, src, dst, 0, mf->n_cells, true);
      marked = prof.on && mf->prof_mark == nullptr;
      mf->prof_mark = nullptr;
      if (st == BP5_OK && !marked) st = prof.mark(2);
      if (st == BP5_OK) st = prof.mark(3);
      if (prof.on) prof.used += 4;
      if (st == BP5_OK) st = phase_mark(mf, 3);
      if (st == BP5_OK) st = bp5_halo_scatter_add(mf, dst); // (fuse.on: dot-product corrections + ghost zeroing inside)
    } else {
      bp5_mf::DevPlan *dp = nullptr;
      st = get_plan_raw(mf, -block_cpt(mf), &dp);
      if (st == BP5_OK) st = halo_streams(mf);
      const int user_variant = mf->apply_variant;
      mf->apply_variant = 56;    // every range takes the block kernel, however few bricks it holds
      mf->defer_combine = true;  // one combine pass per window, launched here
      const bool has_boundary = mf->n_interior < mf->n_cells, in_one_launch = mf->can_wait_value == 1 && has_boundary && mf->n_interior > 0 && mf->blk_two_parts;
      if (st == BP5_OK && in_one_launch) {
        // ONE launch: every workgroup walks its share of the ghost-touching bricks first and counts itself in; the communication
        // stream waits for the count, combines the ghost rows and sends them while the same launch works through the interior bricks
        mf->blk_signal = true;
        st = launch_apply(mf, coef, src, dst, 0, mf->n_cells, true);
        mf->blk_signal = false;
        if (st == BP5_OK && hipStreamWaitValue64(mf->comm_stream, mf->d_signal, mf->signal_target, hipStreamWaitValueGte, ~0ull) != hipSuccess)
          st = fail(BP5_ERR_HIP, "hipStreamWaitValue64");
        if (st == BP5_OK && mf->n_ghost) {
          hipStream_t compute = mf->stream;
          mf->stream = mf->comm_stream; // the ghost-row combine runs on the communication stream, between the wait and the send
          st = launch_combine(mf, dp, dst, true, COMBINE_GHOST);
          mf->stream = compute;
        }
        if (st == BP5_OK) st = scatter_exchange(mf, dst, true, true);
      } else {
        if (st == BP5_OK && has_boundary) {
          const bool two = mf->blk_two_parts;
          mf->blk_two_parts = false; // (range launches)
          st = launch_apply(mf, coef, src, dst, mf->n_interior, mf->n_cells, true); // bricks that touch ghosts
          if (st == BP5_OK && mf->n_ghost) st = launch_combine(mf, dp, dst, true, COMBINE_GHOST);
          if (st == BP5_OK) st = scatter_exchange(mf, dst, true); // send the ghost rows / post the receives: under the interior bricks
          if (st == BP5_OK && mf->n_interior) st = launch_apply(mf, coef, src, dst, 0, mf->n_interior, true);
          mf->blk_two_parts = two;
        } else { // (a rank without ghost-touching cells only receives: post the receives, then all bricks)
          if (st == BP5_OK) st = scatter_exchange(mf, dst, true);
          if (st == BP5_OK) st = launch_apply(mf, coef, src, dst, 0, mf->n_cells, true);
        }
      }
      mf->defer_combine = false;
      mf->apply_variant = user_variant;
      if (st == BP5_OK) st = prof.mark(2); // (the profile brackets the brick launch(es))
      if (st == BP5_OK) st = launch_combine(mf, dp, dst, true, mf->n_ghost ? COMBINE_OWNED : COMBINE_ALL);
      if (st == BP5_OK) st = prof.mark(3);
      if (prof.on) prof.used += 4;
      if (st == BP5_OK) st = phase_mark(mf, 3);
      if (st == BP5_OK) st = bp5_halo_scatter_add_finish(mf, dst);
    }
    *n_cols = mf->fuse.n_cols;
    const bool ghosts_zeroed = mf->fuse.ghosts_zeroed;
    mf->fuse = bp5_mf::Fuse{};
    mf->defer_combine = false;
    BP5_TRY(st);
    if (!ghosts_zeroed) BP5_TRY(bp5_halo_zero_ghosts(mf, src)); // (a rank that owns no interface DoFs launched no unpack kernel; dst: scatter_add_finish)
    // no Dirichlet copy: the write-out stored v = p on this rank's Dirichlet rows and the unpack kernel leaves them alone
    return BP5_OK;
  }
  if (dist) { // phased application: the exchange overlaps the interior cells (apply_overlapped)
    BP5_TRY(prof.mark(0));
    BP5_TRY(prof.mark(1));
    if (prof.on) mf->prof_mark = mf->ev_pool[prof.used + 2];
    const int st = apply_overlapped(mf, coef, src, dst, zero);
    const bool marked = prof.on && mf->prof_mark == nullptr;
    mf->prof_mark = nullptr;
    BP5_TRY(st);
    if (!marked) BP5_TRY(prof.mark(2));
    BP5_TRY(prof.mark(3));
    if (prof.on) prof.used += 4;
    BP5_TRY(bp5_halo_zero_ghosts(mf, src));
    return bp5_copy_constrained(mf, src, dst);
  }
  // kernels that accumulate with atomics need a zeroed target (owner-scatter kernels define every entry themselves)
  const bool owner_scatter = variant_overwrites(mf, effective_variant(mf, 0, mf->n_cells));
  BP5_TRY(prof.mark(0));
  if (zero && !owner_scatter) {
    if (!mf->solver_prezeroed) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream)); // (else: the update kernel stored the zeros)
    zero = false;
  }
  BP5_TRY(prof.mark(1));
  if (prof.on) mf->prof_mark = mf->ev_pool[prof.used + 2];
  if (fusing) { mf->fuse.on = true; mf->fuse.p = src; mf->fuse.r = fuse_r; mf->fuse.n_cols = 0; }
  const int st = launch_apply(mf, coef, src, dst, 0, mf->n_cells, zero);
  if (fusing) { *n_cols = mf->fuse.n_cols; mf->fuse = bp5_mf::Fuse{}; }
  const bool marked = prof.on && mf->prof_mark == nullptr;
  mf->prof_mark = nullptr;
  BP5_TRY(st);
  if (!marked) BP5_TRY(prof.mark(2));
  BP5_TRY(prof.mark(3));
  if (prof.on) prof.used += 4;
  if (fusing) return BP5_OK; // Dirichlet DoFs were written by the fused write-out
  if (mf->solver_copies_dirichlet) return BP5_OK; // ... or will be by the solver's dot-product kernel, which reads src and dst anyway
  return bp5_copy_constrained(mf, src, dst);
}

static int poll_state(bp5_mf *mf)
{
  HIP_TRY(hipMemcpyAsync(mf->h_st, mf->d_st, ST_COUNT * sizeof(int), hipMemcpyDeviceToHost, mf->stream));
  HIP_TRY(hipMemcpyAsync(mf->h_sc, mf->d_sc, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, mf->stream));
  HIP_TRY(hipStreamSynchronize(mf->stream));
  return BP5_OK;
}

// cg.solve(A, x, b, preconditioner): the solvers need nothing of A but vmult (bp5/solver.h:25-30,377,475).  user == nullptr:
// the built-in Poisson operator (coef); otherwise the caller's operator through its callback.
static int cg_solve_impl(bp5_mf *mf, const double *coef, bp5_vmult_fn user, void *user_ctx, const double *diag, const double *b, double *x,
                         const bp5_cg_params *prm, bp5_cg_result *res)
{
  if (!mf || (!user && !coef && mf->geometry_mode != BP5_GEOM_AFFINE) || !b || !x || !prm || !res) return fail(BP5_ERR_INVALID, "null argument");
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
  if (prof.on) { // create the bracketing events before the timed region starts
    const size_t want = 4 * (size_t)std::min(prm->max_iter, ApplyProfile::MAX_PROFILED);
    while (mf->ev_pool.size() < want) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); mf->ev_pool.push_back(e); }
  }
  {
    auto &ph = mf->phase;
    ph.on = prm->profile == 2 && prm->variant == BP5_CG_MERGED;
    ph.it = 0;
    if (ph.on) {
      const size_t want = (size_t)bp5_mf::PhaseProfile::MAX_ITERS * bp5_mf::PhaseProfile::MARKS;
      while (ph.ev.size() < want) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); ph.ev.push_back(e); }
      ph.recorded.assign(bp5_mf::PhaseProfile::MAX_ITERS, 0);
    }
  }
  struct PhaseGuard { bp5_mf *m; ~PhaseGuard() { m->phase.on = false; } } phase_guard{mf};
  const hipEvent_t ev0 = mf->ev_solve[0], ev1 = mf->ev_solve[1];
  // h = A d.  dst is fully defined by the call (the reference zeroes it in update_a* for its atomic scatter)
  auto vmult = [&](double *src, double *dst, const double *fuse_r, uint32_t *n_cols) -> int {
    if (!user) return solver_vmult(mf, coef, src, dst, true, prof, fuse_r, n_cols);
    BP5_TRY(prof.mark(0));
    BP5_TRY(prof.mark(1));
    const int st = user(user_ctx, dst, src);
    if (st != BP5_OK) return fail(st, "the operator's vmult callback reported a failure");
    BP5_TRY(prof.mark(2));
    BP5_TRY(prof.mark(3));
    if (prof.on) prof.used += 4;
    return BP5_OK;
  };
  // scalars: tolerance + iteration cap
  mf->h_sc[SC_TOL] = prm->abs_tol;
  HIP_TRY(hipMemcpyAsync(mf->d_sc + SC_TOL, mf->h_sc + SC_TOL, sizeof(double), hipMemcpyHostToDevice, s));
  mf->h_st[ST_MAXIT] = prm->max_iter;
  HIP_TRY(hipMemcpyAsync(mf->d_st + ST_MAXIT, mf->h_st + ST_MAXIT, sizeof(int), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s)); // pinned staging words are reused below
  HIP_TRY(hipEventRecord(ev0, s));
  const bool plain = prm->variant == BP5_CG_PLAIN;
  bool fused_dots = false;
  mf->fuse = bp5_mf::Fuse{}; // (nothing of an earlier, failed solve survives)
  const int check = prm->check_every;
  int status = BP5_OK;
  // fused dot products: whenever the operator resolves to the packed block kernel on all cells of one rank (merged solver: and D == 1;
  // the plain solver takes only d.h = the quadrature-point energy from the kernel, which no preconditioner enters).
  // Across ranks the fused iteration keeps its dot products in every exchange schedule (solver_vmult): unsplit (bp5_mf_set_overlap 0:
  // gather, one launch, combine, scatter-add on the compute stream), boundary-first (1, the reference's setting: the ghost-touching
  // bricks come first, their rows travel to the owners on the communication stream under the interior bricks), and the automatic
  // choice (2): one launch, ghost rows combined first, the exchange under the owned-row combine.
  const bool dist_solve = mf->comm && !mf->neighbors.empty();
  bool split = false, split_possible = false, late = false;
  if (!user && mf->cg_fusion && (plain || !diag) && block_lpc(mf->degree) != 0 && mf->geometry_mode == BP5_GEOM_MERGED6 &&
      (effective_variant(mf, 0, mf->n_cells) == 56 || (mf->degree == 4 && effective_variant(mf, 0, mf->n_cells) == 63))) {
    bp5_mf::DevPlan *dp = nullptr;
    BP5_TRY(get_plan_raw(mf, -block_cpt(mf), &dp));
    fused_dots = dp->packed && dp->covers_all && (dp->n_shared == 0 || dp->cr_tile);
    if (fused_dots && dist_solve) {
      uint32_t b0_, b1_;
      const bool possible = !mf->h_block_off.empty() && (mf->n_interior == 0 || mf->n_interior == mf->n_cells || block_aligned(mf, 0, mf->n_interior, &b0_, &b1_));
      split_possible = possible;
      split = possible && mf->overlap == 1;
      if (mf->overlap == 1 && !split) fused_dots = false; // explicit overlap on a mesh that cannot run boundary-first: 3-phase schedule, separate dot products
      // automatic: one launch, ghost rows combined first, exchange under the owned-row combine (needs the run-length combine windows)
      late = fused_dots && mf->overlap == 2 && (dp->n_shared == 0 || (dp->cr_tile && !mf->combine_csr));
    }
  }
  struct OverlapGuard { // the fused exchanges choose their streams themselves (solver_vmult), whatever the slab size
    bp5_mf *m; int saved;
    ~OverlapGuard() { m->overlap = saved; m->cg_split = false; m->cg_late = false; m->blk_two_parts = false; m->blk_signal = false; }
  } overlap_guard{mf, mf->overlap};
  if (fused_dots && dist_solve) mf->overlap = 0;
  mf->cg_split = split;
  mf->cg_late = late;
  // whole-range launches of a distributed fused solve walk the ghost-touching bricks first in EITHER exchange schedule: same workgroup
  // ranges, same dot-product columns -- the two schedules then differ only in where the exchange is enqueued and give the same bits
  mf->blk_two_parts = fused_dots && dist_solve && split_possible;
  // one rank, separate dot-product kernel: the two small launches around an operator that scatters with atomics fold into their
  // neighbours -- the update kernel stores the zeros the operator needs in h / v (it holds the values in registers for the last time),
  // the dot-product kernel applies the Dirichlet copy while it reads both vectors (bitmap of the Dirichlet DoFs)
  const bool fold_enabled = mf->tune[BP5_TUNE_FOLD_SMALL] != 0; // A/B knob of the handle
  const bool fold_small = fold_enabled && !fused_dots && !user && !dist_solve;
  const bool prezero = fold_small && mf->n_ghost == 0 && !variant_overwrites(mf, effective_variant(mf, 0, mf->n_cells)); // (the update kernels cover owned entries)
  struct FoldGuard { bp5_mf *m; ~FoldGuard() { m->solver_prezeroed = m->solver_copies_dirichlet = false; } } fold_guard{mf};
  auto folded_vmult = [&](double *src, double *dst) -> int { // the operator between a zero-storing update and a copying dot-product kernel
    mf->solver_prezeroed = prezero;
    mf->solver_copies_dirichlet = fold_small;
    const int st_v = vmult(src, dst, nullptr, nullptr);
    mf->solver_prezeroed = mf->solver_copies_dirichlet = false;
    return st_v;
  };

  if (plain) {
    // g = -b, d = -D g, x = 0   (x0 = 0 short-circuit, bp5/solver.h:375-381)
    hipLaunchKernelGGL(cg_init_kernel, dim3(grid1), dim3(VB), 0, s, b, diag, x, g, d, n, mf->d_partials);
    hipLaunchKernelGGL(finalize_kernel<2>, dim3(2), dim3(VB), 0, s, mf->d_partials, grid1, mf->d_sc + SC_GG, (const int *)nullptr);
    KERNEL_CHECK();
    BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_GG, 2));
    hipLaunchKernelGGL(cg_init_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
    KERNEL_CHECK();
    if (prezero) HIP_TRY(hipMemsetAsync(h, 0, mf->n_local() * sizeof(double), s)); // once: every later zero-fill is stored by cg_update_kernel
    for (int it = 1; it <= prm->max_iter; ++it) {
      if (fused_dots) { // d.h = sum over the cells of the quadrature-point energy (+ d^2 on Dirichlet rows, where h = d): row 0 of the fused sums
        uint32_t n_cols = 0;
        BP5_TRY(vmult(d, h, nullptr, &n_cols)); // (no residual vector: the write-out does not read g)
        hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(VB), 0, s, mf->d_partials, (int)n_cols, mf->d_sc + SC_DH, mf->d_st);
      } else {
        BP5_TRY(folded_vmult(d, h));
        if (fold_small) hipLaunchKernelGGL(cg_dh_kernel, dim3(grid2), dim3(VB), 0, s, d, h, n, mf->d_partials, (const uint32_t *)mf->d_constrained_bits);
        else hipLaunchKernelGGL(dot_kernel, dim3(grid2), dim3(VB), 0, s, d, h, n, mf->d_partials);
        hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(VB), 0, s, mf->d_partials, grid2, mf->d_sc + SC_DH, mf->d_st);
      }
      KERNEL_CHECK();
      BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_DH, 1));
      hipLaunchKernelGGL(cg_update_kernel, dim3(grid2), dim3(VB), 0, s, x, g, d, h, diag, n, mf->d_sc, mf->d_st, mf->d_partials, prezero);
      hipLaunchKernelGGL(finalize_kernel<2>, dim3(2), dim3(VB), 0, s, mf->d_partials, grid2, mf->d_sc + SC_GG, mf->d_st);
      KERNEL_CHECK();
      BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_GG, 2));
      hipLaunchKernelGGL(cg_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
      hipLaunchKernelGGL(cg_direction_kernel, dim3(stream_grid_flat(mf, n, 2)), dim3(VB), 0, s, d, g, diag, n, mf->d_sc, mf->d_st);
      KERNEL_CHECK();
      if (check > 0 && it % check == 0 && it < prm->max_iter) {
        BP5_TRY(poll_state(mf));
        if (mf->h_st[ST_DONE]) break;
      }
    }
  } else {
    // SolverCGFullMerge: g == r, d == p, h == v
    // (v: the fused block kernel of a one-rank solve overwrites every entry before anything reads it; every other path keeps the zero-fill)
    hipLaunchKernelGGL(cgm_init_kernel, dim3(grid1), dim3(VB), 0, s, b, x, g, d, h, diag, n, mf->d_partials, !(fused_dots && !dist_solve));
    hipLaunchKernelGGL(finalize_kernel<2>, dim3(2), dim3(VB), 0, s, mf->d_partials, grid1, mf->d_sc + SC_GG, (const int *)nullptr);
    KERNEL_CHECK();
    BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_GG, 2));
    hipLaunchKernelGGL(cgm_init_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
    KERNEL_CHECK();
    // update kernels: U chunks of 256 pairs per workgroup (flat launch: one trip per workgroup), all loads ahead of the first store.
    // profiles/r4 hbm_sweep: U = 1 flat 5.5-5.9 TB/s, the capped grid-stride grid of rounds 1-3 (U = 4, 2048 workgroups) 4.8-4.9
    const int unroll = mf->tune[BP5_TUNE_UPDATE_UNROLL];
    const int gridu = stream_grid_flat(mf, n, 2 * unroll);
    // v and x non-temporally: at every size since round 4 (-1 = on; profiles/r4 ab_update_*: 3.741 -> 3.707 ms per iteration at 1e8 DoFs with the flat
    // launch, 0.4075 -> 0.4025 at 1e7; rounds 1-3 followed the streaming policy of the metric loads, which is off at 1e8 DoFs)
    const bool streaming = mf->tune[BP5_TUNE_UPDATE_NT] != 0;
    auto launch_update = [&](int mode) {
#define BP5_UPD(M, U, ZV, NT) hipLaunchKernelGGL((cgm_update_kernel<M, U, ZV, NT>), dim3(gridu), dim3(VB), 0, s, d, g, h, x, diag, n, mf->d_sc, mf->d_st)
#define BP5_UPD_U(M, ZV, NT) do { if (unroll == 1) BP5_UPD(M, 1, ZV, NT); else if (unroll == 2) BP5_UPD(M, 2, ZV, NT); else BP5_UPD(M, 4, ZV, NT); } while (0)
      if (mode == 0) { BP5_UPD_U(0, false, false); return; }
      if (prezero) { if (mode == 1) BP5_UPD_U(1, true, false); else BP5_UPD_U(2, true, false); return; }
      if (streaming) { if (mode == 1) BP5_UPD_U(1, false, true); else BP5_UPD_U(2, false, true); return; }
      if (mode == 1) BP5_UPD_U(1, false, false); else BP5_UPD_U(2, false, false);
#undef BP5_UPD_U
#undef BP5_UPD
    };
    const bool fused = fused_dots;
    // fused iteration across ranks: the values of the NEW p at the DoFs this rank sends are computed into the send buffer first, so the
    // ghost gather of p runs on the communication stream underneath the update kernel (which touches owned entries only); the operator
    // then just waits for the event
    const bool early_gather_enabled = mf->tune[BP5_TUNE_EARLY_GATHER] != 0; // A/B knob of the handle
    const bool early_gather = fused && dist_solve && early_gather_enabled;
    auto gather_under_update = [&](int mode) -> int {
      BP5_TRY(halo_streams(mf));
      const uint32_t ns = mf->send_off.back();
      if (ns) {
        const dim3 gr((ns + 255) / 256), bl(256);
        if (mode == 0) hipLaunchKernelGGL(cgm_pack_updated_kernel<0>, gr, bl, 0, s, mf->d_send_idx, ns, d, g, h, diag, mf->d_sc, mf->d_st, mf->d_sendbuf);
        else hipLaunchKernelGGL(cgm_pack_updated_kernel<1>, gr, bl, 0, s, mf->d_send_idx, ns, d, g, h, diag, mf->d_sc, mf->d_st, mf->d_sendbuf);
        KERNEL_CHECK();
      }
      return gather_exchange(mf, d, true);
    };
    int it = 1;
    for (; it <= prm->max_iter; ++it) {
      const int mode = it == 1 ? 0 : it % 2 == 0 ? 1 : 2;
      BP5_TRY(phase_mark(mf, 0));
      if (early_gather) { BP5_TRY(gather_under_update(mode)); mf->fuse.gather_in_flight = true; }
      if (mode != 0) launch_update(mode); // (mode 0, p = -D r: written by cgm_init_kernel already)
      KERNEL_CHECK();
      BP5_TRY(phase_mark(mf, 1));
      const bool one_launch = fused && !mf->comm; // no all-reduce between the local sums and the scalar step
      if (fused) {
        uint32_t n_cols = 0;
        BP5_TRY(vmult(d, h, g, &n_cols));
        BP5_TRY(phase_mark(mf, 4));
        if (one_launch) hipLaunchKernelGGL(cgm_finalize4_kernel<true>, dim3(1), dim3(FIN4_THREADS), 0, s, mf->d_partials, (int)n_cols, mf->d_sc, mf->d_st);
        else hipLaunchKernelGGL(cgm_finalize4_kernel<false>, dim3(1), dim3(FIN4_THREADS), 0, s, mf->d_partials, (int)n_cols, mf->d_sc, mf->d_st);
      } else {
        BP5_TRY(folded_vmult(d, h)); // (h: zeroed by cgm_init_kernel before the first, by the update kernel before every later application)
        BP5_TRY(phase_mark(mf, 4));
        hipLaunchKernelGGL(cgm_dots_kernel, dim3(grid2), dim3(VB), 0, s, d, g, h, diag, n, mf->d_st, mf->d_partials,
                           fold_small ? (const uint32_t *)mf->d_constrained_bits : (const uint32_t *)nullptr);
        hipLaunchKernelGGL(finalize_kernel<7>, dim3(7), dim3(VB), 0, s, mf->d_partials, grid2, mf->d_sc + SC_R0, mf->d_st);
      }
      KERNEL_CHECK();
      BP5_TRY(phase_mark(mf, 5));
      if (!one_launch) {
        BP5_TRY(bp5_comm_allreduce_sum(mf, mf->d_sc + SC_R0, 7));
        BP5_TRY(phase_mark(mf, 6));
        hipLaunchKernelGGL(cgm_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
        KERNEL_CHECK();
      } else
        BP5_TRY(phase_mark(mf, 6));
      BP5_TRY(phase_mark(mf, 7));
      if (mf->phase.on) ++mf->phase.it;
      if (check > 0 && it % check == 0 && it < prm->max_iter) {
        BP5_TRY(poll_state(mf));
        if (mf->h_st[ST_DONE]) break;
      }
    }
    // epilogue x update (solver.h:510-526) runs inside the next update kernel; with max_iter == 0 no
    // iteration has been done and nothing is pending
    if (prm->max_iter > 0) {
      launch_update(1);
      hipLaunchKernelGGL(cgm_control_kernel, dim3(1), dim3(1), 0, s, mf->d_sc, mf->d_st);
      KERNEL_CHECK();
    }
  }
  HIP_TRY(hipEventRecord(ev1, s));
  BP5_TRY(poll_state(mf));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
  res->iterations = mf->h_st[ST_ITER];
  res->residual = mf->h_sc[SC_RES];
  res->initial_residual = mf->h_sc[SC_RES0];
  res->solve_ms = ms;
  res->apply_ms_avg = res->operator_ms_avg = 0.0;
  res->apply_launches = prof.used / 4;
  res->dot_products_fused = fused_dots ? 1 : 0;
  res->exchange_schedule = !dist_solve ? 0 : split ? 2 : late ? 4 : fused_dots ? 1 : overlap_wanted(mf) ? 3 : 1;
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
  memset(res->apply_kernel, 0, sizeof(res->apply_kernel));
  if (!user) strncpy(res->apply_kernel, mf->last_apply_kernel, sizeof(res->apply_kernel) - 1);
  for (double &v : res->phase_ms) v = 0.0;
  if (mf->phase.on) { // averages over the stamped iterations after the first (whose update kernel is the cheap update_a0)
    using PP = bp5_mf::PhaseProfile;
    const int n_it = std::min(mf->phase.it, (int)PP::MAX_ITERS);
    int counted = 0;
    for (int i = n_it > 1 ? 1 : 0; i < n_it; ++i) {
      const uint8_t rec = mf->phase.recorded[i];
      const uint8_t need = (1u << 0) | (1u << 1) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 7);
      if ((rec & need) != need) continue;
      auto ms = [&](int a, int b, double &out) -> int {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, mf->phase.ev[(size_t)i * PP::MARKS + a], mf->phase.ev[(size_t)i * PP::MARKS + b]));
        out += t;
        return BP5_OK;
      };
      const bool g = rec & (1u << 2), x = rec & (1u << 3);
      BP5_TRY(ms(0, 1, res->phase_ms[BP5_PHASE_UPDATE]));
      if (g) BP5_TRY(ms(1, 2, res->phase_ms[BP5_PHASE_GATHER_WAIT]));
      BP5_TRY(ms(g ? 2 : 1, x ? 3 : 4, res->phase_ms[BP5_PHASE_OPERATOR]));
      if (x) BP5_TRY(ms(3, 4, res->phase_ms[BP5_PHASE_EXCHANGE]));
      BP5_TRY(ms(4, 5, res->phase_ms[BP5_PHASE_REDUCE]));
      BP5_TRY(ms(5, 6, res->phase_ms[BP5_PHASE_ALLREDUCE]));
      BP5_TRY(ms(6, 7, res->phase_ms[BP5_PHASE_CONTROL]));
      BP5_TRY(ms(0, 7, res->phase_ms[BP5_PHASE_ITERATION]));
      ++counted;
    }
    if (counted) for (double &v : res->phase_ms) v /= counted;
  }
  if (mf->h_st[ST_BREAKDOWN]) status = fail(BP5_ERR_BREAKDOWN, "CG breakdown: p.Ap is zero or NaN");
  return status;
}

extern "C" int bp5_cg_solve(bp5_mf *mf, const double *coef, const double *diag, const double *b, double *x, const bp5_cg_params *prm,
                            bp5_cg_result *res)
{
  return cg_solve_impl(mf, coef, nullptr, nullptr, diag, b, x, prm, res);
}
extern "C" int bp5_cg_solve_operator(bp5_mf *mf, bp5_vmult_fn vmult, void *ctx, const double *diag, const double *b, double *x,
                                     const bp5_cg_params *prm, bp5_cg_result *res)
{
  if (!vmult) return fail(BP5_ERR_INVALID, "null vmult callback");
  return cg_solve_impl(mf, nullptr, vmult, ctx, diag, b, x, prm, res);
}
