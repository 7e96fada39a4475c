// Shared by the device translation units of libbp5: the handle, the launch templates of the fused operator kernels and the
// per-degree variant dispatch.  bp5_device.hip holds the C ABI; bp5_apply_pN.hip instantiate apply_degree_impl<N> (one
// translation unit per degree so that `make -j` compiles them side by side).
#pragma once
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
#include <tuple>
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

constexpr size_t STREAMING_MAX_DOFS = 24000000; // see bp5_mf_set_streaming (include/bp5.h)
struct bp5_mf {
  int degree = 0, quadrature = 0, coefficient = 0, n = 0, n3 = 0, device = 0;
  uint32_t n_cells = 0, n_interior = 0, n_owned = 0, n_ghost = 0, n_constrained = 0;
  int apply_variant = 0, n_cus = 0, geometry_mode = 0, march_max_steps = 32;
  int operator_kind = 0; // BP5_OP_POISSON | BP5_OP_HELMHOLTZ (seven planes: six merged + the mass plane a JxW)
  mutable int coef_planes_committed = 0; // planes of the metric array the caller has sized (bp5_mf_coef_size) or filled: set_operator may not change the count afterwards
  int n_planes() const { return operator_kind == BP5_OP_HELMHOLTZ ? 7 : 6; }
  uint32_t blk_b0 = 0, blk_b1 = 0; // block range of the next block-kernel launch (0,0 = all blocks)
  bool combine_csr = false; // A/B: per-DoF CSR combine kernel instead of the run-length one
  int block_max_wg = 0; // 0: persistent grid sized from the CU count; > 0: cap (tests force several blocks per workgroup)
  int streaming = -1; // bp5_mf_set_streaming: -1 chosen by size, 0 ordinary accesses, 1 non-temporal accesses to once-used data
  int auto_team = -1;  // -1 not decided; 1: the x-row team plan could be built (p = 1, 3 default)
  int auto_block = -1; // -1 not decided; 1: the caller's cell blocks fit three block-kernel workgroups per CU
  double *d_scalar_plane = nullptr, *d_gcell = nullptr;
  bool force_atomic_scatter = false, block_shared_atomic = false;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Tables tab, tab_gauss;
  // device arrays
  uint32_t *d_l2g = nullptr, *d_constrained = nullptr, *d_send_idx = nullptr;
  uint32_t *d_constrained_bits = nullptr; // bit i: DoF i is a Dirichlet DoF (owned range; the solver's dot-product kernel applies the copy)
  bool solver_prezeroed = false, solver_copies_dirichlet = false; // one operator application inside the merged solver: dst arrives zeroed / the Dirichlet copy follows in the dots kernel
  double *d_coords = nullptr, *d_tab = nullptr, *d_tab_gauss = nullptr;
  // Data mirror (lazy)
  uint32_t *d_l2g_padded = nullptr, *d_constraint_mask = nullptr;
  // hanging nodes (2:1 refinement): per-cell masks and the two 1-D interpolation matrices [2][n*n]; has_hanging: some mask != 0
  // layout of the six metric planes: plane-major [c][cell][q] (round 1) or cell-major [cell][c][q] (one contiguous 48 n^3-byte chunk per cell)
  uint64_t coef_plane_stride = 0, coef_cell_stride = 0;
  bool has_hanging = false;
  uint32_t *d_hang_mask = nullptr;
  double *d_hang_I = nullptr;
  double *d_inv_jac = nullptr, *d_JxW = nullptr, *d_qpoints = nullptr;
  uint32_t pad = 0;
  // halo plan
  std::vector<int> neighbors;
  std::vector<uint32_t> send_off, recv_off;
  double *d_sendbuf = nullptr, *d_recvbuf = nullptr;
  uint8_t *d_send_dirichlet = nullptr; // per send index: the owner DoF is a Dirichlet DoF
  bp5_comm *comm = nullptr;
  // halo exchange on its own stream (overlap with interior cells): created on first use
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_halo[4] = {nullptr, nullptr, nullptr, nullptr}; // packed / gathered / ghosts ready / received
  int overlap = 2;            // MatrixFree::AdditionalData::overlap_communication_computation (bp5/step-64.cu:241): 0 off, 1 on, 2 auto
  bool overlap_now = false;   // the decision for the exchange in flight (set by *_start)
  bool cg_fusion = true;      // SolverCGFullMerge: dot products inside the block kernel's write-out when the plan allows
  bool defer_combine = false; // block kernel on cell ranges: partial slab now, ONE combine pass after the last range
  bool cg_split = false;      // set by a solve with fused dot products across ranks: boundary-first schedule -- the ghost-touching bricks
                              // run first, one combine pass per window, the exchange travels under the interior bricks
  // block kernel launches over ALL bricks of a slab with ghosts: two parts per workgroup (its share of the ghost-touching bricks, then
  // its interior bricks); blk_signal: the launch counts the workgroups whose first part is written out at *d_signal (monotonic:
  // signal_target = the count after the last launch), so that the communication stream can wait for the ghost rows mid-kernel
  bool cg_late = false;       // ... or: all bricks in one launch, the ghost rows combined first, the exchange under the owned-row combine
  bool blk_two_parts = false, blk_signal = false;
  unsigned long long *d_signal = nullptr;
  uint64_t signal_target = 0;
  int wait_value_ok = -1;     // -1 not probed; 1: hipDeviceAttributeCanUseStreamWaitValue AND the producer / consumer self-check saw a mid-kernel release
  int can_wait_value = -1;    // wait_value_ok and not switched off by BP5_TUNE_BOUNDARY_FIRST = 0
  // per-handle tuning / A-B knobs (bp5.h: BP5_TUNE_*): initial values from the environment, read once by bp5_mf_create
  int tune[BP5_TUNE_COUNT] = {1, 1, 0, 1, 1, 1, 1, -1, 16, 1, 0, 1};
  bool cell_interiors_first = false; // the mesh numbers the DoFs strictly inside a cell ahead of all others, cell after cell, x fastest (bp5_mesh_desc.dof_numbering = 2)
  // solver workspace
  double *d_partials = nullptr, *d_sc = nullptr, *d_scalar = nullptr;
  int *d_st = nullptr;
  double *ws_g = nullptr, *ws_d = nullptr, *ws_h = nullptr, *d_evec = nullptr;
  char *ws_base = nullptr;
  unsigned long long *d_stamps = nullptr;
  double *h_sc = nullptr; // pinned
  int *h_st = nullptr;    // pinned
  std::vector<hipEvent_t> ev_pool;
  hipEvent_t ev_solve[2] = {nullptr, nullptr}; // start / stop of a solve (owned by the handle: no leak on error paths)
  hipEvent_t prof_mark = nullptr; // profiling: recorded once before the combine pass (= end of the dominant kernel)
  char last_apply_kernel[96] = ""; // the operator kernel launched last, named as a profiler prints it (bp5_cg_result.apply_kernel)
  // profile == 2: stamps at the phase boundaries of every merged-CG iteration (bp5_cg_result.phase_ms)
  struct PhaseProfile {
    static constexpr int MARKS = 8, MAX_ITERS = 64;
    std::vector<hipEvent_t> ev;   // [MAX_ITERS][MARKS]
    std::vector<uint8_t> recorded; // [MAX_ITERS] bit k: mark k recorded
    bool on = false;
    int it = 0;                   // iteration being stamped (0-based); stamps beyond MAX_ITERS are dropped
  } phase;
  // team plans of the team-assembled kernel, keyed by cells per team
  std::vector<uint32_t> h_l2g;
  struct DevPlan {
    uint32_t *off = nullptr, *dofs = nullptr, *sh_dof = nullptr, *sh_off = nullptr, *sh_slot = nullptr;
    uint16_t *pos = nullptr;
    uint8_t *cell_round = nullptr, *team_rounds = nullptr;
    double *partial = nullptr;
    uint32_t *cell_off = nullptr, *pass_cell = nullptr, *pass_off = nullptr, *run_off = nullptr, *runs = nullptr, *gidx = nullptr;
    uint16_t *packed = nullptr;
    // lattice form of the packed indices (structured bricks: every cell-local DoF's list slot and DoF follow in closed form from the cell's
    // position in its brick -- no per-DoF index stream at all): per block 64 words (bp5_kernels.hpp: BLOCK_LATTICE_WORDS), per cell its position
    uint32_t *lattice = nullptr;
    uint16_t *cell_pos = nullptr;
    uint32_t n_lattice_blocks = 0;
    std::vector<double> h_cost;                       // [n_groups+1] prefix sum of the estimated cost of the blocks (pass units)
    // block ranges of the persistent workgroups, one device array per (n_wg, first block, end block) ever launched: the
    // interior / boundary ranges of the overlapped schedule alternate, nothing is freed or re-uploaded inside a solve
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, uint32_t *> *wg_blocks = nullptr; // key: n_wg, first, end, first block of part 0 (0: one part)
    uint32_t *cr_start = nullptr, *cr_dof0 = nullptr, *cr_soff = nullptr, *cr_slots = nullptr, *cr_tile = nullptr; // run-length combine
    // face carry (BP5_TUNE_FACE_CARRY; bp5_kernels.hpp: BLOCK_CARRY_MAX): which faces stay in LDS depends on the workgroups' block ranges, so the
    // combine tables WITHOUT the carried faces are kept per partition (the key of wg_blocks); cr_active = the tables the last block launch
    // asks its combine pass to use (NULL: the plan's own, every shared DoF)
    struct CombineTables { uint32_t *start = nullptr, *dof0 = nullptr, *soff = nullptr, *slots = nullptr, *tile = nullptr; uint32_t n_shared = 0, n_shared_owned = 0; };
    std::vector<uint32_t> h_cr_start, h_cr_dof0, h_cr_soff, h_cr_slots; // host form of cr_* (runs, not DoFs: small)
    std::vector<uint32_t> h_carry_dof, h_carry_len;                      // [n_groups] first DoF / DoF count of the face block g can hand to block g + 1 (0: none)
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, CombineTables> *cr_carry = nullptr;
    const CombineTables *cr_active = nullptr;
    uint32_t n_shared = 0, n_groups = 0, max_list = 0, max_runs = 0;
    uint32_t n_shared_owned = 0; // shared DoFs are listed in ascending order: ordinals [0, n_shared_owned) are owned DoFs, the rest ghosts
    bool covers_all = false;
  };
  std::vector<bool> h_constrained;   // per local DoF: Dirichlet DoF (run tables carry the flag)
  // fused CG dot products (SolverCGFullMerge on the block kernel): set by the solver around ONE operator application
  struct Fuse {
    bool on = false;
    const double *p = nullptr, *r = nullptr;
    uint32_t n_cols = 0; // columns of d_partials written so far (block kernel workgroups, then the combine pass)
    bool ghosts_zeroed = false; // the exchange's unpack kernel has zeroed the ghost ranges of v and p
    bool gather_in_flight = false; // the solver started the ghost gather of p under its update kernel
  } fuse;
  std::vector<uint32_t> h_block_off; // caller-provided cell blocks (may be empty)
  struct DevMarch { uint32_t *team_off = nullptr, *entries = nullptr; uint32_t n_teams = 0; };
  std::map<int, DevMarch> march_plans; // keyed by cells per team
  std::map<int, DevPlan> plans;
  size_t n_local() const { return (size_t)n_owned + n_ghost; }
};

// Lanes per cell of the block-assembled kernel's default shape (one transpose tile per cell used field after field; when the lanes
// of a cell sit in one wave -- LPC divides 64 -- the tile exchanges are wave-local, else they use the workgroup barrier).
// Cells per pass = 256 / LPC.
// p = 2, 5, 8 (round 3): n^2 = 9 / 36 / 81 lanes per cell -- cells span waves, so the tile exchanges go through the workgroup barrier,
// but 28 / 7 / 3 cells share a pass and hardly a lane idles (round 2: 16 / 64 / 128 lanes per cell, 44 / 44 / 37 % idle).  Same-box A/B
// at the config-4 sizes (profiles/r3 i_*): p = 2 18.1 against 17.7 GDoF/s, p = 5 (6x4x2 bricks) 25.5 against 23.1, p = 8 23.2 against
// 19.7.  The macros are the A/B knobs (-DBP5_LPC_P2=16 -DBP5_LPC_P5=64 -DBP5_LPC_P8=128 rebuilds round 2's shapes).
#ifndef BP5_LPC_P2
#define BP5_LPC_P2 9
#endif
#ifndef BP5_LPC_P5
#define BP5_LPC_P5 36
#endif
#ifndef BP5_LPC_P8
#define BP5_LPC_P8 81
#endif
constexpr int block_lpc(int degree) { return degree == 1 ? 4 : degree == 2 ? BP5_LPC_P2 : degree == 3 ? 16 : degree == 4 ? 32 : degree == 5 ? BP5_LPC_P5 : degree == 6 || degree == 7 ? 64 : degree == 8 ? BP5_LPC_P8 : 0; }
inline int block_cpt(const bp5_mf *mf) { return block_lpc(mf->degree) ? 256 / block_lpc(mf->degree) : 8; }

template <typename T>
inline int upload(T **dptr, const T *host, size_t count)
{
  HIP_TRY(hipMalloc((void **)dptr, std::max<size_t>(count, 1) * sizeof(T)));
  if (count) HIP_TRY(hipMemcpy(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
  return BP5_OK;
}

// defined in bp5_device.hip
// key > 0: uniform teams of `key` cells (team kernel); key < 0: cell blocks walked in passes of
// -key cells (block kernel) -- the caller's blocks if given, else groups of `default_block` cells
int get_plan_raw(bp5_mf *mf, int key, bp5_mf::DevPlan **dpo, int default_block = 64);
int get_plan(bp5_mf *mf, int cpt, bp5::TeamPlan &tp, bp5_mf::DevPlan **dpo);
// window: COMBINE_ALL every shared row; COMBINE_GHOST / COMBINE_OWNED only the rows of ghost / owned DoFs (the boundary-first exchange
// schedule completes the ghost rows before the interior bricks run); the windows need the run-length form of the pass
enum { COMBINE_ALL = 0, COMBINE_GHOST = 1, COMBINE_OWNED = 2, COMBINE_GHOST_THEN_OWNED = 3 }; // 3: one launch, ghost rows first + signal (fused solves)
int launch_combine(bp5_mf *mf, bp5_mf::DevPlan *dp, double *dst, bool set, int window = COMBINE_ALL);
// combine tables of one workgroup partition without the faces its workgroups carry from brick to brick (wb: the block ranges as uploaded for the kernel)
int build_carry_tables(bp5_mf *mf, bp5_mf::DevPlan *dp, const std::vector<uint32_t> &wb, uint32_t n_wg, bool two_parts, bp5_mf::DevPlan::CombineTables *out);
// [c0,c1) == union of whole cell blocks [b0,b1) of the caller's blocking?
bool block_aligned(const bp5_mf *mf, uint32_t c0, uint32_t c1, uint32_t *b0, uint32_t *b1);
// Non-temporal accesses to the data a CG iteration touches once (the operator's metric planes; v and x in the update kernel) keep it from evicting
// the vectors that ARE reused (p, r) from the 256 MB memory-side cache: a gain while those fit there (-6 % per iteration at 1e7 DoFs), a small loss
// beyond (+1 % at 1e8; profiles/r3/README.md, x_*)
inline bool streaming_accesses(const bp5_mf *mf) { return mf->streaming >= 0 ? mf->streaming != 0 : mf->n_local() <= STREAMING_MAX_DOFS; }

// ------------------------------------------------------------------------------------ operator launches
// 1-D tables as the kernels read them: the (anti)symmetric half of N and D (p <= 4), or their even-odd split (mv_even_odd)
template <int n>
inline void pack_even_odd(const double *M, bool anti, double *out)
{
  constexpr int c = (n + 1) / 2, h = n / 2, m = (n - 1) / 2;
  for (int q = 0; q < n; ++q)
    for (int i = 0; i < h; ++i) {
      const double E = 0.5 * (M[q * n + i] + M[q * n + n - 1 - i]), O = 0.5 * (M[q * n + i] - M[q * n + n - 1 - i]);
      if (q < c) out[q * h + i] = anti ? O : E;
      if (q < h) out[c * h + q * h + i] = anti ? E : O;
    }
  if (n & 1)
    for (int q = 0; q < c; ++q) out[c * h + h * h + q] = M[q * n + m];
}
template <int n>
inline void fill_shape(ShapeArg<n> &sh, const bp5_mf *mf)
{
  memcpy(sh.N, mf->tab.N, sizeof(sh.N));
  memcpy(sh.D, mf->tab.D, sizeof(sh.D));
  if constexpr (n >= EO_MIN_N) {
    pack_even_odd<n>(mf->tab.N, false, sh.N);
    pack_even_odd<n>(mf->tab.D, true, sh.D);
  }
}

template <int P, bool COLL, int TW, int LPC, int TPB, bool PF, int ABL = 0>
inline int launch_apply_t(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1)
{
  constexpr int n = P + 1;
  constexpr int CPT = 64 * TW / LPC;
  using L = LdsLayout<n, LPC>;
  ApplyArgs a{};
  a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst;
  a.plane_stride = mf->coef_plane_stride; a.cell_stride = (ABL & 1024) ? (uint64_t)mf->n3 : mf->coef_cell_stride; // (affine builds read ONE scalar plane)
  a.cell_begin = c0; a.cell_end = c1;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  a.hang_mask = mf->d_hang_mask; a.hang_I = mf->d_hang_I;
  if ((ABL & 2097152) && !mf->has_hanging) return fail(BP5_ERR_INVALID, "the hanging-node build needs constraint masks");
  a.n_teams = (c1 - c0 + CPT - 1) / CPT;
  const uint32_t nblk = (a.n_teams + TPB - 1) / TPB;
  a.teams_per_xcd = (nblk + 7) / 8;
  ShapeArg<n> sh;
  fill_shape(sh, mf);
  const size_t lds = (size_t)TPB * CPT * L::CS * sizeof(double);
  snprintf(mf->last_apply_kernel, sizeof(mf->last_apply_kernel), "apply_pencil_kernel<%d,%s,%d,%d,%d,%s,%d>", P, COLL ? "true" : "false", TW, LPC, TPB, PF ? "true" : "false", ABL);
  hipLaunchKernelGGL((apply_pencil_kernel<P, COLL, TW, LPC, TPB, PF, ABL>), dim3(a.teams_per_xcd * 8), dim3(64 * TW * TPB), lds, mf->stream, a,
                     sh);
  KERNEL_CHECK();
  return BP5_OK;
}

// LDS bytes of one block-kernel workgroup: transpose tiles of the cell slots (two per slot where the cells span waves: BlockPass::PP),
// the brick's accumulator (and its staged src: ABL & 524288), two run tables and two lattice tables.  The ONE formula: the launcher
// sizes the launch with it and the library's automatic kernel choice (effective_variant) counts workgroups per CU with it.
template <int P, bool COLL, int LPC, int ABL>
constexpr size_t block_lds_bytes(uint32_t max_list)
{
  return ((size_t)(256 / LPC) * (size_t)BlockPass<P, COLL, LPC, SC_OWNER_SET, ABL>::TILE_CS + ((ABL & 524288) ? 2 : 1) * (size_t)max_list) * sizeof(double) +
         ((ABL & 16384) ? (4 * BLOCK_MAX_RUNS + 2 * BLOCK_LATTICE_WORDS) * sizeof(uint32_t) : 0) + ((ABL & 268435456) ? 2 * BLOCK_CARRY_MAX * sizeof(double) : 0);
}
// ... of the default shape of a degree (sequential tiles, metric loaded in its own pass, run-length write-out, packed indices; the
// Helmholtz, hanging-node, fused-CG and lattice builds have the same tiles)
inline size_t block_default_lds_bytes(int degree, uint32_t max_list)
{
  constexpr int D = 2048 + 8192 + 16384 + 262144;
  switch (degree) {
    case 1: return block_lds_bytes<1, false, block_lpc(1), D>(max_list);
    case 2: return block_lds_bytes<2, false, block_lpc(2), D>(max_list);
    case 3: return block_lds_bytes<3, false, block_lpc(3), D>(max_list);
    case 4: return block_lds_bytes<4, false, block_lpc(4), D>(max_list);
    case 5: return block_lds_bytes<5, false, block_lpc(5), D>(max_list);
    case 6: return block_lds_bytes<6, false, block_lpc(6), D>(max_list);
    case 7: return block_lds_bytes<7, false, block_lpc(7), D>(max_list);
    case 8: return block_lds_bytes<8, false, block_lpc(8), D>(max_list);
  }
  return ~(size_t)0;
}

// block-assembled kernel; falls back to the team kernel path when the range is partial
template <int P, bool COLL, int LPC, int ABL = 0>
inline int launch_block_t(bp5_mf *mf, const double *coef, const double *src, double *dst, bool overwrite)
{
  constexpr int n = P + 1;
  constexpr int CPT = 256 / LPC;
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan_raw(mf, -CPT, &dp));
  const size_t lds = block_lds_bytes<P, COLL, LPC, ABL>(dp->max_list);
  if ((ABL & 16384) && dp->max_runs > (uint32_t)BLOCK_MAX_RUNS) return fail(BP5_ERR_UNSUPPORTED, "too many runs per block for the run-length write-out");
  if ((ABL & 262144) && !dp->packed) return fail(BP5_ERR_UNSUPPORTED, "more than 128 runs per block: packed indices unavailable");
  if ((ABL & 16777216) && !(dp->lattice && dp->n_lattice_blocks == dp->n_groups)) return fail(BP5_ERR_INVALID, "the lattice build needs a plan of lattice blocks only");
  if (lds > 160 * 1024) return fail(BP5_ERR_UNSUPPORTED, "cell block does not fit in LDS; pass smaller cell blocks");
  BlockPlan bp{}; // value-initialised: a field this launcher forgets is null, not garbage
  bp.pass_cell = dp->pass_cell; bp.pass_off = dp->pass_off; bp.off = dp->off; bp.dofs = dp->dofs; bp.pos = dp->pos; bp.gidx = dp->gidx;
  bp.packed = dp->packed;
  bp.lattice = dp->lattice; bp.cell_pos = dp->cell_pos;
  bp.cell_round = dp->cell_round; bp.blk_rounds = dp->team_rounds; bp.partial = dp->partial; bp.n_blocks = dp->n_groups;
  bp.run_off = dp->run_off; bp.runs = dp->runs; bp.max_list = dp->max_list;
  // a block-aligned cell range: only these blocks run, accumulate mode; DoFs shared with other blocks go to dst by
  // atomics (the partial slab + combine pass needs every block of the plan in the launch)
  const bool sub_range = mf->blk_b1 > mf->blk_b0 && (mf->blk_b0 != 0 || mf->blk_b1 != dp->n_groups);
  bp.blk_begin = sub_range ? mf->blk_b0 : 0;
  if (sub_range) bp.n_blocks = mf->blk_b1 - mf->blk_b0;
  // ... unless the caller runs ALL blocks in several range launches and one combine pass after the last one
  // (mf->defer_combine: the overlapped halo schedule): then every launch is the ordinary owner-store kernel
  const bool atomic_shared = mf->block_shared_atomic || (sub_range && !mf->defer_combine);
  if (sub_range && overwrite && !mf->defer_combine) return fail(BP5_ERR_INVALID, "a cell range cannot overwrite dst");
  // persistent grid: two workgroups per CU (LDS budget), a multiple of 8 for the XCD mapping
  if (!mf->n_cus) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, mf->device));
    mf->n_cus = prop.multiProcessorCount;
  }
  constexpr int wg_reg = block_wg_per_cu<P, ABL>(); // what the registers allow (launch bounds of the kernel) ...
  const int wg_per_cu = std::max(1, std::min<int>(wg_reg, (int)(160 * 1024 / std::max<size_t>(lds, 1)))); // ... and what the LDS of this plan allows
  uint32_t n_wg = (uint32_t)(mf->n_cus * wg_per_cu);
  if (mf->block_max_wg > 0) n_wg = std::min<uint32_t>(n_wg, (uint32_t)mf->block_max_wg);
  n_wg = std::max<uint32_t>(8, std::min<uint32_t>(n_wg, (bp.n_blocks + 7) / 8 * 8) / 8 * 8);
  bp.n_wg = n_wg;
  { // block ranges of the persistent workgroups: equal shares of the estimated COST (thin or partial bricks are cheaper per
    // block but dearer per cell than full ones), cached
    const uint32_t B0 = bp.blk_begin, B1 = bp.blk_begin + bp.n_blocks;
    // two parts (whole-range launches on a slab whose ghost-touching bricks come last, [Bs, B1)): every workgroup takes an equal share
    // of the ghost-touching bricks first, then interior bricks up to an equal share of the TOTAL cost
    uint32_t Bs = 0, bs0_ = 0;
    const bool two_parts = mf->blk_two_parts && !sub_range && mf->n_interior > 0 && mf->n_interior < mf->n_cells && block_aligned(mf, 0, mf->n_interior, &bs0_, &Bs) && Bs > B0 && Bs < B1;
    if (!dp->wg_blocks) dp->wg_blocks = new std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, uint32_t *>;
    auto key = std::make_tuple(n_wg, B0, B1, two_parts ? Bs : 0u);
    auto itw = dp->wg_blocks->find(key);
    auto make_ranges = [&]() {
      const std::vector<double> &pc = dp->h_cost;
      std::vector<uint32_t> wb((two_parts ? 2 : 1) * (size_t)(n_wg + 1));
      if (!two_parts) {
        const double c0 = pc[B0], total = pc[B1] - c0;
        for (uint32_t w = 0; w <= n_wg; ++w)
          wb[w] = (uint32_t)(std::lower_bound(pc.begin() + B0, pc.begin() + B1 + 1, c0 + total * w / n_wg - 1e-9) - pc.begin());
        wb[0] = B0; wb[n_wg] = B1;
      } else {
        uint32_t *wa = wb.data(), *wi = wb.data() + n_wg + 1; // part 0: [Bs, B1), part 1: [B0, Bs)
        const double ca = pc[Bs], ta = pc[B1] - ca, ci = pc[B0], ti = pc[Bs] - ci;
        for (uint32_t w = 0; w <= n_wg; ++w) {
          wa[w] = (uint32_t)(std::lower_bound(pc.begin() + Bs, pc.begin() + B1 + 1, ca + ta * w / n_wg - 1e-9) - pc.begin());
          if (w == 0) wa[w] = Bs;
          if (w == n_wg) wa[w] = B1;
          const double want = (ta + ti) * w / n_wg - (pc[wa[w]] - ca); // interior cost the workgroups before w should hold
          wi[w] = (uint32_t)(std::lower_bound(pc.begin() + B0, pc.begin() + Bs + 1, ci + std::max(want, 0.0) - 1e-9) - pc.begin());
          if (w > 0) wi[w] = std::max(wi[w], wi[w - 1]);
        }
        wi[0] = B0; wi[n_wg] = Bs;
      }
      return wb;
    };
    if (itw == dp->wg_blocks->end()) {
      const std::vector<uint32_t> wb = make_ranges();
      uint32_t *dev = nullptr;
      BP5_TRY(upload(&dev, wb.data(), wb.size()));
      itw = dp->wg_blocks->emplace(key, dev).first;
    }
    bp.wg_block = itw->second;
    // face carry: whole-range owner-store launches of a carry build whose plan found faces to carry; the combine pass of THIS launch then
    // takes the tables of this partition (dp->cr_active), every other launch the plan's own
    dp->cr_active = nullptr;
    if constexpr ((ABL & 268435456) != 0) {
      const bool atomic_shared_ = mf->block_shared_atomic || (sub_range && !mf->defer_combine);
      if (mf->tune[BP5_TUNE_FACE_CARRY] && !sub_range && !atomic_shared_ && dp->cr_tile && !mf->combine_csr && !dp->h_carry_len.empty()) {
        if (!dp->cr_carry) dp->cr_carry = new std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, bp5_mf::DevPlan::CombineTables>;
        auto itc = dp->cr_carry->find(key);
        if (itc == dp->cr_carry->end()) {
          bp5_mf::DevPlan::CombineTables ct;
          BP5_TRY(build_carry_tables(mf, dp, make_ranges(), n_wg, two_parts, &ct));
          itc = dp->cr_carry->emplace(key, ct).first;
        }
        if (itc->second.n_shared < dp->n_shared) { // (equal: no face of this partition can be carried)
          dp->cr_active = &itc->second;
          bp.carry = 1u;
        }
      }
    }
    bp.n_parts = two_parts ? 2u : 1u;
    bp.signal = nullptr;
    if (mf->blk_signal) {
      if (!two_parts || !mf->d_signal) return fail(BP5_ERR_INVALID, "boundary-first signal: the launch has no separate ghost-touching part");
      bp.signal = mf->d_signal;
    }
  }
  bp.cg_r = mf->fuse.r; bp.dot_partials = mf->d_partials; bp.n_owned = mf->n_owned; bp.cg_state = mf->d_st;
  bp.stamps = nullptr;
  if (ABL & 4096) {
    if (!mf->d_stamps) HIP_TRY(hipMalloc((void **)&mf->d_stamps, 4096 * 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(mf->d_stamps, 0, 4096 * 16 * sizeof(unsigned long long), mf->stream));
    bp.stamps = mf->d_stamps;
  }
  ApplyArgs a{};
  a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst;
  a.plane_stride = mf->coef_plane_stride; a.cell_stride = (ABL & 1024) ? (uint64_t)mf->n3 : mf->coef_cell_stride; // (affine builds read ONE scalar plane)
  a.cell_begin = 0; a.cell_end = mf->n_cells; a.n_teams = dp->n_groups; a.teams_per_xcd = 0;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  a.hang_mask = mf->d_hang_mask; a.hang_I = mf->d_hang_I;
  if ((ABL & 2097152) && !mf->has_hanging) return fail(BP5_ERR_INVALID, "the hanging-node build needs constraint masks");
  ShapeArg<n> sh;
  fill_shape(sh, mf);
  const bool set = overwrite && dp->covers_all;
  if (overwrite && !set) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
  if (bp.signal && atomic_shared) return fail(BP5_ERR_INVALID, "boundary-first signal: owner-store launches only");
  const dim3 grid(n_wg), block(256);
  snprintf(mf->last_apply_kernel, sizeof(mf->last_apply_kernel), "apply_block_kernel<%d,%s,%d,%d,%d>", P, COLL ? "true" : "false", LPC,
           atomic_shared ? (set ? SC_OWNER_SET_ATOMIC : SC_OWNER_ADD_ATOMIC) : (set ? SC_OWNER_SET : SC_OWNER_ADD), ABL);
  if constexpr ((ABL & 1048576) != 0) { // fused CG dot products: overwrite mode, every DoF touched; the whole range in one launch, or
    // (boundary-first exchange schedule) in block ranges that together cover it, with ONE deferred combine pass
    if (!set || atomic_shared || (sub_range && !mf->defer_combine) || (mf->defer_combine && !mf->cg_split && !mf->cg_late) || !mf->fuse.on)
      return fail(BP5_ERR_INVALID, "fused dot products need overwrite launches that cover the whole range");
    if (mf->fuse.n_cols + n_wg > (uint32_t)PARTIAL_STRIDE / 2) return fail(BP5_ERR_UNSUPPORTED, "too many workgroups for the partial-sum rows");
    bp.dot_col0 = mf->fuse.n_cols;
    mf->fuse.n_cols += n_wg;
  }
  if constexpr ((ABL & 1048576) == 0) if (atomic_shared) {
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
  } else if constexpr ((ABL & 1048576) == 0) {
    auto kern = apply_block_kernel<P, COLL, LPC, SC_OWNER_ADD, ABL>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, mf->stream, a, bp, sh);
  }
  KERNEL_CHECK();
  if (bp.signal) mf->signal_target += n_wg; // the launch was accepted: every workgroup counts itself in once; the caller waits for this value
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
  if (mf->defer_combine) return BP5_OK; // the caller runs launch_combine after its last range
  return launch_combine(mf, dp, dst, set);
}

// overwrite == true: dst need not be zeroed by the caller, the launch defines every entry
template <int P, bool COLL, int TW, int LPC, bool PF, int OPT = 0>
inline int launch_team_t(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1, bool overwrite)
{
  constexpr int n = P + 1;
  constexpr int CPT = 64 * TW / LPC;
  using L = LdsLayout<n, LPC>;
  TeamPlan tp{};
  bp5_mf::DevPlan *dp = nullptr;
  BP5_TRY(get_plan(mf, CPT, tp, &dp));
  ApplyArgs a{};
  a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst;
  a.plane_stride = mf->coef_plane_stride; a.cell_stride = (OPT & 1024) ? (uint64_t)mf->n3 : mf->coef_cell_stride; // (affine builds read ONE scalar plane)
  a.cell_begin = c0; a.cell_end = c1;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  a.n_teams = (c1 + CPT - 1) / CPT - c0 / CPT;
  a.teams_per_xcd = (a.n_teams + 7) / 8;
  ShapeArg<n> sh;
  fill_shape(sh, mf);
  const size_t lds = (size_t)CPT * L::CS * sizeof(double);
  const dim3 grid(a.teams_per_xcd * 8), block(64 * TW);
  const bool whole = (c0 == 0 && c1 == mf->n_cells);
  snprintf(mf->last_apply_kernel, sizeof(mf->last_apply_kernel), "apply_team_kernel<%d,%s,%d,%d,%s,", P, COLL ? "true" : "false", TW, LPC, PF ? "true" : "false");
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
  BP5_CASE(P, V)                                                                                                   \
    return coll ? launch_team_t<P, true, TW, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite)                       \
                : launch_team_t<P, false, TW, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite)

// z-marching kernel (whole cell range only; partial ranges take the plain pencil kernel)
template <int P, bool COLL, int TW, int LPC, bool PF, int ABL = 0>
inline int launch_march_t(bp5_mf *mf, const double *coef, const double *src, double *dst)
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
  a.plane_stride = mf->coef_plane_stride; a.cell_stride = (ABL & 1024) ? (uint64_t)mf->n3 : mf->coef_cell_stride; // (affine builds read ONE scalar plane)
  a.cell_begin = 0; a.cell_end = mf->n_cells; a.n_teams = mp.n_teams; a.teams_per_xcd = mp.teams_per_xcd;
  a.gcell = mf->d_gcell; a.n_cells_total = mf->n_cells;
  ShapeArg<n> sh;
  fill_shape(sh, mf);
  const size_t lds = (size_t)CPT * L::CS * sizeof(double);
  snprintf(mf->last_apply_kernel, sizeof(mf->last_apply_kernel), "apply_march_kernel<%d,%s,%d,%d,%s,%d>", P, COLL ? "true" : "false", TW, LPC, PF ? "true" : "false", ABL);
  hipLaunchKernelGGL((apply_march_kernel<P, COLL, TW, LPC, PF, ABL>), dim3(mp.teams_per_xcd * 8), dim3(64 * TW), lds, mf->stream, a, mp, sh);
  KERNEL_CHECK();
  return BP5_OK;
}

// variant table: (degree, variant) -> (TW, LPC, TPB, PF); variant 0 = default for the degree
#define APPLY_CASE(P, V, TW, LPC, TPB, PF)                                                                         \
  BP5_CASE(P, V) {                                                                                                 \
    if (overwrite && hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream) != hipSuccess)             \
      return fail(BP5_ERR_HIP, "hipMemsetAsync");                                                                  \
    return coll ? launch_apply_t<P, true, TW, LPC, TPB, PF>(mf, coef, src, dst, c0, c1)                            \
                : launch_apply_t<P, false, TW, LPC, TPB, PF>(mf, coef, src, dst, c0, c1);                          \
  }

// overwrite: the launch must leave dst = A src (no prior zeroing by the caller); otherwise dst += A src
template <int P>
inline int launch_affine(bp5_mf *mf, const double *src, double *dst, uint32_t c0, uint32_t c1)
{ // TW = 4 teams when n^2 lanes per cell pack well into 256 threads, as for the 6-plane default
  constexpr int n2 = (P + 1) * (P + 1);
  constexpr int LPC = n2;
  constexpr bool PF = true;
  return mf->quadrature == BP5_QUAD_GLL ? launch_apply_t<P, true, 4, LPC, 1, PF, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1)
                                        : launch_apply_t<P, false, 4, LPC, 1, PF, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1);
}

// (degree, variant) -> kernel.  One instantiation per degree: only the cases of DEG are compiled into it.
#define BP5_CASE(P, V) if constexpr (DEG == (P)) if (variant == (V))
template <int DEG>
int apply_degree_impl(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t c0, uint32_t c1, bool overwrite)
{
  if (mf->operator_kind == BP5_OP_HELMHOLTZ) {
    // step-64's Helmholtz operator (step-64/step-64.cu:154-160,201-219) as a build of the same fused kernels: the degree's default pencil
    // shape (any mesh), or the deterministic block kernel on cell bricks (variant 56; with the CG dot products fused when the solver asks)
    if (mf->has_hanging || mf->geometry_mode == BP5_GEOM_AFFINE) return fail(BP5_ERR_UNSUPPORTED, "the Helmholtz operator needs a conforming mesh and the six-plane geometry");
    const bool coll_ = mf->quadrature == BP5_QUAD_GLL;
    constexpr int HELM = 8388608;
    if (mf->apply_variant == 56) {
      if constexpr (block_lpc(DEG) != 0) {
        constexpr int LPCB = block_lpc(DEG);
        if (c1 <= c0) { if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream)); return BP5_OK; }
        if (!block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) return fail(BP5_ERR_INVALID, "variant 56 needs a cell range aligned with the cell blocks");
        struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; } } reset{mf};
        bp5_mf::DevPlan *dp_ = nullptr;
        BP5_TRY(get_plan_raw(mf, -(256 / LPCB), &dp_));
        if (!dp_->packed) return fail(BP5_ERR_UNSUPPORTED, "variant 56 needs packed indices (<= 128 runs per cell block)");
        if (mf->fuse.on)
          return coll_ ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576 + HELM>(mf, coef, src, dst, overwrite)
                       : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576 + HELM>(mf, coef, src, dst, overwrite);
        return coll_ ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144 + HELM>(mf, coef, src, dst, overwrite)
                     : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + HELM>(mf, coef, src, dst, overwrite);
      }
    }
#ifdef BP5_TIMING_BUILDS
    // timing-only ablations of the Helmholtz block kernel at p = 3 (wrong results; profiles/r4 j_*): 91 no write-out (and no combine pass), 93 no plane loads, 95 no gather
    if constexpr (DEG == 3) if (mf->apply_variant == 91 || mf->apply_variant == 93 || mf->apply_variant == 95) {
      if (!block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) return fail(BP5_ERR_INVALID, "needs aligned cell blocks");
      struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; } } reset{mf};
      if (mf->apply_variant == 91) return launch_block_t<3, false, 16, 2048 + 8192 + 16384 + 262144 + HELM + 1>(mf, coef, src, dst, true);
      if (mf->apply_variant == 93) return launch_block_t<3, false, 16, 2048 + 8192 + 16384 + 262144 + HELM + 2>(mf, coef, src, dst, true);
      return launch_block_t<3, false, 16, 2048 + 8192 + 16384 + 262144 + HELM + 4>(mf, coef, src, dst, true);
    }
#endif
    if (mf->apply_variant != 0) return fail(BP5_ERR_UNSUPPORTED, "the Helmholtz operator runs apply variants 0 (pencil kernel) and 56 (block kernel)");
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    if (c1 <= c0) return BP5_OK;
    constexpr int n2e = (DEG + 1) * (DEG + 1);
    constexpr int TWE = DEG <= 3 ? 1 : 4, TPBE = DEG <= 3 ? 4 : 1;
    return coll_ ? launch_apply_t<DEG, true, TWE, n2e, TPBE, true, HELM>(mf, coef, src, dst, c0, c1)
                 : launch_apply_t<DEG, false, TWE, n2e, TPBE, true, HELM>(mf, coef, src, dst, c0, c1);
  }
  if (mf->has_hanging) {
    // 2:1 refined meshes (resolve_hanging_nodes, bp5/fe_evaluation_gl.h:150-151,167-168): the hanging-node fix-up after the gather and its
    // adjoint before the scatter.  Variant 56: the deterministic block kernel (cell blocks, packed indices; the CG dot products fused when
    // the solver asks); variant 90: the degree's default pencil shape with atomics (any mesh; also the affine geometry mode)
    if (mf->apply_variant == 56 && mf->geometry_mode != BP5_GEOM_AFFINE) {
      if constexpr (block_lpc(DEG) != 0) {
        constexpr int LPCB = block_lpc(DEG);
        constexpr int HANG = 2097152;
        const bool coll_ = mf->quadrature == BP5_QUAD_GLL;
        if (c1 <= c0) { if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream)); return BP5_OK; }
        if (!block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) return fail(BP5_ERR_INVALID, "variant 56 needs a cell range aligned with the cell blocks");
        struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; } } reset{mf};
        bp5_mf::DevPlan *dp_ = nullptr;
        BP5_TRY(get_plan_raw(mf, -(256 / LPCB), &dp_));
        if (!dp_->packed) return fail(BP5_ERR_UNSUPPORTED, "variant 56 needs packed indices (<= 128 runs per cell block)");
        if constexpr (DEG == 4) if (!coll_ && streaming_accesses(mf)) { // streaming policy (non-temporal metric loads on small meshes), as for conforming meshes
          if (mf->fuse.on) return launch_block_t<4, false, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576 + HANG + 32768>(mf, coef, src, dst, overwrite);
          return launch_block_t<4, false, LPCB, 2048 + 8192 + 16384 + 262144 + HANG + 32768>(mf, coef, src, dst, overwrite);
        }
        if (mf->fuse.on)
          return coll_ ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576 + HANG>(mf, coef, src, dst, overwrite)
                       : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576 + HANG>(mf, coef, src, dst, overwrite);
        return coll_ ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144 + HANG>(mf, coef, src, dst, overwrite)
                     : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + HANG>(mf, coef, src, dst, overwrite);
      }
    }
    if (mf->apply_variant != 90) return fail(BP5_ERR_UNSUPPORTED, "meshes with hanging nodes run apply variants 90 (pencil kernel) and 56 (block kernel)");
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    if (c1 <= c0) return BP5_OK;
    constexpr int n2h = (DEG + 1) * (DEG + 1);
    constexpr int TWH = DEG <= 3 ? 1 : 4, TPBH = DEG <= 3 ? 4 : 1;
    if (mf->geometry_mode == BP5_GEOM_AFFINE) // all cells affine (undeformed 2:1 meshes): per-cell K K^T + one scalar plane
      return mf->quadrature == BP5_QUAD_GLL ? launch_apply_t<DEG, true, TWH, n2h, TPBH, true, 2097152 + 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1)
                                            : launch_apply_t<DEG, false, TWH, n2h, TPBH, true, 2097152 + 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1);
    return mf->quadrature == BP5_QUAD_GLL ? launch_apply_t<DEG, true, TWH, n2h, TPBH, true, 2097152>(mf, coef, src, dst, c0, c1)
                                          : launch_apply_t<DEG, false, TWH, n2h, TPBH, true, 2097152>(mf, coef, src, dst, c0, c1);
  }
  if (mf->geometry_mode == BP5_GEOM_AFFINE && c1 > c0) {
    const bool coll_ = mf->quadrature == BP5_QUAD_GLL;
    const bool whole = c0 == 0 && c1 == mf->n_cells;
    if constexpr (DEG == 4) {
    if (mf->apply_variant % 100 == 10) {
      mf->force_atomic_scatter = mf->apply_variant >= 100;
      return coll_ ? launch_team_t<4, true, 4, 25, true, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1, overwrite)
                   : launch_team_t<4, false, 4, 25, true, 1024>(mf, mf->d_scalar_plane, src, dst, c0, c1, overwrite);
    }
    if (mf->apply_variant == 56) { // the default block-kernel shape on the scalar plane + per-cell K K^T
      if (!block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) return fail(BP5_ERR_INVALID, "variant 56 needs a cell range aligned with the cell blocks");
      struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; } } reset{mf};
      return coll_ ? launch_block_t<4, true, 32, 1024 + 2048 + 8192 + 16384 + 262144>(mf, mf->d_scalar_plane, src, dst, overwrite)
                   : launch_block_t<4, false, 32, 1024 + 2048 + 8192 + 16384 + 262144>(mf, mf->d_scalar_plane, src, dst, overwrite);
    }
    if (whole && (mf->apply_variant == 54 || mf->apply_variant == 55)) {
      mf->block_shared_atomic = true;
      const int st_ = mf->apply_variant == 54 ? (coll_ ? launch_block_t<4, true, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                                       : launch_block_t<4, false, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite))
                                              : (coll_ ? launch_block_t<4, true, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                                       : launch_block_t<4, false, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite));
      mf->block_shared_atomic = false;
      return st_;
    }
    if (whole && (mf->apply_variant == 50 || mf->apply_variant == 51))
      return mf->apply_variant == 50 ? (coll_ ? launch_block_t<4, true, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                              : launch_block_t<4, false, 25, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite))
                                     : (coll_ ? launch_block_t<4, true, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite)
                                              : launch_block_t<4, false, 32, 1024>(mf, mf->d_scalar_plane, src, dst, overwrite));
#ifdef BP5_TIMING_BUILDS
    if (mf->apply_variant == 85) // timing only: affine, no scatter atomics -> compute/latency floor of the pencil kernel
      return launch_apply_t<4, false, 4, 25, 1, true, 1025>(mf, mf->d_scalar_plane, src, dst, c0, c1);
#endif
    }
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    return launch_affine<DEG>(mf, src, dst, c0, c1);
  }
  if (c1 <= c0) {
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    return BP5_OK;
  }
  const bool coll = mf->quadrature == BP5_QUAD_GLL;
  // variants >= 100: the team kernel of (variant - 100) with the global-atomic scatter (A/B tests)
  mf->force_atomic_scatter = mf->apply_variant >= 100;
  int variant = mf->apply_variant % 100;
  // cell-interior DoFs numbered ahead of all others (recognised by bp5_mf_create): the default pencil kernel of p >= 5 stores the entries a cell owns alone
  // plainly -- (p-1)^3 of (p+1)^3 atomics less per cell (47 % at p = 8), and no store ever meets an atomic in one cache line
  if constexpr (DEG >= 5) if (variant == 0 && !mf->force_atomic_scatter && mf->cell_interiors_first && mf->tune[BP5_TUNE_INTERIOR_STORES]) {
    if (overwrite) HIP_TRY(hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream));
    return coll ? launch_apply_t<DEG, true, 4, (DEG + 1) * (DEG + 1), 1, true, 32>(mf, coef, src, dst, c0, c1)
                : launch_apply_t<DEG, false, 4, (DEG + 1) * (DEG + 1), 1, true, 32>(mf, coef, src, dst, c0, c1);
  }
  {
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
#ifdef BP5_TIMING_BUILDS
    // timing-only ablations of variant 3 (results are wrong by construction): 20 + ABL mask
#define ABL_CASE(M) BP5_CASE(4, 20 + (M)) return launch_apply_t<4, false, 4, 25, 1, true, M>(mf, coef, src, dst, c0, c1)
#define ABL_CASE_HI(P, L, M) BP5_CASE(P, 20 + (M)) return launch_apply_t<P, false, 4, L, 1, true, M>(mf, coef, src, dst, c0, c1)
    ABL_CASE_HI(8, 81, 1); ABL_CASE_HI(8, 81, 2); ABL_CASE_HI(8, 81, 4); ABL_CASE_HI(8, 81, 8); ABL_CASE_HI(8, 81, 9); ABL_CASE_HI(8, 81, 11);
    ABL_CASE_HI(6, 49, 1); ABL_CASE_HI(6, 49, 2); ABL_CASE_HI(6, 49, 8); ABL_CASE_HI(6, 49, 9);
    BP5_CASE(4, 7) return coll ? launch_apply_t<4, true, 4, 25, 1, true, 256>(mf, coef, src, dst, c0, c1) : launch_apply_t<4, false, 4, 25, 1, true, 256>(mf, coef, src, dst, c0, c1);
    BP5_CASE(4, 8) return coll ? launch_apply_t<4, true, 4, 25, 1, true, 512>(mf, coef, src, dst, c0, c1) : launch_apply_t<4, false, 4, 25, 1, true, 512>(mf, coef, src, dst, c0, c1);
    BP5_CASE(4, 9) return coll ? launch_apply_t<4, true, 2, 25, 1, true, 512>(mf, coef, src, dst, c0, c1) : launch_apply_t<4, false, 2, 25, 1, true, 512>(mf, coef, src, dst, c0, c1);
    BP5_CASE(4, 82) return launch_apply_t<4, false, 4, 25, 1, true, 257>(mf, coef, src, dst, c0, c1);
    BP5_CASE(4, 83) return launch_apply_t<4, false, 4, 25, 1, true, 4096>(mf, coef, src, dst, c0, c1);
    BP5_CASE(4, 84) return launch_apply_t<4, false, 4, 25, 1, true, 8192>(mf, coef, src, dst, c0, c1);
    BP5_CASE(4, 80) return launch_apply_t<4, false, 4, 25, 1, true, 64>(mf, coef, src, dst, c0, c1);
    BP5_CASE(4, 81) { // E-vector stores need a big scratch target
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 128>(mf, coef, src, mf->d_evec, c0, c1); }
    BP5_CASE(4, 90) {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, ((size_t)mf->n_cells * mf->n3 + 4096 * 5) * sizeof(double) * 2));
      return launch_apply_t<4, false, 4, 25, 1, true, 262144>(mf, coef, src, mf->d_evec, c0, c1); }
    BP5_CASE(4, 88) {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 128 + 65536>(mf, coef, src, mf->d_evec, c0, c1); }
    BP5_CASE(4, 89) {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 1 + 131072>(mf, coef, src, mf->d_evec, c0, c1); }
    BP5_CASE(4, 86) {
      if (!mf->d_evec) HIP_TRY(hipMalloc((void **)&mf->d_evec, (size_t)mf->n_cells * mf->n3 * sizeof(double)));
      return launch_apply_t<4, false, 4, 25, 1, true, 128 + 16384>(mf, coef, src, mf->d_evec, c0, c1); }
    ABL_CASE(1); ABL_CASE(2); ABL_CASE(3); ABL_CASE(4); ABL_CASE(5); ABL_CASE(7); ABL_CASE(8); ABL_CASE(9); ABL_CASE(15); ABL_CASE(14); ABL_CASE(13); ABL_CASE(11);
#endif
    // z-marching kernel, variants 70+ (atomic scatter: dst must be zero-filled like for the pencil kernel)
#define MARCH_CASE(P, V, TW, LPC, PF)                                                                              \
  BP5_CASE(P, V) {                                                                                                 \
    if (overwrite && hipMemsetAsync(dst, 0, mf->n_local() * sizeof(double), mf->stream) != hipSuccess)             \
      return fail(BP5_ERR_HIP, "hipMemsetAsync");                                                                  \
    if (c0 != 0 || c1 != mf->n_cells)                                                                              \
      return coll ? launch_apply_t<P, true, TW, LPC, 1, PF>(mf, coef, src, dst, c0, c1)                            \
                  : launch_apply_t<P, false, TW, LPC, 1, PF>(mf, coef, src, dst, c0, c1);                          \
    return coll ? launch_march_t<P, true, TW, LPC, PF>(mf, coef, src, dst) : launch_march_t<P, false, TW, LPC, PF>(mf, coef, src, dst); \
  }
#ifdef BP5_TIMING_BUILDS
    BP5_CASE(4, 73) return launch_march_t<4, false, 4, 25, true, 1>(mf, coef, src, dst); // timing only: march, no scatter
#endif
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
  BP5_CASE(P, V) {                                                                                                 \
    if (c0 != 0 || c1 != mf->n_cells)                                                                              \
      return coll ? launch_team_t<P, true, TW_FALLBACK, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite)            \
                  : launch_team_t<P, false, TW_FALLBACK, LPC, PF>(mf, coef, src, dst, c0, c1, overwrite);          \
    return coll ? launch_block_t<P, true, LPC>(mf, coef, src, dst, overwrite) : launch_block_t<P, false, LPC>(mf, coef, src, dst, overwrite); \
  }
    BLOCK_CASE(1, 50, 4, 4, true);
    BLOCK_CASE(2, 50, 9, 4, true);
    BLOCK_CASE(3, 50, 16, 4, true);
    BLOCK_CASE(4, 50, 25, 4, true);
    BLOCK_CASE(4, 51, 32, 4, true);
    if constexpr (DEG == 4) if (variant == 54 || variant == 55) {
      if (c0 == 0 && c1 == mf->n_cells) {
        mf->block_shared_atomic = true;
        const int st_ = variant == 54 ? (coll ? launch_block_t<4, true, 32>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32>(mf, coef, src, dst, overwrite))
                                             : (coll ? launch_block_t<4, true, 25>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 25>(mf, coef, src, dst, overwrite));
        mf->block_shared_atomic = false;
        return st_;
      }
      return coll ? launch_team_t<4, true, 4, 25, true>(mf, coef, src, dst, c0, c1, overwrite) : launch_team_t<4, false, 4, 25, true>(mf, coef, src, dst, c0, c1, overwrite);
    }
    // 56 on the other degrees with a wave-local cell shape (p = 2, 3, 5, 6, 7): sequential tiles, run-length write-out, packed
    // indices; p >= 5 keep two workgroups per CU (registers), fused CG dot products when the solver asks for them
#ifdef BP5_TIMING_BUILDS
    if constexpr (DEG == 6 || DEG == 8 || DEG == 5) if (variant == 99 || variant == 91 || variant == 93) { // cycle stamps / no write-out / no metric loads
      constexpr int LPCB = block_lpc(DEG);
      if (!block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) return fail(BP5_ERR_INVALID, "needs aligned cell blocks");
      struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; } } reset{mf};
      if (variant == 99) return launch_block_t<DEG, false, LPCB, 4096 + 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, true);
      if (variant == 91) return launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + 1>(mf, coef, src, dst, true);
      return launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + 2>(mf, coef, src, dst, true);
    }
#endif
    if constexpr (DEG != 4 && block_lpc(DEG) != 0) if (variant == 56) {
      constexpr int LPCB = block_lpc(DEG);
      if (!block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) return fail(BP5_ERR_INVALID, "variant 56 needs a cell range aligned with the cell blocks");
      struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; } } reset{mf};
      bp5_mf::DevPlan *dp_ = nullptr;
      BP5_TRY(get_plan_raw(mf, -(256 / LPCB), &dp_));
      if (!dp_->packed) return fail(BP5_ERR_UNSUPPORTED, "variant 56 needs packed indices (<= 128 runs per cell block) at this degree");
      constexpr int LATT = 16777216; // every block a lattice block: closed-form indices, no per-DoF index stream
      const bool lattice = dp_->lattice && dp_->n_lattice_blocks == dp_->n_groups;
      if (mf->fuse.on && lattice)
        return coll ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576 + LATT>(mf, coef, src, dst, overwrite)
                    : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576 + LATT>(mf, coef, src, dst, overwrite);
      if (mf->fuse.on)
        return coll ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576>(mf, coef, src, dst, overwrite)
                    : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + 1048576>(mf, coef, src, dst, overwrite);
      if (lattice)
        return coll ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144 + LATT>(mf, coef, src, dst, overwrite)
                    : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144 + LATT>(mf, coef, src, dst, overwrite);
      return coll ? launch_block_t<DEG, true, LPCB, 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, overwrite)
                  : launch_block_t<DEG, false, LPCB, 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, overwrite);
    }
    // 48 = 56 with the per-DoF CSR combine kernel instead of the run-length one (A/B)
    // 49 = 56 with run-length write-out but without packed indices (A/B)
    // 60 = 56 with the brick's src staged once in LDS (cells gather from LDS; needs the packed indices)
    // 61 = 56 with non-temporal metric loads (A/B: the once-read metric stream then evicts less of a brick's src from L2)
    // 62 = 56 with ds_add_f64 for the accumulation into the LDS vector (A/B)
    // 63 = 56 with the rolling metric prefetch (BlockPass::ROLL; lattice blocks only) -- libbp5_timing.so only: a measured loss (profiles/r4 d_*)
    if constexpr (DEG == 4) if (variant == 48 || variant == 49 || variant == 56 || variant == 60 || variant == 61 || variant == 62 || variant == 63) { if (block_aligned(mf, c0, c1, &mf->blk_b0, &mf->blk_b1)) {
        struct Reset { bp5_mf *m; ~Reset() { m->blk_b0 = m->blk_b1 = 0; m->combine_csr = false; } } reset{mf};
        mf->combine_csr = variant == 48;
        bp5_mf::DevPlan *dp_ = nullptr;
        BP5_TRY(get_plan_raw(mf, -8, &dp_)); // p = 4: 32 lanes per cell, 8 cells per pass
        if (variant == 60) {
          if (!dp_->packed) return fail(BP5_ERR_UNSUPPORTED, "variant 60 needs packed indices (<= 128 runs per cell block)");
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144 + 524288>(mf, coef, src, dst, overwrite)
                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 524288>(mf, coef, src, dst, overwrite);
        }
        if (variant == 62) {
          if (!dp_->packed) return fail(BP5_ERR_UNSUPPORTED, "variant 62 needs packed indices (<= 128 runs per cell block)");
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144 + 4194304>(mf, coef, src, dst, overwrite)
                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 4194304>(mf, coef, src, dst, overwrite);
        }
        if (variant == 61) {
          if (!dp_->packed) return fail(BP5_ERR_UNSUPPORTED, "variant 61 needs packed indices (<= 128 runs per cell block)");
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144 + 32768>(mf, coef, src, dst, overwrite)
                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 32768>(mf, coef, src, dst, overwrite);
        }
        constexpr int LATT = 16777216; // every block a lattice block: closed-form indices, no per-DoF index stream
        constexpr int LATC = LATT + 268435456; // ... with the face carry compiled in (BP5_TUNE_FACE_CARRY switches it per launch)
        const bool lattice = (variant == 56 || variant == 63) && dp_->packed && dp_->lattice && dp_->n_lattice_blocks == dp_->n_groups;
        const bool ntm = streaming_accesses(mf); // non-temporal metric loads
#ifndef BP5_TIMING_BUILDS
        if (variant == 63) return fail(BP5_ERR_INVALID, "variant 63 lives in libbp5_timing.so");
#else
        constexpr int ROLL = 67108864;
        if (variant == 63) {
          if (!lattice || coll) return fail(BP5_ERR_UNSUPPORTED, "variant 63 (rolling metric prefetch) needs lattice blocks and Gauss quadrature");
          if (mf->fuse.on) return ntm ? launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 1048576 + LATT + 32768 + ROLL>(mf, coef, src, dst, overwrite)
                                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 1048576 + LATT + ROLL>(mf, coef, src, dst, overwrite);
          return ntm ? launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + LATT + 32768 + ROLL>(mf, coef, src, dst, overwrite)
                     : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + LATT + ROLL>(mf, coef, src, dst, overwrite);
        }
#endif
        if (mf->fuse.on) { // the solver asked for the fused dot products (only ever with the packed default shape)
          if (!dp_->packed || variant != 56) return fail(BP5_ERR_INVALID, "fused dot products need the packed block kernel");
          if (lattice && ntm && !coll) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 1048576 + LATC + 32768>(mf, coef, src, dst, overwrite);
          if (lattice)
            return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144 + 1048576 + LATC>(mf, coef, src, dst, overwrite)
                        : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 1048576 + LATC>(mf, coef, src, dst, overwrite);
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144 + 1048576>(mf, coef, src, dst, overwrite)
                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 1048576>(mf, coef, src, dst, overwrite);
        }
        if (lattice && ntm && !coll) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + LATC + 32768>(mf, coef, src, dst, overwrite);
        if (lattice)
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144 + LATC>(mf, coef, src, dst, overwrite)
                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + LATC>(mf, coef, src, dst, overwrite);
        if (dp_->packed && variant != 49) // few long runs (block-major numbering): one packed u16 per cell-local DoF, no local_to_global stream
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, overwrite)
                      : launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, overwrite);
        if (dp_->max_runs <= (uint32_t)BLOCK_MAX_RUNS) // write-out without list loads
          return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 16384>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192 + 16384>(mf, coef, src, dst, overwrite);
        return coll ? launch_block_t<4, true, 32, 2048 + 8192>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192>(mf, coef, src, dst, overwrite);
      }
      return fail(BP5_ERR_INVALID, "variant 56 needs a cell range aligned with the cell blocks");
    }
    BP5_CASE(4, 59) { if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 2048 + 8192>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 59 needs the whole cell range"); }
    BP5_CASE(4, 57) { if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 8192>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 8192>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 57 needs the whole cell range"); }
    BP5_CASE(4, 58) { if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 2048 + 8192 + 32768>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048 + 8192 + 32768>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 58 needs the whole cell range"); }
#ifdef BP5_TIMING_BUILDS
    BP5_CASE(4, 99) return launch_block_t<4, false, 32, 4096 + 2048 + 8192 + 16384 + 262144>(mf, coef, src, dst, true);   // stamps of the default shape (sequential tiles, 3 WG/CU, run write-out, packed indices)
#endif
    BP5_CASE(4, 52) { if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 32, 2048>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 32, 2048>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 52 needs the whole cell range"); }
    BP5_CASE(4, 53) { if (c0 == 0 && c1 == mf->n_cells) return coll ? launch_block_t<4, true, 25, 2048>(mf, coef, src, dst, overwrite) : launch_block_t<4, false, 25, 2048>(mf, coef, src, dst, overwrite);
      return fail(BP5_ERR_INVALID, "variant 53 needs the whole cell range"); }
#ifdef BP5_TIMING_BUILDS
    BP5_CASE(4, 64) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 16777216>(mf, coef, src, dst, true);              // the lattice build (reference point of the two probes below)
    BP5_CASE(4, 65) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 16777216 + 33554432>(mf, coef, src, dst, true);   // ... metric as whole aligned lines, 12 instructions
    BP5_CASE(4, 66) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 262144 + 16777216 + 134217728>(mf, coef, src, dst, true);  // ... tails paired, 15 instructions
    BP5_CASE(4, 87) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 65536>(mf, coef, src, dst, true);  // variant 56 with plain (not non-temporal) stores
    BP5_CASE(4, 91) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 1>(mf, coef, src, dst, true);  // variant 56 without write-out (and combine)
    BP5_CASE(4, 93) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 2>(mf, coef, src, dst, true);  // ... without metric loads
    BP5_CASE(4, 95) return launch_block_t<4, false, 32, 2048 + 8192 + 16384 + 4>(mf, coef, src, dst, true);  // ... without gather
    BP5_CASE(4, 97) return launch_block_t<4, false, 32, 4096>(mf, coef, src, dst, true);          // stamps, double-buffered
    BP5_CASE(4, 98) return launch_block_t<4, false, 32, 4096 + 2048>(mf, coef, src, dst, true);   // stamps, single-buffered
    BP5_CASE(4, 92) return launch_block_t<4, false, 32, 2049>(mf, coef, src, dst, true);
    BP5_CASE(4, 96) return launch_block_t<4, false, 32, 2053>(mf, coef, src, dst, true);
#endif
    BLOCK_CASE(5, 50, 36, 4, true);
    BLOCK_CASE(6, 50, 49, 4, false);
    BLOCK_CASE(7, 50, 64, 4, false);
    BLOCK_CASE(8, 50, 81, 4, false);
#ifdef BP5_TIMING_BUILDS
#define BABL_CASE(M) BP5_CASE(4, 60 + (M)) return launch_block_t<4, false, 25, M>(mf, coef, src, dst, true)
    BABL_CASE(16); BABL_CASE(1); BABL_CASE(2); BABL_CASE(3); BABL_CASE(4); BABL_CASE(5); BABL_CASE(7); BABL_CASE(8); BABL_CASE(9); BABL_CASE(15);
    // timing-only ablations of the team kernel (SET mode): 40 + mask (1: no scatter stage, 4: no gather stage)
#define TABL_CASE(M)                                                                                               \
  BP5_CASE(4, 40 + (M)) {                                                                                         \
    TeamPlan tp; bp5_mf::DevPlan *dp = nullptr;                                                                    \
    BP5_TRY(get_plan(mf, 10, tp, &dp));                                                                            \
    ApplyArgs a; a.l2g = mf->d_l2g; a.coef = coef; a.src = src; a.dst = dst; a.plane_stride = mf->coef_plane_stride; a.cell_stride = mf->coef_cell_stride; \
    a.cell_begin = c0; a.cell_end = c1; a.n_teams = (c1 + 9) / 10 - c0 / 10; a.teams_per_xcd = (a.n_teams + 7) / 8;  \
    ShapeArg<5> sh; fill_shape(sh, mf);                  \
    hipLaunchKernelGGL((apply_team_kernel<4, false, 4, 25, true, SC_OWNER_SET, M>), dim3(a.teams_per_xcd * 8), dim3(256), \
                       (10 * LdsLayout<5, 25>::CS * sizeof(double)), mf->stream, a, tp, sh);                          \
    KERNEL_CHECK(); return BP5_OK; }
    TABL_CASE(0); TABL_CASE(1); TABL_CASE(4); TABL_CASE(5);
#endif
    // team-assembled kernel (LDS-staged gather + scatter), variants 10+
    TEAM_CASE(1, 10, 4, 4, true);
    TEAM_CASE(2, 10, 4, 9, true);
    TEAM_CASE(3, 10, 4, 16, true);
    TEAM_CASE(4, 10, 4, 25, true);
    TEAM_CASE(4, 11, 8, 25, true);
    TEAM_CASE(4, 12, 4, 25, false);
    TEAM_CASE(4, 13, 2, 25, true);
    BP5_CASE(4, 14) return coll ? launch_team_t<4, true, 4, 25, true, 32>(mf, coef, src, dst, c0, c1, overwrite)
                          : launch_team_t<4, false, 4, 25, true, 32>(mf, coef, src, dst, c0, c1, overwrite);
    TEAM_CASE(5, 10, 4, 36, true);
    TEAM_CASE(6, 10, 4, 49, false);
    TEAM_CASE(7, 10, 4, 64, false);
    TEAM_CASE(8, 10, 4, 81, false);
  }
  return fail(BP5_ERR_INVALID, "unknown (degree, apply variant)");
}

#define BP5_EXTERN_DEGREE(N) extern template int apply_degree_impl<N>(bp5_mf *, const double *, const double *, double *, uint32_t, uint32_t, bool);
BP5_EXTERN_DEGREE(1) BP5_EXTERN_DEGREE(2) BP5_EXTERN_DEGREE(3) BP5_EXTERN_DEGREE(4) BP5_EXTERN_DEGREE(5) BP5_EXTERN_DEGREE(6) BP5_EXTERN_DEGREE(7) BP5_EXTERN_DEGREE(8)
