/* bp5.h -- C ABI of the MI355X-native BP5 matrix-free operator + CG hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one piece of the reference's
 * (peterrum/deal-and-ceed-on-gpu) hot path, cited as file:line under the reference root.  The
 * reference has no FFI of its own (it is C++ calling deal.II templates); a maintainer binds
 * these symbols from the C++ host code as shown in INTEGRATION.md, or uses the header-only
 * facade include/bp5_dealii_facade.hpp which re-creates the deal.II class names on top of them.
 *
 * Conventions
 *   - every function returns an int status (BP5_OK == 0); no exception crosses the boundary
 *     (the reference throws: AssertThrow/AssertCuda, bp5/solver.h:396-397,539-540);
 *   - pointers are DEVICE pointers unless the parameter name ends in `_host`;
 *   - handles are opaque, one per device; calls on one handle are not thread-safe;
 *   - all device work is enqueued on the handle's stream (bp5_mf_set_stream) and is
 *     asynchronous unless stated otherwise; nothing allocates inside the hot loop; the only other
 *     stream the library uses is the handle's own communication stream (RCCL halo traffic that
 *     overlaps compute), ordered against the handle's stream by events in both directions;
 *   - FP64 arithmetic, 32-bit DoF indices (types::global_dof_index, bp5/fe_evaluation_gl.h:84);
 *   - vectors are plain arrays of n_owned + n_ghost doubles, owned range first
 *     (LinearAlgebra::distributed::Vector<double, MemorySpace::CUDA>, bp5/step-64.cu:321-323).
 */
#ifndef BP5_H
#define BP5_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* status codes                                                                                */
enum {
  BP5_OK = 0,
  BP5_ERR_INVALID = 1,        /* bad argument / unsupported degree                             */
  BP5_ERR_HIP = 2,            /* a HIP runtime call failed (AssertCuda, bp5/solver.h:396-397)  */
  BP5_ERR_NO_DEVICE = 3,      /* no gfx950 device visible: there is NO CPU fallback            */
  BP5_ERR_RCCL = 4,           /* an RCCL call failed                                           */
  BP5_ERR_UNSUPPORTED = 5,    /* e.g. a cell block that does not fit in LDS for a forced variant */
  BP5_ERR_BREAKDOWN = 6,      /* CG breakdown: p.Ap == 0 or NaN (ExcDivideByZero, solver.h:501)*/
  BP5_ERR_NO_CONVERGENCE = 7  /* SolverControl::NoConvergence, bp5/solver.h:539-540            */
};
const char *bp5_strerror(int status);
const char *bp5_last_error(void); /* detail of the last failure on this thread */

enum { BP5_QUAD_GAUSS = 0, /* QGauss<1>(p+1), bp5/step-64.cu:246 (reference default) */
       BP5_QUAD_GLL = 1    /* QGaussLobatto<1>(p+1), bp5/step-64.cu:244 (COLLOCATION) */ };

enum { BP5_COEF_ONE = 0,   /* kappa == 1: the reference folds only JxW, bp5/step-64.cu:107-113 */
       BP5_COEF_STEP64 = 1 /* kappa = 10/(0.05+2|x|^2), step-64/step-64.cu:117 */ };

#define BP5_MAX_DEGREE 8

/* ------------------------------------------------------------------------------------------ */
/* host-only helpers (run without a GPU)                                                       */

/* 1-D data of FE_Q(degree) on GLL nodes with the chosen quadrature, all on [0,1]:
 * nodes[n], pts[n], w[n], N[q*n+i] = phi_i(x_q), D[q*n+i] = phi_i'(x_q), n = degree+1.
 * Replaces the shape_values/shape_gradients tables MatrixFree::reinit builds
 * (call site bp5/step-64.cu:243-248). */
int bp5_shape_tables(int degree, int quadrature, double *nodes_host, double *pts_host, double *w_host,
                     double *N_host, double *D_host);

/* Structured hex-mesh generator standing in for GridGenerator::subdivided_hyper_rectangle +
 * refine_global + DoFHandler::distribute_dofs + zero Dirichlet constraints
 * (bp5/step-64.cu:341-358, 629-663); slab partition along z replaces the p4est partition. */
typedef struct bp5_mesh bp5_mesh;
typedef struct {
  int degree;          /* p */
  uint32_t cells[3];   /* global number of cells per direction */
  double h;            /* cell side (reference: 2^-refine) */
  double deform_amp;   /* 0 = cubes; >0 = smooth boundary-preserving sine displacement */
  int rank, n_ranks;   /* z-slab decomposition; interface planes owned by the lower rank */
  uint32_t cell_block[3]; /* cells are emitted block by block (bx x by x bz cells, x fastest inside a
                             block) so that consecutive cells form compact bricks; 0 = plain
                             lexicographic order */
  int dof_numbering;   /* 0: lexicographic I + NX (J + NY K) over the owned range;
                          1: block-major (needs cell_block): the DoFs strictly inside a cell block are
                             numbered contiguously, then block faces, edges, vertices -- every entity
                             contiguous.  A brick's gather/scatter then touches a few long runs instead
                             of many short rows.  global_ids_host always gives the lexicographic id.
                          2: the owned DoFs strictly INSIDE a cell first, cell after cell in the order the cells are handed over
                             ((p-1)^3 consecutive DoFs per cell, x fastest), then all other owned DoFs as in 0 (as in 1 where
                             cell_block is given).  bp5_mf_create recognises this property of a mesh (whoever numbered it) and the
                             pencil kernel of p >= 5 then stores cell-interior entries plainly instead of adding them atomically:
                             they share no cache line with an atomically updated DoF (half the atomics at p = 8). */
  int cell_block_order; /* order of the cells inside a block: 0 lexicographic (x fastest); 1 parity class by
                             parity class ((x&1, y&1, z&1) relative to the block corner, lexicographic inside a
                             class): consecutive cells then share no DoF, which is the order the block-assembled
                             kernel walks them in (its passes read consecutive cells) and spreads the atomics of
                             the pencil kernel */
} bp5_mesh_desc;

typedef struct {
  int degree;
  uint32_t n_cells;           /* locally owned cells                                            */
  uint32_t n_interior_cells;  /* cells [0,n_interior_cells) touch no ghost DoF                  */
  uint32_t n_owned, n_ghost;  /* local DoFs: owned first, then ghosts                           */
  uint64_t n_global_dofs;
  uint32_t global_dofs_per_dir[3];
  const uint32_t *local_to_global_host; /* [n_cells*(p+1)^3], local indices, i + n(j + n k)     */
  const double *node_coords_host;       /* [n_owned+n_ghost][3]                                 */
  const uint64_t *global_ids_host;      /* [n_owned+n_ghost] lexicographic global DoF id        */
  const uint32_t *constrained_host;     /* local indices of Dirichlet DoFs (owned + ghost)      */
  uint32_t n_constrained;
  /* halo plan (Utilities::MPI::Partitioner): per neighbour, owned indices to send for the
   * ghost gather, and the contiguous ghost range received from it */
  int n_neighbors;
  const int *neighbor_rank_host;        /* [n_neighbors]                                        */
  const uint32_t *send_offsets_host;    /* [n_neighbors+1] into send_indices                    */
  const uint32_t *send_indices_host;    /* owned local indices                                  */
  const uint32_t *recv_offsets_host;    /* [n_neighbors+1] offsets into the ghost range         */
  uint32_t n_cell_blocks;               /* cell blocks: block b = cells [off[b], off[b+1])       */
  const uint32_t *cell_block_offsets_host; /* [n_cell_blocks+1]                                  */
} bp5_mesh_view;

int bp5_mesh_create_brick(const bp5_mesh_desc *desc, bp5_mesh **out);
int bp5_mesh_view_get(const bp5_mesh *mesh, bp5_mesh_view *out);
void bp5_mesh_destroy(bp5_mesh *mesh);

/* ------------------------------------------------------------------------------------------ */
/* device + vectors                                                                            */
int bp5_device_count(int *count);
int bp5_vec_alloc(size_t n, double **out);            /* hipMalloc, zero-filled                 */
int bp5_vec_free(double *v);
int bp5_copy_h2d(void *dst, const void *src_host, size_t bytes);   /* synchronous               */
int bp5_copy_d2h(void *dst_host, const void *src, size_t bytes);   /* synchronous               */

/* ------------------------------------------------------------------------------------------ */
/* matrix-free engine: CUDAWrappers::MatrixFree<3,double> as the reference uses it             */
typedef struct bp5_mf bp5_mf;

typedef struct {
  int dim;                 /* must be 3                                                         */
  int degree;              /* 1..BP5_MAX_DEGREE                                                 */
  int quadrature;          /* BP5_QUAD_*                                                        */
  int coefficient;         /* BP5_COEF_*                                                        */
  uint32_t n_cells, n_interior_cells, n_owned, n_ghost;
  const uint32_t *local_to_global_host;
  const double *node_coords_host;       /* MappingQGeneric(degree) through the FE_Q nodes,
                                           bp5/step-64.cu:234                                   */
  const uint32_t *constrained_host;
  uint32_t n_constrained;
  int n_neighbors;
  const int *neighbor_rank_host;
  const uint32_t *send_offsets_host, *send_indices_host, *recv_offsets_host;
  int device;              /* HIP device ordinal                                                */
  void *stream;            /* hipStream_t; NULL = the HIP default (null) stream                 */
  /* optional: groups of consecutive cells that one workgroup assembles in LDS (compact bricks give
   * the fewest DoFs shared between groups); NULL = the library groups 64 consecutive cells      */
  uint32_t n_cell_blocks;
  const uint32_t *cell_block_offsets_host; /* [n_cell_blocks+1], first 0, last n_cells           */
  /* optional: hanging-node constraints of 2:1 refined meshes, one mask per cell (BP5_HANG_* below); NULL = conforming mesh.
   * == MatrixFree::Data::constraint_mask as consumed by resolve_hanging_nodes, bp5/fe_evaluation_gl.h:150-151,167-168 */
  const uint32_t *constraint_mask_host;   /* [n_cells]                                          */
} bp5_mf_desc;

/* constraint_mask bits of a fine cell that touches coarser cells ("hanging" faces / edges of 2:1 refined meshes):
 *   BP5_HANG_FACE_d   the face normal to direction d lies on a coarser neighbour (any subset: cells on the rim of a refined
 *                     region have one, at its edges two, at its corners three);
 *   BP5_HANG_EDGE_d   the edge along d lies on a coarser cell's edge although neither face through it is constrained
 *                     (re-entrant corners of the refined region);
 *   BP5_HANG_SIDE_e   position of the cell in its parent along e (0 / 1): constrained faces normal to e sit at xi_e = that side,
 *                     a constrained edge along d at the corner (SIDE_e1, SIDE_e2) of the two other directions;
 *   BP5_HANG_HALF_d   the same position, as selector of the interpolation ALONG d: the cell covers the upper half [1/2, 1] of
 *                     the coarse face / edge (else the lower half).  (A cell with one constrained face may name only the SIDE
 *                     of the normal and the HALF of the two tangential directions; where both are used they must agree.)
 * local_to_global of the (p+1)^2 entries ON a constrained face names the COARSE face's DoFs in the same orientation, the p+1
 * entries on a constrained edge the coarse edge's.  Gathers (read_dof_values, and the node coordinates of the geometry) apply,
 * per direction d, the 1-D matrix I_h[a][b] = phi_b(xi_a / 2 + h / 2), h = HALF_d, to every cell-local line along d that lies on
 * a constrained face tangential to d or on the constrained edge along d; scatters (distribute_local_to_global, RHS
 * assembly) apply the adjoint.  (deal.II's own bit layout is not part of the reference repository; this one is the library's.) */
enum { BP5_HANG_FACE_X = 1, BP5_HANG_FACE_Y = 2, BP5_HANG_FACE_Z = 4,
       BP5_HANG_SIDE_X = 8, BP5_HANG_SIDE_Y = 16, BP5_HANG_SIDE_Z = 32,
       BP5_HANG_HALF_X = 64, BP5_HANG_HALF_Y = 128, BP5_HANG_HALF_Z = 256,
       BP5_HANG_EDGE_X = 512, BP5_HANG_EDGE_Y = 1024, BP5_HANG_EDGE_Z = 2048 };

/* == MatrixFree::reinit(mapping, dof_handler, constraints, quad, additional_data),
 *    bp5/step-64.cu:234-248.  Uploads the flat arrays; computes nothing yet. */
int bp5_mf_create(const bp5_mf_desc *desc, bp5_mf **out);
int bp5_mf_destroy(bp5_mf *mf);
int bp5_mf_set_stream(bp5_mf *mf, void *hip_stream);
int bp5_mf_sync(bp5_mf *mf); /* hipStreamSynchronize */

/* Which operator the handle applies (bp5_apply*, bp5_cg_solve, all exchange schedules, fused dot products):
 *   BP5_OP_POISSON    (default) LocalPoissonOperator, bp5/step-64.cu:147-194:  (grad v, kappa grad u), six merged planes;
 *   BP5_OP_HELMHOLTZ  step-64's LocalHelmholtzOperator + HelmholtzOperatorQuad (step-64/step-64.cu:154-160,201-219):
 *                     (grad v, grad u) + (v, a(x) u) -- evaluate(true, true), submit_value(a * get_value()) +
 *                     submit_gradient(get_gradient()), integrate(true, true) -- as a native fused kernel: the value path costs one
 *                     more 1-D contraction each way and one more plane.  The metric array then holds SEVEN planes: the six merged
 *                     planes JxW K K^T (coefficient 1) and the mass plane a(x_q) JxW, a = the handle's BP5_COEF_* function
 *                     (VaryingCoefficientFunctor, step-64/step-64.cu:99-118, with JxW folded in): G = 7 doubles per q-point.
 *                     Conforming meshes, BP5_GEOM_MERGED6; apply variants 0 (pencil kernel) and 56 (block kernel on cell bricks).
 * Set before bp5_mf_coef_size / bp5_mf_compute_merged_metric. */
enum { BP5_OP_POISSON = 0, BP5_OP_HELMHOLTZ = 1 };
int bp5_mf_set_operator(bp5_mf *mf, int op);
/* number of doubles of the merged-metric array: 6 * n_cells * (p+1)^3  (bp5/step-64.cu:253-254); 7 planes for BP5_OP_HELMHOLTZ */
int bp5_mf_coef_size(const bp5_mf *mf, size_t *n_doubles);

/* == mf_data.evaluate_coefficients(JacobianFunctor), bp5/step-64.cu:84-114,256-258:
 *    coef_c = kappa * JxW * (K K^T)_c, c in {00,11,22,01,02,12}.
 *    Device layout (this library's own; the reference's is [c][cell][q]; bp5_mf_metric_to_reference_layout converts):
 *      coef[c*n_cells*nq + cell*nq + off(qi, qj + n*qk)],   n = p+1, nq = n^3, ab = qj + n*qk,
 *      off(qi, ab) = (qi/2)*2n^2 + 2ab + (qi&1)  for qi < 2(n/2);   (n/2)*2n^2 + ab  for the last qi of odd n
 *    i.e. per cell the x-pencils of the n^2 (qj,qk) positions, stored as pairs (qi, qi+1) position after position:
 *    the kernels fetch 16 bytes per lane and a wave's load is one contiguous run. */
int bp5_mf_compute_merged_metric(bp5_mf *mf, double *coef);
/* Geometry representation used by bp5_apply / bp5_cg_solve:
 *   BP5_GEOM_MERGED6  the reference's six stored planes per q-point (G = 6 doubles per q-point), `coef`
 *                     from bp5_mf_compute_merged_metric -- the default;
 *   BP5_GEOM_AFFINE   for meshes whose cells are all affine (the reference's meshes always are: congruent
 *                     cubes, bp5/step-64.cu:656-663): K K^T is constant per cell, so the library keeps six
 *                     doubles per CELL and one scalar plane kappa*JxW per q-point (G = 1).  Same operator,
 *                     same results to rounding; the `coef` argument of apply/solve is then ignored.
 *                     Fails with BP5_ERR_UNSUPPORTED if K K^T varies by more than 1e-10 (relative) inside a cell. */
enum { BP5_GEOM_MERGED6 = 0, BP5_GEOM_AFFINE = 1 };
int bp5_mf_set_geometry_mode(bp5_mf *mf, int mode);
/* permute to the reference layout [c][cell][qi + n(qj + n qk)] (tests / interop) */
int bp5_mf_metric_to_reference_layout(bp5_mf *mf, const double *coef, double *coef_ref);

/* MatrixFree::Data mirror (bp5/fe_evaluation_gl.h:112-120, bp5/step-64.cu:94-97):
 * the unmerged geometry inv_jacobian (9 SoA planes) and JxW with deal.II's padding.
 * Arrays are created on first request. */
typedef struct {
  const uint32_t *local_to_global; /* [n_cells*padding_length]                                  */
  const double *inv_jacobian;      /* [9][n_cells*padding_length], plane d*3+e = d xi_d/d x_e   */
  const double *JxW;               /* [n_cells*padding_length]                                  */
  const double *q_points;          /* [3][n_cells*padding_length]                               */
  const uint32_t *constraint_mask; /* [n_cells], BP5_HANG_* bits (all zero on conforming meshes) */
  uint32_t n_cells, padding_length, row_start;
  int use_coloring;
} bp5_mf_data;
int bp5_mf_get_data(bp5_mf *mf, int color, bp5_mf_data *out);

/* == PoissonOperator::vmult(dst, src), bp5/step-64.cu:263-276:
 *    [dst = 0 if zero_dst] ; dst += sum_cells P^T B^T S B P src ; dst[c] = src[c] on Dirichlet DoFs.
 *    Single-rank form (no halo exchange). */
int bp5_apply(bp5_mf *mf, const double *coef, const double *src, double *dst, int zero_dst);
/* pieces, for hosts that overlap the halo exchange themselves (MatrixFree::cell_loop,
 * bp5/step-64.cu:274): cell range [cell_begin, cell_end) only, no zeroing, no Dirichlet copy */
int bp5_apply_cells(bp5_mf *mf, const double *coef, const double *src, double *dst, uint32_t cell_begin,
                    uint32_t cell_end);
/* (with cell blocks: ranges that are unions of whole blocks keep the block-assembled kernel -- DoFs shared with
 * blocks outside the range are then added atomically; [0, n_interior_cells) of bp5_mesh_create_brick is such a range) */
/* == mf_data.copy_constrained_values(src, dst), bp5/step-64.cu:275 */
int bp5_copy_constrained(bp5_mf *mf, const double *src, double *dst);
/* == MatrixFree::set_constrained_values(val, dst) [upstream] */
int bp5_set_constrained(bp5_mf *mf, double value, double *dst);

/* kernel variant selection for the fused operator (tuning / A-B tests):
 * 0 = library default: the measured best kernel for the degree, the geometry mode and the way the
 * cells were handed over (a mesh given in cell blocks runs the block-assembled kernel at p = 4) */
int bp5_mf_set_apply_variant(bp5_mf *mf, int variant);
/* cap on the persistent grid of the block-assembled kernel (0 = sized from the CU count; tuning / tests: a small cap
 * makes every workgroup walk several blocks even on a small mesh) */
int bp5_mf_set_block_workgroups(bp5_mf *mf, int max_workgroups);
/* cache policy for the data a CG iteration touches ONCE (the metric planes in the p = 4 lattice block kernel; v and x in the merged solver's
 * update kernel; results are the same bits either way): 1 = non-temporal accesses, so that this data does not evict the vectors that are
 * reused (p, r) from the 256 MB memory-side cache (-6 % per CG iteration at 1e7 DoFs, +1 % at 1e8), 0 = ordinary accesses,
 * -1 = chosen from the local size (default: non-temporal up to 2.4e7 local DoFs) */
int bp5_mf_set_streaming(bp5_mf *mf, int policy);
/* Per-handle tuning / A-B knobs (results are the same bits for every setting; two handles of one process may differ).  The
 * environment variables named below are read ONCE, by bp5_mf_create, as the handle's initial values -- nothing on the apply or
 * solve path reads the environment.
 *   BP5_TUNE_LATTICE_INDICES   (env BP5_LATTICE_INDICES, default 1) closed-form indices for lattice blocks; 0 keeps the packed
 *                              u16 stream.  Set before the block plan is built (first apply / solve / bp5_mf_block_plan_*).
 *   BP5_TUNE_EARLY_GATHER      (env BP5_EARLY_GATHER, default 1) fused merged CG across ranks: the ghost gather of the new search
 *                              direction travels under the update kernel (0: after it).
 *   BP5_TUNE_COMBINE_SIGNAL    (env BP5_COMBINE_SIGNAL, default 0) default exchange schedule with ONE combine launch: the ghost rows
 *                              first, the exchange released mid-launch by a stream wait-value (needs the capability below).
 *   BP5_TUNE_BOUNDARY_FIRST    (env BP5_BOUNDARY_FIRST = signal | launches, default 1 = signal) boundary-first schedule inside one
 *                              launch (stream wait-value) or as two launches (0).  The in-launch form is used only when the device
 *                              reports hipDeviceAttributeCanUseStreamWaitValue AND a producer / consumer self-check, run once when the
 *                              handle creates its communication stream, has seen the waiting stream released while the producing kernel
 *                              was still running (on plain device memory the runtime implements the wait by polling; if that ever
 *                              stops working mid-kernel the library falls back to two launches by itself).
 *   BP5_TUNE_FOLD_SMALL        (env BP5_FOLD_SMALL, default 1) merged CG on the atomic kernels: zero-fill and Dirichlet copy folded
 *                              into the neighbouring launches.
 *   BP5_TUNE_UPDATE_UNROLL     (env BP5_UPDATE_UNROLL, 1 | 2 | 4, default 1) 16-byte accesses per lane, stream and loop trip of the merged
 *                              solver's update kernels.
 *   BP5_TUNE_UPDATE_FLAT       (env BP5_UPDATE_FLAT, default 1) streaming kernels without a reduction (update kernels, vector updates)
 *                              are launched with ONE trip per workgroup instead of a capped grid-stride grid: the stores of a capped
 *                              grid drift apart and lose 15-35 % of the HBM write rate (profiles/r4: hbm_sweep).
 *   BP5_TUNE_UPDATE_NT         (env BP5_UPDATE_NT, -1 | 0 | 1, default -1) non-temporal accesses to v and x in the update kernels:
 *                              -1 = the library's default (on at every size: -0.9 % per iteration at 1e8 DoFs, -1.2 % at 1e7).
 *   BP5_TUNE_COMBINE_WG_PER_CU (env BP5_COMBINE_WG_PER_CU, 0 ... 32, default 16) workgroups per CU of the fused solver's combine pass (a fixed grid that walks
 *                              the tiles of brick-surface DoFs, one dot-product column per workgroup); 0 = as many workgroups as columns are free
 *                              (rounds 2-3).  The dot products are summed over another column layout (rounding-level differences), v is the same bits.
 *   BP5_TUNE_GHOST_COMBINE_ON_COMM (env BP5_GHOST_COMBINE_ON_COMM, default 0) default exchange schedule (overlap 2): the ghost-row window of the combine pass on the
 *                              communication stream (in front of the send) instead of the compute stream, so that the owned-row window starts right behind the brick kernel.
 *   BP5_TUNE_INTERIOR_STORES   (env BP5_INTERIOR_STORES, default 1) atomic pencil kernel of p >= 5 on a mesh whose cell-interior DoFs are numbered ahead of all
 *                              others (bp5_mesh_desc.dof_numbering = 2; detected at bp5_mf_create): plain stores for the (p-1)^3 entries a cell owns alone.
 *   BP5_TUNE_FACE_CARRY        (env BP5_FACE_CARRY, default 1) block kernel on lattice bricks (p = 4): the interior of the face two CONSECUTIVE bricks of one
 *                              workgroup's range share stays in LDS from the first brick's write-out to the second's, which stores the sum as an owner store;
 *                              the combine pass of that launch skips those DoFs.  v is the same bits as without (a + b == b + a); the fused dot products are
 *                              summed over other workgroups (rounding-level differences, reproducible run to run). */
enum { BP5_TUNE_LATTICE_INDICES = 0, BP5_TUNE_EARLY_GATHER = 1, BP5_TUNE_COMBINE_SIGNAL = 2, BP5_TUNE_BOUNDARY_FIRST = 3,
       BP5_TUNE_FOLD_SMALL = 4, BP5_TUNE_UPDATE_UNROLL = 5, BP5_TUNE_UPDATE_FLAT = 6, BP5_TUNE_UPDATE_NT = 7, BP5_TUNE_COMBINE_WG_PER_CU = 8,
       BP5_TUNE_INTERIOR_STORES = 9, BP5_TUNE_GHOST_COMBINE_ON_COMM = 10, BP5_TUNE_FACE_CARRY = 11, BP5_TUNE_COUNT = 12 };
int bp5_mf_set_tuning(bp5_mf *mf, int knob, int value);
int bp5_mf_get_tuning(const bp5_mf *mf, int knob, int *value);
/* 1 when the in-launch stream wait-value schedules are available on this handle (capability + self-check, see BP5_TUNE_BOUNDARY_FIRST);
 * creates the handle's communication stream if it does not exist yet */
int bp5_mf_wait_value_available(bp5_mf *mf, int *available);
/* facts about the block kernel's plan for this handle (builds it): number of cell blocks, longest run-length list of a
 * block, and whether the packed one-u16-per-DoF index form is available (<= 128 runs per block; brick-major numbering
 * gives ~30, a slab's boundary bricks with their ghost rows ~70) -- every rank of a multi-GPU run should report 1 */
int bp5_mf_block_plan_info(bp5_mf *mf, uint32_t *n_blocks, uint32_t *max_runs, int *packed_indices);
/* ... and how many of its cell blocks are LATTICE blocks: full boxes of cells whose DoFs are numbered entity by entity (what a brick-major
 * numbering produces; recognised topologically and verified entry by entry when the plan is built).  For those the kernel computes every
 * entry's list slot and DoF in closed form from the cell's position in its block and reads no per-DoF index at all (the packed stream of
 * the other blocks costs 2 bytes per cell-local DoF); the results are bitwise the same either way. */
int bp5_mf_block_plan_lattice(bp5_mf *mf, uint32_t *n_lattice_blocks);
/* ... and the face carry (BP5_TUNE_FACE_CARRY): faces the plan found that a workgroup may keep in LDS from one block to the next, the plan's
 * brick-surface (shared) DoFs, and how many of those the combine pass of the LAST block-kernel launch on this handle had in its tables
 * (== n_shared when that launch carried nothing). */
int bp5_mf_block_plan_carry(bp5_mf *mf, uint32_t *n_faces, uint32_t *n_shared, uint32_t *n_shared_last_launch);
/* the variant a whole-range application resolves to (what "0" means for this handle) */
int bp5_mf_get_apply_variant(bp5_mf *mf, int *effective);

/* diag(A_eff) of the operator bp5_apply applies (1 on Dirichlet DoFs), or its reciprocal when invert != 0: the
 * Jacobi preconditioner for the `diag` slot every solver kernel of the reference already threads through
 * (bp5/solver.h:68,100,131,170; bp5/step-64.cu:428-432 fills it with ones).  Setup-time, matrix-free
 * (sum-factorised, no element matrices); ghost contributions are sent to their owners when a communicator is set.
 * diag: owned + ghost storage, ghost entries are left zero.  BP5_OP_HELMHOLTZ: the mass term is included. */
int bp5_compute_diagonal(bp5_mf *mf, const double *coef, double *diag, int invert);

/* b_i = int phi_i with Gauss(p+1), constrained rows 0 (assemble_rhs, bp5/step-64.cu:372-418) */
int bp5_assemble_rhs(bp5_mf *mf, double *b);
/* ||u_h||_L2 by Gauss(p+1) quadrature (output_results, bp5/step-64.cu:602-616); synchronous.  With a communicator and
 * neighbours the ghost range of u is refreshed from its owners first (the reference integrates a ghosted copy,
 * ghost_solution_host) and zeroed again afterwards -- the owned entries are never written. */
int bp5_l2_norm_solution(bp5_mf *mf, const double *u, double *result_host);

/* ------------------------------------------------------------------------------------------ */
/* vector BLAS-1 used by the solvers (LinearAlgebra::distributed::Vector surface,
 * bp5/solver.h:369-382,417-421,511,528).  n = number of OWNED entries.                        */
int bp5_vec_fill(bp5_mf *mf, double *v, double value, size_t n);
int bp5_vec_axpy(bp5_mf *mf, double *y, double a, const double *x, size_t n);              /* y += a x   (add)  */
int bp5_vec_equ(bp5_mf *mf, double *y, double a, const double *x, size_t n);               /* y  = a x   (equ)  */
int bp5_vec_sadd(bp5_mf *mf, double *y, double s, double a, const double *x, size_t n);    /* y = s y + a x     */
int bp5_vec_dot(bp5_mf *mf, const double *x, const double *y, size_t n, double *result_host); /* synchronous, local */
/* Vector::l2_norm() / Vector::all_zero() (bp5/solver.h:369-382, bp5/step-64.cu:467): over the first n (= owned)
 * entries of every rank -- one on-stream RCCL all-reduce when a communicator is attached; synchronous */
int bp5_vec_l2_norm(bp5_mf *mf, const double *x, size_t n, double *result_host);
int bp5_vec_all_zero(bp5_mf *mf, const double *x, size_t n, int *result_host);

/* ------------------------------------------------------------------------------------------ */
/* communication: one rank per GPU, RCCL over xGMI                                             */
typedef struct bp5_comm bp5_comm;
#define BP5_UNIQUE_ID_BYTES 128
int bp5_comm_unique_id(char *id_host /*[BP5_UNIQUE_ID_BYTES]*/);    /* rank 0; broadcast by the host */
int bp5_comm_create(const char *id_host, int rank, int n_ranks, bp5_comm **out);
int bp5_comm_destroy(bp5_comm *comm);
int bp5_mf_set_comm(bp5_mf *mf, bp5_comm *comm);
/* sum-allreduce of n doubles in place on the handle's stream
 * (replaces cudaMemcpy D2H + MPI_Allreduce, bp5/solver.h:488-494) */
int bp5_comm_allreduce_sum(bp5_mf *mf, double *buf, size_t n);
/* == src.update_ghost_values_start/finish, dst.compress_start/finish(add), zero_out_ghosts
 *    (inside cell_loop, bp5/step-64.cu:274).
 *    *_start: the owners' values are packed on the handle's stream, then the RCCL send/recv group runs on the handle's own
 *    communication stream (ordered by events); *_finish: the handle's stream waits for the transfer (scatter-add: and adds
 *    the received contributions to the owned entries, ghosts zeroed).  Work enqueued on the handle's stream between a start
 *    and its finish overlaps the transfer; it must not touch the entries in flight (gather: the ghost range of v;
 *    scatter-add: the ghost range of v and nothing else).  One exchange of each kind may be in flight per handle.
 *    bp5_halo_gather / bp5_halo_scatter_add = start immediately followed by finish. */
int bp5_halo_gather_start(bp5_mf *mf, double *v);
int bp5_halo_gather_finish(bp5_mf *mf, double *v);
int bp5_halo_scatter_add_start(bp5_mf *mf, double *v);
int bp5_halo_scatter_add_finish(bp5_mf *mf, double *v);
int bp5_halo_gather(bp5_mf *mf, double *v);
int bp5_halo_scatter_add(bp5_mf *mf, double *v);
int bp5_halo_zero_ghosts(bp5_mf *mf, double *v);
/* == MatrixFree::AdditionalData::overlap_communication_computation (bp5/step-64.cu:241).  mode 1: on (the reference's setting);
 *    0: off -- the exchange stays on the handle's stream and the cell loop runs unsplit; 2 (default): the library decides.
 *    Solvers on the block kernel (fused dot products), mode 1: BOUNDARY-FIRST -- every workgroup of the (single) launch walks its share
 *    of the ghost-touching bricks first and counts itself in; the communication stream waits for the count (hipStreamWaitValue64),
 *    combines the ghost rows and sends them to their owners while the same launch works through the interior bricks.  Mode 2 there:
 *    one launch, then the ghost rows of the combine pass, the exchange on the communication stream UNDER the owned rows of the combine
 *    pass (a transfer that runs beside the bandwidth-bound brick kernel is slow and slows it; profiles/r3).  In every mode the ghost
 *    gather of the search direction travels under the vector update, and the results are bitwise the same.
 *    bp5_apply_distributed and the atomic kernels use the three-phase split (interior, boundary, interior): automatic from 1e6
 *    interior cells on -- it costs three launches and four cross-stream dependencies per application (profiles/r2, r3 READMEs) */
int bp5_mf_set_overlap(bp5_mf *mf, int mode);
/* distributed vmult == PoissonOperator::vmult on more than one rank (bp5/step-64.cu:263-276 with the cell_loop of :274).
 *    With the overlapped schedule (bp5_mf_set_overlap): ghost gather started; first part of the interior cells [0, n_interior_cells) underneath it; gather finished; the cells
 *    that touch ghosts; ghost contributions sent to their owners (atomic kernels: under the rest of the interior cells; the
 *    block kernel: after its single combine pass, which completes the ghost entries) and added; ghosts of src zeroed;
 *    Dirichlet copy.  Same kernels and, with the block kernel, bitwise the same result as the unsplit application (gather,
 *    all cells, scatter-add on the handle's stream), which is what runs when the overlap policy says off. */
int bp5_apply_distributed(bp5_mf *mf, const double *coef, double *src, double *dst, int zero_dst);

/* ------------------------------------------------------------------------------------------ */
/* Krylov solvers                                                                              */
enum { BP5_CG_PLAIN = 0,  /* deal.II SolverCG (bp5/step-64.cu:446-453): the parity target      */
       BP5_CG_MERGED = 1  /* SolverCGFullMerge (bp5/solver.h:343-542), x schedule fixed         */ };

typedef struct {
  int variant;       /* BP5_CG_*                                                                */
  int max_iter;      /* IterationNumberControl(max_iter, abs_tol), bp5/step-64.cu:443-445       */
  double abs_tol;    /* stop when ||r|| <= abs_tol                                              */
  int check_every;   /* host looks at the device-side convergence flag every k iterations
                        (0 = only at the end); the iterate is frozen on device at convergence
                        either way, so the result does not depend on it                         */
  int profile;       /* 1 = bracket every operator launch with HIP events; 2 = also stamp the phases of every
                        iteration (bp5_cg_result.phase_ms; diagnostic: the stamps themselves cost a few us) */
} bp5_cg_params;

typedef struct {
  int iterations;          /* SolverControl::last_step()                                        */
  double residual;         /* last ||r||                                                        */
  double initial_residual;
  double solve_ms;         /* HIP-event time of the whole solve on the stream                   */
  double apply_ms_avg;     /* profile=1: average duration of one launch of the cell kernel alone  */
  int apply_launches;
  double operator_ms_avg;  /* profile=1: zero-fill + cell kernel + combine pass (one A*x without halo) */
  int dot_products_fused;  /* 1: the solve formed its dot products inside the operator kernels (bp5_mf_set_cg_fusion) */
  int exchange_schedule;   /* halo exchange of the operator applications: 0 none (one rank), 1 unsplit (gather, all cells, scatter-add),
                              2 boundary-first (ghost-touching bricks, exchange on the communication stream under the interior bricks),
                              3 three-phase (atomic kernels / separate dot products: interior, boundary, interior),
                              4 all bricks in one launch, ghost rows combined first, exchange under the owned-row combine pass    */
  char apply_kernel[96];   /* the operator kernel the solve launched last, as a profiler prints it (e.g.
                              "apply_block_kernel<4,false,32,1,1337344>"): what rocprofv3 rows belong to this solve               */
  /* profile=2, BP5_CG_MERGED: average HIP-event time of the phases of one iteration on the handle's stream (ms), BP5_PHASE_* */
  double phase_ms[8];
} bp5_cg_result;
enum { BP5_PHASE_UPDATE = 0,      /* vector update (+ packing / starting the ghost gather of the new search direction)       */
       BP5_PHASE_GATHER_WAIT = 1, /* stream waits for the ghost values (exposed part of the gather)                          */
       BP5_PHASE_OPERATOR = 2,    /* cell kernels + combine pass (boundary-first: + ghost-row combine, exchange start)        */
       BP5_PHASE_EXCHANGE = 3,    /* exposed part of the scatter-add exchange + unpack (unsplit: the whole exchange)          */
       BP5_PHASE_REDUCE = 4,      /* local reduction of the partial sums                                                      */
       BP5_PHASE_ALLREDUCE = 5,   /* RCCL all-reduce of the 7 sums                                                            */
       BP5_PHASE_CONTROL = 6,     /* scalar step (alpha, beta, stopping test)                                                 */
       BP5_PHASE_ITERATION = 7    /* whole iteration                                                                          */ };

/* == cg.solve(A, x, b, preconditioner) with DiagonalMatrix (bp5/step-64.cu:446-453,488-495).
 *    diag may be NULL (== 1, bp5/step-64.cu:432).  x is overwritten (x0 = 0, bp5/step-64.cu:449).
 *    Returns BP5_OK also when max_iter is reached (IterationNumberControl reports success). */
int bp5_cg_solve(bp5_mf *mf, const double *coef, const double *diag, const double *b, double *x,
                 const bp5_cg_params *params, bp5_cg_result *result_host);

/* The same solvers for ANY operator: cg.solve(A, x, b, preconditioner) uses nothing of A but A.vmult(dst, src)
 * (bp5/solver.h:25-30,377,475; the reference solves step-64's Helmholtz operator with them, step-64/step-64.cu:505-530).
 * The callback must ENQUEUE dst = A src on the handle's stream (no host synchronisation needed), define every owned entry
 * of dst (Dirichlet rows included), may use the ghost range of src and dst as scratch, and returns BP5_OK or an error code
 * (which aborts the solve and is returned).  `mf` supplies the vector layout (n_owned), the stream, the communicator for
 * the dot products and the work vectors; b, x, diag as for bp5_cg_solve. */
typedef int (*bp5_vmult_fn)(void *ctx, double *dst, double *src);
int bp5_cg_solve_operator(bp5_mf *mf, bp5_vmult_fn vmult, void *ctx, const double *diag, const double *b, double *x,
                          const bp5_cg_params *params, bp5_cg_result *result_host);

/* BP5_CG_MERGED on the packed block kernel (cell bricks, one rank's cells, diag == NULL): by default the operator's
 * write-out and combine pass also form the v-dependent dot products of update_b (bp5/solver.h:142-311: p.v, v.v, r.v, r.r)
 * and write the Dirichlet rows, so the separate pass over p, r, v and the copy_constrained launch disappear.  Same
 * arithmetic, different summation order (still fixed: bitwise reproducible).  BP5_CG_PLAIN on that kernel takes d.h (its one
 * dot product with the operator's result; any diag) from it the same way: the sum over the cells of the quadrature-point
 * energy, so the pass over d and h disappears.  0 switches back to the separate kernels. */
int bp5_mf_set_cg_fusion(bp5_mf *mf, int on);

/* event helpers so a host in another language can time on the handle's stream */
typedef struct bp5_event bp5_event;
int bp5_event_create(bp5_event **out);
int bp5_event_record(bp5_mf *mf, bp5_event *ev);
int bp5_event_elapsed_ms(bp5_event *start, bp5_event *stop, double *ms_host); /* synchronises on stop */
int bp5_event_destroy(bp5_event *ev);

#ifdef __cplusplus
}
#endif
#endif /* BP5_H */
