// C++ host program written against the deal.II-shaped facade (include/bp5_dealii_facade.hpp):
// the BP5 Poisson operator and the step-64 Helmholtz operator as user device functors, the fused
// library path, and the two CG solvers -- the host-side shape of the reference's
// bp5/step-64.cu (PoissonProblem) and step-64/step-64.cu (HelmholtzProblem).
//
//   bp5_step64 check <degree> <nx> <ny> <nz> <deform> <prefix>   functor path vs fused path; dumps vectors
//   bp5_step64 bench <degree> <n> <iterations> <repetitions>     prints pcg-standard / pcg-merged / vmult lines
//   bp5_step64 hanging <degree> <prefix>                         externally generated 2:1 refined mesh (<prefix>_l2g/_coords/_constrained/
//                                                                _mask/_src.bin): functor path with resolve_hanging_nodes vs library kernel
//   bp5_step64 helmholtz_native <degree> <n> <prefix>            the same through the library's native fused Helmholtz kernel
//   bp5_step64 helmholtz <degree> <n> <prefix>                   HelmholtzProblem::solve of step-64/step-64.cu on n^3 cells of the
//                                                                unit cube: the functor operator inside the library's CG solvers
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bp5_dealii_facade.hpp"

using namespace bp5::dealii_facade;

// ---- merged-metric functor (role of JacobianFunctor, bp5/step-64.cu:60-114)
template <int dim, int fe_degree>
class MetricFunctor {
public:
  MetricFunctor(double *c, unsigned int nc) : coef(c), n_cells(nc) {}
  static const unsigned int n_dofs_1d = fe_degree + 1;
  static const unsigned int n_q_points = Utilities::pow(n_dofs_1d, dim);
  __device__ void operator()(const unsigned int cell, const typename CUDAWrappers::MatrixFree<dim, double>::Data *gpu_data)
  {
    const unsigned int q = CUDAWrappers::q_point_id_in_cell<dim>(n_dofs_1d);
    const size_t plane = (size_t)gpu_data->n_cells * gpu_data->padding_length;
    const size_t at = (size_t)cell * gpu_data->padding_length + q;
    double K[dim][dim];
    for (int d = 0; d < dim; ++d)
      for (int e = 0; e < dim; ++e) K[d][e] = gpu_data->inv_jacobian[at + plane * (d * dim + e)];
    const double jxw = gpu_data->JxW[at];
    const size_t o = q + (size_t)cell * n_q_points, stride = (size_t)n_cells * n_q_points;
    unsigned int c = 0;
    for (int d = 0; d < dim; ++d, ++c) coef[o + c * stride] = jxw * (K[d][0] * K[d][0] + K[d][1] * K[d][1] + K[d][2] * K[d][2]);
    for (int d = 0; d < dim; ++d)
      for (int e = d + 1; e < dim; ++e, ++c) coef[o + c * stride] = jxw * (K[d][0] * K[e][0] + K[d][1] * K[e][1] + K[d][2] * K[e][2]);
  }

private:
  double *coef;
  const unsigned int n_cells;
};

// ---- BP5 cell operator through FEEvaluation (role of LocalPoissonOperator, bp5/step-64.cu:118-194).
// MERGED: read the six planes written by MetricFunctor (WITH the cell offset the reference forgets,
// SURVEY 0.5); otherwise the unmerged submit_gradient(get_gradient()) path.
template <int dim, int fe_degree, bool MERGED>
class LocalPoisson {
public:
  LocalPoisson(const double *c, unsigned int nc) : coef(c), n_cells(nc) {}
  static const unsigned int n_dofs_1d = fe_degree + 1;
  static const unsigned int n_local_dofs = Utilities::pow(fe_degree + 1, dim);
  static const unsigned int n_q_points = Utilities::pow(fe_degree + 1, dim);
  __device__ void operator()(const unsigned int cell, const typename CUDAWrappers::MatrixFree<dim, double>::Data *gpu_data,
                             CUDAWrappers::SharedData<dim, double> *shared_data, const double *src, double *dst) const
  {
    CUDAWrappers::FEEvaluation<dim, fe_degree, fe_degree + 1, 1, double> fe_eval(cell, gpu_data, shared_data);
    fe_eval.read_dof_values(src);
    fe_eval.evaluate(false, true);
    if (MERGED) {
      const size_t stride = (size_t)n_q_points * n_cells;
      const unsigned int q = CUDAWrappers::internal::compute_index<dim, fe_degree + 1>();
      const double *S = coef + (size_t)cell * n_q_points + q;
      const double a = shared_data->gradients[0][q], b = shared_data->gradients[1][q], c = shared_data->gradients[2][q];
      shared_data->gradients[0][q] = S[0] * a + S[3 * stride] * b + S[4 * stride] * c;
      shared_data->gradients[1][q] = S[3 * stride] * a + S[stride] * b + S[5 * stride] * c;
      shared_data->gradients[2][q] = S[4 * stride] * a + S[5 * stride] * b + S[2 * stride] * c;
      __syncthreads();
    } else {
      fe_eval.submit_gradient(fe_eval.get_gradient());
    }
    fe_eval.integrate(false, true);
    fe_eval.distribute_local_to_global(dst);
  }

private:
  const double *coef;
  const unsigned int n_cells;
};

// ---- step-64 Helmholtz: (grad v, grad u) + (v, a(x) u), a = 10/(0.05 + 2|x|^2) (step-64/step-64.cu:99-219)
template <int dim, int fe_degree>
class VaryingCoefficient {
public:
  explicit VaryingCoefficient(double *c) : coef(c) {}
  static const unsigned int n_dofs_1d = fe_degree + 1;
  static const unsigned int n_q_points = Utilities::pow(n_dofs_1d, dim);
  __device__ void operator()(const unsigned int cell, const typename CUDAWrappers::MatrixFree<dim, double>::Data *gpu_data)
  {
    const unsigned int pos = CUDAWrappers::local_q_point_id<dim, double>(cell, gpu_data, n_dofs_1d, n_q_points);
    const auto x = CUDAWrappers::get_quadrature_point<dim, double>(cell, gpu_data, n_dofs_1d);
    double r2 = 0;
    for (int d = 0; d < dim; ++d) r2 += x[d] * x[d];
    coef[pos] = 10.0 / (0.05 + 2.0 * r2);
  }

private:
  double *coef;
};
template <int dim, int fe_degree>
class HelmholtzQuad {
public:
  explicit __device__ HelmholtzQuad(double c) : coef(c) {}
  __device__ void operator()(CUDAWrappers::FEEvaluation<dim, fe_degree> *fe_eval, const unsigned int q) const
  {
    fe_eval->submit_value(coef * fe_eval->get_value(q), q);
    fe_eval->submit_gradient(fe_eval->get_gradient(q), q);
  }

private:
  double coef;
};
template <int dim, int fe_degree>
class LocalHelmholtz {
public:
  explicit LocalHelmholtz(const double *c) : coef(c) {}
  static const unsigned int n_dofs_1d = fe_degree + 1;
  static const unsigned int n_local_dofs = Utilities::pow(fe_degree + 1, dim);
  static const unsigned int n_q_points = Utilities::pow(fe_degree + 1, dim);
  __device__ void operator()(const unsigned int cell, const typename CUDAWrappers::MatrixFree<dim, double>::Data *gpu_data,
                             CUDAWrappers::SharedData<dim, double> *shared_data, const double *src, double *dst) const
  {
    const unsigned int pos = CUDAWrappers::local_q_point_id<dim, double>(cell, gpu_data, n_dofs_1d, n_q_points);
    CUDAWrappers::FEEvaluation<dim, fe_degree> fe_eval(cell, gpu_data, shared_data);
    fe_eval.read_dof_values(src);
    fe_eval.evaluate(true, true);
    fe_eval.apply_quad_point_operations(HelmholtzQuad<dim, fe_degree>(coef[pos]));
    fe_eval.integrate(true, true);
    fe_eval.distribute_local_to_global(dst);
  }

private:
  const double *coef;
};

using DeviceVector = LinearAlgebra::distributed::Vector<double, MemorySpace::CUDA>; // bp5/step-64.cu:349

// ---- operator wrapper with the reference's public surface (bp5/step-64.cu:198-276)
template <int dim, int fe_degree>
class PoissonOperator {
public:
  PoissonOperator(const bp5_mesh_view &mv, int quadrature, bool use_functor_path, const uint32_t *constraint_mask_host = nullptr)
    : functor_path(use_functor_path), do_zero_out(true)
  {
    bp5_mf_desc d{};
    d.dim = dim; d.degree = fe_degree; d.quadrature = quadrature; d.coefficient = BP5_COEF_ONE;
    d.n_cells = mv.n_cells; d.n_interior_cells = mv.n_interior_cells; d.n_owned = mv.n_owned; d.n_ghost = mv.n_ghost;
    d.local_to_global_host = mv.local_to_global_host; d.node_coords_host = mv.node_coords_host;
    d.constrained_host = mv.constrained_host; d.n_constrained = mv.n_constrained;
    d.n_cell_blocks = mv.n_cell_blocks; d.cell_block_offsets_host = mv.cell_block_offsets_host; // cells handed over in bricks (if any)
    d.constraint_mask_host = constraint_mask_host; // hanging nodes of a 2:1 refined mesh (BP5_HANG_* bits), or NULL
    mf_data.reinit(d);
    n_owned_cells = mv.n_cells;
    n_local = mv.n_owned + mv.n_ghost;
    size_t nc;
    check(bp5_mf_coef_size(mf_data.handle(), &nc));
    check(bp5_vec_alloc(nc, &coef_fast));
    check(bp5_mf_compute_merged_metric(mf_data.handle(), coef_fast)); // library layout, fused kernels
    check(bp5_vec_alloc(nc, &coef_ref));
    mf_data.evaluate_coefficients(MetricFunctor<dim, fe_degree>(coef_ref, n_owned_cells)); // reference layout, functor path
  }
  ~PoissonOperator() { bp5_vec_free(coef_fast); bp5_vec_free(coef_ref); }
  void vmult(double *dst, const double *src) const
  {
    if (!functor_path) { check(bp5_apply(mf_data.handle(), coef_fast, src, dst, do_zero_out)); return; }
    if (do_zero_out) check(bp5_vec_fill(mf_data.handle(), dst, 0.0, n_local));
    mf_data.cell_loop(LocalPoisson<dim, fe_degree, true>(coef_ref, n_owned_cells), src, dst);
    mf_data.copy_constrained_values(src, dst);
  }
  void vmult_unmerged(double *dst, const double *src) const
  {
    check(bp5_vec_fill(mf_data.handle(), dst, 0.0, n_local));
    mf_data.cell_loop(LocalPoisson<dim, fe_degree, false>(nullptr, n_owned_cells), src, dst);
    mf_data.copy_constrained_values(src, dst);
  }
  void initialize_dof_vector(double **v) const { mf_data.initialize_dof_vector(v); }
  void initialize_dof_vector(DeviceVector &v) const { mf_data.initialize_dof_vector(v); } // bp5/step-64.cu:214
  void vmult(DeviceVector &dst, const DeviceVector &src) const { vmult(dst.get_values(), static_cast<const double *>(src.get_values())); }
  bp5_mf *handle() const { return mf_data.handle(); }
  const double *coef() const { return coef_fast; }
  const double *coef_reference_layout() const { return coef_ref; }
  CUDAWrappers::MatrixFree<dim, double> mf_data;
  bool functor_path;

private:
  double *coef_fast = nullptr, *coef_ref = nullptr;
  unsigned int n_owned_cells = 0, n_local = 0;

public:
  bool do_zero_out;
};

// ---- step-64's HelmholtzOperator (step-64/step-64.cu:226-300): owns the MatrixFree object and the coefficient array, applies the
// user functor through cell_loop.  It has NO coef(): the solvers treat it as a foreign operator and only ever call vmult.
template <int dim, int fe_degree>
class HelmholtzOperator {
public:
  explicit HelmholtzOperator(const bp5_mesh_view &mv)
  {
    bp5_mf_desc d{};
    d.dim = dim; d.degree = fe_degree; d.quadrature = BP5_QUAD_GAUSS; d.coefficient = BP5_COEF_ONE;
    d.n_cells = mv.n_cells; d.n_interior_cells = mv.n_interior_cells; d.n_owned = mv.n_owned; d.n_ghost = mv.n_ghost;
    d.local_to_global_host = mv.local_to_global_host; d.node_coords_host = mv.node_coords_host;
    d.constrained_host = mv.constrained_host; d.n_constrained = mv.n_constrained;
    mf_data.reinit(d);
    check(bp5_vec_alloc((size_t)mv.n_cells * Utilities::pow(fe_degree + 1, dim), &coef));
    mf_data.evaluate_coefficients(VaryingCoefficient<dim, fe_degree>(coef)); // step-64/step-64.cu:249-251
  }
  ~HelmholtzOperator() { bp5_vec_free(coef); }
  void vmult(DeviceVector &dst, const DeviceVector &src) const
  { // step-64/step-64.cu:283-300: dst = 0; cell_loop; copy_constrained_values
    dst = 0.;
    mf_data.cell_loop(LocalHelmholtz<dim, fe_degree>(coef), static_cast<const double *>(src.get_values()), dst.get_values());
    mf_data.copy_constrained_values(static_cast<const double *>(src.get_values()), dst.get_values());
  }
  void initialize_dof_vector(DeviceVector &v) const { mf_data.initialize_dof_vector(v); }
  bp5_mf *handle() const { return mf_data.handle(); }
  CUDAWrappers::MatrixFree<dim, double> mf_data;

private:
  double *coef = nullptr;
};

// ---- the same operator on the library's NATIVE fused Helmholtz kernel (bp5_mf_set_operator(BP5_OP_HELMHOLTZ)): what a maintainer of
// step-64 switches to once the functor version works -- same class surface, plus coef(): the solvers then run it as the library's own
// operator (fused kernels, CG dot products inside the operator on cell bricks) instead of calling back into vmult.
template <int dim, int fe_degree>
class HelmholtzOperatorNative {
public:
  explicit HelmholtzOperatorNative(const bp5_mesh_view &mv)
  {
    bp5_mf_desc d{};
    d.dim = dim; d.degree = fe_degree; d.quadrature = BP5_QUAD_GAUSS; d.coefficient = BP5_COEF_STEP64; // a(x) of VaryingCoefficientFunctor
    d.n_cells = mv.n_cells; d.n_interior_cells = mv.n_interior_cells; d.n_owned = mv.n_owned; d.n_ghost = mv.n_ghost;
    d.local_to_global_host = mv.local_to_global_host; d.node_coords_host = mv.node_coords_host;
    d.constrained_host = mv.constrained_host; d.n_constrained = mv.n_constrained;
    d.n_cell_blocks = mv.n_cell_blocks; d.cell_block_offsets_host = mv.cell_block_offsets_host;
    mf_data.reinit(d);
    check(bp5_mf_set_operator(mf_data.handle(), BP5_OP_HELMHOLTZ));
    size_t nc;
    check(bp5_mf_coef_size(mf_data.handle(), &nc)); // seven planes: six merged + a JxW
    check(bp5_vec_alloc(nc, &coef7));
    check(bp5_mf_compute_merged_metric(mf_data.handle(), coef7));
  }
  ~HelmholtzOperatorNative() { bp5_vec_free(coef7); }
  void vmult(DeviceVector &dst, const DeviceVector &src) const
  { // HelmholtzOperator::vmult, step-64/step-64.cu:283-300, in one call
    check(bp5_apply(mf_data.handle(), coef7, static_cast<const double *>(src.get_values()), dst.get_values(), 1));
  }
  void initialize_dof_vector(DeviceVector &v) const { mf_data.initialize_dof_vector(v); }
  bp5_mf *handle() const { return mf_data.handle(); }
  const double *coef() const { return coef7; }
  CUDAWrappers::MatrixFree<dim, double> mf_data;

private:
  double *coef7 = nullptr;
};

static std::vector<double> download(const double *d, size_t n)
{
  std::vector<double> h(n);
  check(bp5_copy_d2h(h.data(), d, n * sizeof(double)));
  return h;
}
static double rel_diff(const std::vector<double> &a, const std::vector<double> &b)
{
  double num = 0, den = 0;
  for (size_t i = 0; i < a.size(); ++i) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
  return std::sqrt(num / den);
}
static void dump(const std::string &path, const std::vector<double> &v)
{
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  fwrite(v.data(), sizeof(double), v.size(), f);
  fclose(f);
}

template <int fe_degree>
static int run_check(uint32_t nx, uint32_t ny, uint32_t nz, double deform, const std::string &prefix)
{
  constexpr int dim = 3;
  bp5_mesh_desc md{};
  md.degree = fe_degree; md.cells[0] = nx; md.cells[1] = ny; md.cells[2] = nz; md.h = 1.0; md.deform_amp = deform; md.rank = 0; md.n_ranks = 1;
  bp5_mesh *mesh;
  check(bp5_mesh_create_brick(&md, &mesh));
  bp5_mesh_view mv;
  check(bp5_mesh_view_get(mesh, &mv));
  const size_t n = mv.n_owned;
  PoissonOperator<dim, fe_degree> fast(mv, BP5_QUAD_GAUSS, false), generic(mv, BP5_QUAD_GAUSS, true);
  // deterministic source with non-zero boundary values
  std::vector<double> s(n);
  uint64_t state = 88172645463325252ull;
  for (auto &x : s) { state ^= state << 13; state ^= state >> 7; state ^= state << 17; x = (double)(state >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
  double *src, *d1, *d2, *d3;
  fast.initialize_dof_vector(&src); fast.initialize_dof_vector(&d1); fast.initialize_dof_vector(&d2); fast.initialize_dof_vector(&d3);
  check(bp5_copy_h2d(src, s.data(), n * sizeof(double)));
  fast.vmult(d1, src);
  generic.vmult(d2, src);
  generic.vmult_unmerged(d3, src);
  check(bp5_mf_sync(fast.handle()));
  check(bp5_mf_sync(generic.handle()));
  const auto h1 = download(d1, n), h2 = download(d2, n), h3 = download(d3, n);
  // metric: functor (reference layout) vs library (converted)
  size_t nc;
  check(bp5_mf_coef_size(fast.handle(), &nc));
  double *cref;
  check(bp5_vec_alloc(nc, &cref));
  check(bp5_mf_metric_to_reference_layout(fast.handle(), fast.coef(), cref));
  check(bp5_mf_sync(fast.handle()));
  const double e_metric = rel_diff(download(generic.coef_reference_layout(), nc), download(cref, nc));
  // Helmholtz through the functor path
  double *hcoef, *dh;
  check(bp5_vec_alloc((size_t)mv.n_cells * Utilities::pow(fe_degree + 1, dim), &hcoef));
  fast.initialize_dof_vector(&dh);
  generic.mf_data.evaluate_coefficients(VaryingCoefficient<dim, fe_degree>(hcoef));
  generic.mf_data.cell_loop(LocalHelmholtz<dim, fe_degree>(hcoef), src, dh);
  generic.mf_data.copy_constrained_values(src, dh);
  check(bp5_mf_sync(generic.handle()));
  // CG through the facade solvers
  double *b, *x1, *x2;
  fast.initialize_dof_vector(&b); fast.initialize_dof_vector(&x1); fast.initialize_dof_vector(&x2);
  check(bp5_assemble_rhs(fast.handle(), b));
  IterationNumberControl c1(10, 0.0), c2(10, 0.0);
  SolverCG cg(c1);
  cg.solve(fast, x1, b, DiagonalMatrix());
  SolverCGFullMerge cgm(c2);
  cgm.solve(fast, x2, b, DiagonalMatrix());
  const double e_cg = rel_diff(download(x2, n), download(x1, n));
  // the same solve written against the vector class, as PoissonProblem::solve does (bp5/step-64.cu:428-453,467)
  double e_vec = 0;
  {
    DeviceVector solution, system_rhs, tmp;
    fast.initialize_dof_vector(solution);
    fast.initialize_dof_vector(system_rhs);
    if (!solution.all_zero() || solution.local_size() != n || solution.size() != n) throw std::runtime_error("DeviceVector: reinit/all_zero/size");
    LinearAlgebra::ReadWriteVector<double> rw(n);
    const auto hb = download(b, n);
    for (size_t i = 0; i < n; ++i) rw[i] = hb[i];
    system_rhs.import(rw, VectorOperation::insert);
    double nb = 0;
    for (double v : hb) nb += v * v;
    if (std::abs(system_rhs.l2_norm() - std::sqrt(nb)) > 1e-13 * std::sqrt(nb) || system_rhs.all_zero()) throw std::runtime_error("DeviceVector: import/l2_norm");
    IterationNumberControl c3(10, 1e-6 * system_rhs.l2_norm());
    SolverCGFullMerge cgv(c3);
    solution = 0;
    cgv.solve(fast, solution, system_rhs, DiagonalMatrix());
    e_vec = rel_diff(download(solution.get_values(), n), download(x2, n));
    // add / equ / sadd / vmult on vectors: tmp = A solution - b, then ||tmp|| = CG residual
    tmp.reinit(solution);
    fast.vmult(tmp, solution);
    tmp.add(-1.0, system_rhs);
    const double res = tmp.l2_norm();
    tmp.equ(2.0, system_rhs);
    tmp.sadd(0.5, -1.0, system_rhs);
    if (!(tmp.l2_norm() < 1e-14 * system_rhs.l2_norm())) throw std::runtime_error("DeviceVector: equ/sadd");
    solution.update_ghost_values(); solution.compress(VectorOperation::add); solution.zero_out_ghosts(); // one rank: no-ops
    printf("vector_api_cg %.3e\nvector_api_residual %.6e\nsolver_residual %.6e\n", e_vec, res, c3.last_value());
  }
  printf("check p=%d cells=%ux%ux%u dofs=%zu\n", fe_degree, nx, ny, nz, n);
  printf("functor_vs_fused %.3e\nunmerged_vs_fused %.3e\nmetric_functor_vs_library %.3e\nmerged_vs_plain_cg %.3e iterations %u %u\n",
         rel_diff(h2, h1), rel_diff(h3, h1), e_metric, e_cg, c1.last_step(), c2.last_step());
  dump(prefix + "_src.bin", s);
  dump(prefix + "_poisson.bin", h1);
  dump(prefix + "_poisson_functor.bin", h2);
  dump(prefix + "_helmholtz.bin", download(dh, n));
  dump(prefix + "_cg.bin", download(x1, n));
  bp5_mesh_destroy(mesh);
  return 0;
}

template <int fe_degree>
static int run_bench(uint32_t ncell, int n_iterations, int n_repetitions, const uint32_t *cells3 = nullptr, double h = 0.0)
{ // measurement protocol of PoissonProblem::solve, bp5/step-64.cu:422-561
  constexpr int dim = 3;
  bp5_mesh_desc md{};
  md.degree = fe_degree; md.cells[0] = md.cells[1] = md.cells[2] = ncell; md.h = 1.0 / ncell; md.n_ranks = 1;
  if (cells3) { md.cells[0] = cells3[0]; md.cells[1] = cells3[1]; md.cells[2] = cells3[2]; md.h = h; }
  // the cell order is the host's choice (MatrixFree::reinit reorders cells as well): the bricks bench.py uses for this degree
  const bool small = (uint64_t)md.cells[0] * md.cells[1] * md.cells[2] < 400000;
  const uint32_t bricks[9][3] = {{0, 0, 0}, {8, 8, 8}, {0, 0, 0}, {8, 4, 4}, {4, 4, small ? 2u : 4u}, {8, 8, 8}, {4, 4, 2}, {4, 2, 2}, {8, 8, 8}};
  if (bricks[fe_degree][0]) {
    for (int d = 0; d < 3; ++d) md.cell_block[d] = bricks[fe_degree][d];
    md.dof_numbering = 1; md.cell_block_order = 1;
  }
  bp5_mesh *mesh;
  check(bp5_mesh_create_brick(&md, &mesh));
  bp5_mesh_view mv;
  check(bp5_mesh_view_get(mesh, &mv));
  PoissonOperator<dim, fe_degree> A(mv, BP5_QUAD_GAUSS, false);
  double *b, *x;
  A.initialize_dof_vector(&b); A.initialize_dof_vector(&x);
  check(bp5_assemble_rhs(A.handle(), b));
  double bb;
  check(bp5_vec_dot(A.handle(), b, b, mv.n_owned, &bb));
  printf("   Number of active cells:       %u\n   Number of degrees of freedom: %llu\n\n", mv.n_cells, (unsigned long long)mv.n_global_dofs);
  for (int variant = 0; variant < 2; ++variant) {
    double best = 0;
    for (int r = 0; r < n_repetitions; ++r) {
      IterationNumberControl control(n_iterations, 1e-6 * std::sqrt(bb));
      bp5_cg_result res;
      if (variant == 0) { SolverCG cg(control); cg.solve(A, x, b, DiagonalMatrix()); res = cg.result; }
      else { SolverCGFullMerge cg(control); cg.solve(A, x, b, DiagonalMatrix()); res = cg.result; }
      const double thr = (double)mv.n_global_dofs * control.last_step() / (res.solve_ms * 1e-3);
      best = std::max(best, thr);
      double xx;
      check(bp5_vec_dot(A.handle(), x, x, mv.n_owned, &xx));
      printf("   Solved in %u iterations with time %g and DoFs/s %g norm %.12g\n", control.last_step(), res.solve_ms * 1e-3, thr, std::sqrt(xx));
    }
    printf("%s %llu %g\n\n", variant == 0 ? "pcg-standard" : "pcg-merged", (unsigned long long)mv.n_global_dofs, best);
  }
  {
    double best = 0;
    bp5_event *e0, *e1;
    check(bp5_event_create(&e0)); check(bp5_event_create(&e1));
    for (int r = 0; r < n_repetitions; ++r) {
      check(bp5_event_record(A.handle(), e0));
      for (int t = 0; t < n_iterations; ++t) A.vmult(x, b);
      check(bp5_event_record(A.handle(), e1));
      double ms;
      check(bp5_event_elapsed_ms(e0, e1, &ms));
      const double thr = (double)mv.n_global_dofs * n_iterations / (ms * 1e-3);
      best = std::max(best, thr);
      printf("   %d mat-vecs in time %g and DoFs/s %g\n", n_iterations, ms * 1e-3, thr);
    }
    printf("vmult %llu %g\n\n", (unsigned long long)mv.n_global_dofs, best);
    bp5_event_destroy(e0); bp5_event_destroy(e1);
  }
  bp5_mesh_destroy(mesh);
  return 0;
}

// PoissonProblem<dim, degree>::run(cycle_min, cycle_max, n_iterations, n_repetitions, min_run) of the reference's main program
// (bp5/step-64.cu:619-700,724-730: degree 5, cycles 7...40, 200 iterations, 10 repetitions): the same mesh family -- a brick of
// (1|2|3) x (1|2) x (1|2) unit cells by the cycle's remainder mod 6, refined globally cycle / 6 times (:633-663) -- and the same
// output lines per cycle.  min_run skips meshes with fewer DoFs (as the reference's per-rank lower bound does).
template <int fe_degree>
static int run_cycles(int cycle_min, int cycle_max, int n_iterations, int n_repetitions, unsigned long long min_run, unsigned long long max_run)
{
  for (int cycle = cycle_min; cycle <= cycle_max; ++cycle) {
    // coarse brick by cycle mod 6; the 12-cell brick of remainder 1 takes one global refinement less (cycle 1 itself is the single cube)
    static const uint32_t family[6][3] = {{1, 1, 1}, {3, 2, 2}, {2, 1, 1}, {3, 1, 1}, {2, 2, 1}, {3, 2, 1}};
    const int remainder = cycle % 6;
    const bool big_brick = remainder == 1 && cycle > 1;
    const int n_refine = cycle / 6 - (big_brick ? 1 : 0);
    const uint32_t sub[3] = {remainder == 1 && !big_brick ? 1u : family[remainder][0], remainder == 1 && !big_brick ? 1u : family[remainder][1],
                             remainder == 1 && !big_brick ? 1u : family[remainder][2]};
    const uint32_t cells[3] = {sub[0] << n_refine, sub[1] << n_refine, sub[2] << n_refine};
    const unsigned long long n_dofs = (unsigned long long)(cells[0] * fe_degree + 1) * (cells[1] * fe_degree + 1) * (cells[2] * fe_degree + 1);
    if (n_dofs < min_run || n_dofs > max_run) continue;
    printf("Cycle %d\n", cycle);
    const int st = run_bench<fe_degree>(0, n_iterations, n_repetitions, cells, 1.0 / (1u << n_refine));
    if (st) return st;
    fflush(stdout);
  }
  return 0;
}

static void dump(const std::string &path, const std::vector<double> &v);
// HelmholtzProblem<dim, fe_degree>::run for ONE cycle (step-64/step-64.cu:505-530,602-616,634-663): n^3 cells of the unit cube,
// f == 1, zero Dirichlet values, a(x) = 10 / (0.05 + 2 |x|^2), identity preconditioner, tolerance 1e-12 ||b||, at most n_dofs
// iterations; prints the iteration counts and the L2 norm of the solution for SolverCG and SolverCGFullMerge.
template <int fe_degree, bool NATIVE = false>
static int run_helmholtz(uint32_t ncell, const std::string &prefix)
{
  constexpr int dim = 3;
  bp5_mesh_desc md{};
  md.degree = fe_degree; md.cells[0] = md.cells[1] = md.cells[2] = ncell; md.h = 1.0 / ncell; md.n_ranks = 1;
  bp5_mesh *mesh;
  check(bp5_mesh_create_brick(&md, &mesh));
  bp5_mesh_view mv;
  check(bp5_mesh_view_get(mesh, &mv));
  typename std::conditional<NATIVE, HelmholtzOperatorNative<dim, fe_degree>, HelmholtzOperator<dim, fe_degree>>::type system_matrix_dev(mv);
  DeviceVector solution_dev, system_rhs_dev;
  system_matrix_dev.initialize_dof_vector(solution_dev);
  system_matrix_dev.initialize_dof_vector(system_rhs_dev);
  check(bp5_assemble_rhs(system_matrix_dev.handle(), system_rhs_dev.get_values())); // (phi_i, 1), constrained rows 0: step-64.cu:443-481
  printf("   Number of active cells:       %u\n   Number of degrees of freedom: %llu\n", mv.n_cells, (unsigned long long)mv.n_global_dofs);
  const size_t n = mv.n_owned;
  for (int merged = 0; merged < 2; ++merged) {
    SolverControl solver_control((unsigned int)system_rhs_dev.size(), 1e-12 * system_rhs_dev.l2_norm());
    solution_dev = 0.;
    if (merged) { SolverCGFullMerge cg(solver_control); cg.solve(system_matrix_dev, solution_dev, system_rhs_dev, DiagonalMatrix()); }
    else { SolverCG cg(solver_control); cg.solve(system_matrix_dev, solution_dev, system_rhs_dev, DiagonalMatrix()); }
    double norm = 0;
    check(bp5_l2_norm_solution(system_matrix_dev.handle(), solution_dev.get_values(), &norm));
    printf("  %s: Solved in %u iterations.\n  solution norm: %.10g\n", merged ? "SolverCGFullMerge" : "SolverCG", solver_control.last_step(), norm);
    printf("helmholtz_%s_iterations %u\nhelmholtz_%s_norm %.12e\nhelmholtz_%s_residual %.6e\n", merged ? "merged" : "plain", solver_control.last_step(),
           merged ? "merged" : "plain", norm, merged ? "merged" : "plain", solver_control.last_value());
    dump(prefix + (merged ? "_helmholtz_x_merged.bin" : "_helmholtz_x_plain.bin"), download(solution_dev.get_values(), n));
  }
  dump(prefix + "_helmholtz_b.bin", download(system_rhs_dev.get_values(), n));
  bp5_mesh_destroy(mesh);
  return 0;
}

template <typename T>
static std::vector<T> slurp(const std::string &path)
{
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot read " + path);
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<T> v(bytes / sizeof(T));
  if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) { fclose(f); throw std::runtime_error("short read: " + path); }
  fclose(f);
  return v;
}
// A mesh with hanging nodes handed over as flat arrays (what a host sitting under real deal.II would extract): the user functor
// goes through FEEvaluation::read_dof_values / distribute_local_to_global, which resolve the hanging-node constraints from
// Data::constraint_mask (bp5/fe_evaluation_gl.h:150-151,167-168); the library's own kernel (apply variant 90) does the same.
template <int fe_degree>
static int run_hanging(const std::string &prefix)
{
  constexpr int dim = 3, n3 = (fe_degree + 1) * (fe_degree + 1) * (fe_degree + 1);
  const auto l2g = slurp<uint32_t>(prefix + "_l2g.bin"), constrained = slurp<uint32_t>(prefix + "_constrained.bin"), mask = slurp<uint32_t>(prefix + "_mask.bin");
  const auto coords = slurp<double>(prefix + "_coords.bin"), s = slurp<double>(prefix + "_src.bin");
  bp5_mesh_view mv{};
  mv.degree = fe_degree; mv.n_cells = (uint32_t)(l2g.size() / n3); mv.n_interior_cells = mv.n_cells;
  mv.n_owned = (uint32_t)(coords.size() / 3); mv.n_ghost = 0; mv.n_global_dofs = mv.n_owned;
  mv.local_to_global_host = l2g.data(); mv.node_coords_host = coords.data();
  mv.constrained_host = constrained.data(); mv.n_constrained = (uint32_t)constrained.size();
  if (mask.size() != mv.n_cells || s.size() != mv.n_owned) throw std::runtime_error("hanging: array sizes do not match");
  PoissonOperator<dim, fe_degree> fast(mv, BP5_QUAD_GAUSS, false, mask.data()), generic(mv, BP5_QUAD_GAUSS, true, mask.data());
  const size_t n = mv.n_owned;
  double *src, *d1, *d2, *d3;
  fast.initialize_dof_vector(&src); fast.initialize_dof_vector(&d1); fast.initialize_dof_vector(&d2); fast.initialize_dof_vector(&d3);
  check(bp5_copy_h2d(src, s.data(), n * sizeof(double)));
  fast.vmult(d1, src);
  generic.vmult(d2, src);
  generic.vmult_unmerged(d3, src);
  check(bp5_mf_sync(fast.handle()));
  check(bp5_mf_sync(generic.handle()));
  const auto h1 = download(d1, n), h2 = download(d2, n), h3 = download(d3, n);
  printf("hanging p=%d cells=%u dofs=%zu\nfunctor_vs_library %.3e\nunmerged_functor_vs_library %.3e\n", fe_degree, mv.n_cells, n, rel_diff(h2, h1), rel_diff(h3, h1));
  dump(prefix + "_out_library.bin", h1);
  dump(prefix + "_out_functor.bin", h2);
  dump(prefix + "_out_functor_unmerged.bin", h3);
  return 0;
}

int main(int argc, char **argv)
{
  try {
    if (argc >= 8 && !strcmp(argv[1], "check")) {
      const int p = atoi(argv[2]);
      const uint32_t nx = atoi(argv[3]), ny = atoi(argv[4]), nz = atoi(argv[5]);
      const double deform = atof(argv[6]);
      switch (p) {
        case 2: return run_check<2>(nx, ny, nz, deform, argv[7]);
        case 3: return run_check<3>(nx, ny, nz, deform, argv[7]);
        case 4: return run_check<4>(nx, ny, nz, deform, argv[7]);
      }
    } else if (argc >= 4 && !strcmp(argv[1], "hanging")) {
      switch (atoi(argv[2])) {
        case 2: return run_hanging<2>(argv[3]);
        case 3: return run_hanging<3>(argv[3]);
      }
    } else if (argc >= 5 && !strcmp(argv[1], "helmholtz_native")) {
      switch (atoi(argv[2])) {
        case 2: return run_helmholtz<2, true>(atoi(argv[3]), argv[4]);
        case 3: return run_helmholtz<3, true>(atoi(argv[3]), argv[4]);
        case 4: return run_helmholtz<4, true>(atoi(argv[3]), argv[4]);
      }
    } else if (argc >= 5 && !strcmp(argv[1], "helmholtz")) {
      switch (atoi(argv[2])) {
        case 2: return run_helmholtz<2>(atoi(argv[3]), argv[4]);
        case 3: return run_helmholtz<3>(atoi(argv[3]), argv[4]);
        case 4: return run_helmholtz<4>(atoi(argv[3]), argv[4]);
      }
    } else if (argc >= 6 && !strcmp(argv[1], "bench")) {
      const int p = atoi(argv[2]);
      switch (p) {
        case 4: return run_bench<4>(atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
        case 5: return run_bench<5>(atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
      }
    }
    else if (argc >= 7 && !strcmp(argv[1], "run")) { // the reference's main(): run <degree> cycle_min cycle_max iterations repetitions [min_dofs [max_dofs]]
      const int c0 = atoi(argv[3]), c1 = atoi(argv[4]), it = atoi(argv[5]), rep = atoi(argv[6]);
      const unsigned long long lo = argc >= 8 ? strtoull(argv[7], nullptr, 10) : 0ull, hi = argc >= 9 ? strtoull(argv[8], nullptr, 10) : ~0ull;
      switch (atoi(argv[2])) {
        case 4: return run_cycles<4>(c0, c1, it, rep, lo, hi);
        case 5: return run_cycles<5>(c0, c1, it, rep, lo, hi);
      }
    }
    fprintf(stderr, "usage: %s check <2|3|4> nx ny nz deform prefix | bench <4|5> n iterations repetitions | helmholtz <2|3|4> n prefix | "
                    "run <4|5> cycle_min cycle_max iterations repetitions [min_dofs [max_dofs]]\n", argv[0]);
    return 2;
  } catch (const std::exception &e) {
    // same shape as the reference's top-level handler, bp5/step-64.cu:735-759
    fprintf(stderr, "\n----------------------------------------------------\nException on processing:\n%s\nAborting!\n", e.what());
    return 1;
  }
}
