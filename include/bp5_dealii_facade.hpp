// bp5_dealii_facade.hpp -- header-only C++ facade over the C ABI (bp5.h) that re-creates the
// deal.II names the reference's hot path is written against, so that a device functor shaped like
// the reference's LocalPoissonOperator / JacobianFunctor / HelmholtzOperatorQuad
// (bp5/step-64.cu:60-194, step-64/step-64.cu:69-219) compiles under hipcc unchanged in structure:
//
//   CUDAWrappers::MatrixFree<dim,Number>::{AdditionalData, Data, reinit, cell_loop,
//        evaluate_coefficients, copy_constrained_values, set_constrained_values,
//        initialize_dof_vector, get_data}
//   CUDAWrappers::FEEvaluation<dim,fe_degree,n_q_points_1d,1,Number>
//        (read_dof_values, distribute_local_to_global, evaluate, integrate, get_value,
//         submit_value, get_gradient, submit_gradient, apply_quad_point_operations; both the
//         q-point-argument generation of the API, step-64/step-64.cu:154-160, and the
//         no-argument one, bp5/step-64.cu:190)
//   CUDAWrappers::SharedData, q_point_id_in_cell, local_q_point_id, get_quadrature_point,
//   internal::compute_index, block_size, chunk_size, Tensor<1,dim,Number>,
//   SolverControl / IterationNumberControl / SolverCG / SolverCGFullMerge / DiagonalMatrix,
//   LinearAlgebra::distributed::Vector<Number, MemorySpace::CUDA> (the part of its surface the path uses:
//        reinit, = scalar, all_zero, add, equ, sadd, l2_norm, get_values, local_size, size, import,
//        update_ghost_values, compress(add), zero_out_ghosts) and LinearAlgebra::ReadWriteVector.
//
// This generic path is the functional twin of the reference's apply_kernel_shmem (one thread per
// local DoF = per q-point, values + gradients[dim] in shared memory, FP64 atomics): it exists for
// user-defined physics.  The BP5 operator itself should go through bp5_apply (fused kernels).
// deal.II mesh ingestion (mapping / dof_handler / constraints objects) is NOT re-created: reinit
// takes the flat arrays of bp5_mf_desc, which is exactly what deal.II's reinit extracts.
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "bp5.h"

namespace bp5 {
namespace dealii_facade {

inline void check(int s)
{ // the reference throws (AssertThrow / AssertCuda); the C ABI returns codes
  if (s != BP5_OK) throw std::runtime_error(std::string(bp5_strerror(s)) + ": " + bp5_last_error());
}

template <int rank, int dim, typename Number = double>
struct Tensor; // only rank 1 is needed on this path
template <int dim, typename Number>
struct Tensor<1, dim, Number> {
  Number v[dim];
  __host__ __device__ Number &operator[](int i) { return v[i]; }
  __host__ __device__ const Number &operator[](int i) const { return v[i]; }
};

namespace Utilities {
template <typename T>
constexpr T pow(T b, int e) { return e == 0 ? T(1) : b * pow(b, e - 1); }
} // namespace Utilities

namespace CUDAWrappers {

constexpr unsigned int block_size = 512; // bp5/solver.h:61
constexpr unsigned int chunk_size = 1;   // bp5/solver.h:61
constexpr int max_n1d = BP5_MAX_DEGREE + 1;

// 1-D tables, dof-major like deal.II's global_shape_values: [i * n_q + q]
static __constant__ double global_shape_values[max_n1d * max_n1d];
static __constant__ double global_shape_gradients[max_n1d * max_n1d];
// hanging nodes: I[h][a][b] = phi_b(xi_a / 2 + h / 2), the coarse 1-D basis at the nodes of the lower / upper half
static __constant__ double global_hanging_interpolation[2 * max_n1d * max_n1d];

namespace internal {
// resolve_hanging_nodes<dim, fe_degree, transpose>(constraint_mask, values) -- the call of bp5/fe_evaluation_gl.h:150-151,167-168
// (upstream code in deal.II); mask bits: bp5.h BP5_HANG_*.  One thread per local DoF, `values` in shared memory, whole block calls.
template <int dim, int fe_degree, bool transpose, typename Number>
__device__ inline void resolve_hanging_nodes(const unsigned int constraint_mask, Number *values)
{
  static_assert(dim == 3, "dim == 3");
  constexpr unsigned int any = BP5_HANG_FACE_X | BP5_HANG_FACE_Y | BP5_HANG_FACE_Z | BP5_HANG_EDGE_X | BP5_HANG_EDGE_Y | BP5_HANG_EDGE_Z;
  if (!(constraint_mask & any)) return; // block-uniform
  constexpr int n = fe_degree + 1;
  const int idx[3] = {(int)(threadIdx.x % n), (int)threadIdx.y, (int)threadIdx.z};
  const int q = idx[0] + n * (idx[1] + n * idx[2]);
  bool at_side[3]; // this entry lies on the side of the cell that the mask's position bits name
  for (int e = 0; e < 3; ++e) at_side[e] = idx[e] == (int)((constraint_mask >> (3 + e)) & 1u) * (n - 1);
  for (int t = 0; t < 3; ++t) { // one sweep per direction: the lines along t on constrained faces tangential to t, and the constrained edge along t
    const int e1 = t == 0 ? 1 : 0, e2 = t == 2 ? 1 : 2;
    const bool face1 = (constraint_mask >> e1) & 1u, face2 = (constraint_mask >> e2) & 1u, edge = (constraint_mask >> (9 + t)) & 1u;
    if (!(face1 || face2 || edge)) continue; // block-uniform
    const bool on = (face1 && at_side[e1]) || (face2 && at_side[e2]) || (edge && at_side[e1] && at_side[e2]);
    const double *M = global_hanging_interpolation + ((constraint_mask >> (6 + t)) & 1u) * n * n;
    Number acc = values[q];
    if (on) {
      acc = 0;
      for (int r = 0; r < n; ++r) {
        int e[3] = {idx[0], idx[1], idx[2]};
        e[t] = r;
        acc += (transpose ? M[r * n + idx[t]] : M[idx[t] * n + r]) * values[e[0] + n * (e[1] + n * e[2])];
      }
    }
    __syncthreads();
    values[q] = acc;
    __syncthreads();
  }
}
} // namespace internal

template <int dim, typename Number>
struct SharedData { // [upstream] C5
  Number *values;
  Number *gradients[dim];
};

namespace internal {
template <int dim, int n>
__device__ inline unsigned int compute_index()
{ // [upstream] C7
  return dim == 1 ? threadIdx.x % n : dim == 2 ? threadIdx.x % n + n * threadIdx.y : threadIdx.x % n + n * (threadIdx.y + n * threadIdx.z);
}
} // namespace internal

template <int dim>
__device__ inline unsigned int q_point_id_in_cell(const unsigned int n_q_points_1d)
{
  return dim == 1 ? threadIdx.x % n_q_points_1d
                  : dim == 2 ? threadIdx.x % n_q_points_1d + n_q_points_1d * threadIdx.y
                             : threadIdx.x % n_q_points_1d + n_q_points_1d * (threadIdx.y + n_q_points_1d * threadIdx.z);
}

template <int dim, typename Number = double>
class MatrixFree {
public:
  struct AdditionalData { // [upstream] C1
    enum ParallelizationScheme { parallel_in_elem, parallel_over_elem };
    ParallelizationScheme parallelization_scheme = parallel_in_elem;
    unsigned int mapping_update_flags = 0;
    bool use_coloring = false;
    bool overlap_communication_computation = false;
  };
  struct Data { // bp5/fe_evaluation_gl.h:112-120, bp5/step-64.cu:94-97
    Number *q_points;               // [dim][n_cells*padding_length]
    unsigned int *local_to_global;  // [n_cells*padding_length]
    Number *inv_jacobian;           // [dim*dim][n_cells*padding_length]
    Number *JxW;                    // [n_cells*padding_length]
    unsigned int n_cells, padding_length, row_start;
    unsigned int *constraint_mask;
    bool use_coloring;
  };

  ~MatrixFree() { if (mf) bp5_mf_destroy(mf); }

  // == reinit(mapping, dof_handler, constraints, quad, additional_data), bp5/step-64.cu:248:
  //    the flat arrays deal.II would extract are passed in directly
  void reinit(const bp5_mf_desc &desc, const AdditionalData & = AdditionalData())
  {
    static_assert(dim == 3, "this build covers dim == 3");
    if (mf) { bp5_mf_destroy(mf); mf = nullptr; }
    check(bp5_mf_create(&desc, &mf));
    degree = desc.degree;
    n_owned = desc.n_owned;
    n_ghost = desc.n_ghost;
    n_cells = desc.n_cells;
    stream = (hipStream_t)desc.stream;
    const int n = degree + 1;
    std::vector<double> N(n * n), D(n * n), sv(n * n), sg(n * n);
    check(bp5_shape_tables(degree, desc.quadrature, nullptr, nullptr, nullptr, N.data(), D.data()));
    for (int q = 0; q < n; ++q)
      for (int i = 0; i < n; ++i) { sv[i * n + q] = N[q * n + i]; sg[i * n + q] = D[q * n + i]; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(global_shape_values), sv.data(), n * n * sizeof(double)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(global_shape_gradients), sg.data(), n * n * sizeof(double)) != hipSuccess)
      throw std::runtime_error("hipMemcpyToSymbol failed");
    { // hanging-node interpolation matrices from the FE_Q nodes
      std::vector<double> nodes(n), I(2 * n * n);
      check(bp5_shape_tables(degree, desc.quadrature, nodes.data(), nullptr, nullptr, nullptr, nullptr));
      for (int h = 0; h < 2; ++h)
        for (int a = 0; a < n; ++a)
          for (int b = 0; b < n; ++b) {
            long double num = 1, den = 1;
            const long double x = 0.5L * nodes[a] + 0.5L * h;
            for (int c = 0; c < n; ++c) if (c != b) { num *= x - (long double)nodes[c]; den *= (long double)nodes[b] - (long double)nodes[c]; }
            I[(h * n + a) * n + b] = (double)(num / den);
          }
      if (hipMemcpyToSymbol(HIP_SYMBOL(global_hanging_interpolation), I.data(), 2 * n * n * sizeof(double)) != hipSuccess)
        throw std::runtime_error("hipMemcpyToSymbol failed");
    }
    bp5_mf_data d;
    check(bp5_mf_get_data(mf, 0, &d));
    data.q_points = const_cast<Number *>(d.q_points);
    data.local_to_global = const_cast<unsigned int *>(d.local_to_global);
    data.inv_jacobian = const_cast<Number *>(d.inv_jacobian);
    data.JxW = const_cast<Number *>(d.JxW);
    data.n_cells = d.n_cells; data.padding_length = d.padding_length; data.row_start = d.row_start;
    data.constraint_mask = const_cast<unsigned int *>(d.constraint_mask);
    data.use_coloring = d.use_coloring != 0;
  }

  Data get_data(unsigned int /*color*/ = 0) const { return data; }
  bp5_mf *handle() const { return mf; }
  unsigned int n_local() const { return n_owned + n_ghost; }

  // launch geometry of [upstream] C4
  static unsigned int cells_per_block_shmem(int fe_degree) { return dim == 3 ? (fe_degree == 1 ? 8 : fe_degree == 2 ? 2 : 1) : 1; }

  template <typename Functor>
  void cell_loop(const Functor &func, const Number *src, Number *dst) const;
  template <typename Functor>
  void evaluate_coefficients(Functor func) const;

  void copy_constrained_values(const Number *src, Number *dst) const { check(bp5_copy_constrained(mf, src, dst)); }
  void set_constrained_values(Number val, Number *dst) const { check(bp5_set_constrained(mf, val, dst)); }
  // == initialize_dof_vector(vec): owned + ghost storage
  void initialize_dof_vector(Number **vec) const { check(bp5_vec_alloc(n_local(), vec)); }
  template <typename VectorType>
  void initialize_dof_vector(VectorType &vec) const { vec.reinit(mf, n_owned, n_ghost); }

private:
  bp5_mf *mf = nullptr;
  Data data{};
  int degree = 0;
  unsigned int n_owned = 0, n_ghost = 0, n_cells = 0;
  hipStream_t stream = nullptr;
};

template <int dim, typename Number>
__device__ inline unsigned int local_q_point_id(const unsigned int cell, const typename MatrixFree<dim, Number>::Data *data,
                                                const unsigned int n_q_points_1d, const unsigned int n_q_points)
{
  return data->row_start + cell * n_q_points + q_point_id_in_cell<dim>(n_q_points_1d);
}
template <int dim, typename Number>
__device__ inline Tensor<1, dim, Number> get_quadrature_point(const unsigned int cell, const typename MatrixFree<dim, Number>::Data *data,
                                                              const unsigned int n_q_points_1d)
{
  Tensor<1, dim, Number> p;
  const unsigned int at = data->padding_length * cell + q_point_id_in_cell<dim>(n_q_points_1d);
  for (int d = 0; d < dim; ++d) p[d] = data->q_points[(size_t)d * data->n_cells * data->padding_length + at];
  return p;
}

// ---- kernels ([upstream] apply_kernel_shmem / evaluate_coeff)
template <int dim, typename Number, typename Functor>
__global__ void apply_kernel_shmem(Functor func, const typename MatrixFree<dim, Number>::Data gpu_data, const Number *src, Number *dst)
{
  constexpr unsigned int n3 = Functor::n_local_dofs;
  constexpr unsigned int n1 = Functor::n_dofs_1d;
  const unsigned int cells_per_block = blockDim.x / n1;
  extern __shared__ double facade_smem[];
  Number *values = facade_smem;
  Number *gradients = facade_smem + cells_per_block * n3;
  const unsigned int local_cell = threadIdx.x / n1;
  const unsigned int cell = local_cell + cells_per_block * (blockIdx.x + gridDim.x * blockIdx.y);
  SharedData<dim, Number> shared_data;
  shared_data.values = values + local_cell * n3;
  for (int d = 0; d < dim; ++d) shared_data.gradients[d] = gradients + (size_t)d * cells_per_block * n3 + local_cell * n3;
  // all threads of a block must reach the functor's barriers: a tail block repeats its last cell
  // with the scatter disabled through the cell index
  if (cell < gpu_data.n_cells) func(cell, &gpu_data, &shared_data, src, dst);
}
template <int dim, typename Number, typename Functor>
__global__ void evaluate_coeff(Functor func, const typename MatrixFree<dim, Number>::Data gpu_data)
{
  constexpr unsigned int n1 = Functor::n_dofs_1d;
  const unsigned int cells_per_block = blockDim.x / n1;
  const unsigned int cell = threadIdx.x / n1 + cells_per_block * (blockIdx.x + gridDim.x * blockIdx.y);
  if (cell < gpu_data.n_cells) func(cell, &gpu_data);
}

template <int dim, typename Number>
template <typename Functor>
void MatrixFree<dim, Number>::cell_loop(const Functor &func, const Number *src, Number *dst) const
{
  constexpr unsigned int n1 = Functor::n_dofs_1d, n3 = Functor::n_local_dofs;
  // cells_per_block must divide n_cells here (a partial block would leave barriers unmatched)
  unsigned int cpb = cells_per_block_shmem(n1 - 1);
  while (n_cells % cpb) cpb /= 2;
  const unsigned int n_blocks = n_cells / cpb;
  const dim3 block(n1 * cpb, n1, n1), grid(n_blocks, 1);
  const size_t smem = (size_t)(1 + dim) * cpb * n3 * sizeof(Number);
  hipLaunchKernelGGL((apply_kernel_shmem<dim, Number, Functor>), grid, block, smem, stream, func, data, src, dst);
  if (hipGetLastError() != hipSuccess) throw std::runtime_error("apply_kernel_shmem launch failed");
}
template <int dim, typename Number>
template <typename Functor>
void MatrixFree<dim, Number>::evaluate_coefficients(Functor func) const
{
  constexpr unsigned int n1 = Functor::n_dofs_1d;
  unsigned int cpb = cells_per_block_shmem(n1 - 1);
  while (n_cells % cpb) cpb /= 2;
  const dim3 block(n1 * cpb, n1, n1), grid(n_cells / cpb, 1);
  hipLaunchKernelGGL((evaluate_coeff<dim, Number, Functor>), grid, block, 0, stream, func, data);
  if (hipGetLastError() != hipSuccess) throw std::runtime_error("evaluate_coeff launch failed");
}

// ---- device-side cell evaluator (bp5/fe_evaluation_gl.h is a clone of the upstream class)
template <int dim, int fe_degree, int n_q_points_1d = fe_degree + 1, int n_components_ = 1, typename Number = double>
class FEEvaluation {
public:
  using value_type = Number;
  using gradient_type = Tensor<1, dim, Number>;
  using data_type = typename MatrixFree<dim, Number>::Data;
  static constexpr unsigned int dimension = dim;
  static constexpr unsigned int n_components = n_components_;
  static constexpr unsigned int n_q_points = Utilities::pow(n_q_points_1d, dim);
  static constexpr unsigned int tensor_dofs_per_cell = Utilities::pow(fe_degree + 1, dim);
  static_assert(dim == 3 && n_components_ == 1 && n_q_points_1d == fe_degree + 1, "facade covers dim 3, scalar, n_q = p+1");

  __device__ FEEvaluation(const unsigned int cell_id, const data_type *data, SharedData<dim, Number> *shdata)
    : n_cells(data->n_cells), padding_length(data->padding_length), constraint_mask(data->constraint_mask[cell_id]),
      use_coloring(data->use_coloring), values(shdata->values)
  {
    local_to_global = data->local_to_global + padding_length * cell_id;
    inv_jac = data->inv_jacobian + padding_length * cell_id;
    JxW = data->JxW + padding_length * cell_id;
    for (unsigned int i = 0; i < dim; ++i) gradients[i] = shdata->gradients[i];
  }

  __device__ void read_dof_values(const Number *src)
  { // bp5/fe_evaluation_gl.h:128-152
    const unsigned int idx = internal::compute_index<dim, n_q_points_1d>();
    values[idx] = src[local_to_global[idx]];
    __syncthreads();
    internal::resolve_hanging_nodes<dim, fe_degree, false>(constraint_mask, values);
  }
  __device__ void distribute_local_to_global(Number *dst) const
  { // bp5/fe_evaluation_gl.h:156-181
    internal::resolve_hanging_nodes<dim, fe_degree, true>(constraint_mask, values);
    const unsigned int idx = internal::compute_index<dim, n_q_points_1d>();
    if (use_coloring) dst[local_to_global[idx]] += values[idx];
    else __hip_atomic_fetch_add(dst + local_to_global[idx], values[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  __device__ void evaluate(const bool evaluate_val, const bool evaluate_grad)
  { // [upstream] C6
    if (evaluate_grad) {
      // x: g0 = D u, g1 = N u, g2 = N u
      const Number u_dx = contract<0, true, false>(global_shape_gradients, values);
      const Number u_nx = contract<0, true, false>(global_shape_values, values);
      gradients[0][index()] = u_dx; gradients[1][index()] = u_nx; gradients[2][index()] = u_nx;
      __syncthreads();
      inplace<1, true>(global_shape_values, gradients[0]);
      inplace<1, true>(global_shape_gradients, gradients[1]);
      inplace<1, true>(global_shape_values, gradients[2]);
      inplace<2, true>(global_shape_values, gradients[0]);
      inplace<2, true>(global_shape_values, gradients[1]);
      inplace<2, true>(global_shape_gradients, gradients[2]);
    }
    if (evaluate_val) {
      inplace<0, true>(global_shape_values, values);
      inplace<1, true>(global_shape_values, values);
      inplace<2, true>(global_shape_values, values);
    }
  }
  __device__ void integrate(const bool integrate_val, const bool integrate_grad)
  {
    if (integrate_val) {
      inplace<0, false>(global_shape_values, values);
      inplace<1, false>(global_shape_values, values);
      inplace<2, false>(global_shape_values, values);
    }
    if (integrate_grad) {
      inplace<2, false>(global_shape_values, gradients[0]);
      inplace<2, false>(global_shape_values, gradients[1]);
      inplace<2, false>(global_shape_gradients, gradients[2]);
      inplace<1, false>(global_shape_values, gradients[0]);
      inplace<1, false>(global_shape_gradients, gradients[1]);
      inplace<1, false>(global_shape_values, gradients[2]);
      const Number s = contract<0, false, false>(global_shape_gradients, gradients[0]) + contract<0, false, false>(global_shape_values, gradients[1]) +
                       contract<0, false, false>(global_shape_values, gradients[2]);
      __syncthreads();
      if (integrate_val) values[index()] += s; else values[index()] = s;
      __syncthreads();
    }
  }

  __device__ value_type get_value(const unsigned int q_point) const { return values[q_point]; }
  __device__ value_type get_value() const { return values[index()]; }
  __device__ value_type get_dof_value(const unsigned int dof) const { return values[dof]; }
  __device__ void submit_value(const value_type &val_in, const unsigned int q_point) { values[q_point] = val_in * JxW[q_point]; }
  __device__ void submit_value(const value_type &val_in) { submit_value(val_in, index()); }
  __device__ void submit_dof_value(const value_type &val_in, const unsigned int dof) { values[dof] = val_in; }

  // physical gradient at a quadrature point from the reference-space one: grad_x u = K^T ghat, K[r][c] = d xi_r / d x_c
  // (the nine SoA planes of MatrixFree::Data; semantics of bp5/fe_evaluation_gl.h:318-343)
  __device__ gradient_type get_gradient(const unsigned int q_point) const
  {
    const Number gh[3] = {gradients[0][q_point], gradients[1][q_point], gradients[2][q_point]};
    gradient_type g;
    for (int c = 0; c < dim; ++c) g[c] = K(0, c, q_point) * gh[0] + K(1, c, q_point) * gh[1] + K(2, c, q_point) * gh[2];
    return g;
  }
  __device__ gradient_type get_gradient() const { return get_gradient(index()); }
  // test-function side: what integrate() contracts with is JxW K grad (semantics of bp5/fe_evaluation_gl.h:354-369)
  __device__ void submit_gradient(const gradient_type &grad_in, const unsigned int q_point)
  {
    const Number w = JxW[q_point];
    for (int r = 0; r < dim; ++r) gradients[r][q_point] = w * (K(r, 0, q_point) * grad_in[0] + K(r, 1, q_point) * grad_in[1] + K(r, 2, q_point) * grad_in[2]);
  }
  __device__ void submit_gradient(const gradient_type &grad_in)
  {
    submit_gradient(grad_in, index());
    __syncthreads();
  }
  template <typename Functor>
  __device__ void apply_quad_point_operations(const Functor &func)
  { // bp5/fe_evaluation_gl.h:381-393
    func(this, index());
    __syncthreads();
  }

private:
  static __device__ unsigned int index() { return internal::compute_index<dim, n_q_points_1d>(); }
  // entry (r, c) of the inverse Jacobian at this cell's quadrature point q: plane r * dim + c of inv_jacobian
  __device__ Number K(int r, int c, unsigned int q) const { return inv_jac[(size_t)(r * dim + c) * n_cells * padding_length + q]; }

  // one 1-D contraction along `direction` evaluated at this thread's point; shape is dof-major
  // [i * n + q]; dof_to_quad selects shape[k * n + q] (evaluation) or shape[q * n + k] (integration)
  template <int direction, bool dof_to_quad, bool /*add*/>
  __device__ Number contract(const double *shape, const Number *in) const
  {
    constexpr int n = n_q_points_1d;
    const int i = threadIdx.x % n, j = threadIdx.y, k = threadIdx.z;
    const int q = direction == 0 ? i : direction == 1 ? j : k;
    Number t = 0;
    for (int m = 0; m < n; ++m) {
      const int src = direction == 0 ? m + n * (j + n * k) : direction == 1 ? i + n * (m + n * k) : i + n * (j + n * m);
      t += shape[dof_to_quad ? m * n + q : q * n + m] * in[src];
    }
    return t;
  }
  template <int direction, bool dof_to_quad>
  __device__ void inplace(const double *shape, Number *array) const
  {
    const Number t = contract<direction, dof_to_quad, false>(shape, array);
    __syncthreads();
    array[index()] = t;
    __syncthreads();
  }

  unsigned int *local_to_global;
  unsigned int n_cells;
  unsigned int padding_length;
  const unsigned int constraint_mask;
  const bool use_coloring;
  Number *inv_jac;
  Number *JxW;
  Number *values;
  Number *gradients[dim];
};

} // namespace CUDAWrappers

// ---- device vector: the part of LinearAlgebra::distributed::Vector<Number, MemorySpace::CUDA> the reference's
//      path uses (bp5/solver.h:369-382,417-421,511,528; bp5/step-64.cu:349,366-367,431-432,445,449,467).
//      Storage: owned entries then ghost entries, contiguous (the deal.II layout); global reductions and the
//      halo calls go through the MatrixFree handle the vector was initialised from.
namespace MemorySpace {
struct Host {};
struct CUDA {};
} // namespace MemorySpace
namespace VectorOperation {
enum values { unknown, insert, add };
}
namespace LinearAlgebra {
template <typename Number>
class ReadWriteVector { // host staging vector for Vector::import (bp5/step-64.cu:413-416)
public:
  ReadWriteVector() = default;
  explicit ReadWriteVector(size_t n) : v(n, Number(0)) {}
  void reinit(size_t n) { v.assign(n, Number(0)); }
  size_t size() const { return v.size(); }
  Number &operator[](size_t i) { return v[i]; }
  const Number &operator[](size_t i) const { return v[i]; }
  Number *data() { return v.data(); }
  const Number *data() const { return v.data(); }

private:
  std::vector<Number> v;
};
namespace distributed {
template <typename Number, typename MemorySpaceType = MemorySpace::CUDA>
class Vector {
  static_assert(sizeof(Number) == sizeof(double), "the library computes in FP64");

public:
  using value_type = Number;
  using size_type = size_t;
  Vector() = default;
  Vector(const Vector &) = delete;
  Vector &operator=(const Vector &) = delete;
  Vector(Vector &&o) noexcept : mf(o.mf), val(o.val), n_owned(o.n_owned), n_ghost(o.n_ghost), owns(o.owns) { o.val = nullptr; }
  ~Vector() { if (val && owns) bp5_vec_free(val); }
  // non-owning view of n_owned + n_ghost device doubles laid out like the vectors of `handle` (the solvers hand their work
  // vectors to a user operator's vmult(Vector &, const Vector &) this way)
  static Vector view(bp5_mf *handle, Number *values, size_t n_owned_, size_t n_ghost_)
  {
    Vector v;
    v.mf = handle; v.val = values; v.n_owned = n_owned_; v.n_ghost = n_ghost_; v.owns = false;
    return v;
  }
  bp5_mf *handle() const { return mf; }

  // == reinit(locally_owned, ghost, comm): the index sets are those the handle was created with
  void reinit(bp5_mf *handle, size_t n_owned_, size_t n_ghost_)
  {
    if (val && owns) bp5_vec_free(val);
    val = nullptr; owns = true;
    mf = handle; n_owned = n_owned_; n_ghost = n_ghost_;
    check(bp5_vec_alloc(n_owned + n_ghost, &val)); // zero-filled
  }
  // == reinit(v, omit_zeroing_entries)
  void reinit(const Vector &v, bool omit_zeroing_entries = false)
  {
    if (!val || n_owned != v.n_owned || n_ghost != v.n_ghost || mf != v.mf) reinit(v.mf, v.n_owned, v.n_ghost);
    else if (!omit_zeroing_entries) *this = Number(0);
  }
  Vector &operator=(Number s)
  {
    check(bp5_vec_fill(mf, val, s, n_owned + n_ghost));
    return *this;
  }
  bool all_zero() const
  {
    int z = 0;
    check(bp5_vec_all_zero(mf, val, n_owned, &z));
    return z != 0;
  }
  void add(Number a, const Vector &v) { check(bp5_vec_axpy(mf, val, a, v.val, n_owned)); }
  void equ(Number a, const Vector &v) { check(bp5_vec_equ(mf, val, a, v.val, n_owned)); }
  void sadd(Number s, Number a, const Vector &v) { check(bp5_vec_sadd(mf, val, s, a, v.val, n_owned)); }
  Number l2_norm() const
  {
    double r = 0;
    check(bp5_vec_l2_norm(mf, val, n_owned, &r));
    return r;
  }
  Number *get_values() const { return val; }
  size_type local_size() const { return n_owned; }
  size_type n_ghost_entries() const { return n_ghost; }
  // global size: sum of the owned ranges (one tiny all-reduce, cached)
  size_type size() const
  {
    if (!n_global) {
      double *tmp = nullptr, h = (double)n_owned;
      check(bp5_vec_alloc(2, &tmp));
      check(bp5_copy_h2d(tmp, &h, sizeof(double)));
      check(bp5_comm_allreduce_sum(mf, tmp, 1));
      check(bp5_mf_sync(mf));
      check(bp5_copy_d2h(&h, tmp, sizeof(double)));
      bp5_vec_free(tmp);
      n_global = (size_type)(h + 0.5);
    }
    return n_global;
  }
  // == import(ReadWriteVector, VectorOperation::insert): host values of the owned range
  void import(const ReadWriteVector<Number> &rw, VectorOperation::values op = VectorOperation::insert)
  {
    if (op != VectorOperation::insert || rw.size() != n_owned) throw std::runtime_error("Vector::import: insert of the owned range only");
    check(bp5_copy_h2d(val, rw.data(), n_owned * sizeof(Number)));
  }
  void extract(ReadWriteVector<Number> &rw) const
  {
    rw.reinit(n_owned);
    check(bp5_copy_d2h(rw.data(), val, n_owned * sizeof(Number)));
  }
  void update_ghost_values() const { check(bp5_halo_gather(mf, val)); }
  void update_ghost_values_start() const { check(bp5_halo_gather_start(mf, val)); }
  void update_ghost_values_finish() const { check(bp5_halo_gather_finish(mf, val)); }
  void compress_start(VectorOperation::values op)
  {
    if (op != VectorOperation::add) throw std::runtime_error("Vector::compress_start: add only");
    check(bp5_halo_scatter_add_start(mf, val));
  }
  void compress_finish(VectorOperation::values) { check(bp5_halo_scatter_add_finish(mf, val)); }
  void compress(VectorOperation::values op)
  {
    if (op != VectorOperation::add) throw std::runtime_error("Vector::compress: add only");
    check(bp5_halo_scatter_add(mf, val));
  }
  void zero_out_ghosts() const { check(bp5_halo_zero_ghosts(mf, val)); }

private:
  bp5_mf *mf = nullptr;
  Number *val = nullptr;
  size_t n_owned = 0, n_ghost = 0;
  bool owns = true;
  mutable size_type n_global = 0;
};
} // namespace distributed
} // namespace LinearAlgebra

// ---- solver-side names (host): thin wrappers over bp5_cg_solve
class SolverControl {
public:
  SolverControl(unsigned int n = 100, double tol = 1e-10) : max_steps(n), tolerance(tol) {}
  unsigned int last_step() const { return lstep; }
  double last_value() const { return lvalue; }
  unsigned int max_steps;
  double tolerance;
  unsigned int lstep = 0;
  double lvalue = 0;
};
class IterationNumberControl : public SolverControl { // bp5/step-64.cu:443-445
public:
  using SolverControl::SolverControl;
};
struct DiagonalMatrix { // bp5/step-64.cu:428-432; nullptr == identity
  const double *diag = nullptr;
  const double *get_vector() const { return diag; }
};
namespace internal {
template <typename T, typename = void> struct has_coef : std::false_type {};
template <typename T> struct has_coef<T, decltype((void)std::declval<const T &>().coef())> : std::true_type {};
} // namespace internal
// cg.solve(A, x, b, preconditioner), bp5/solver.h:25-30: the solvers use nothing of A but A.vmult(dst, src).
//  * an operator that exposes  bp5_mf* handle()  and  const double* coef()  is the library's own Poisson operator: the
//    whole solve runs inside bp5_cg_solve (fused operator kernels, dot products inside the block kernel where possible);
//  * ANY other MatrixType with  vmult(VectorType &dst, const VectorType &src)  (e.g. the step-64 Helmholtz operator written
//    as a device functor, examples/bp5_step64.hip) goes through bp5_cg_solve_operator: same solver kernels, A.vmult called
//    once per iteration on non-owning views of the solver's work vectors.
template <int VARIANT>
class SolverCGBase {
public:
  explicit SolverCGBase(SolverControl &cn) : control(cn) {}
  template <typename MatrixType>
  void solve(const MatrixType &A, double *x, const double *b, const DiagonalMatrix &preconditioner)
  {
    static_assert(internal::has_coef<MatrixType>::value, "raw-pointer solve: the library's own operator; use the Vector overload for others");
    bp5_cg_params prm{VARIANT, (int)control.max_steps, control.tolerance, 0, 0};
    bp5_cg_result res{};
    const int s = bp5_cg_solve(A.handle(), A.coef(), preconditioner.get_vector(), b, x, &prm, &res);
    finish(res, s);
  }
  // VectorType = LinearAlgebra::distributed::Vector<double, MemorySpace::CUDA> (bp5/step-64.cu:450-453)
  template <typename MatrixType, typename VectorType>
  auto solve(const MatrixType &A, VectorType &x, const VectorType &b, const DiagonalMatrix &preconditioner) -> decltype((void)x.get_values())
  {
    if constexpr (internal::has_coef<MatrixType>::value)
      solve(A, x.get_values(), static_cast<const double *>(b.get_values()), preconditioner);
    else {
      struct Ctx { const MatrixType *A; bp5_mf *mf; size_t n_owned, n_ghost; std::string what; } ctx{&A, x.handle(), x.local_size(), x.n_ghost_entries(), {}};
      bp5_vmult_fn tramp = [](void *c, double *dst, double *src) -> int {
        Ctx *q = static_cast<Ctx *>(c);
        try { // no exception may cross the C boundary
          VectorType vd = VectorType::view(q->mf, dst, q->n_owned, q->n_ghost), vs = VectorType::view(q->mf, src, q->n_owned, q->n_ghost);
          q->A->vmult(vd, vs);
          return BP5_OK;
        } catch (const std::exception &e) { q->what = e.what(); return BP5_ERR_INVALID; }
      };
      bp5_cg_params prm{VARIANT, (int)control.max_steps, control.tolerance, 0, 0};
      bp5_cg_result res{};
      const int s = bp5_cg_solve_operator(x.handle(), tramp, &ctx, preconditioner.get_vector(), static_cast<const double *>(b.get_values()), x.get_values(), &prm, &res);
      if (s != BP5_OK && !ctx.what.empty()) throw std::runtime_error("operator vmult failed inside the solver: " + ctx.what);
      finish(res, s);
    }
  }
  bp5_cg_result result{};

private:
  void finish(const bp5_cg_result &res, int status)
  {
    control.lstep = res.iterations;
    control.lvalue = res.residual;
    result = res;
    check(status);
  }
  SolverControl &control;
};
using SolverCG = SolverCGBase<BP5_CG_PLAIN>;           // bp5/step-64.cu:446-453
using SolverCGFullMerge = SolverCGBase<BP5_CG_MERGED>; // bp5/solver.h:16-30

} // namespace dealii_facade
} // namespace bp5
