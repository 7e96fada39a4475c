"""Host-side mirror of the reference's operator/solver interface for the BP5 path, on top of the
C ABI (include/bp5.h).  Names, argument meaning and error behaviour follow the reference:

  MatrixFree ............. CUDAWrappers::MatrixFree<3,double> as used at bp5/step-64.cu:234-275
  PoissonOperator ........ bp5/step-64.cu:198-276 (vmult, initialize_dof_vector, do_zero_out)
  DiagonalMatrix ......... bp5/step-64.cu:428-432 (get_vector)
  IterationNumberControl . bp5/step-64.cu:443-445 (last_step)
  SolverCG ............... deal.II SolverCG, call site bp5/step-64.cu:446-453
  SolverCGFullMerge ...... bp5/solver.h:16-30,343-542 (x-update schedule fixed)

Vectors are torch float64 CUDA tensors of n_owned + n_ghost entries (torch is plumbing for
device memory / streams / the process group only -- no torch op is on the hot path)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BP5Error, CG_MERGED, CG_PLAIN, COEF_ONE, QUAD_GAUSS


def _torch():
    import torch
    return torch


def _ptr(t, n_min=0):
    torch = _torch()
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise BP5Error(1, "expected a contiguous float64 CUDA tensor")
    if t.numel() < n_min:
        raise BP5Error(1, f"vector has {t.numel()} entries, need {n_min}")
    return C.c_void_p(t.data_ptr())


class Communicator:
    """RCCL communicator, one rank per GPU.  The 128-byte unique id is created on rank 0 and
    broadcast by the host (here: torch.distributed, any backend)."""

    def __init__(self, rank=0, n_ranks=1, unique_id=None):
        L = _lib.lib()
        if unique_id is None:
            buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
            _lib.check(L.bp5_comm_unique_id(buf))
            unique_id = buf.raw
        self.rank, self.n_ranks = rank, n_ranks
        self._h = C.c_void_p()
        _lib.check(L.bp5_comm_create(C.c_char_p(unique_id), rank, n_ranks, C.byref(self._h)))

    @classmethod
    def from_torch_distributed(cls):
        import torch.distributed as dist
        rank, n = dist.get_rank(), dist.get_world_size()
        box = [None]
        if rank == 0:
            buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
            _lib.check(_lib.lib().bp5_comm_unique_id(buf))
            box[0] = buf.raw
        dist.broadcast_object_list(box, src=0)
        return cls(rank, n, box[0])

    def close(self):
        if self._h:
            _lib.lib().bp5_comm_destroy(self._h)
            self._h = C.c_void_p()


class MatrixFree:
    """== CUDAWrappers::MatrixFree<3,double>; reinit uploads the flat per-cell arrays."""

    def __init__(self):
        self._h = None
        self.mesh = None

    def reinit(self, mesh, quadrature=QUAD_GAUSS, coefficient=COEF_ONE, device=0, stream=None, comm=None):
        """== mf_data.reinit(mapping, dof_handler, constraints, quad, additional_data),
        bp5/step-64.cu:234-248."""
        torch = _torch()
        L = _lib.lib()
        self.mesh, self.quadrature, self.coefficient, self.device = mesh, quadrature, coefficient, device
        if stream is None:
            stream = torch.cuda.current_stream(device).cuda_stream
        d = _lib.MFDesc()
        d.dim, d.degree, d.quadrature, d.coefficient = 3, mesh.degree, quadrature, coefficient
        d.n_cells, d.n_interior_cells, d.n_owned, d.n_ghost = mesh.n_cells, mesh.n_interior_cells, mesh.n_owned, mesh.n_ghost
        keep = [np.ascontiguousarray(mesh.l2g, dtype=np.uint32), np.ascontiguousarray(mesh.coords, dtype=np.float64),
                np.ascontiguousarray(mesh.constrained, dtype=np.uint32), np.ascontiguousarray(mesh.neighbor_rank, dtype=np.int32),
                np.ascontiguousarray(mesh.send_offsets, dtype=np.uint32), np.ascontiguousarray(mesh.send_indices, dtype=np.uint32),
                np.ascontiguousarray(mesh.recv_offsets, dtype=np.uint32)]
        d.local_to_global_host, d.node_coords_host, d.constrained_host = keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data
        d.n_constrained = keep[2].size
        d.n_neighbors = int(mesh.n_neighbors)
        d.neighbor_rank_host, d.send_offsets_host = keep[3].ctypes.data, keep[4].ctypes.data
        d.send_indices_host, d.recv_offsets_host = keep[5].ctypes.data, keep[6].ctypes.data
        d.device, d.stream = device, stream
        blocks = getattr(mesh, "cell_block_offsets", None)
        if blocks is not None:
            keep.append(np.ascontiguousarray(blocks, dtype=np.uint32))
            d.n_cell_blocks, d.cell_block_offsets_host = keep[-1].size - 1, keep[-1].ctypes.data
        cmask = getattr(mesh, "constraint_mask", None)      # hanging-node masks of 2:1 refined meshes (BP5_HANG_* bits)
        if cmask is not None:
            keep.append(np.ascontiguousarray(cmask, dtype=np.uint32))
            d.constraint_mask_host = keep[-1].ctypes.data
        h = C.c_void_p()
        _lib.check(L.bp5_mf_create(C.byref(d), C.byref(h)))
        self._h = h
        self.n_owned, self.n_ghost, self.n_local = mesh.n_owned, mesh.n_ghost, mesh.n_owned + mesh.n_ghost
        self.comm = comm
        if comm is not None:
            _lib.check(L.bp5_mf_set_comm(h, comm._h))
        return self

    # -- handle plumbing
    @property
    def handle(self):
        if not self._h:
            raise BP5Error(1, "MatrixFree.reinit has not been called")
        return self._h

    def close(self):
        if self._h:
            _lib.lib().bp5_mf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(_lib.lib().bp5_mf_sync(self.handle))

    def set_geometry_mode(self, mode):
        """BP5_GEOM_MERGED6 (reference representation, default) or BP5_GEOM_AFFINE (affine meshes)."""
        _lib.check(_lib.lib().bp5_mf_set_geometry_mode(self.handle, int(mode)))

    def set_overlap(self, mode=True):
        """== AdditionalData::overlap_communication_computation (bp5/step-64.cu:241): True / 1 on, False / 0 off, 2 = the
        library decides by the slab's size (default)."""
        _lib.check(_lib.lib().bp5_mf_set_overlap(self.handle, int(mode)))

    def set_cg_fusion(self, on=True):
        """SolverCGFullMerge: dot products inside the block kernel's write-out (default) or as a separate kernel."""
        _lib.check(_lib.lib().bp5_mf_set_cg_fusion(self.handle, 1 if on else 0))

    def set_apply_variant(self, v):
        _lib.check(_lib.lib().bp5_mf_set_apply_variant(self.handle, int(v)))

    def set_operator(self, op):
        """OP_POISSON (bp5/step-64.cu:147-194, default) or OP_HELMHOLTZ (step-64/step-64.cu:154-160,201-219: seven planes)."""
        _lib.check(_lib.lib().bp5_mf_set_operator(self.handle, int(op)))

    def set_block_workgroups(self, n):
        _lib.check(_lib.lib().bp5_mf_set_block_workgroups(self.handle, int(n)))

    def set_streaming(self, policy):
        """1: non-temporal accesses to once-used data (metric planes; v, x in the update kernel), 0: ordinary, -1: by local size (default)."""
        _lib.check(_lib.lib().bp5_mf_set_streaming(self.handle, int(policy)))

    TUNE = {"lattice_indices": 0, "early_gather": 1, "combine_signal": 2, "boundary_first": 3, "fold_small": 4, "update_unroll": 5,
            "update_flat": 6, "update_nt": 7, "combine_wg_per_cu": 8, "interior_stores": 9, "ghost_combine_on_comm": 10, "face_carry": 11}   # bp5.h: BP5_TUNE_*

    def set_tuning(self, knob, value):
        """Per-handle A/B knob (bp5.h BP5_TUNE_*; same bits for every setting); knob by name or number."""
        _lib.check(_lib.lib().bp5_mf_set_tuning(self.handle, int(self.TUNE.get(knob, knob)), int(value)))

    def get_tuning(self, knob):
        v = C.c_int()
        _lib.check(_lib.lib().bp5_mf_get_tuning(self.handle, int(self.TUNE.get(knob, knob)), C.byref(v)))
        return v.value

    def wait_value_available(self):
        """True when the in-launch stream wait-value schedules passed the handle's self-check (bp5.h)."""
        v = C.c_int()
        _lib.check(_lib.lib().bp5_mf_wait_value_available(self.handle, C.byref(v)))
        return bool(v.value)

    def block_plan_info(self):
        """(n_blocks, max_runs, packed_indices) of the block kernel's plan."""
        nb, mr, pk = C.c_uint32(), C.c_uint32(), C.c_int()
        _lib.check(_lib.lib().bp5_mf_block_plan_info(self.handle, C.byref(nb), C.byref(mr), C.byref(pk)))
        return nb.value, mr.value, bool(pk.value)

    def block_plan_lattice(self):
        """number of the block plan's LATTICE blocks (closed-form indices: no per-DoF index stream)"""
        n = C.c_uint32()
        _lib.check(_lib.lib().bp5_mf_block_plan_lattice(self.handle, C.byref(n)))
        return n.value

    def block_plan_carry(self):
        """(faces the plan can carry in LDS from block to block, brick-surface DoFs, brick-surface DoFs in the combine tables of the last block launch)"""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(_lib.lib().bp5_mf_block_plan_carry(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_apply_variant(self):
        """The kernel variant a whole-range application resolves to (what 0 = default means here)."""
        v = C.c_int()
        _lib.check(_lib.lib().bp5_mf_get_apply_variant(self.handle, C.byref(v)))
        return v.value

    # -- reference API
    def initialize_dof_vector(self, vec=None):
        """== mf_data.initialize_dof_vector(vec), bp5/step-64.cu:214: owned + ghost storage."""
        torch = _torch()
        return torch.zeros(self.n_local, dtype=torch.float64, device=f"cuda:{self.device}")

    def coef_size(self):
        n = C.c_size_t()
        _lib.check(_lib.lib().bp5_mf_coef_size(self.handle, C.byref(n)))
        return n.value

    def evaluate_coefficients(self, coef=None):
        """== mf_data.evaluate_coefficients(JacobianFunctor), bp5/step-64.cu:256-258."""
        torch = _torch()
        if coef is None:
            coef = torch.empty(max(self.coef_size(), 1), dtype=torch.float64, device=f"cuda:{self.device}")
        _lib.check(_lib.lib().bp5_mf_compute_merged_metric(self.handle, _ptr(coef, self.coef_size())))
        return coef

    def coef_reference_layout(self, coef):
        torch = _torch()
        out = torch.empty_like(coef)
        _lib.check(_lib.lib().bp5_mf_metric_to_reference_layout(self.handle, _ptr(coef), _ptr(out)))
        return out

    def cell_loop(self, coef, src, dst, cell_begin=0, cell_end=None):
        """== mf_data.cell_loop(LocalPoissonOperator, src, dst), bp5/step-64.cu:274 (one range)."""
        if cell_end is None:
            cell_end = self.mesh.n_cells
        _lib.check(_lib.lib().bp5_apply_cells(self.handle, _ptr(coef), _ptr(src, self.n_local), _ptr(dst, self.n_local),
                                              cell_begin, cell_end))

    def copy_constrained_values(self, src, dst):
        _lib.check(_lib.lib().bp5_copy_constrained(self.handle, _ptr(src, self.n_local), _ptr(dst, self.n_local)))

    def set_constrained_values(self, value, dst):
        _lib.check(_lib.lib().bp5_set_constrained(self.handle, float(value), _ptr(dst, self.n_local)))

    def get_data(self, color=0):
        d = _lib.MFData()
        _lib.check(_lib.lib().bp5_mf_get_data(self.handle, color, C.byref(d)))
        return d


class PoissonOperator:
    """== BP5::PoissonOperator<3,fe_degree>, bp5/step-64.cu:198-276."""

    def __init__(self, mesh, quadrature=QUAD_GAUSS, coefficient=COEF_ONE, device=0, comm=None, stream=None, geometry=0):
        self.mf_data = MatrixFree().reinit(mesh, quadrature, coefficient, device, stream, comm)
        self.geometry = geometry
        if geometry == _lib.GEOM_AFFINE:          # per-cell metric + one scalar plane, no 6-plane array at all
            self.mf_data.set_geometry_mode(geometry)
            self.coef = None
        else:
            self.coef = self.mf_data.evaluate_coefficients()
        self.n_owned_cells = mesh.n_cells
        self.do_zero_out = True                      # bp5/step-64.cu:223,232
        self.distributed = comm is not None and comm.n_ranks > 1

    def initialize_dof_vector(self, vec=None):
        return self.mf_data.initialize_dof_vector(vec)

    def vmult(self, dst, src):
        """dst = [0 +] A src; dst[c] = src[c] on Dirichlet DoFs (bp5/step-64.cu:263-276)."""
        L, mf = _lib.lib(), self.mf_data
        dst, src = _vals(dst), _vals(src)
        fn = L.bp5_apply_distributed if self.distributed else L.bp5_apply
        _lib.check(fn(mf.handle, _ptr(self.coef), _ptr(src, mf.n_local), _ptr(dst, mf.n_local), 1 if self.do_zero_out else 0))

    def compute_diagonal(self, invert=False):
        """diag(A_eff) (1 on Dirichlet DoFs) or, with invert=True, the Jacobi preconditioner vector for
        DiagonalMatrix (the `diag` the reference's solver kernels multiply by, bp5/solver.h:68,100,131,170)."""
        d = self.initialize_dof_vector()
        _lib.check(_lib.lib().bp5_compute_diagonal(self.mf_data.handle, _ptr(self.coef), _ptr(d), 1 if invert else 0))
        return d

    def assemble_rhs(self):
        b = self.initialize_dof_vector()
        _lib.check(_lib.lib().bp5_assemble_rhs(self.mf_data.handle, _ptr(b)))
        return b

    def l2_norm_solution(self, u):
        r = C.c_double()
        _lib.check(_lib.lib().bp5_l2_norm_solution(self.mf_data.handle, _ptr(u, self.mf_data.n_local), C.byref(r)))
        return r.value


class HelmholtzOperator(PoissonOperator):
    """== Step64::HelmholtzOperator<3,fe_degree>, step-64/step-64.cu:233-305: (grad v, grad u) + (v, a(x) u) with
    a = 10 / (0.05 + 2 |x|^2) (VaryingCoefficientFunctor, :99-118) as the library's native fused kernel
    (bp5_mf_set_operator(BP5_OP_HELMHOLTZ)): `coef` holds the six merged planes and the mass plane a JxW.  vmult, the solvers
    (cg.solve(A, x, b, P): the PoissonOperator branch of Solver.solve -- same handle, same entry points) and the halo exchange
    are the ones of the Poisson operator."""

    def __init__(self, mesh, quadrature=QUAD_GAUSS, coefficient=1, device=0, comm=None, stream=None):
        self.mf_data = MatrixFree().reinit(mesh, quadrature, coefficient, device, stream, comm)
        self.mf_data.set_operator(_lib.OP_HELMHOLTZ)
        self.geometry = 0
        self.coef = self.mf_data.evaluate_coefficients()
        self.n_owned_cells = mesh.n_cells
        self.do_zero_out = True
        self.distributed = comm is not None and comm.n_ranks > 1


class Vector:
    """The part of LinearAlgebra::distributed::Vector<double, MemorySpace::CUDA> the reference's path uses
    (bp5/solver.h:369-382,417-421,511,528; bp5/step-64.cu:349,366-367,431-432,445,449,467), over a torch CUDA
    tensor (owned entries, then ghosts) and the library's BLAS-1 / halo entry points.  Solvers and operators
    accept either this class or the bare tensor (`.values`)."""

    def __init__(self, mf=None):
        self.mf, self.values = mf, None
        if mf is not None:
            self.reinit(mf)

    def reinit(self, other, omit_zeroing_entries=False):
        """reinit(MatrixFree) == initialize_dof_vector; reinit(Vector) == same layout as `other`."""
        mf = other.mf if isinstance(other, Vector) else other
        if self.values is None or self.mf is not mf:
            self.mf = mf
            self.values = mf.initialize_dof_vector()
        elif not omit_zeroing_entries:
            self.assign(0.0)
        return self

    def assign(self, s):
        """== operator=(scalar)"""
        _lib.check(_lib.lib().bp5_vec_fill(self.mf.handle, _ptr(self.values), float(s), self.mf.n_local))
        return self

    def all_zero(self):
        z = C.c_int()
        _lib.check(_lib.lib().bp5_vec_all_zero(self.mf.handle, _ptr(self.values), self.local_size(), C.byref(z)))
        return bool(z.value)

    def add(self, a, v):
        _lib.check(_lib.lib().bp5_vec_axpy(self.mf.handle, _ptr(self.values), float(a), _ptr(v.values), self.local_size()))

    def equ(self, a, v):
        _lib.check(_lib.lib().bp5_vec_equ(self.mf.handle, _ptr(self.values), float(a), _ptr(v.values), self.local_size()))

    def sadd(self, s, a, v):
        _lib.check(_lib.lib().bp5_vec_sadd(self.mf.handle, _ptr(self.values), float(s), float(a), _ptr(v.values), self.local_size()))

    def l2_norm(self):
        r = C.c_double()
        _lib.check(_lib.lib().bp5_vec_l2_norm(self.mf.handle, _ptr(self.values), self.local_size(), C.byref(r)))
        return r.value

    def get_values(self):
        return self.values

    def local_size(self):
        return self.mf.mesh.n_owned

    def size(self):
        return int(self.mf.mesh.n_global_dofs)

    def import_(self, host_values):
        """== import(ReadWriteVector, VectorOperation::insert): host values of the owned range."""
        torch = _torch()
        h = torch.as_tensor(host_values, dtype=torch.float64)
        if h.numel() != self.local_size():
            raise BP5Error(1, "import: need the owned range")
        self.values[:self.local_size()].copy_(h)

    def update_ghost_values(self):
        _lib.check(_lib.lib().bp5_halo_gather(self.mf.handle, _ptr(self.values)))

    def update_ghost_values_start(self):
        _lib.check(_lib.lib().bp5_halo_gather_start(self.mf.handle, _ptr(self.values)))

    def update_ghost_values_finish(self):
        _lib.check(_lib.lib().bp5_halo_gather_finish(self.mf.handle, _ptr(self.values)))

    def compress_start(self):
        """== compress_start(VectorOperation::add)"""
        _lib.check(_lib.lib().bp5_halo_scatter_add_start(self.mf.handle, _ptr(self.values)))

    def compress_finish(self):
        _lib.check(_lib.lib().bp5_halo_scatter_add_finish(self.mf.handle, _ptr(self.values)))

    def compress_add(self):
        """== compress(VectorOperation::add)"""
        _lib.check(_lib.lib().bp5_halo_scatter_add(self.mf.handle, _ptr(self.values)))

    def zero_out_ghosts(self):
        _lib.check(_lib.lib().bp5_halo_zero_ghosts(self.mf.handle, _ptr(self.values)))


def _vals(v):
    return v.values if isinstance(v, Vector) else v


class DiagonalMatrix:
    """== DiagonalMatrix<Vector>; `None` vector == identity (the reference sets it to 1,
    bp5/step-64.cu:432, and still streams it; here identity costs no bytes)."""

    def __init__(self, vector=None):
        self._v = vector

    def get_vector(self):
        return self._v


class SolverControl:
    def __init__(self, max_steps=100, tolerance=1e-10):
        self.max_steps, self.tolerance = int(max_steps), float(tolerance)
        self._last_step, self._last_value, self._initial_value = 0, float("nan"), float("nan")
        self.solve_ms = self.apply_ms_avg = self.operator_ms_avg = 0.0
        self.apply_launches = 0
        self.dot_products_fused = False
        self.exchange_schedule, self.apply_kernel, self.phase_ms = 0, "", [0.0] * 8

    def last_step(self):
        return self._last_step

    def last_value(self):
        return self._last_value

    def initial_value(self):
        return self._initial_value


class IterationNumberControl(SolverControl):
    """Stops at max_steps or when the residual drops below tolerance, reports success either way
    (bp5/step-64.cu:443-445)."""


class _DeviceArray:
    """__cuda_array_interface__ carrier: lets torch alias device memory the library owns (the solvers' work vectors)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class _SolverBase:
    variant = CG_PLAIN

    def __init__(self, control, check_every=0, profile=False):
        self.control, self.check_every, self.profile = control, check_every, profile

    def solve(self, A, x, b, preconditioner=None):
        """== cg.solve(A, x, b, preconditioner), bp5/step-64.cu:450-453,492-495.  x0 = 0.
        A PoissonOperator runs entirely inside bp5_cg_solve; any other object with `mf_data` (vector layout, stream) and
        `vmult(dst, src)` is solved through bp5_cg_solve_operator -- the solvers need nothing of A but vmult
        (bp5/solver.h:25-30,377,475)."""
        mf = A.mf_data
        x, b = _vals(x), _vals(b)
        diag = _vals(preconditioner.get_vector()) if preconditioner is not None else None
        prm = _lib.CGParams(self.variant, self.control.max_steps, self.control.tolerance, self.check_every,
                            int(self.profile))   # False / True / 2 (phase stamps)
        res = _lib.CGResult()
        dptr = _ptr(diag, mf.n_owned) if diag is not None else None
        if isinstance(A, PoissonOperator):
            status = _lib.lib().bp5_cg_solve(mf.handle, _ptr(A.coef), dptr, _ptr(b, mf.n_local), _ptr(x, mf.n_local), C.byref(prm), C.byref(res))
        else:
            torch, failure = _torch(), []

            def view(p):
                return torch.as_tensor(_DeviceArray(p, mf.n_local), device=f"cuda:{mf.device}")

            def callback(_ctx, dst, src):      # enqueues on the current stream; no exception may cross the C boundary
                try:
                    A.vmult(view(dst), view(src))
                    return 0
                except Exception as e:         # noqa: BLE001
                    failure.append(e)
                    return 1

            cb = _lib.VMULT_FN(callback)
            status = _lib.lib().bp5_cg_solve_operator(mf.handle, cb, None, dptr, _ptr(b, mf.n_local), _ptr(x, mf.n_local), C.byref(prm), C.byref(res))
            if failure:
                raise failure[0]
        c = self.control
        c._last_step, c._last_value, c._initial_value = res.iterations, res.residual, res.initial_residual
        c.solve_ms, c.apply_ms_avg, c.apply_launches = res.solve_ms, res.apply_ms_avg, res.apply_launches
        c.operator_ms_avg = res.operator_ms_avg
        c.dot_products_fused = bool(res.dot_products_fused)
        c.exchange_schedule, c.apply_kernel, c.phase_ms = res.exchange_schedule, res.apply_kernel.decode(), list(res.phase_ms)
        _lib.check(status)
        return res


class SolverCG(_SolverBase):
    variant = CG_PLAIN


class SolverCGFullMerge(_SolverBase):
    variant = CG_MERGED
