"""ctypes binding of libbp5.so (include/bp5.h).  No torch types cross this boundary."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = None

QUAD_GAUSS, QUAD_GLL = 0, 1
COEF_ONE, COEF_STEP64 = 0, 1
CG_PLAIN, CG_MERGED = 0, 1
GEOM_MERGED6, GEOM_AFFINE = 0, 1
OP_POISSON, OP_HELMHOLTZ = 0, 1
UNIQUE_ID_BYTES = 128


class BP5Error(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"bp5 status {status}: {detail}")
        self.status = status


def lib_path():
    return os.environ.get("BP5_LIB", os.path.join(_HERE, "libbp5.so"))


def build(verbose=False):
    """Compile libbp5.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-j8", "-C", os.path.join(_HERE, "csrc")], stdout=out)
    return lib_path()


def _header_symbols():
    text = open(os.path.join(_ROOT, "include", "bp5.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bp5_[a-z0-9_]+)\s*\(", text)))


HEADER_SYMBOLS = _header_symbols()


class MeshDesc(C.Structure):
    _fields_ = [("degree", C.c_int), ("cells", C.c_uint32 * 3), ("h", C.c_double), ("deform_amp", C.c_double),
                ("rank", C.c_int), ("n_ranks", C.c_int), ("cell_block", C.c_uint32 * 3), ("dof_numbering", C.c_int), ("cell_block_order", C.c_int)]


class MeshView(C.Structure):
    _fields_ = [("degree", C.c_int), ("n_cells", C.c_uint32), ("n_interior_cells", C.c_uint32), ("n_owned", C.c_uint32),
                ("n_ghost", C.c_uint32), ("n_global_dofs", C.c_uint64), ("global_dofs_per_dir", C.c_uint32 * 3),
                ("local_to_global_host", C.POINTER(C.c_uint32)), ("node_coords_host", C.POINTER(C.c_double)),
                ("global_ids_host", C.POINTER(C.c_uint64)), ("constrained_host", C.POINTER(C.c_uint32)),
                ("n_constrained", C.c_uint32), ("n_neighbors", C.c_int), ("neighbor_rank_host", C.POINTER(C.c_int)),
                ("send_offsets_host", C.POINTER(C.c_uint32)), ("send_indices_host", C.POINTER(C.c_uint32)),
                ("recv_offsets_host", C.POINTER(C.c_uint32)), ("n_cell_blocks", C.c_uint32),
                ("cell_block_offsets_host", C.POINTER(C.c_uint32))]


class MFDesc(C.Structure):
    _fields_ = [("dim", C.c_int), ("degree", C.c_int), ("quadrature", C.c_int), ("coefficient", C.c_int),
                ("n_cells", C.c_uint32), ("n_interior_cells", C.c_uint32), ("n_owned", C.c_uint32), ("n_ghost", C.c_uint32),
                ("local_to_global_host", C.c_void_p), ("node_coords_host", C.c_void_p), ("constrained_host", C.c_void_p),
                ("n_constrained", C.c_uint32), ("n_neighbors", C.c_int), ("neighbor_rank_host", C.c_void_p),
                ("send_offsets_host", C.c_void_p), ("send_indices_host", C.c_void_p), ("recv_offsets_host", C.c_void_p),
                ("device", C.c_int), ("stream", C.c_void_p), ("n_cell_blocks", C.c_uint32),
                ("cell_block_offsets_host", C.c_void_p), ("constraint_mask_host", C.c_void_p)]


class MFData(C.Structure):
    _fields_ = [("local_to_global", C.c_void_p), ("inv_jacobian", C.c_void_p), ("JxW", C.c_void_p), ("q_points", C.c_void_p),
                ("constraint_mask", C.c_void_p), ("n_cells", C.c_uint32), ("padding_length", C.c_uint32),
                ("row_start", C.c_uint32), ("use_coloring", C.c_int)]


class CGParams(C.Structure):
    _fields_ = [("variant", C.c_int), ("max_iter", C.c_int), ("abs_tol", C.c_double), ("check_every", C.c_int),
                ("profile", C.c_int)]


class CGResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("residual", C.c_double), ("initial_residual", C.c_double), ("solve_ms", C.c_double),
                ("apply_ms_avg", C.c_double), ("apply_launches", C.c_int), ("operator_ms_avg", C.c_double), ("dot_products_fused", C.c_int),
                ("exchange_schedule", C.c_int), ("apply_kernel", C.c_char * 96), ("phase_ms", C.c_double * 8)]


VMULT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)   # bp5_vmult_fn(ctx, dst, src)


def lib():
    """Load libbp5.so; fails loudly if it has not been built (no fallback of any kind)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise BP5Error(-1, f"{path} is missing: run __graft_entry__.build() (make -C deal-and-ceed-on-gpu_amd/csrc)")
    try:  # share torch's HIP/RCCL runtime when torch is in the process (same SONAMEs)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, u32, i32, f64, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_double, C.c_size_t
    protos = {
        "bp5_strerror": (C.c_char_p, [i32]),
        "bp5_last_error": (C.c_char_p, []),
        "bp5_shape_tables": (i32, [i32, i32, vp, vp, vp, vp, vp]),
        "bp5_mesh_create_brick": (i32, [C.POINTER(MeshDesc), C.POINTER(vp)]),
        "bp5_mesh_view_get": (i32, [vp, C.POINTER(MeshView)]),
        "bp5_mesh_destroy": (None, [vp]),
        "bp5_device_count": (i32, [C.POINTER(i32)]),
        "bp5_vec_alloc": (i32, [sz, C.POINTER(vp)]),
        "bp5_vec_free": (i32, [vp]),
        "bp5_copy_h2d": (i32, [vp, vp, sz]),
        "bp5_copy_d2h": (i32, [vp, vp, sz]),
        "bp5_mf_create": (i32, [C.POINTER(MFDesc), C.POINTER(vp)]),
        "bp5_mf_destroy": (i32, [vp]),
        "bp5_mf_set_stream": (i32, [vp, vp]),
        "bp5_mf_sync": (i32, [vp]),
        "bp5_mf_coef_size": (i32, [vp, C.POINTER(sz)]),
        "bp5_mf_compute_merged_metric": (i32, [vp, vp]),
        "bp5_mf_metric_to_reference_layout": (i32, [vp, vp, vp]),
        "bp5_mf_set_geometry_mode": (i32, [vp, i32]),
        "bp5_mf_get_data": (i32, [vp, i32, C.POINTER(MFData)]),
        "bp5_apply": (i32, [vp, vp, vp, vp, i32]),
        "bp5_apply_cells": (i32, [vp, vp, vp, vp, u32, u32]),
        "bp5_copy_constrained": (i32, [vp, vp, vp]),
        "bp5_set_constrained": (i32, [vp, f64, vp]),
        "bp5_mf_set_apply_variant": (i32, [vp, i32]),
        "bp5_mf_get_apply_variant": (i32, [vp, C.POINTER(C.c_int)]),
        "bp5_mf_set_block_workgroups": (i32, [vp, i32]),
        "bp5_mf_set_streaming": (i32, [vp, i32]),
        "bp5_mf_set_tuning": (i32, [vp, i32, i32]),
        "bp5_mf_get_tuning": (i32, [vp, i32, C.POINTER(C.c_int)]),
        "bp5_mf_wait_value_available": (i32, [vp, C.POINTER(C.c_int)]),
        "bp5_mf_block_plan_info": (i32, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
        "bp5_assemble_rhs": (i32, [vp, vp]),
        "bp5_compute_diagonal": (i32, [vp, vp, vp, i32]),
        "bp5_l2_norm_solution": (i32, [vp, vp, C.POINTER(f64)]),
        "bp5_vec_fill": (i32, [vp, vp, f64, sz]),
        "bp5_vec_axpy": (i32, [vp, vp, f64, vp, sz]),
        "bp5_vec_equ": (i32, [vp, vp, f64, vp, sz]),
        "bp5_vec_sadd": (i32, [vp, vp, f64, f64, vp, sz]),
        "bp5_vec_dot": (i32, [vp, vp, vp, sz, C.POINTER(f64)]),
        "bp5_vec_l2_norm": (i32, [vp, vp, C.c_size_t, C.POINTER(C.c_double)]),
        "bp5_vec_all_zero": (i32, [vp, vp, C.c_size_t, C.POINTER(C.c_int)]),
        "bp5_comm_unique_id": (i32, [vp]),
        "bp5_comm_create": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "bp5_comm_destroy": (i32, [vp]),
        "bp5_mf_set_comm": (i32, [vp, vp]),
        "bp5_comm_allreduce_sum": (i32, [vp, vp, sz]),
        "bp5_halo_gather": (i32, [vp, vp]),
        "bp5_halo_gather_start": (i32, [vp, vp]),
        "bp5_halo_gather_finish": (i32, [vp, vp]),
        "bp5_halo_scatter_add_start": (i32, [vp, vp]),
        "bp5_halo_scatter_add_finish": (i32, [vp, vp]),
        "bp5_mf_set_overlap": (i32, [vp, i32]),
        "bp5_mf_set_cg_fusion": (i32, [vp, i32]),
        "bp5_mf_set_operator": (i32, [vp, i32]),
        "bp5_mf_block_plan_lattice": (i32, [vp, vp]),
        "bp5_mf_block_plan_carry": (i32, [vp, vp, vp, vp]),
        "bp5_halo_scatter_add": (i32, [vp, vp]),
        "bp5_halo_zero_ghosts": (i32, [vp, vp]),
        "bp5_apply_distributed": (i32, [vp, vp, vp, vp, i32]),
        "bp5_cg_solve": (i32, [vp, vp, vp, vp, vp, C.POINTER(CGParams), C.POINTER(CGResult)]),
        "bp5_cg_solve_operator": (i32, [vp, VMULT_FN, vp, vp, vp, vp, C.POINTER(CGParams), C.POINTER(CGResult)]),
        "bp5_event_create": (i32, [C.POINTER(vp)]),
        "bp5_event_record": (i32, [vp, vp]),
        "bp5_event_elapsed_ms": (i32, [vp, vp, C.POINTER(f64)]),
        "bp5_event_destroy": (i32, [vp]),
    }
    for name, (res, args) in protos.items():
        if os.environ.get("BP5_LIB") and not hasattr(L, name):
            continue  # A/B runs against an older build (tools only); the default library must export everything
        fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    L._protos = protos
    _LIB = L
    return L


def check(status):
    if status != 0:
        L = lib()
        raise BP5Error(status, f"{L.bp5_strerror(status).decode()}: {L.bp5_last_error().decode()}")


def shape_tables(degree, quadrature):
    """(nodes, pts, w, N, D) as numpy arrays -- host-only, works without a GPU."""
    n = degree + 1
    nodes, pts, w = (np.zeros(n) for _ in range(3))
    N, D = np.zeros((n, n)), np.zeros((n, n))
    check(lib().bp5_shape_tables(degree, quadrature, nodes.ctypes.data, pts.ctypes.data, w.ctypes.data,
                                 N.ctypes.data, D.ctypes.data))
    return nodes, pts, w, N, D
