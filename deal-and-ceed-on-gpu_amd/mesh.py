"""Structured hex mesh from the library's host generator (bp5_mesh_create_brick): stands in for
GridGenerator::subdivided_hyper_rectangle + refine_global + distribute_dofs + boundary
constraints (bp5/step-64.cu:341-358,629-663) with a z-slab partition instead of p4est."""
import ctypes as C

import numpy as np

from . import _lib


class BrickMesh:
    def __init__(self, degree, cells, h=1.0, deform_amp=0.0, rank=0, n_ranks=1, cell_block=(0, 0, 0), dof_numbering=0, cell_block_order=0):
        L = _lib.lib()
        d = _lib.MeshDesc(degree, (C.c_uint32 * 3)(*[int(c) for c in cells]), float(h), float(deform_amp), rank, n_ranks,
                          (C.c_uint32 * 3)(*[int(b) for b in cell_block]), int(dof_numbering), int(cell_block_order))
        self._h = C.c_void_p()
        _lib.check(L.bp5_mesh_create_brick(C.byref(d), C.byref(self._h)))
        v = _lib.MeshView()
        _lib.check(L.bp5_mesh_view_get(self._h, C.byref(v)))
        self.view = v
        self.degree, self.cells, self.h, self.deform_amp = degree, tuple(cells), h, deform_amp
        self.rank, self.n_ranks = rank, n_ranks
        self.n = degree + 1
        self.n_cells, self.n_interior_cells = v.n_cells, v.n_interior_cells
        self.n_owned, self.n_ghost, self.n_global_dofs = v.n_owned, v.n_ghost, v.n_global_dofs
        self.n_local = self.n_owned + self.n_ghost
        n3 = self.n ** 3

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,))

        self.l2g = arr(v.local_to_global_host, self.n_cells * n3, np.uint32).reshape(self.n_cells, n3)
        self.coords = arr(v.node_coords_host, self.n_local * 3, np.float64).reshape(self.n_local, 3)
        self.global_ids = arr(v.global_ids_host, self.n_local, np.uint64)
        self.constrained = arr(v.constrained_host, v.n_constrained, np.uint32)
        self.n_neighbors = v.n_neighbors
        self.neighbor_rank = arr(v.neighbor_rank_host, v.n_neighbors, np.int32)
        self.send_offsets = arr(v.send_offsets_host, v.n_neighbors + 1, np.uint32)
        self.send_indices = arr(v.send_indices_host, int(self.send_offsets[-1]) if v.n_neighbors else 0, np.uint32)
        self.recv_offsets = arr(v.recv_offsets_host, v.n_neighbors + 1, np.uint32)
        self.cell_block = tuple(cell_block)
        self.dof_numbering = dof_numbering
        self.cell_block_order = cell_block_order
        self.cell_block_offsets = arr(v.cell_block_offsets_host, v.n_cell_blocks + 1, np.uint32) if v.n_cell_blocks else None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().bp5_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
