"""ctypes binding of oracle/libbp5_oracle.so (CPU ORACLE -- test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libbp5_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libbp5_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_tables.argtypes = [C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p]
        L.orc_geometry.argtypes = [C.c_int, C.c_int, C.c_uint32, _u32p, _f64p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_plan_create.restype = C.c_void_p
        L.orc_plan_create.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, _u32p]
        L.orc_plan_destroy.argtypes = [C.c_void_p]
        L.orc_apply.argtypes = [C.c_void_p, _f64p, _f64p, _f64p]
        L.orc_vmult.argtypes = [C.c_void_p, _f64p, _u32p, C.c_uint32, _f64p, _f64p]
        L.orc_rhs.argtypes = [C.c_int, C.c_uint32, C.c_uint32, _u32p, _f64p, _u32p, C.c_uint32, _f64p]
        L.orc_cg_plain.argtypes = [C.c_void_p, _f64p, _u32p, C.c_uint32, _f64p, _f64p, C.c_int, C.c_double,
                                   C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def tables(p, quadrature):
    n = p + 1
    nodes, pts, w = (np.zeros(n) for _ in range(3))
    N, D = np.zeros((n, n)), np.zeros((n, n))
    assert lib().orc_tables(p, quadrature, nodes, pts, w, N, D) == 0
    return nodes, pts, w, N, D


class CProblem:
    """C-oracle counterpart of bp5_oracle.Problem on explicit mesh arrays."""

    def __init__(self, p, quadrature, l2g, coords, constrained, kappa_mode=0):
        self.p, self.quadrature = p, quadrature
        self.l2g = np.ascontiguousarray(l2g, dtype=np.uint32).reshape(-1)
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.constrained = np.ascontiguousarray(constrained, dtype=np.uint32)
        self.n_dofs = self.coords.shape[0]
        self.nq = (p + 1) ** 3
        self.n_cells = self.l2g.size // self.nq
        self.coef = np.zeros((6, self.n_cells, self.nq))
        L = lib()
        assert L.orc_geometry(p, quadrature, self.n_cells, self.l2g, self.coords, kappa_mode,
                              self.coef.ctypes.data, None, None) == 0
        self.plan = L.orc_plan_create(p, quadrature, self.n_cells, self.n_dofs, self.l2g)
        assert self.plan

    def __del__(self):
        if getattr(self, "plan", None):
            lib().orc_plan_destroy(self.plan)
            self.plan = None

    def geometry(self):
        K = np.zeros((self.n_cells, self.nq, 3, 3))
        JxW = np.zeros((self.n_cells, self.nq))
        lib().orc_geometry(self.p, self.quadrature, self.n_cells, self.l2g, self.coords, 0, None,
                           K.ctypes.data, JxW.ctypes.data)
        return K, JxW

    def apply(self, src):
        dst = np.zeros(self.n_dofs)
        lib().orc_apply(self.plan, self.coef.reshape(-1), np.ascontiguousarray(src), dst)
        return dst

    def vmult(self, src):
        dst = np.zeros(self.n_dofs)
        lib().orc_vmult(self.plan, self.coef.reshape(-1), self.constrained, self.constrained.size,
                        np.ascontiguousarray(src), dst)
        return dst

    def rhs(self):
        b = np.zeros(self.n_dofs)
        assert lib().orc_rhs(self.p, self.n_cells, self.n_dofs, self.l2g, self.coords, self.constrained,
                             self.constrained.size, b) == 0
        return b

    def cg_plain(self, b, max_iter, tol=0.0):
        x = np.zeros(self.n_dofs)
        it, res = C.c_int(0), C.c_double(0.0)
        lib().orc_cg_plain(self.plan, self.coef.reshape(-1), self.constrained, self.constrained.size,
                           np.ascontiguousarray(b), x, max_iter, tol, C.byref(it), C.byref(res))
        return x, it.value, res.value
