"""MI355X-native BP5 matrix-free operator + CG hot path (host-side mirror of the reference's
operator interface: PoissonOperator / MatrixFree / SolverCG / SolverCGFullMerge,
peterrum/deal-and-ceed-on-gpu bp5/step-64.cu, bp5/solver.h).

All compute goes through the C ABI of include/bp5.h (libbp5.so, hand-written HIP for gfx950).
There is NO CPU fallback: importing works anywhere, compute calls fail loudly without the
library or without a GPU."""
from ._lib import (BP5Error, QUAD_GAUSS, QUAD_GLL, COEF_ONE, COEF_STEP64, CG_PLAIN, CG_MERGED, GEOM_MERGED6, GEOM_AFFINE, OP_POISSON, OP_HELMHOLTZ, build, lib,
                   lib_path, shape_tables, HEADER_SYMBOLS)
from .mesh import BrickMesh
from .matrix_free import (MatrixFree, PoissonOperator, HelmholtzOperator, DiagonalMatrix, IterationNumberControl, SolverControl,
                          SolverCG, SolverCGFullMerge, Communicator, Vector)

__all__ = ["BP5Error", "QUAD_GAUSS", "QUAD_GLL", "COEF_ONE", "COEF_STEP64", "CG_PLAIN", "CG_MERGED", "GEOM_MERGED6", "GEOM_AFFINE", "OP_POISSON", "OP_HELMHOLTZ", "build", "lib",
           "lib_path", "shape_tables", "HEADER_SYMBOLS", "BrickMesh", "MatrixFree", "PoissonOperator", "HelmholtzOperator",
           "DiagonalMatrix", "IterationNumberControl", "SolverControl", "SolverCG", "SolverCGFullMerge",
           "Communicator", "Vector"]
