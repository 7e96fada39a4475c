import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bp5_pkg
pkg = bp5_pkg.load()
cells, block, wgs = (9, 6, 7), (4, 4, 4), 8
mesh = pkg.BrickMesh(4, cells, h=0.125, deform_amp=0.03, cell_block=block, dof_numbering=1, cell_block_order=1)
op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
b = op.assemble_rhs()
bn = float(torch.linalg.norm(b))
for variant in (56, 3):
  op.mf_data.set_apply_variant(variant)
  op.mf_data.set_block_workgroups(wgs)
  for tol in (1e-4, 1e-6, 1e-8):
    for solver in (pkg.SolverCGFullMerge, pkg.SolverCG):
      for fused in (True, False):
        op.mf_data.set_cg_fusion(fused)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(400, tol * bn)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        Ax = op.initialize_dof_vector()
        op.vmult(Ax, x)
        print(variant, tol, solver.__name__, fused, ctl.last_step(), ctl.last_value() / bn, float(torch.linalg.norm(Ax - b)) / bn)
