import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bp5_pkg
pkg = bp5_pkg.load()
mesh = pkg.BrickMesh(4, (9, 8, 6), h=0.2, deform_amp=0.03, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
xs = {}
b = None
for tag, lattice, carry in (("lat", 1, 0), ("pack", 0, 0), ("lat2", 1, 0), ("latc", 1, 1)):
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    op.mf_data.set_tuning("lattice_indices", lattice)
    op.mf_data.set_tuning("face_carry", carry)
    op.mf_data.set_apply_variant(56)
    op.mf_data.set_block_workgroups(8)
    if b is None:
        b = op.assemble_rhs()
    for its in (1, 2, 5):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        xs[(tag, its)] = x
    print(tag, ctl.apply_kernel, op.mf_data.block_plan_info(), op.mf_data.block_plan_carry(), flush=True)
for its in (1, 2, 5):
    r = xs[("lat", its)]
    for tag in ("pack", "lat2", "latc"):
        d = (xs[(tag, its)] - r).abs().max().item()
        print(its, tag, "max abs diff vs lat", d, "rel", d / r.abs().max().item())
