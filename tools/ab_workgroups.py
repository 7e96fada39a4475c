#!/usr/bin/env python3
"""merged CG per-iteration time against the cap on the block kernel's persistent grid (quantisation of bricks per workgroup on small meshes)
usage: python tools/ab_workgroups.py --cells 54 54 54 --cell-block 4 4 2 --caps 0 512 640 704"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bp5_pkg
pkg = bp5_pkg.load()
ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, nargs=3, default=[54, 54, 54])
ap.add_argument("--cell-block", type=int, nargs=3, default=[4, 4, 2])
ap.add_argument("--caps", type=int, nargs="+", default=[0, 512])
ap.add_argument("--iters", type=int, default=100)
a = ap.parse_args()
mesh = pkg.BrickMesh(4, a.cells, h=1.0 / a.cells[0], cell_block=a.cell_block, dof_numbering=1, cell_block_order=1)
op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
op.mf_data.set_apply_variant(56)
b, x = op.assemble_rhs(), op.initialize_dof_vector()
print("bricks", op.mf_data.block_plan_info()[0], "dofs", mesh.n_owned)
res = {}
for rnd in range(4):
    for cap in a.caps:
        op.mf_data.set_block_workgroups(cap)
        ctl = pkg.IterationNumberControl(a.iters, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pkg.SolverCGFullMerge(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
        torch.cuda.synchronize()
        if rnd:
            res.setdefault(cap, []).append(((time.perf_counter() - t0) / a.iters * 1e3, ctl.apply_ms_avg))
for cap, r in res.items():
    print(f"cap {cap}: {min(r)[0]:.4f} ms/iteration, cell kernel {min(x[1] for x in r):.4f} ms")
