#!/usr/bin/env python3
"""A/B in ONE process: merged CG with the dot products fused into the block kernel vs the separate kernels, and the library's
default operator variant vs a forced one, on one mesh.  usage: python tools/ab_fusion.py --cells 116 116 116 [--variants 0 56]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bp5_pkg
pkg = bp5_pkg.load()
ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, nargs=3, default=[116, 116, 116])
ap.add_argument("--cell-block", type=int, nargs=3, default=[4, 4, 4])
ap.add_argument("--variants", type=int, nargs="+", default=[0])
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--solver", choices=["merged", "plain"], default="merged")
ap.add_argument("--degree", type=int, default=4)
a = ap.parse_args()
mesh = pkg.BrickMesh(a.degree, a.cells, h=1.0 / a.cells[0], cell_block=a.cell_block, dof_numbering=1, cell_block_order=1)
op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
b, x = op.assemble_rhs(), op.initialize_dof_vector()
res = {}
for rnd in range(a.rounds + 1):
    for v in a.variants:
        for fused in (True, False):
            op.mf_data.set_apply_variant(v)
            op.mf_data.set_cg_fusion(fused)
            ctl = pkg.IterationNumberControl(a.iters, 0.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            (pkg.SolverCGFullMerge if a.solver == "merged" else pkg.SolverCG)(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.iters * 1e3
            if rnd:
                res.setdefault((v, fused), []).append((dt, ctl.apply_ms_avg, ctl.operator_ms_avg, op.mf_data.get_apply_variant()))
print(f"cells {a.cells} block {a.cell_block} dofs {mesh.n_owned}")
for (v, fused), r in res.items():
    best = min(r)
    print(f"variant {v} (effective {best[3]}) fused={fused}: {best[0]:.4f} ms/iteration = {mesh.n_owned / best[0] / 1e6:.2f} GDoF/s, "
          f"cell kernel {best[1]:.3f} ms, operator {best[2]:.3f} ms")
