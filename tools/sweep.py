#!/usr/bin/env python3
"""BASELINE configs 4 and 5: operator + CG throughput for p = 1..8 at ~5e7 DoFs (and p = 6 deformed),
both quadratures.  usage: python tools/sweep.py [--dofs 5e7] [--degrees 1 2 ...]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bp5_pkg
pkg = bp5_pkg.load()

ap = argparse.ArgumentParser()
ap.add_argument("--dofs", type=float, default=5e7)
ap.add_argument("--degrees", type=int, nargs="+", default=list(range(1, 9)))
ap.add_argument("--variants", type=int, nargs="+", default=[0])
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--deform", type=float, default=0.0)
ap.add_argument("--cell-block", type=int, nargs=3, default=[0, 0, 0])
ap.add_argument("--numbering", type=int, default=0)
a = ap.parse_args()
rows = []
for p in a.degrees:
    ncell = max(2, int(round((a.dofs ** (1 / 3) - 1) / p)))
    mesh = pkg.BrickMesh(p, (ncell,) * 3, h=1.0 / ncell, deform_amp=a.deform, cell_block=a.cell_block, dof_numbering=a.numbering)
    n = mesh.n_owned
    r = mesh.n_cells * (p + 1) ** 3 / n
    B_op, B = 16 + 52 * r, 16 + 52 * r + 88
    for quad, qn in ((pkg.QUAD_GAUSS, "gauss"), (pkg.QUAD_GLL, "gll")):
        op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64)
        b = op.assemble_rhs()
        x = op.initialize_dof_vector()
        for v in a.variants:
            try:
                op.mf_data.set_apply_variant(v)
                pkg.SolverCGFullMerge(pkg.IterationNumberControl(3, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
            except pkg.BP5Error as e:
                print(f"p={p} {qn} variant {v}: {e}")
                continue
            ctl = pkg.IterationNumberControl(a.iters, 0.0)
            pkg.SolverCGFullMerge(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
            cg = n * ctl.last_step() / (ctl.solve_ms * 1e-3)
            row = dict(p=p, quad=qn, variant=v, cells=ncell, dofs=n, r=round(r, 3), apply_ms=round(ctl.apply_ms_avg, 3),
                       apply_GDoFs=round(n / ctl.apply_ms_avg / 1e6, 2), apply_frac=round(B_op * n / ctl.apply_ms_avg / 1e6 / 8000, 3),
                       cg_GDoFs=round(cg / 1e9, 2), cg_frac=round(cg * B / 8e12, 3))
            rows.append(row)
            print(json.dumps(row), flush=True)
        del op, b, x
        torch.cuda.empty_cache()
