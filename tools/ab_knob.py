#!/usr/bin/env python3
"""merged CG per-iteration time with a knob of the library at several values, interleaved in ONE process: a tuning knob of the handle
(--knob update_flat -> mf.set_tuning("update_flat", int(value)); several knobs at once: --knob update_flat,update_unroll --values 0,4 1,1 1,2)
or a setter of MatrixFree (--knob streaming -> mf.set_streaming(int(value)))
usage: python tools/ab_knob.py --knob streaming --values 0 1 --cells 54 54 54 --cell-block 4 4 2"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bp5_pkg
pkg = bp5_pkg.load()
ap = argparse.ArgumentParser()
ap.add_argument("--knob", default="streaming")
ap.add_argument("--values", nargs="+", default=["0", "1"])
ap.add_argument("--cells", type=int, nargs=3, default=[116, 116, 116])
ap.add_argument("--cell-block", type=int, nargs=3, default=[4, 4, 4])
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--degree", type=int, default=4)
ap.add_argument("--numbering", type=int, default=1)
ap.add_argument("--quadrature", choices=["gauss", "gll"], default="gauss")
a = ap.parse_args()
mesh = pkg.BrickMesh(a.degree, a.cells, h=1.0 / a.cells[0], cell_block=a.cell_block, dof_numbering=a.numbering, cell_block_order=1)
op = pkg.PoissonOperator(mesh, pkg.QUAD_GLL if a.quadrature == "gll" else pkg.QUAD_GAUSS, pkg.COEF_STEP64)
b, x = op.assemble_rhs(), op.initialize_dof_vector()
try:
    print("bricks", op.mf_data.block_plan_info()[0], "dofs", mesh.n_owned, flush=True)
except pkg.BP5Error:   # (cell blocks too large for the block kernel: the pencil kernel's meshes)
    print("no block plan; dofs", mesh.n_owned, flush=True)
res = {}
for rnd in range(a.rounds + 1):
    for val in a.values:
        for kn, v in zip(a.knob.split(","), val.split(",")):
            if kn in op.mf_data.TUNE:
                op.mf_data.set_tuning(kn, int(v))
            else:
                getattr(op.mf_data, "set_" + kn)(int(v))
        ctl = pkg.IterationNumberControl(a.iters, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pkg.SolverCGFullMerge(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
        torch.cuda.synchronize()
        if rnd:
            res.setdefault(val, []).append(((time.perf_counter() - t0) / a.iters * 1e3, ctl.apply_ms_avg, ctl.apply_kernel))
for val, r in res.items():
    ms = sorted(x[0] for x in r)
    print(f"{a.knob}={val}: median {ms[len(ms) // 2]:.4f} min {ms[0]:.4f} ms/iteration, cell kernel {min(x[1] for x in r):.4f} ms  {r[0][2]}")
