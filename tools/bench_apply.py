#!/usr/bin/env python3
"""A/B timing of operator-kernel variants in ONE process, interleaved rounds (guide rule 24).
usage: python tools/bench_apply.py --degree 4 --cells 116 116 116 --variants 0 1 2 3 4 5
The timing-only ablation variants (wrong results by construction: 20+mask, 40+mask, 60+mask, 80-99, ...) exist only in the separate
library: `make -C deal-and-ceed-on-gpu_amd/csrc timing` and run with BP5_LIB=deal-and-ceed-on-gpu_amd/libbp5_timing.so."""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bp5_pkg
pkg = bp5_pkg.load()

ap = argparse.ArgumentParser()
ap.add_argument("--degree", type=int, default=4)
ap.add_argument("--cells", type=int, nargs=3, default=[116, 116, 116])
ap.add_argument("--variants", type=int, nargs="+", default=[0])
ap.add_argument("--quadrature", default="gauss")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--deform", type=float, default=0.0)
ap.add_argument("--cell-block", type=int, nargs=3, default=[0, 0, 0])
ap.add_argument("--numbering", type=int, default=0)
ap.add_argument("--block-order", type=int, default=0, help="1: parity-class-major cell order inside a block")
ap.add_argument("--geometry", choices=["merged6", "affine"], default="merged6")
ap.add_argument("--overwrite", action="store_true", help="time vmult with zero_dst=1 instead of the accumulating cell loop")
ap.add_argument("--operator", choices=["poisson", "helmholtz"], default="poisson", help="helmholtz: step-64's operator on the native fused kernel (seven planes)")
a = ap.parse_args()
p = a.degree
mesh = pkg.BrickMesh(p, a.cells, h=1.0 / a.cells[0], deform_amp=a.deform, cell_block=a.cell_block, dof_numbering=a.numbering, cell_block_order=a.block_order)
quad = pkg.QUAD_GAUSS if a.quadrature == "gauss" else pkg.QUAD_GLL
op = pkg.HelmholtzOperator(mesh, quad, pkg.COEF_STEP64) if a.operator == "helmholtz" else \
    pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64, geometry=pkg.GEOM_AFFINE if a.geometry == 'affine' else pkg.GEOM_MERGED6)
mf = op.mf_data
n = mesh.n_owned
r = mesh.n_cells * (p + 1) ** 3 / n
B_op = 16 + 4 * r + ((56 if a.operator == 'helmholtz' else 48) if a.geometry == 'merged6' else 8) * r
src = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
dst = mf.initialize_dof_vector()
times = {v: [] for v in a.variants}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rnd in range(a.rounds + 1):
    for v in a.variants:
        mf.set_apply_variant(v)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(a.reps):
            if a.overwrite:
                op.vmult(dst, src)
            else:
                mf.cell_loop(op.coef, src, dst)
        ev[1].record()
        torch.cuda.synchronize()
        if rnd:
            times[v].append(ev[0].elapsed_time(ev[1]) / a.reps)
print(f"p={p} cells={a.cells} dofs={n} r={r:.4f} B_op={B_op:.1f} B/DoF quad={a.quadrature}")
for v in a.variants:
    t = np.array(times[v])
    med = np.median(t)
    print(f"variant {v}: median {med:.3f} ms  min {t.min():.3f} ms  -> {n / med / 1e6:.2f} GDoF/s  {B_op * n / med / 1e6:.0f} GB/s alg ({B_op * n / med / 1e6 / 80:.1f}% of 8 TB/s)")
