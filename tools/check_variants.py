#!/usr/bin/env python3
"""Quick A/B correctness of operator variants against a reference variant on one mesh (same process):
usage: python tools/check_variants.py --cells 20 19 9 --cell-block 4 4 1 --ref 3 --variants 56 60"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bp5_pkg
pkg = bp5_pkg.load()
ap = argparse.ArgumentParser()
ap.add_argument("--degree", type=int, default=4)
ap.add_argument("--cells", type=int, nargs=3, default=[20, 19, 9])
ap.add_argument("--cell-block", type=int, nargs=3, default=[4, 4, 4])
ap.add_argument("--ref", type=int, default=3)
ap.add_argument("--variants", type=int, nargs="+", default=[56])
ap.add_argument("--workgroups", type=int, default=0)
ap.add_argument("--deform", type=float, default=0.03)
a = ap.parse_args()
mesh = pkg.BrickMesh(a.degree, a.cells, deform_amp=a.deform, cell_block=a.cell_block, dof_numbering=1, cell_block_order=1)
op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
mf = op.mf_data
print("plan (n_blocks, max_runs, packed):", mf.block_plan_info())
if a.workgroups:
    mf.set_block_workgroups(a.workgroups)
g = torch.Generator(device="cuda:0").manual_seed(3)
src = torch.rand(mesh.n_local, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
outs = {}
for v in [a.ref] + a.variants:
    mf.set_apply_variant(v)
    d = mf.initialize_dof_vector()
    d.fill_(float("nan"))
    op.vmult(d, src)
    d2 = mf.initialize_dof_vector()
    op.vmult(d2, src)
    outs[v] = d
    print(f"variant {v}: rel diff vs {a.ref} = {float((d - outs[a.ref]).abs().max() / outs[a.ref].abs().max()):.3e}, repeatable = {bool(torch.equal(d, d2))}",
          flush=True)
if 56 in outs:
    for v in a.variants:
        print(f"variant {v} bitwise equal to 56: {bool(torch.equal(outs[v], outs[56]))}")
