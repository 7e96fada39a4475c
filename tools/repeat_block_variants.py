"""Repeat block-kernel variants on a brick-ordered mesh with partial bricks (CELLS env, default 86 84 82) against the atomic
pencil kernel: counts wrong or non-reproducible launches.  Used to track down the two block-kernel races of round 1
(profiles/r1/README.md, h_*).  usage: python tools/repeat_block_variants.py 56 59 50 ..."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bp5_pkg
pkg = bp5_pkg.load()
import os
cells = tuple(int(x) for x in os.environ.get('CELLS', '86 84 82').split())
mesh = pkg.BrickMesh(4, cells, h=1.0 / cells[0], deform_amp=0.03, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
n = mesh.n_owned
g = torch.Generator(device="cuda:0").manual_seed(1)
u = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
ref = op.initialize_dof_vector()
op.mf_data.set_apply_variant(3)
op.vmult(ref, u)
tol = 1e-10 * float(ref.abs().max())
for v in [int(x) for x in sys.argv[1:]]:
    op.mf_data.set_apply_variant(v)
    fails = []
    first = None
    for rep in range(12):
        d = op.initialize_dof_vector(); d.fill_(float("nan"))
        op.vmult(d, u)
        if first is None: first = d.clone()
        bad = torch.nonzero(~((d - ref).abs() < tol)).flatten()
        if bad.numel(): fails.append((rep, bad.numel(), bad[:6].tolist()))
        if not torch.equal(d, first) and not bad.numel(): fails.append((rep, "not bitwise equal to first run"))
    print("variant", v, "failures", len(fails), fails[:4], flush=True)
