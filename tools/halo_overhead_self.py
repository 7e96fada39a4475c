"""Per-iteration cost of the halo exchange + all-reduce code path on ONE GPU: the rank-1-of-2 slab of the bench workload with its
ghost plane, exchanged with itself through RCCL (send/recv to self, one-rank all-reduce).  Not an xGMI measurement: it shows the
launch/latency overhead the sequential exchange adds to an iteration."""
import sys, time
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, ".")
import bp5_pkg
pkg = bp5_pkg.load()
p, n = 4, 116
kw = dict(cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
m1 = pkg.BrickMesh(p, (n, n, 2 * n), h=1.0 / n, rank=1, n_ranks=2, **kw)
ng, no = m1.n_ghost, m1.n_owned
mesh = SimpleNamespace(degree=p, n=p + 1, cells=(n, n, 2 * n), n_cells=m1.n_cells, n_interior_cells=m1.n_interior_cells, n_owned=no, n_ghost=ng,
                       n_local=no + ng, n_global_dofs=no, l2g=m1.l2g, coords=m1.coords, global_ids=m1.global_ids, constrained=m1.constrained,
                       n_neighbors=1, neighbor_rank=np.zeros(1, np.int32), send_offsets=np.asarray([0, ng], np.uint32),
                       send_indices=np.arange(no - ng, no, dtype=np.uint32), recv_offsets=np.asarray([0, ng], np.uint32),
                       cell_block_offsets=m1.cell_block_offsets, rank=0, n_ranks=1, h=1.0 / n, deform_amp=0.0)
res = {}
import os
for name in (("slab with ghost plane + exchange",) if os.environ.get("SLAB_ONLY") else ("slab with ghost plane + exchange", "same size, one rank, no exchange")):
    if name.startswith("slab"):
        comm, msh = pkg.Communicator(0, 1), mesh
    else:
        comm, msh = None, pkg.BrickMesh(p, (n, n, n), h=1.0 / n, **kw)
    op = pkg.PoissonOperator(msh, 0, pkg.COEF_STEP64, comm=comm)
    b = op.assemble_rhs()
    x = op.initialize_dof_vector()
    pkg.SolverCGFullMerge(pkg.IterationNumberControl(5, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        ctl = pkg.IterationNumberControl(50, 0.0)
        t0 = time.perf_counter()
        pkg.SolverCGFullMerge(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
    res[name] = best
    print(f"{name}: {best:.3f} ms per iteration, operator {ctl.operator_ms_avg:.3f} ms, variant {op.mf_data.get_apply_variant()}, "
          f"cells {msh.n_cells}, owned {msh.n_owned}, ghosts {msh.n_ghost}", flush=True)
    op.mf_data.close()
if not os.environ.get("SLAB_ONLY"):
  a, b_ = res["slab with ghost plane + exchange"], res["same size, one rank, no exchange"]
  print(f"difference: {(a - b_) * 1e3:.0f} us per iteration ({(a / b_ - 1) * 100:.1f} %)")
