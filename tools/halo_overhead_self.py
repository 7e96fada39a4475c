"""Per-iteration cost of the halo exchange + all-reduce code path on ONE GPU.

The slab a rank > 0 owns in an N-rank run of the bench workload (default: rank 3 of 8 of the 116x116x120 problem = BASELINE
config 3 under strong scaling, ~1.26e7 DoFs) with its ghost plane exchanged with ITSELF through RCCL (send/recv to self,
one-rank all-reduce): one plane out and one plane in per exchange, like a middle rank.  Not an xGMI measurement: it shows
the launch / latency / scheduling overhead the exchange adds to an iteration, overlapped (the library default, exchange on
the communication stream under the interior cells) against sequential (bp5_mf_set_overlap(0)) against a mesh of the same
size without any exchange.

  python tools/halo_overhead_self.py [--ranks 8] [--rank 3] [--weak] [--iters 50]
"""
import argparse
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bp5_pkg

pkg = bp5_pkg.load()

def consistent_self_glue(m1):
    """send indices for a self-neighbour exchange: ghost DoF j is glued to an owned DoF with the SAME Dirichlet status (as on a real
    partitioned mesh, where a ghost is a copy of its owner): free ghosts to the last free owned DoFs, Dirichlet ghosts to Dirichlet ones"""
    no, ng = m1.n_owned, m1.n_ghost
    con = np.zeros(no + ng, bool)
    con[m1.constrained.astype(np.int64)] = True
    free_owned = np.nonzero(~con[:no])[0][::-1]
    dir_owned = np.nonzero(con[:no])[0][::-1]
    send, kf, kd = np.zeros(ng, np.uint32), 0, 0
    for j in range(ng):
        if con[no + j]:
            send[j] = dir_owned[kd]; kd += 1
        else:
            send[j] = free_owned[kf]; kf += 1
    return send
ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--weak", action="store_true", help="every rank owns 116 cell layers (round-1 weak-scaling shape) instead of 116/ranks")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--solver", choices=["merged", "plain"], default="merged")
ap.add_argument("--cell-block", type=int, nargs=3, default=[4, 4, 4])
ap.add_argument("--modes", default="auto,overlapped,launches,sequential,none", help="comma list of: overlapped (boundary-first inside one launch), launches (boundary-first, two launches), sequential, none")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cells", type=int, nargs=3, default=[116, 116, 120], help="cells of the WHOLE problem (the slab of --rank of --ranks is cut from it)")
args = ap.parse_args()
p = 4
n, ny_, nz_ = args.cells
kw = dict(cell_block=tuple(args.cell_block), dof_numbering=1, cell_block_order=1)
nz = nz_ * args.ranks if args.weak else nz_
m1 = pkg.BrickMesh(p, (n, ny_, nz), h=1.0 / n, rank=args.rank, n_ranks=args.ranks, **kw)
ng, no = m1.n_ghost, m1.n_owned
layers = m1.n_cells // (n * ny_)
mesh = SimpleNamespace(degree=p, n=p + 1, cells=(n, ny_, nz), n_cells=m1.n_cells, n_interior_cells=m1.n_interior_cells, n_owned=no, n_ghost=ng,
                       n_local=no + ng, n_global_dofs=no, l2g=m1.l2g, coords=m1.coords, global_ids=m1.global_ids, constrained=m1.constrained,
                       n_neighbors=1, neighbor_rank=np.zeros(1, np.int32), send_offsets=np.asarray([0, ng], np.uint32),
                       send_indices=consistent_self_glue(m1), recv_offsets=np.asarray([0, ng], np.uint32),
                       cell_block_offsets=m1.cell_block_offsets, rank=0, n_ranks=1, h=1.0 / n, deform_amp=0.0)
Solver = pkg.SolverCGFullMerge if args.solver == "merged" else pkg.SolverCG
res = {}
all_modes = {"auto": "slab + exchange, automatic", "overlapped": "slab + exchange, overlapped", "launches": "slab + exchange, overlapped (two launches)",
             "sequential": "slab + exchange, sequential", "none": "same size, one rank, no exchange"}
for name in [all_modes[m] for m in args.modes.split(",")]:
    if name.startswith("slab"):
        comm, msh = pkg.Communicator(0, 1), mesh
    else:
        comm, msh = None, pkg.BrickMesh(p, (n, ny_, layers), h=1.0 / n, **kw)
    op = pkg.PoissonOperator(msh, 0, pkg.COEF_STEP64, comm=comm)
    op.mf_data.set_tuning("boundary_first", 0 if "two launches" in name else 1)
    if name.startswith("slab"):
        op.mf_data.set_overlap(0 if name.endswith("sequential") else 2 if name.endswith("automatic") else 1)
    b = op.assemble_rhs()
    x = op.initialize_dof_vector()
    Solver(pkg.IterationNumberControl(5, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(args.reps):
        ctl = pkg.IterationNumberControl(args.iters, 0.0)
        t0 = time.perf_counter()
        Solver(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / args.iters * 1e3)
    res[name] = best
    info = op.mf_data.block_plan_info() if op.mf_data.get_apply_variant() == 56 else None
    print(f"{name}: plan (bricks, max runs, packed) {info}, dot products fused {ctl.dot_products_fused}")
    print(f"{name}: exchange schedule {ctl.exchange_schedule}, kernel {ctl.apply_kernel}")
    print(f"{name}: {best:.3f} ms per iteration ({msh.n_owned / best / 1e6:.2f} GDoF/s), operator {ctl.operator_ms_avg:.3f} ms, "
          f"variant {op.mf_data.get_apply_variant()}, cells {msh.n_cells}, owned {msh.n_owned}, ghosts {msh.n_ghost}", flush=True)
    op.mf_data.close()
    if comm is not None:
        comm.close()
ref = res.get("same size, one rank, no exchange")
for k in ("slab + exchange, automatic", "slab + exchange, overlapped", "slab + exchange, overlapped (two launches)", "slab + exchange, sequential"):
    if ref is None or k not in res:
        continue
    print(f"{k}: +{(res[k] - ref) * 1e3:.0f} us per iteration ({(res[k] / ref - 1) * 100:.1f} %) over the mesh without exchange")
