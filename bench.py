#!/usr/bin/env python3
"""BP5 benchmark: DoFs/s per CG iteration on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

A "step" is one CG iteration (operator apply + vector updates + dot products) of the fused
solver on the p=4, ~1e8-DoF synthetic hex mesh; W warm-up iterations, then a solve of exactly K
iterations is timed between barrier + synchronize pairs, max over ranks (the reference's protocol
times the whole cg.solve too: bp5/step-64.cu:442-463).  N > 1 is weak scaling: every rank owns a
z-slab of the same size as the N = 1 problem.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_dof(p, n_cells, n_dofs, G=6, I=1, operator_only=False):
    """B = 16 + I*4*r + G*8*r + 88 with the exact r of the run."""
    r = n_cells * (p + 1) ** 3 / n_dofs
    return 16.0 + I * 4.0 * r + G * 8.0 * r + (0.0 if operator_only else 88.0)


def measured_traffic(key):
    """HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes
    (profiles/*/traffic.json); None when this workload has not been profiled."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    for rnd in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, rnd, "traffic.json")
        if os.path.exists(f):
            for e in json.load(open(f))["entries"]:
                if e["key"] == key:
                    best = e
    return best


def cpu_baseline(p, quad, cells, iters, deform, km):
    """CPU restatement (oracle/bp5_oracle.c, OpenMP) timed on the host cores: a reported
    baseline ("port"), not deal.II and not the optimisation target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bp5_oracle as O
    import c_oracle as CO
    m = O.BrickMesh(p, cells, deform_amp=deform)
    cp = CO.CProblem(p, quad, m.l2g, m.coords, m.constrained, km)
    b = cp.rhs()
    cp.cg_plain(b, 1)
    t0 = time.perf_counter()
    _, k, _ = cp.cg_plain(b, iters)
    dt = time.perf_counter() - t0
    return {"value": m.n_dofs * k / dt, "unit": "DoF/s", "cores": CO.lib().orc_num_threads(), "kind": "port",
            "sample": f"CPU restatement (not deal.II): plain CG, p={p}, {cells[0]}x{cells[1]}x{cells[2]} cells, "
                      f"{m.n_dofs} DoFs, {k} iterations, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--cells", type=int, nargs=3, default=None, help="cells per direction PER GPU (default: ~1e8 DoFs)")
    ap.add_argument("--quadrature", choices=["gauss", "gll"], default="gauss")
    ap.add_argument("--coefficient", choices=["one", "step64"], default="step64")
    ap.add_argument("--deform", type=float, default=0.0)
    ap.add_argument("--variant", choices=["merged", "plain"], default="merged")
    ap.add_argument("--apply-variant", type=int, default=0)
    ap.add_argument("--geometry", choices=["merged6", "affine"], default="merged6",
                    help="merged6: the reference's six stored planes per q-point (G=6, default); affine: per-cell metric + one scalar plane (G=1), affine meshes only")
    ap.add_argument("--cell-block", type=int, nargs=3, default=None,
                    help="hand the cells over in bricks of this many cells (default: 4 4 4 at p=4 -> block-assembled kernel, 8 8 8 at p>=5; "
                         "0 0 0 = lexicographic cell order -> pencil kernel with atomics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import bp5_pkg
    pkg = bp5_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        comm = pkg.Communicator.from_torch_distributed()   # library-side RCCL communicator (id broadcast through torch)

    p = args.degree
    # p = 4: the headline size (config 3: 116^3 cells, 100 544 625 DoFs); other degrees: config 4 (~5e7 DoFs)
    n1 = {1: 367, 2: 184, 3: 122, 4: 116, 5: 73, 6: 61, 7: 52, 8: 46}[p]
    cells_per_gpu = tuple(args.cells) if args.cells else (n1, n1, n1)
    cells = (cells_per_gpu[0], cells_per_gpu[1], cells_per_gpu[2] * world)  # weak scaling: z-slabs
    quad = pkg.QUAD_GAUSS if args.quadrature == "gauss" else pkg.QUAD_GLL
    km = pkg.COEF_STEP64 if args.coefficient == "step64" else pkg.COEF_ONE

    # cell order / DoF numbering are the host's choice (the reference's MatrixFree::reinit reorders cells too): bricks of
    # 4x4x4 cells, parity-class order inside a brick, brick-major DoF numbering -> the library picks its block kernel
    # (p >= 5: the atomic pencil kernel gains 3-6 % from 8x8x8 parity-class bricks, profiles/r1 g_sweep_order_degrees.txt)
    block = tuple(args.cell_block) if args.cell_block else ((4, 4, 4) if p == 4 else (8, 8, 8) if p >= 5 else (0, 0, 0))
    blocked = all(b > 0 for b in block)
    mesh = pkg.BrickMesh(p, cells, h=1.0 / cells[0], deform_amp=args.deform, rank=rank, n_ranks=world,
                         cell_block=block if blocked else (0, 0, 0), dof_numbering=1 if blocked else 0, cell_block_order=1 if blocked else 0)
    G = 6 if args.geometry == "merged6" else 1
    op = pkg.PoissonOperator(mesh, quad, km, device=local_rank, comm=comm,
                             geometry=pkg.GEOM_MERGED6 if G == 6 else pkg.GEOM_AFFINE)
    op.mf_data.set_apply_variant(args.apply_variant)
    b = op.assemble_rhs()
    x = op.initialize_dof_vector()
    Solver = pkg.SolverCGFullMerge if args.variant == "merged" else pkg.SolverCG
    precond = pkg.DiagonalMatrix()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up
    Solver(pkg.IterationNumberControl(max(args.warmup, 1), 0.0), profile=True).solve(op, x, b, precond)
    barrier()
    ctl = pkg.IterationNumberControl(args.steps, 0.0)
    solver = Solver(ctl, profile=True)
    barrier()
    t0 = time.perf_counter()
    solver.solve(op, x, b, precond)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    iters = ctl.last_step()
    n_global = int(mesh.n_global_dofs)
    value = n_global * iters / dt

    # achievable-stream figure: device copy y = 1.0 * x over the solver's vectors, read 8 + write 8 B per entry
    import ctypes as C
    ya, xa = op.initialize_dof_vector(), op.initialize_dof_vector()
    L, hnd = pkg.lib(), op.mf_data.handle
    pv = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):
        L.bp5_vec_equ(hnd, pv(ya), 1.0, pv(xa), mesh.n_owned)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.bp5_vec_equ(hnd, pv(ya), 1.0, pv(xa), mesh.n_owned)
    e1.record()
    torch.cuda.synchronize()
    stream_copy_gbs = 16.0 * mesh.n_owned * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del ya, xa

    if rank == 0:
        n_cells_local, n_dofs_local = mesh.n_cells, mesh.n_owned
        B = algorithmic_bytes_per_dof(p, n_cells_local, n_dofs_local, G=G)
        B_op = algorithmic_bytes_per_dof(p, n_cells_local, n_dofs_local, G=G, operator_only=True)
        apply_s = ctl.apply_ms_avg * 1e-3
        achieved = B_op * n_dofs_local / apply_s / 1e9 if apply_s > 0 else 0.0
        ev = op.mf_data.get_apply_variant()
        key = f"p{p}_{args.quadrature}_{cells_per_gpu[0]}x{cells_per_gpu[1]}x{cells_per_gpu[2]}_{args.geometry}_v{ev}"
        tr = measured_traffic(key) if args.deform == 0.0 else None
        out = {
            "metric": "BP5 DoFs/sec per CG iter (p=4, ~1e8 DoFs) + % HBM roofline at 1/2/4/8 GPUs",
            "value": value, "unit": "DoF/s", "n_gpus": world, "steps": iters, "warmup": args.warmup,
            "ms_per_step": dt / max(iters, 1) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BP5 p={p} {args.quadrature}(p+1) quadrature, {cells[0]}x{cells[1]}x{cells[2]} hex cells, "
                                   f"{n_global} DoFs, coefficient={args.coefficient}, deform={args.deform}, "
                                   f"CG={args.variant} (identity preconditioner), G={G} I=1 ({args.geometry} geometry)",
                       "dofs_per_gpu": n_dofs_local, "parallelism": f"z-slab x{world}",
                       "cell_block": list(block) if blocked else None, "apply_variant": ev},
            "value_per_gpu": value / world,   # the reference's convention divides by the rank count (bp5/step-64.cu:457-461)
            "roofline_cg": {"bytes_per_dof": B, "achieved_GBs_per_gpu": value / world * B / 1e9,
                            "frac_of_hbm_peak": value / world * B / 1e9 / HBM_PEAK_GBS,
                            "stream_copy_GBs": stream_copy_gbs, "frac_of_stream_copy": value / world * B / 1e9 / stream_copy_gbs},
            # `achieved`: algorithmic bytes of ONE operator application (B_op x DoFs) / average duration of the cell kernel
            # (HIP events on the solver's stream around that launch, inside the timed solve); `operator_ms` is the whole
            # application: zero-fill (atomic kernels) + cell kernel + combine pass (owner-scatter kernels); `traffic`: PMC bytes
            "roofline": {"bound": "hbm", "kernel": tr["kernel"] if tr else {0: "apply_pencil_kernel", 10: "apply_team_kernel", 56: "apply_block_kernel<4,false,32,1,288768>"}.get(ev, f"apply variant {ev}"), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr["traffic_bytes"] if tr else None, "algorithmic_bytes": B_op * n_dofs_local,
                         "bytes_per_dof": B_op, "avg_launch_ms": ctl.apply_ms_avg, "launches": ctl.apply_launches,
                         "operator_ms": ctl.operator_ms_avg,
                         "algorithmic_formula": f"16 + I*4r + G*8r B/DoF with I=1, G={G} (SURVEY 8d)"
                                                + ("; the block kernel's own index stream is 2r B/DoF (packed run/offset), local_to_global is not read" if ev == 56 else "")},
        }
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            # bounded sample of the same workload family: ~10 s of host CPU work
            out["cpu_baseline"] = cpu_baseline(p, quad, (48, 48, 48) if p <= 4 else (24, 24, 24), 120, args.deform, km)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
        op.mf_data.close()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
