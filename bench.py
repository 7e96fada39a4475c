#!/usr/bin/env python3
"""BP5 benchmark: DoFs/s per CG iteration on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is one CG iteration (operator apply + vector updates + dot products) of the fused solver on the
p=4, 116x116x120-cell, 104 004 225-DoF synthetic hex mesh (BASELINE config 3; at N = 1 the whole problem sits on one
GPU; rounds 1-3: 116^3 cells -- 120 layers split into 2, 4 and 8 z-slabs of EQUAL height, 116 gave one of eight ranks 15 layers
against a mean of 14.5; `--cells 116 116 116` is the old mesh, same DoF/s at N = 1: profiles/r4).  W warm-up iterations, then a solve of exactly K iterations is timed between barrier + synchronize pairs,
max over ranks (the reference times the whole cg.solve too: bp5/step-64.cu:442-463).  Prints ONE JSON line on rank 0.

N > 1: one rank per GPU, RCCL over xGMI.  Either the caller starts the ranks (torch.distributed.run sets
RANK/WORLD_SIZE), or this script starts them itself -- as a CHILD process, before anything here touches a GPU.
Default = STRONG scaling: the SAME 116x116x120 problem split into N z-slabs (the reference reports the throughput of one
~1e8-DoF problem spread over its ranks, bp5/step-64.cu:457-461); `--scaling weak` gives every rank a 116x116x120 slab.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HEADLINE_CELLS = (116, 116, 120)  # BASELINE config 3 ("p = 4, ~1e8 DoFs"): 465 x 465 x 481 = 104 004 225 DoFs; 120 cell layers = 2 x 60 = 4 x 30 = 8 x 15


def algorithmic_bytes_per_dof(p, n_cells, n_dofs, G=6, I=1, operator_only=False):
    """B = 16 + I*4*r + G*8*r + 88 with the exact r of the run."""
    r = n_cells * (p + 1) ** 3 / n_dofs
    return 16.0 + I * 4.0 * r + G * 8.0 * r + (0.0 if operator_only else 88.0)


def measured_traffic(key, kernel):
    """HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes (profiles/*/traffic.json, the latest
    round that profiled THIS kernel build on this workload wins); None when it has not been profiled."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    for rnd in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, rnd, "traffic.json")
        if os.path.exists(f):
            for e in json.load(open(f))["entries"]:
                if e["key"] == key and e.get("kernel") == kernel:
                    best = dict(e, source=f"profiles/{rnd}/traffic.json")
    return best


def live_traffic(kernel, workload_args, timeout_s=90):
    """HBM bytes per launch of `kernel`, measured NOW: two child passes of this very script under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE: one counter per pass, `--kernel-trace` only, the program directly after `--`), 3 CG iterations each, counters averaged over
    the kernel's launches; FETCH_SIZE x 2 on gfx950, both in KiB (MI355X_MICROARCH.md, HBM section).  None if the profiler is not there or a
    pass fails: the caller then falls back to the committed passes of profiles/*/traffic.json."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):
        return None    # this invocation is being profiled itself: no nested profiler
    want = kernel.replace(" ", "")
    kib = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bp5_pmc_", dir="/tmp")
        try:
            # the program after `--` is the running interpreter itself (a real ELF binary: no shim that would exec under the profiler's preload)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", os.path.realpath(sys.executable),
                   os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-mesh-116", "--sustained-iters", "0",
                   "--no-traffic-pass"] + workload_args
            child = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)
            try:
                rc = child.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)    # the profiler AND the python grandchild that holds the GPU (its own session: nothing else)
                child.wait()
                return None
            files = glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if want in row["Kernel_Name"].replace(" ", "") and row.get("Counter_Name", counter) == counter]
            if not vals:
                return None
            kib[counter] = sum(vals) / len(vals)
        except (OSError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"traffic_bytes": int(round((2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024.0)), "fetch_size_kib": kib["FETCH_SIZE"],
            "write_size_kib": kib["WRITE_SIZE"],
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this bench invocation (3 iterations each; FETCH x2 on gfx950, KiB)"}


def cpu_baseline(mesh, p, quad, km, budget_s, max_iters):
    """CPU restatement (oracle/bp5_oracle.c, OpenMP) timed on the host cores ON THE BENCH'S OWN MESH (same cells, same
    DoF numbering, same coefficient): a reported baseline ("port"), not deal.II and not the optimisation target.
    Bounded: one timed iteration sizes a plain-CG run of about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as CO
    cp = CO.CProblem(p, quad, mesh.l2g, mesh.coords, mesh.constrained, km)
    b = cp.rhs()
    t0 = time.perf_counter()
    cp.cg_plain(b, 1)
    t1 = time.perf_counter() - t0
    iters = int(max(2, min(max_iters, budget_s / max(t1, 1e-6))))
    t0 = time.perf_counter()
    _, k, _ = cp.cg_plain(b, iters)
    dt = time.perf_counter() - t0
    c = mesh.cells
    return {"value": mesh.n_owned * k / dt, "unit": "DoF/s", "cores": CO.lib().orc_num_threads(), "kind": "port",
            "sample": f"CPU restatement (not deal.II) on the bench's own mesh: plain CG, p={p}, {c[0]}x{c[1]}x{c[2]} cells, "
                      f"{mesh.n_owned} DoFs, {k} iterations, {dt:.1f} s"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run BEFORE this
    process imports torch or touches a GPU (a process that has initialised the GPU must never be replaced), relay the
    child's output and exit with its code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


PHASES = ["update", "gather_wait", "operator", "exchange", "reduce_local", "allreduce", "control", "iteration"]   # bp5.h BP5_PHASE_*
SCHEDULES = {0: "none (one rank)", 1: "unsplit", 2: "boundary-first", 3: "three-phase", 4: "under-combine"}       # bp5_cg_result.exchange_schedule
CONFIG_SIZES = {1: 367, 2: 184, 3: 122, 4: 92, 5: 73, 6: 61, 7: 52, 8: 46}    # BASELINE config 4: ~5e7 DoFs per degree


def default_cell_block(p, cells_per_rank):
    """Cell order / DoF numbering are the host's choice (the reference's MatrixFree::reinit reorders cells too): bricks sized for the
    block kernel's LDS accumulator, parity-class order inside a brick, brick-major DoF numbering -> the library picks its deterministic
    block kernel.  Small problems (config 2's 54^3; the 116x116x14.5 slab of one of 8 ranks): 4x4x2 bricks give the persistent
    workgroups twice as many bricks to balance (profiles/r2: 0.427 vs 0.439 ms per iteration at 54^3).  p = 2, 5 (round 3: n^2 lanes
    per cell, 28 / 7 cells per pass -- bricks whose parity classes fill a pass: 8x8x4, 6x4x2) run the deterministic block kernel too, on
    par with the atomic pencil kernel (profiles/r3 i_*); p = 8: the atomic pencil kernel is still 17 % ahead of the block kernel and
    gains 3-6 % from 8x8x8 parity-class bricks (`--cell-block 2 2 2 --apply-variant 56` runs the deterministic kernel there)."""
    small = cells_per_rank < 400000
    return {1: (8, 8, 8), 2: (8, 8, 4), 3: (8, 4, 4), 4: (4, 4, 2) if small else (4, 4, 4), 5: (6, 4, 2), 6: (4, 4, 2), 7: (4, 2, 2), 8: (8, 8, 8)}.get(p, (0, 0, 0))


def default_numbering(p, blocked):
    """DoF numbering handed to the library (bp5_mesh_desc.dof_numbering; the operator accepts any): block-major with cell bricks (the deterministic block kernel's
    lattice blocks).  `--numbering 2` (cell interiors first: the atomic pencil kernel of p >= 5 then stores them plainly) was measured at p = 8 in round 4 and buys
    nothing there -- 1.130 against 1.125 ms, the kernel is bound by its arithmetic, not by the count of its atomics any more (profiles/r4 o_*)."""
    return 1 if blocked else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed CG iterations (default 50; --config 1: 10)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4, 5], default=None,
                    help="BASELINE.json configuration as ONE flag: 1 = p 2, 8^3 cells, 10 iterations (plumbing); 2 = p 4, 54^3 (1.02e7 DoFs), "
                         "variable coefficient; 3 = p 4, 116x116x120 (1.04e8 DoFs; the default, what --gpus N splits into equal z-slabs); 4 = degree sweep p = 1..8 at ~5e7 "
                         "DoFs (value = the degree of --degree, all eight under 'sweep'); 5 = p 6, 61^3, deformed mesh")
    ap.add_argument("--degree", type=int, default=None)
    ap.add_argument("--cells", type=int, nargs=3, default=None,
                    help="cells per direction of the WHOLE problem (strong scaling) / per GPU (weak scaling); default: config 3 / 4 sizes")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the same problem split into N z-slabs (default, BASELINE config 3); weak = one such problem per GPU")
    ap.add_argument("--quadrature", choices=["gauss", "gll"], default="gauss")
    ap.add_argument("--coefficient", choices=["one", "step64"], default="step64")
    ap.add_argument("--deform", type=float, default=None)
    ap.add_argument("--variant", choices=["merged", "plain"], default="merged")
    ap.add_argument("--operator", choices=["poisson", "helmholtz"], default="poisson",
                    help="poisson: BP5 (the metric); helmholtz: step-64's (grad v, grad u) + (v, a u) on the native fused kernel (SURVEY 8 f1: seven planes, G = 7)")
    ap.add_argument("--apply-variant", type=int, default=0)
    ap.add_argument("--overlap", type=int, choices=[0, 1, 2], default=2,
                    help="N > 1: halo-exchange schedule of the timed solve (bp5_mf_set_overlap): 0 unsplit, 1 boundary-first, 2 the library decides (exchange under the owned-row combine)")
    ap.add_argument("--geometry", choices=["merged6", "affine"], default="merged6",
                    help="merged6: the reference's six stored planes per q-point (G=6, default); affine: per-cell metric + one scalar plane (G=1), affine meshes only")
    ap.add_argument("--cell-block", type=int, nargs=3, default=None,
                    help="hand the cells over in bricks of this many cells (default: 4 4 4 at p=4 -> block-assembled kernel, 8 8 8 at p>=5; "
                         "0 0 0 = lexicographic cell order -> pencil kernel with atomics)")
    ap.add_argument("--numbering", type=int, choices=[0, 1, 2], default=None,
                    help="bp5_mesh_desc.dof_numbering: 0 lexicographic, 1 block-major (needs cell bricks), 2 cell interiors first; default: 1 with bricks, 2 at p = 8")
    ap.add_argument("--sustained-iters", type=int, default=200, help="reference protocol: iterations per repetition (bp5/step-64.cu:729); 0 = skip")
    ap.add_argument("--sustained-reps", type=int, default=3, help="reference protocol: repetitions, best one reported (bp5/step-64.cu:457-463)")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep-other-quadrature", action="store_true", help="--config 4: only the quadrature of --quadrature per degree (default: both, the other one under its name)")
    ap.add_argument("--no-mesh-116", action="store_true", help="N = 1, default mesh: skip the extra solve on the 116^3 mesh of rounds 1-3 (reported as mesh_116_cubed)")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic (N = 1 only); the committed passes of "
                         "profiles/*/traffic.json are quoted instead.  Needed under an outer profiler.")
    ap.add_argument("--post-budget", type=float, default=None,
                    help="seconds the legs AFTER the timed region may take (sustained solves, unfused solve, exchange A/B, copy stream, traffic passes, CPU "
                         "baseline); then the line is printed with what is there and the process exits with code 4 (default: 600 at N = 1, 240 at N > 1) -- a hang in a "
                         "diagnostic leg of the first real multi-GPU run must not cost the measurement")
    ap.add_argument("--control-backend", choices=["gloo", "nccl"], default="gloo",
                    help="N > 1: torch.distributed backend of the bench's CONTROL plane (barriers, max over ranks of the wall clock, the id of the library's "
                         "communicator); the data path is the library's own RCCL communicator either way")
    ap.add_argument("--no-exchange-ab", action="store_true",
                    help="N > 1: skip the diagnostic solves after the timed region (unsplit vs boundary-first exchange, phase stamps)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="control-flow rehearsal of an N > 1 run on ONE GPU (tests only): every rank on cuda:0, torch.distributed over gloo, "
                         "needs BP5_LIB = libbp5_loopback.so (RCCL refuses two ranks on one device); the JSON line is marked, its numbers mean nothing")
    ap.add_argument("--dry-run", action="store_true",
                    help="host only (no GPU, no process group): every rank builds its slab of the mesh and prints its partition as one JSON line")
    args = ap.parse_args()
    # BASELINE configurations as one flag each (explicit flags still win)
    cfg = {1: dict(degree=2, cells=[8, 8, 8], steps=10, deform=0.0), 2: dict(degree=4, cells=[54, 54, 54], deform=0.0),
           3: dict(degree=4, cells=list(HEADLINE_CELLS), deform=0.0), 4: dict(deform=0.0), 5: dict(degree=6, cells=[61, 61, 61], deform=0.05)}.get(args.config, {})
    for k, v in cfg.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    args.degree = 4 if args.degree is None else args.degree
    args.steps = 50 if args.steps is None else args.steps
    args.deform = 0.0 if args.deform is None else args.deform

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    # host threads of this rank's set-up (mesh generation, index plans: OpenMP in libbp5.so).  torch.distributed.run exports OMP_NUM_THREADS=1
    # to its workers unless the caller has set it, which makes every rank build its slab on ONE core (11.6 s instead of a few for the mesh of
    # one of two ranks): give each rank its share of the node's cores instead.  BP5_HOST_THREADS overrides; must happen before the library loads.
    n_ranks_env = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BP5_HOST_THREADS"):
        os.environ["OMP_NUM_THREADS"] = os.environ["BP5_HOST_THREADS"]
    elif n_ranks_env > 1 and os.environ.get("OMP_NUM_THREADS", "1") == "1":
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // n_ranks_env))

    import bp5_pkg
    pkg = bp5_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    def die(stage, err):
        """a rank-tagged, stage-tagged message before the non-zero exit: what a failed first multi-GPU run needs to be diagnosed from its log"""
        sys.stderr.write(f"[bench rank {rank}/{world} local_rank {local_rank}] FAILED in {stage}: {type(err).__name__}: {err}\n")
        sys.stderr.flush()
        os._exit(3)   # (no atexit collectives: the other ranks may be stuck in theirs)

    if args.rehearsal:
        if not os.environ.get("BP5_LIB", "").endswith("libbp5_loopback.so"):
            raise SystemExit("--rehearsal needs BP5_LIB=.../libbp5_loopback.so")
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    comm = None
    if not args.dry_run:
        import torch
        try:
            torch.cuda.set_device(local_rank)
        except Exception as e:   # noqa: BLE001
            die(f"torch.cuda.set_device({local_rank}) [visible devices: {torch.cuda.device_count()}]", e)
    if world > 1 and not args.dry_run:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed carries ONLY the bench's control plane (the barrier around the timed region, max / min over ranks of a float, the 128-byte
        # id of the library's communicator).  The data path -- halo exchange and the per-iteration all-reduce -- is the library's own RCCL communicator.
        # Default `--control-backend gloo`: then that communicator is the only RCCL communicator on the device (a second one, torch's, would meet it for
        # the first time in the first real multi-GPU run), and it is the combination the N > 1 rehearsals of this repository have run; `nccl` on request.
        control = "gloo" if args.rehearsal else args.control_backend
        if control == "gloo":
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: the container's hostname may not resolve
            try:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            except Exception as e:   # noqa: BLE001
                # (every rank of one node fails here alike -- no interface, no rendezvous --, so all of them fall back together)
                sys.stderr.write(f"[bench rank {rank}/{world}] gloo control plane unavailable ({type(e).__name__}: {e}); falling back to nccl\n")
                if args.rehearsal:
                    die("torch.distributed.init_process_group (backend gloo)", e)
                control = "nccl"
        if control == "nccl":
            try:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
            except Exception as e:   # noqa: BLE001
                die("torch.distributed.init_process_group (backend nccl: rendezvous / RCCL bootstrap of the bench's control plane)", e)
        control_on_cpu = control == "gloo"
        args.control_backend = control
        try:
            comm = pkg.Communicator.from_torch_distributed()   # library-side RCCL communicator (id broadcast through torch)
        except Exception as e:   # noqa: BLE001
            die("bp5_comm_create (ncclCommInitRank of the library's communicator; the unique id travels through torch.distributed)", e)

    t_setup0 = time.perf_counter()
    p = args.degree
    # p = 4: the headline size (config 3: 116x116x120 cells, 104 004 225 DoFs); other degrees: config 4 (~5e7 DoFs)
    n1 = {**CONFIG_SIZES, 4: 92}[p]
    base = tuple(args.cells) if args.cells else (HEADLINE_CELLS if (p == 4 and args.config != 4) else (n1, n1, n1))
    strong = args.scaling == "strong"
    cells = base if strong else (base[0], base[1], base[2] * world)   # z-slabs either way
    quad = pkg.QUAD_GAUSS if args.quadrature == "gauss" else pkg.QUAD_GLL
    km = pkg.COEF_STEP64 if args.coefficient == "step64" else pkg.COEF_ONE
    G = (7 if args.operator == "helmholtz" else 6) if args.geometry == "merged6" else 1
    Solver = pkg.SolverCGFullMerge if args.variant == "merged" else pkg.SolverCG
    precond = pkg.DiagonalMatrix()

    def build(p_, cells_, block_=None):
        """mesh of this rank + operator (host work: mesh generation, index plans, upload; device: the merged metric)"""
        blk = tuple(block_) if block_ else default_cell_block(p_, cells_[0] * cells_[1] * cells_[2] // (world if strong else 1))
        blocked_ = all(b > 0 for b in blk)
        mesh_ = pkg.BrickMesh(p_, cells_, h=1.0 / cells_[0], deform_amp=args.deform, rank=rank, n_ranks=world,
                              cell_block=blk if blocked_ else (0, 0, 0), dof_numbering=default_numbering(p_, blocked_) if args.numbering is None else args.numbering,
                              cell_block_order=1 if blocked_ else 0)
        return mesh_, blk, blocked_

    mesh, block, blocked = build(p, cells, args.cell_block)
    if args.dry_run:
        print(json.dumps({"rank": rank, "world": world, "scaling": args.scaling, "cells": list(cells), "n_cells": int(mesh.n_cells),
                          "n_interior_cells": int(mesh.n_interior_cells), "n_owned": int(mesh.n_owned), "n_ghost": int(mesh.n_ghost),
                          "n_global_dofs": int(mesh.n_global_dofs), "neighbors": [int(r) for r in mesh.neighbor_rank]}), flush=True)
        return
    try:
        if args.operator == "helmholtz":
            op = pkg.HelmholtzOperator(mesh, quad, km, device=local_rank, comm=comm)
        else:
            op = pkg.PoissonOperator(mesh, quad, km, device=local_rank, comm=comm,
                                     geometry=pkg.GEOM_MERGED6 if G == 6 else pkg.GEOM_AFFINE)
        op.mf_data.set_apply_variant(args.apply_variant)
        op.mf_data.set_overlap(args.overlap)
        b = op.assemble_rhs()
        x = op.initialize_dof_vector()
        torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001
        die("operator set-up (bp5_mf_create / merged metric / RHS with compress(add): the FIRST halo exchange of the run)", e)
    setup_s = time.perf_counter() - t_setup0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_ranks(v, how):
        if world == 1:
            return v
        import torch.distributed as dist
        tt = torch.tensor([float(v)], dtype=torch.float64, device="cpu" if control_on_cpu else f"cuda:{local_rank}")
        dist.all_reduce(tt, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[how])
        return float(tt.item())

    def timed_solve(iters, op_=None, x_=None, b_=None, profile=True):
        """one solve of exactly `iters` iterations between barrier + synchronize pairs; wall clock, max over ranks"""
        op_, x_, b_ = op if op_ is None else op_, x if x_ is None else x_, b if b_ is None else b_
        ctl_ = pkg.IterationNumberControl(iters, 0.0)
        sv = Solver(ctl_, profile=profile)
        barrier()
        t0_ = time.perf_counter()
        sv.solve(op_, x_, b_, precond)
        barrier()
        return ctl_, reduce_ranks(time.perf_counter() - t0_, "max")

    # warm-up: also the first solve with its gather / scatter-add groups and all-reduces
    try:
        Solver(pkg.IterationNumberControl(max(args.warmup, 1), 0.0), profile=True).solve(op, x, b, precond)
        barrier()
    except Exception as e:   # noqa: BLE001
        die("warm-up solve (first RCCL send/recv groups of the operator and first 7-double all-reduce)", e)
    ctl, dt = timed_solve(args.steps)
    iters = ctl.last_step()
    n_global = int(mesh.n_global_dofs)
    value = n_global * iters / dt

    # Everything below is reporting around the measurement above.  The legs fill `extras`; compose() (defined further down) builds the line
    # from whatever is there.  A watchdog prints the line and ends the process (every rank, exit code 4: the measurement is on stdout, the
    # overrun -- possibly a hung GPU leg -- shows in the run record) if the legs take longer than --post-budget: a diagnostic leg that hangs
    # (a collective only some ranks entered, a schedule the node does not support) must not cost the run its measurement.
    import copy
    import threading

    class Extras:
        """what the reporting legs have produced so far; the watchdog thread composes the line from a snapshot taken under the lock (legs
        publish finished values only: a nested list or dict is re-assigned as a copy, never mutated in place)"""
        def __init__(self, **kw):
            self._d, self._lock = dict(kw), threading.Lock()
        def __setitem__(self, k, v):
            with self._lock:
                self._d[k] = v
        def __getitem__(self, k):
            with self._lock:
                return self._d[k]
        def snapshot(self):
            with self._lock:
                return copy.deepcopy(self._d)
    extras = extras_live = Extras(stage="sustained solves", fused_all=bool(ctl.dot_products_fused), fused_any=bool(ctl.dot_products_fused))
    post_budget = args.post_budget if args.post_budget is not None else (600.0 if world == 1 else 240.0)
    compose_ref, printed = [], threading.Lock()
    apply_variant_used = op.mf_data.get_apply_variant()   # (read here: compose() may run on the watchdog thread and must not enter the library)

    def watchdog_fire():
        if rank == 0 and compose_ref and printed.acquire(blocking=False):
            try:
                print(json.dumps(compose_ref[0](False)), flush=True)
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench rank 0] watchdog could not compose the line: {type(e).__name__}: {e}\n")
        sys.stderr.write(f"[bench rank {rank}/{world}] post-processing exceeded {post_budget:.0f} s in stage '{extras['stage']}': line printed without it, exiting with code 4\n")
        sys.stderr.flush()
        os._exit(4)
    watchdog = threading.Timer(post_budget, watchdog_fire)
    def compose(final):
        """the JSON line from the timed solve and whatever the later legs have put into `extras` (watchdog: final = False)"""
        extras = extras_live.snapshot()
        n_cells_local, n_dofs_local = mesh.n_cells, mesh.n_owned
        r = n_cells_local * (p + 1) ** 3 / n_dofs_local
        B = algorithmic_bytes_per_dof(p, n_cells_local, n_dofs_local, G=G)
        B_op = algorithmic_bytes_per_dof(p, n_cells_local, n_dofs_local, G=G, operator_only=True)
        apply_s = ctl.apply_ms_avg * 1e-3
        ev = apply_variant_used
        block_kernel = ctl.apply_kernel.startswith("apply_block_kernel")
        # SolverCGFullMerge on the packed block kernel: the dot products of update_b (contract: "dot reads p,r,v",
        # 24 B/DoF of the formula's 88) are formed inside the operator's write-out (reported by the solve itself)
        fused = bool(ctl.dot_products_fused)
        B_kernel = B_op + (24.0 if fused else 0.0)
        achieved = B_kernel * n_dofs_local / apply_s / 1e9 if apply_s > 0 else 0.0
        # what the launched kernel has to move for its own representation (block kernel: one packed u16 per cell-local DoF instead of the
        # 4r of local_to_global; fused: r at the stored DoFs only -- p.v comes from the quadrature-point energy, v.v from LDS)
        kname = ctl.apply_kernel or f"apply variant {ev}"     # reported by the solve: the kernel it launched, as a profiler prints it
        # index bytes the block kernel reads: one packed u16 per cell-local DoF, or -- lattice build (kernel id bit 16777216): closed-form
        # indices -- one u32 per CELL
        try:
            lattice_kernel = bool(int(kname.split(",")[-1].rstrip(">")) & 16777216)
        except ValueError:
            lattice_kernel = False
        idx_moved = 4.0 * r / (p + 1) ** 3 if lattice_kernel else 2.0 * r
        B_moved = (16.0 + idx_moved + G * 8.0 * r + (8.0 if fused else 0.0)) if block_kernel else B_kernel
        sustained, unfused_ms, exchange_ab, sweep = extras.get("sustained"), extras.get("unfused_ms"), extras.get("exchange_ab"), extras.get("sweep")
        fused_all, fused_any, stream_copy_gbs, tr = extras["fused_all"], extras["fused_any"], extras.get("stream_copy_gbs"), extras.get("traffic")
        out = {
            "metric": "BP5 DoFs/sec per CG iter (p=4, ~1e8 DoFs) + % HBM roofline at 1/2/4/8 GPUs",
            "value": value, "unit": "DoF/s", "n_gpus": world, "steps": iters, "warmup": args.warmup,
            "ms_per_step": dt / max(iters, 1) * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("step-64 Helmholtz operator (NOT the BP5 metric's operator; SURVEY 8 f1), " if args.operator == "helmholtz" else "") +
                                   f"BP5 p={p} {args.quadrature}(p+1) quadrature, {cells[0]}x{cells[1]}x{cells[2]} hex cells, "
                                   f"{n_global} DoFs, coefficient={args.coefficient}, deform={args.deform}, "
                                   f"CG={args.variant} (identity preconditioner), G={G} I=1 ({args.geometry} geometry)",
                       "baseline_config": args.config if args.config else (3 if (p == 4 and base == HEADLINE_CELLS) else None),
                       "dofs_per_gpu": n_dofs_local, "parallelism": f"z-slab x{world} ({args.scaling} scaling)",
                       "collectives": None if world == 1 else ("data path: the library's RCCL communicator (halo send/recv, one 7-double all-reduce per iteration); "
                                                               f"bench control plane (barrier, max over ranks): torch.distributed over {'gloo' if (args.rehearsal or args.control_backend == 'gloo') else 'nccl'}"),
                       "cell_block": list(block) if blocked else None, "apply_variant": ev, "cg_dot_products_fused": fused,
                       "cg_dot_products_fused_on_every_rank": fused_all, "cg_dot_products_fused_on_some_rank": fused_any,
                       "exchange_schedule": SCHEDULES.get(ctl.exchange_schedule, "?"), "overlap_policy": args.overlap},
            "value_per_gpu": value / world,   # the reference's convention divides by the rank count (bp5/step-64.cu:457-461)
            # the all-reduced residual norms of the TIMED solve (x0 = 0, exactly `steps` iterations): the same problem at every N, so the lines of an
            # N = 1, 2, 4, 8 sweep must agree in these to rounding (the partition only changes summation orders) -- parity across ranks on the
            # hardware the sweep runs on, for free
            "solve_check": {"initial_residual": ctl.initial_value(), "residual_after_timed_solve": ctl.last_value(), "iterations": iters,
                            "note": "strong scaling: identical across N up to rounding (relative 1e-10)"},
            "host_setup_s": setup_s,          # this rank: mesh generation, index plans, upload, merged metric, RHS (every rank does its own slab in parallel)
            # whole iteration against the roofline, priced by the CONTRACT FORMULA of SURVEY 8(d) (16 + 4r + 48r + 88 B/DoF): a formula-based
            # figure, not a measured bandwidth -- the fused iteration moves fewer bytes than the formula credits (2r index stream, 8 B instead
            # of 24 B for the dot products), which is why frac_of_stream_copy can exceed 1; bytes_moved_per_dof prices what is really moved
            "roofline_cg": {"basis": "contract formula (SURVEY 8d), not measured bytes", "bytes_per_dof": B,
                            "achieved_GBs_per_gpu": value / world * B / 1e9, "frac_of_hbm_peak": value / world * B / 1e9 / HBM_PEAK_GBS,
                            "stream_copy_GBs": stream_copy_gbs, "frac_of_stream_copy": (value / world * B / 1e9 / stream_copy_gbs) if stream_copy_gbs else None,
                            # merged CG as implemented: update kernels 40 / 56 B/DoF alternating (the operator overwrites v: no v write, x every
                            # second iteration) = 48 on average instead of the formula's 64; separate dot-product pass 24 when not fused
                            "bytes_moved_per_dof": (B_moved + 48.0 + (0.0 if fused else 24.0)) if args.variant == "merged" else None,
                            "frac_moved_of_hbm_peak": (value / world * (B_moved + 48.0 + (0.0 if fused else 24.0)) / 1e9 / HBM_PEAK_GBS) if args.variant == "merged" else None},
            # `achieved`: algorithmic bytes of ONE launch of the dominant kernel / its average duration (HIP events on the solver's
            # stream around that launch, inside the timed solve).  Unfused: one operator application, B_op x DoFs of this rank.
            # Fused (default at p = 4 on bricks): the same kernel also does the solver's dot-product pass, so its algorithmic
            # bytes are B_op + 24 B/DoF (the contract formula's "dot reads p, r, v"); `frac_operator_only` prices the same
            # duration against B_op alone, `frac_moved` against the bytes this kernel's representation has to move.
            # `operator_ms`: zero-fill (atomic kernels) + cell kernel + combine pass.
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr["traffic_bytes"] if tr else None,
                         "traffic_source": tr["source"] if tr else None,
                         "algorithmic_bytes": B_kernel * n_dofs_local,
                         "bytes_per_dof": B_kernel, "avg_launch_ms": ctl.apply_ms_avg, "launches": ctl.apply_launches,
                         "operator_ms": ctl.operator_ms_avg,
                         "frac_operator_only": B_op * n_dofs_local / apply_s / 1e9 / HBM_PEAK_GBS if apply_s > 0 else 0.0,
                         "bytes_moved_per_dof": B_moved,
                         "frac_moved": B_moved * n_dofs_local / apply_s / 1e9 / HBM_PEAK_GBS if apply_s > 0 else 0.0,
                         # the same operator kernel WITHOUT the fused dot products (10 untimed iterations after the timed region): the
                         # SURVEY 8(d) operator-only figure, B_op x DoFs / its average launch duration
                         "operator_kernel_unfused": ({"avg_launch_ms": unfused_ms, "bytes_per_dof": B_op,
                                                      "frac": B_op * n_dofs_local / (unfused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS} if unfused_ms else None),
                         "algorithmic_formula": (f"16 + I*4r + G*8r" + (" + 24 [fused dot products: p, r, v]" if fused else "") +
                                                 f" B/DoF with I=1 (contract)" + ("; the kernel itself reads one u32 per CELL" if lattice_kernel else
                                                                                  "; the kernel itself reads one packed u16 per local DoF" if block_kernel else "") +
                                                 f", G={G}, r={r:.4f} (SURVEY 8d)"),
                         # what this kernel has to move for its own representation (the contract formula credits I = 1 and 24 B for the dots)
                         "bytes_moved_formula": ((("16 + 4r/(p+1)^3 + G*8r" if lattice_kernel else "16 + 2r + G*8r") +
                                                  (" + 8 [r at the stored DoFs; p.v comes from the quadrature-point energy, v.v from LDS]" if fused else "") +
                                                  f" = {B_moved:.1f} B/DoF: " + ("lattice blocks, one u32 per CELL (closed-form indices) " if lattice_kernel else
                                                                                "one packed u16 (run, offset) per cell-local DoF ") +
                                                  "instead of the 4r of local_to_global") if block_kernel else
                                                 f"16 + 4r + G*8r = {B_op:.1f} B/DoF (local_to_global is read)")},
        }
        if args.rehearsal:
            out["rehearsal"] = "N ranks on ONE GPU over the loopback transport (tests only): control flow, not a measurement"
            out["metric"] = "REHEARSAL (not a measurement): " + out["metric"]
        if sustained:
            out["sustained"] = sustained
        if extras.get("mesh_116_cubed"):
            out["mesh_116_cubed"] = extras["mesh_116_cubed"]
        if exchange_ab:
            out["exchange_ab"] = exchange_ab
        if sweep:
            out["sweep"] = sweep
        if extras.get("cpu_baseline"):
            out["cpu_baseline"] = extras["cpu_baseline"]
        out["post_processing"] = {"completed": bool(final), "stage_reached": None if final else extras["stage"], "budget_s": post_budget}
        return out
    compose_ref.append(compose)
    watchdog.daemon = True
    watchdog.start()

    # the reference's own protocol (bp5/step-64.cu:443-463,724-730): solves of 200 iterations, wall clock + device sync
    # around each, the BEST repetition is reported.  Minutes-long runs clock ~5 % below short bursts (profiles/r1 j_*).
    sustained = None
    if args.sustained_iters > 0 and args.sustained_reps > 0:
        best = 0.0
        for _ in range(args.sustained_reps):
            sctl, sdt = timed_solve(args.sustained_iters, profile=False)
            best = max(best, n_global * sctl.last_step() / sdt)
        sustained = {"value": best, "unit": "DoF/s", "iterations": args.sustained_iters, "repetitions": args.sustained_reps,
                     "protocol": "reference: best of n repetitions of one solve, wall clock incl. device sync (bp5/step-64.cu:457-463,724-730)"}
    extras["sustained"] = sustained
    extras["stage"] = "fused / unfused agreement across ranks"

    # every rank decides fused / unfused from its own brick plan: agree before anything collective branches on it
    fused_all = bool(reduce_ranks(1.0 if ctl.dot_products_fused else 0.0, "min"))
    fused_any = bool(reduce_ranks(1.0 if ctl.dot_products_fused else 0.0, "max"))
    extras["fused_all"], extras["fused_any"] = fused_all, fused_any
    extras["stage"] = "short solve with the separate dot-product kernel"

    # the bare operator kernel in the same run (when the timed solve fused the dot products into it): a short solve with the
    # separate dot-product kernel, so that the operator-only roofline figure of SURVEY 8(d) can be read off the same box
    unfused_ms = None
    if fused_all:
        op.mf_data.set_cg_fusion(False)
        uctl, _ = timed_solve(10)
        unfused_ms = uctl.apply_ms_avg
        op.mf_data.set_cg_fusion(True)
    extras["unfused_ms"] = unfused_ms

    # N > 1 diagnostics: the same solve in both exchange schedules, each once plainly timed and once with phase stamps (HIP events at
    # the phase boundaries of every iteration on the solver's stream; the stamps cost a few us each, so the stamped run is not the timed
    # one).  One driver run then tells how much of an iteration is kernels, exposed exchange, and all-reduce -- per rank extremes.
    exchange_ab = None
    if world > 1 and not args.no_exchange_ab and args.variant == "merged":
        exchange_ab = {}
        k_ab = max(10, min(args.steps, 40))
        extras["exchange_ab"] = {}            # (published leg by leg: the watchdog reports the legs that finished)
        # (the default's legs first; the last two need stream wait-value support)
        for mode, name in ((0, "unsplit"), (2, "automatic"), (2, "automatic_one_combine_launch"), (1, "boundary_first")):
            extras["stage"] = f"exchange A/B leg '{name}'"
            try:
                op.mf_data.set_tuning("combine_signal", 1 if name == "automatic_one_combine_launch" else 0)   # (every rank, in step)
                op.mf_data.set_overlap(mode)
                timed_solve(3)
                actl, adt = timed_solve(k_ab, profile=False)
                pctl, _ = timed_solve(min(k_ab, 32), profile=2)
                entry = {"ms_per_iteration": adt / max(actl.last_step(), 1) * 1e3, "iterations": actl.last_step(),
                         "schedule_rank0": SCHEDULES.get(actl.exchange_schedule, "?"), "dot_products_fused_rank0": bool(actl.dot_products_fused),
                         "phases_ms_max_over_ranks": {}, "phases_ms_min_over_ranks": {}}
                for i, ph in enumerate(PHASES):
                    entry["phases_ms_max_over_ranks"][ph] = reduce_ranks(pctl.phase_ms[i], "max")
                    entry["phases_ms_min_over_ranks"][ph] = reduce_ranks(pctl.phase_ms[i], "min")
                exchange_ab[name] = entry
                extras["exchange_ab"] = dict(exchange_ab)
            except Exception as e:   # noqa: BLE001
                # a diagnostic leg: report, skip the remaining legs (the other ranks may be inside a collective of this one: the watchdog ends that)
                sys.stderr.write(f"[bench rank {rank}/{world}] exchange A/B leg '{name}' (bp5_mf_set_overlap({mode})) failed: {type(e).__name__}: {e}\n")
                exchange_ab[name] = {"error": f"{type(e).__name__}: {e}"}
                break
        op.mf_data.set_tuning("combine_signal", 0)
        op.mf_data.set_overlap(args.overlap)
        exchange_ab["note"] = ("same problem, same kernels, same bits; unsplit = gather, one launch, combine, scatter-add on the compute stream; "
                               "boundary_first = ghost-touching bricks first inside the launch, ghost rows + scatter-add on the communication stream "
                               "under the interior bricks; automatic = the library's default: one launch, ghost rows combined first, scatter-add on "
                               "the communication stream under the owned-row combine; automatic_one_combine_launch = the same with ghost rows and owned rows in ONE "
                               "combine launch, the exchange released by a stream wait-value (BP5_TUNE_COMBINE_SIGNAL: slower on one GPU, profiles/r3 z_*).  phases: HIP "
                               "events on the solver's stream (exchange = exposed part incl. unpack; gather_wait = exposed part of the ghost "
                               "gather that travels under the vector update)")
        extras["exchange_ab"] = dict(exchange_ab)

    # BASELINE.md's restatement of config 3 names the 116^3 mesh (100 544 625 DoFs; the headline of rounds 1-3).  The default became 116x116x120 so that eight
    # z-slabs are equally high; the same solve on the 116^3 mesh is reported beside it (N = 1, default mesh only), so that rounds stay comparable
    if world == 1 and base == HEADLINE_CELLS and not args.cells and p == 4 and args.config in (None, 3) and not args.no_mesh_116:
        extras["stage"] = "the same solve on the 116^3 mesh of rounds 1-3"
        m3, _, _ = build(p, (116, 116, 116))
        o3 = pkg.HelmholtzOperator(m3, quad, km, device=local_rank) if args.operator == "helmholtz" else \
            pkg.PoissonOperator(m3, quad, km, device=local_rank, geometry=pkg.GEOM_MERGED6 if G == 6 else pkg.GEOM_AFFINE)
        o3.mf_data.set_apply_variant(args.apply_variant)
        b3, x3 = o3.assemble_rhs(), o3.initialize_dof_vector()
        timed_solve(max(args.warmup, 1), o3, x3, b3)
        c3, d3 = timed_solve(args.steps, o3, x3, b3)
        v3 = int(m3.n_global_dofs) * c3.last_step() / d3
        extras["mesh_116_cubed"] = {"cells": [116, 116, 116], "dofs": int(m3.n_global_dofs), "value": v3, "unit": "DoF/s", "steps": c3.last_step(),
                                    "ms_per_step": d3 / max(c3.last_step(), 1) * 1e3, "kernel": c3.apply_kernel, "kernel_avg_launch_ms": c3.apply_ms_avg,
                                    "frac_of_hbm_peak": v3 * algorithmic_bytes_per_dof(p, m3.n_cells, m3.n_owned, G=G) / 1e9 / HBM_PEAK_GBS,
                                    "note": "BASELINE.md section 3 restates config 3 as this mesh; same code, same process, after the timed region"}
        o3.mf_data.close()
        del o3, b3, x3, m3
        torch.cuda.empty_cache()

    # achievable-stream figure: device copy y = 1.0 * x over the solver's vectors, read 8 + write 8 B per entry
    extras["stage"] = "copy-stream measurement"
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
    extras["stream_copy_gbs"] = stream_copy_gbs

    # BASELINE config 4 as one line: the other degrees at their ~5e7-DoF sizes, one short solve each (N = 1)
    # Both quadratures per degree (the reference's COLLOCATION switch, bp5/step-64.cu:48,243-247): Gauss(p+1) is the
    # reference default and the line's `value`; GLL(p+1) -- true BP5: collocated, one third of the contractions on the same bytes -- is
    # reported beside it per degree ("gll").
    sweep = None
    if args.config == 4 and world == 1:
        sweep = []
        extras["stage"] = "config-4 degree sweep"
        other_quad = pkg.QUAD_GLL if quad == pkg.QUAD_GAUSS else pkg.QUAD_GAUSS
        quad_name = {pkg.QUAD_GAUSS: "gauss", pkg.QUAD_GLL: "gll"}

        def one(q, mq, qd):
            oq = pkg.HelmholtzOperator(mq, qd, km, device=local_rank) if args.operator == "helmholtz" else \
                pkg.PoissonOperator(mq, qd, km, device=local_rank, geometry=pkg.GEOM_MERGED6 if G == 6 else pkg.GEOM_AFFINE)
            bq, xq = oq.assemble_rhs(), oq.initialize_dof_vector()
            timed_solve(3, oq, xq, bq)
            timed_solve(8, oq, xq, bq)   # (a second warm-up: the first entry of the sweep read 8 % low behind a single short one -- fresh allocations, clocks; profiles/r4 e_*, i_*)
            # best of two solves (the reference's own protocol takes the best repetition, bp5/step-64.cu:457-463): single solves of the sweep's first entry still
            # read up to 13 % low on some boxes (p = 1 Gauss 9.8 with GLL at 11.2 on the same mesh a minute later; profiles/r4 y_bench_config4_final.json)
            cq, dq = min((timed_solve(max(10, min(args.steps, 30)), oq, xq, bq) for _ in range(2)), key=lambda cd: cd[1])
            vq = int(mq.n_global_dofs) * cq.last_step() / dq
            e = {"value": vq, "ms_per_step": dq / max(cq.last_step(), 1) * 1e3, "kernel": cq.apply_kernel, "dot_products_fused": bool(cq.dot_products_fused),
                 "frac_of_hbm_peak": vq * algorithmic_bytes_per_dof(q, mq.n_cells, mq.n_owned, G=G) / 1e9 / HBM_PEAK_GBS, "repetitions": 2}
            oq.mf_data.close()
            del oq, bq, xq
            torch.cuda.empty_cache()
            return e
        for q in range(1, 9):
            extras["stage"] = f"config-4 degree sweep, p = {q}"
            if q == p and not args.cells:
                nq, mq = cells[0], mesh
                e = {"value": value, "ms_per_step": dt / max(iters, 1) * 1e3, "kernel": ctl.apply_kernel, "dot_products_fused": bool(ctl.dot_products_fused),
                     "frac_of_hbm_peak": value * algorithmic_bytes_per_dof(q, mesh.n_cells, mesh.n_owned, G=G) / 1e9 / HBM_PEAK_GBS}
            else:
                nq = CONFIG_SIZES[q]
                mq, _, _ = build(q, (nq, nq, nq))
                e = one(q, mq, quad)
            entry = {"degree": q, "cells": [nq, nq, nq], "dofs": int(mq.n_global_dofs), "quadrature": quad_name[quad]}
            entry.update(e)
            if not args.no_sweep_other_quadrature:
                entry[quad_name[other_quad]] = one(q, mq, other_quad)
            sweep.append(entry)
            extras["sweep"] = list(sweep)
            if mq is not mesh:
                del mq

    if rank == 0:
        extras["stage"] = "HBM traffic passes (rocprofv3 --pmc children)"
        ev = apply_variant_used
        key = f"p{p}_{args.quadrature}_{base[0]}x{base[1]}x{base[2]}_{args.geometry}_v{ev}" + ("_helmholtz" if args.operator == "helmholtz" else "")
        kname = ctl.apply_kernel or f"apply variant {ev}"
        tr = None
        if world == 1 and not args.no_traffic_pass and not args.rehearsal:
            wl = ["--operator", args.operator, "--degree", str(p), "--quadrature", args.quadrature, "--coefficient", args.coefficient, "--deform", str(args.deform),
                  "--variant", args.variant, "--geometry", args.geometry, "--apply-variant", str(args.apply_variant),
                  "--cells", str(cells[0]), str(cells[1]), str(cells[2]), "--cell-block", str(block[0]), str(block[1]), str(block[2])]
            tr = live_traffic(kname, wl)
        if tr is None and args.deform == 0.0 and world == 1:
            tr = measured_traffic(key, kname)
            if tr:
                tr["source"] += ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed (not measured in this run)"
        extras["traffic"] = tr
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            extras["stage"] = "CPU baseline"
            extras["cpu_baseline"] = cpu_baseline(mesh, p, quad, km, args.cpu_budget, max(args.steps, 2))
    watchdog.cancel()
    if rank == 0 and printed.acquire(blocking=False):
        print(json.dumps(compose(True)), flush=True)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
        op.mf_data.close()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
