"""N ranks of the library as N PROCESSES ON ONE GPU against the oracle on the undivided mesh.

RCCL refuses two ranks on one device, and the driver's multi-GPU node is not available to the tests: so the library is built a
second time (libbp5_loopback.so, csrc/Makefile target `loopback`) with its nine RCCL calls renamed to a host-shared-memory
transport (tests/loopback/loopback_rccl.cpp: stream-ordered, grouped send/recv, rank-ordered all-reduce).  Every other line
is the product source: z-slab meshes and halo plans of first / middle / last ranks, pack + unpack kernels, gather /
scatter-add in all three schedules (sequential, 3-phase overlapped, automatic), the block kernel's fused dot products with the
owners' corrections between DIFFERENT ranks, the 7-value all-reduce of every iteration, compress(add) of the RHS and the
diagonal, the ghost refresh of the L2 norm.  What this cannot show is xGMI behaviour or RCCL itself (those: the self-neighbour
tests of test_gpu_parity.py and the driver's scaling run)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import bp5_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOPBACK_LIB = os.path.join(ROOT, "deal-and-ceed-on-gpu_amd", "libbp5_loopback.so")
WORKER = os.path.join(ROOT, "tests", "loopback", "worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(world, args, out, timeout=420, worker=WORKER, delay_us=0, extra_env=None):
    """delay_us: every transfer of the loopback transport lags that long behind its stream (BP5_LOOPBACK_DELAY_US); receives poison their
    destination with NaN until the message has landed (default of the transport) -- a missing cross-stream wait gives a wrong result"""
    assert os.path.exists(LOOPBACK_LIB), "libbp5_loopback.so missing: run __graft_entry__.build() (make -C .../csrc loopback)"
    env = dict(os.environ, BP5_LIB=LOOPBACK_LIB, HSA_ENABLE_IPC_MODE_LEGACY="0", BP5_LOOPBACK_DELAY_US=str(delay_us), **(extra_env or {}))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), out] + [str(a) for a in args], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs, failed = [], False
    for r, pr in enumerate(procs):
        try:
            o, _ = pr.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:                      # exactly the processes started above
                q.kill()
            o, _ = pr.communicate()
            failed = True
        logs.append(f"--- rank {r} (exit {pr.returncode}) ---\n{o[-3000:]}")
        failed = failed or pr.returncode != 0
    assert not failed, "\n".join(logs)


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _stop_tolerance(pr, k_min=8, k_max=40):
    """(k, tol): the first iteration k >= k_min of the oracle's plain CG whose residual undercuts every earlier one by 8 %, and a tolerance
    half way (geometrically) between that residual and the lowest earlier one: 4 % of margin on either side, rounding (1e-13) cannot
    move the stopping iteration"""
    hist = []
    O.cg_plain(pr.vmult, pr.rhs(), k_max, history=hist)
    res = [float(np.linalg.norm(pr.rhs()))] + hist            # res[k]: after k iterations
    for k in range(k_min, k_max + 1):
        low = min(res[:k])
        if res[k] < 0.92 * low:
            return k, float(np.sqrt(res[k] * low))
    raise AssertionError("no clear record low in the oracle's residual history")


@pytest.mark.parametrize("world,p,cells,block,numbering,variant,delay_us", [
    (2, 4, (8, 8, 12), (4, 4, 4), 1, 56, 0),  # the bench's configuration: parity-class bricks, block kernel, fused dot products
    (2, 4, (8, 8, 12), (4, 4, 4), 1, 56, 400),  # ... with every transfer lagging 0.4 ms behind its stream (cross-stream ordering of both exchange schedules)
    (3, 4, (8, 4, 13), (4, 4, 2), 1, 56, 0),  # first / middle / last rank, ragged slabs (13 layers over 3 ranks), thin bricks
    (3, 4, (8, 4, 13), (4, 4, 2), 1, 56, 250),  # ... lagging
    (2, 4, (8, 8, 12), (4, 4, 4), 1, 0, 0),   # the library's own choice at this size (too few bricks for the persistent grid: pencil kernel)
    (3, 2, (3, 3, 7), (0, 0, 0), 0, 0, 0),    # lexicographic cells, atomic pencil kernel
    (3, 2, (3, 3, 7), (0, 0, 0), 0, 0, 250),  # ... lagging (3-phase schedule of the atomic kernels)
    (2, 6, (4, 4, 5), (4, 4, 2), 1, 56, 0),   # another degree on the block kernel
    (4, 4, (4, 4, 9), (4, 4, 2), 1, 56, 0),   # four ranks: two middle ranks, slabs of 3 / 2 / 2 / 2 layers (thinner than a brick)
    (3, 3, (5, 4, 7), (2, 2, 2), 1, 10, 0),   # team kernel (LDS-staged atomics) behind the exchange
    (2, 1, (6, 5, 6), (0, 0, 0), 0, 0, 0),    # p = 1
    (5, 4, (4, 4, 10), (4, 4, 2), 1, 56, 0),  # five ranks (the most this box lets one test run beside its own process): three middle ranks, equal slabs of one brick layer
    (2, 2, (8, 8, 8), (8, 8, 4), 1, 56, 0),   # p = 2 on the block kernel with its cells packed wave by wave (round 4), one brick layer per rank
    (2, 1, (8, 8, 16), (8, 8, 8), 1, 56, 0),  # p = 1 on the block kernel at four workgroups per CU (round 4)
])
def test_ranks_on_one_gpu_match_the_undivided_problem(tmp_path, world, p, cells, block, numbering, variant, delay_us):
    iters = 8
    pr = O.Problem(p, cells, O.QUAD_GAUSS, deform_amp=0.03, kappa=O.kappa_step64)
    # tolerance stop: a tolerance half way (geometrically) between two consecutive residuals of the oracle's own history, so that rounding
    # cannot move the stopping iteration
    pr1 = O.Problem(p, cells, O.QUAD_GAUSS, deform_amp=0.03)      # the tolerance-stop solves: constant coefficient (converges in tens of iterations)
    k_stop, stop_tol = _stop_tolerance(pr1)
    _run_ranks(world, [p, *cells, *block, numbering, iters, variant, repr(stop_tol)], str(tmp_path), delay_us=delay_us)
    nd = pr.mesh.n_dofs
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    keys = [k for k in ranks[0].files if k[0] in "bAx" or k == "inv_diag"]
    full = {k: np.full(nd, np.nan) for k in keys}
    for z in ranks:
        gid = z["gid"].astype(np.int64)
        assert np.isnan(full["b"][gid]).all()                            # every DoF owned by exactly one rank
        for k in keys:
            full[k][gid] = z[k]
    assert not any(np.isnan(v).any() for v in full.values())
    b_ref = pr.rhs()
    assert _rel(full["b"], b_ref) < 1e-13
    A_ref = pr.vmult(O.deterministic_src(nd, seed=21))
    for mode in (0, 1, 2):
        assert _rel(full[f"A{mode}"], A_ref) < 1e-13, mode
    x_plain, _, res_plain = O.cg_plain(pr.vmult, b_ref, iters)
    x_merged, _, res_merged = O.cg_merged(pr.vmult, b_ref, iters)
    assert _rel(full["x_plain"], x_plain) < 1e-11 and _rel(full["x_plain_overlapped"], x_plain) < 1e-11
    merged = ["merged_unsplit", "merged_overlapped", "merged_default", "merged_unfused", "merged_unsplit_again", "merged_overlapped_again",
              "merged_unsplit_late_gather", "merged_overlapped_late_gather", "merged_default_one_combine_launch", "merged_default_ghost_combine_on_comm"]
    for k in merged:
        assert _rel(full["x_" + k], x_merged) < 1e-11, k
    on_block_kernel = int(ranks[0]["variant"]) == 56
    assert on_block_kernel == (variant == 56)
    if on_block_kernel:
        assert np.array_equal(full["x_merged_unsplit"], full["x_merged_unsplit_again"])     # fixed summation orders: reproducible
        assert np.array_equal(full["x_merged_overlapped"], full["x_merged_overlapped_again"])  # ... in the boundary-first schedule too
        # every schedule runs the same kernels over the same workgroup ranges and sums the same columns: the same bits
        assert np.array_equal(full["x_merged_overlapped"], full["x_merged_unsplit"]) and np.array_equal(full["x_merged_default"], full["x_merged_unsplit"])
        # the ghost gather of the new search direction under the vector update (cgm_pack_updated_kernel recomputes p at the interface DoFs:
        # same arithmetic, explicit fma) or after it (BP5_EARLY_GATHER=0): the same bits
        assert np.array_equal(full["x_merged_unsplit_late_gather"], full["x_merged_unsplit"])
        assert np.array_equal(full["x_merged_overlapped_late_gather"], full["x_merged_unsplit"])
        # BP5_COMBINE_SIGNAL=1: ghost rows and owned rows of the combine pass in ONE launch, the exchange released by a stream wait-value
        assert np.array_equal(full["x_merged_default_one_combine_launch"], full["x_merged_unsplit"])
        assert np.array_equal(full["x_merged_default_ghost_combine_on_comm"], full["x_merged_unsplit"])   # ghost-row combine on the communication stream
    # tolerance stop across the ranks: the same iteration as the oracle on the undivided mesh, on every rank, in every schedule; iterate frozen there
    x_stop, k_ref, _ = O.cg_plain(pr1.vmult, pr1.rhs(), 400, tol=stop_tol)
    assert k_ref == k_stop
    for key in ("stop_plain", "stop_merged_default", "stop_merged_overlapped", "stop_merged_unsplit"):
        assert all(int(z["its_" + key]) == k_ref for z in ranks), (key, [int(z["its_" + key]) for z in ranks], k_ref)
        assert all(float(z["res_" + key]) <= stop_tol for z in ranks)
        assert _rel(full["x_" + key], x_stop) < 1e-10, key
    for z in ranks:
        # block kernel: the dot products stay fused in BOTH exchange schedules (1 unsplit, 2 boundary-first); atomic kernels: 3-phase split
        assert bool(z["fused_merged_unsplit"]) == on_block_kernel and bool(z["fused_merged_overlapped"]) == on_block_kernel and not bool(z["fused_merged_unfused"])
        assert int(z["sched_merged_unsplit"]) == 1 and int(z["sched_merged_overlapped"]) == (2 if on_block_kernel else 3)
        assert int(z["sched_merged_default"]) == (4 if on_block_kernel else 1)
        assert np.array_equal(z["norms"], ranks[0]["norms"])             # every rank sees the same all-reduced residual
    assert abs(ranks[0]["norms"][0] - res_plain) < 1e-9 * np.linalg.norm(b_ref)
    assert abs(ranks[0]["norms"][1] - res_merged) < 1e-9 * np.linalg.norm(b_ref)
    d_ref = O.operator_diagonal(pr.mesh, pr.coef, pr.N, pr.D)
    assert _rel(full["inv_diag"], 1.0 / d_ref) < 1e-13
    x_jac, _, _ = O.cg_merged(pr.vmult, b_ref, iters, diag=1.0 / d_ref)
    assert _rel(full["x_jacobi"], x_jac) < 1e-11
    l2 = O.l2_norm_solution(pr.mesh, full["x_merged_default"])
    for z in ranks:
        assert abs(float(z["l2"]) - l2) < 1e-12 * l2
    # step-64's Helmholtz operator (native kernel) across the ranks: operator and merged CG against the oracle on the undivided mesh
    c = pr.mesh.constrained.astype(np.int64)

    def Ah(s):
        d = O.apply_helmholtz_cells(pr.mesh, pr.N, pr.D, pr.w, s)
        d[c] = s[c]
        return d

    assert _rel(full["Ah"], Ah(O.deterministic_src(nd, seed=21))) < 1e-13
    xh, _, _ = O.cg_merged(Ah, b_ref, iters)
    assert _rel(full["x_helmholtz"], xh) < 1e-11
    for z in ranks:
        assert bool(z["fused_helmholtz"]) == on_block_kernel


def test_bench_with_two_ranks_as_the_driver_launches_it():
    """bench.py under `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` (the driver's N > 1 command form):
    rendezvous, slab meshes, library communicator, barrier + max-over-ranks timing, rank-0 JSON line, teardown.  `--rehearsal`
    puts both ranks on cuda:0 over the loopback transport; the line is marked as not being a measurement."""
    import json
    assert os.path.exists(LOOPBACK_LIB)
    env = dict(os.environ, BP5_LIB=LOOPBACK_LIB, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--cells", "16", "16", "16", "--sustained-iters", "0", "--apply-variant", "56", "--rehearsal"]
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "strong" and "rehearsal" in out
    assert out["config"]["dofs_per_gpu"] < 65 ** 3 and "274625 DoFs" in out["config"]["workload"]      # ONE 16^3-cell problem split over the ranks
    assert out["config"]["apply_variant"] == 56 and out["config"]["cg_dot_products_fused"] is True      # fused dot products on both ranks
    assert out["config"]["cg_dot_products_fused_on_every_rank"] is True and out["config"]["exchange_schedule"] == "under-combine"
    assert out["value"] > 0 and "cpu_baseline" not in out and out["host_setup_s"] > 0
    assert out["roofline"]["kernel"].startswith("apply_block_kernel<4,false,32,1,")                     # the name the solve itself reports
    # the N > 1 diagnostics: both exchange schedules timed, every phase of an iteration stamped (max / min over the ranks)
    ab = out["exchange_ab"]
    for name, sched in (("unsplit", "unsplit"), ("boundary_first", "boundary-first"), ("automatic", "under-combine"),
                        ("automatic_one_combine_launch", "under-combine")):
        e = ab[name]
        assert e["schedule_rank0"] == sched and e["dot_products_fused_rank0"] is True and e["ms_per_iteration"] > 0
        for k in ("update", "gather_wait", "operator", "exchange", "reduce_local", "allreduce", "control", "iteration"):
            assert e["phases_ms_max_over_ranks"][k] >= e["phases_ms_min_over_ranks"][k] >= 0.0
        assert e["phases_ms_max_over_ranks"]["operator"] > 0 and e["phases_ms_max_over_ranks"]["iteration"] > 0
    # solve_check: the all-reduced residual norms of the timed solve are those of the SAME problem on one rank (what lets the lines of an
    # N = 1, 2, 4, 8 sweep vouch for each other)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--cells", "16", "16", "16",
                          "--sustained-iters", "0", "--apply-variant", "56", "--no-cpu-baseline", "--no-traffic-pass"],
                         env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])["solve_check"]
    got = out["solve_check"]
    assert got["iterations"] == ref["iterations"] == 6
    assert abs(got["initial_residual"] - ref["initial_residual"]) <= 1e-12 * ref["initial_residual"]
    assert abs(got["residual_after_timed_solve"] - ref["residual_after_timed_solve"]) <= 1e-10 * ref["residual_after_timed_solve"]


@pytest.mark.parametrize("world,p,amp", [(2, 2, 0.03), (3, 3, 0.0), (2, 4, 0.02)])
def test_refined_mesh_with_hanging_nodes_across_ranks(tmp_path, world, p, amp):
    """A 2:1 refined mesh (staircase-shaped refined region: constrained faces in every number, constrained edges) cut into slabs along x
    (tests/loopback/partition.py): constrained faces whose coarse DoFs are GHOSTS of the rank, ragged interfaces, a rank with a single
    interior cell.  Hanging-node kernel (apply variant 90) behind the exchange in both schedules, RHS and diagonal with
    compress(add), the solvers, the ghosted L2 norm -- against the oracle on the undivided mesh."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "loopback"))
    from partition import partition
    r = np.zeros((2, 2, 3), bool)
    r[0, 0, 0] = r[0, 0, 1] = r[0, 1, 0] = r[1, 0, 0] = r[1, 1, 2] = True
    m = O.RefinedBrickMesh(p, (3, 2, 2), r, H=0.5, deform_amp=amp)
    x_cell = m.cell_node_coords().reshape(m.n_cells, -1, 3).mean(1)[:, 0]
    cuts = [0.6, 1.1][:world - 1] if world == 3 else [0.8]
    cell_rank = sum((x_cell > c).astype(int) for c in cuts)
    pieces = partition(m, cell_rank, world)
    assert all(pc["n_ghost"] > 0 for pc in pieces[1:]) and any((pc["constraint_mask"] != 0).any() for pc in pieces[1:])
    for rk, pc in enumerate(pieces):
        np.savez(os.path.join(str(tmp_path), f"mesh{rk}.npz"), **pc)
    iters = 6
    _run_ranks(world, [p, iters], str(tmp_path), worker=os.path.join(ROOT, "tests", "loopback", "worker_mesh.py"))
    _, _, w, N, D = O.shape_tables(p, O.QUAD_GAUSS)
    coef = O.merged_metric(m, N, D, w, O.kappa_step64)
    c = m.constrained.astype(np.int64)

    def A(s):
        d = O.apply_cells(m, coef, N, D, s)
        d[c] = s[c]
        return d

    ranks = [np.load(os.path.join(str(tmp_path), f"rank{rk}.npz")) for rk in range(world)]
    keys = [k for k in ranks[0].files if k[0] in "bAx" or k == "inv_diag"]
    full = {k: np.full(m.n_dofs, np.nan) for k in keys}
    for z in ranks:
        gid = z["gid"].astype(np.int64)
        assert int(z["variant"]) == 90 and np.isnan(full["b"][gid]).all()
        for k in keys:
            full[k][gid] = z[k]
    assert not any(np.isnan(v).any() for v in full.values())
    b_ref = O.assemble_rhs(m)
    assert _rel(full["b"], b_ref) < 1e-13
    A_ref = A(O.deterministic_src(m.n_dofs, seed=23))
    assert _rel(full["A0"], A_ref) < 1e-13 and _rel(full["A1"], A_ref) < 1e-13
    x_plain, _, res_plain = O.cg_plain(A, b_ref, iters)
    x_merged, _, _ = O.cg_merged(A, b_ref, iters)
    assert _rel(full["x_plain"], x_plain) < 1e-11 and _rel(full["x_merged"], x_merged) < 1e-11
    for z in ranks:
        assert np.array_equal(z["norms"], ranks[0]["norms"])
    assert abs(ranks[0]["norms"][0] - res_plain) < 1e-9 * np.linalg.norm(b_ref)
    d_ref = O.operator_diagonal(m, coef, N, D)
    assert _rel(full["inv_diag"], 1.0 / d_ref) < 1e-13
    x_jac, _, _ = O.cg_merged(A, b_ref, iters, diag=1.0 / d_ref)
    assert _rel(full["x_jacobi"], x_jac) < 1e-11
    l2 = O.l2_norm_solution(m, full["x_merged"])
    for z in ranks:
        assert abs(float(z["l2"]) - l2) < 1e-12 * l2


@pytest.mark.parametrize("layout", ["quadrants", "random"])
def test_quadrant_partition_with_two_way_and_diagonal_neighbours(tmp_path, layout):
    """A conforming mesh cut into 2 x 2 quadrants in (x, y), four ranks: every rank has neighbours it both sends to and receives from, and
    ranks 0 and 3 share only an edge of DoFs -- halo plans the library's own slab generator never produces (it is what a
    p4est-style host hands over).  Same checks as above; the mask array is all zero (conforming), so the default kernels run."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "loopback"))
    from partition import partition
    p, cells, world, iters = 3, (4, 4, 3), 4, 6
    pr = O.Problem(p, cells, O.QUAD_GAUSS, deform_amp=0.03, kappa=O.kappa_step64)
    m = pr.mesh
    n = p + 1
    cen = m.coords[m.l2g.astype(np.int64)].reshape(m.n_cells, -1, 3).mean(1)
    if layout == "quadrants":
        cell_rank = (cen[:, 0] > 2.0).astype(int) + 2 * (cen[:, 1] > 2.0).astype(int)
        assert sorted(np.bincount(cell_rank)) == [12, 12, 12, 12]
    else:               # cells dealt out at random: every rank neighbours every other, hardly any interior cell, scattered ghost sets
        cell_rank = np.random.default_rng(11).integers(0, world, m.n_cells)
    pieces = partition(m, cell_rank, world, owner_rule="alternate")
    if layout == "quadrants":
        assert [pc["n_neighbors"] for pc in pieces] == [3, 2, 2, 3]  # (the DoFs of the common edge belong to rank 0 or 3: ranks 1 and 2 never talk)
    else:
        assert all(pc["n_neighbors"] == 3 for pc in pieces)
    assert any(pc["send_offsets"][k + 1] > pc["send_offsets"][k] and pc["recv_offsets"][k + 1] > pc["recv_offsets"][k]
               for pc in pieces for k in range(pc["n_neighbors"]))                       # a two-way neighbour exists
    for rk, pc in enumerate(pieces):
        np.savez(os.path.join(str(tmp_path), f"mesh{rk}.npz"), **pc)
    _run_ranks(world, [p, iters], str(tmp_path), worker=os.path.join(ROOT, "tests", "loopback", "worker_mesh.py"))
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{rk}.npz")) for rk in range(world)]
    keys = [k for k in ranks[0].files if k[0] in "bAx" or k == "inv_diag"]
    full = {k: np.full(m.n_dofs, np.nan) for k in keys}
    for z in ranks:
        gid = z["gid"].astype(np.int64)
        assert np.isnan(full["b"][gid]).all()
        for k in keys:
            full[k][gid] = z[k]
    assert not any(np.isnan(v).any() for v in full.values())
    b_ref = pr.rhs()
    assert _rel(full["b"], b_ref) < 1e-13
    A_ref = pr.vmult(O.deterministic_src(m.n_dofs, seed=23))
    assert _rel(full["A0"], A_ref) < 1e-13 and _rel(full["A1"], A_ref) < 1e-13
    x_plain, _, _ = O.cg_plain(pr.vmult, b_ref, iters)
    x_merged, _, _ = O.cg_merged(pr.vmult, b_ref, iters)
    assert _rel(full["x_plain"], x_plain) < 1e-11 and _rel(full["x_merged"], x_merged) < 1e-11
    d_ref = O.operator_diagonal(m, pr.coef, pr.N, pr.D)
    assert _rel(full["inv_diag"], 1.0 / d_ref) < 1e-13
    x_jac, _, _ = O.cg_merged(pr.vmult, b_ref, iters, diag=1.0 / d_ref)
    assert _rel(full["x_jacobi"], x_jac) < 1e-11
