"""The deal.II-shaped C++ facade (include/bp5_dealii_facade.hpp) on the GPU: user device functors
written like the reference's (examples/bp5_step64.hip) against the fused library path and the
oracle.  Runs the compiled example as a child process."""
import os
import re
import subprocess

import numpy as np
import pytest

import bp5_oracle as O
import bp5_pkg

pytestmark = pytest.mark.gpu
EXE = os.path.join(bp5_pkg.ROOT, "examples", "bp5_step64")


@pytest.mark.parametrize("p,cells,deform", [(2, (4, 3, 2), 0.04), (3, (3, 2, 2), 0.0), (4, (3, 2, 2), 0.05)])
def test_facade_functors(tmp_path, p, cells, deform):
    assert os.path.exists(EXE), "examples/bp5_step64 missing: run __graft_entry__.build()"
    prefix = str(tmp_path / "out")
    r = subprocess.run([EXE, "check", str(p), *map(str, cells), str(deform), prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vals = {k: float(v) for k, v in re.findall(r"^(\w+) ([0-9.e+-]+)", r.stdout, flags=re.M)}
    assert vals["functor_vs_fused"] < 1e-13            # LocalPoisson functor == fused kernel
    assert vals["unmerged_vs_fused"] < 1e-13           # submit_gradient(get_gradient()) == merged metric
    assert vals["metric_functor_vs_library"] < 1e-13   # evaluate_coefficients(functor) == library metric
    assert vals["merged_vs_plain_cg"] < 1e-11
    # the same solve through LinearAlgebra::distributed::Vector<double, MemorySpace::CUDA> (reinit, import, l2_norm,
    # all_zero, add, equ, sadd, = scalar) gives the same iterate; its true residual agrees with the recurrence
    assert vals["vector_api_cg"] < 1e-14
    assert abs(vals["vector_api_residual"] - vals["solver_residual"]) < 1e-6 * vals["solver_residual"] + 1e-12
    pr = O.Problem(p, cells, O.QUAD_GAUSS, deform_amp=deform)
    s = np.fromfile(prefix + "_src.bin")
    ref = pr.vmult(s)
    got = np.fromfile(prefix + "_poisson_functor.bin")
    assert np.linalg.norm(got - ref) < 1e-13 * np.linalg.norm(ref)
    # step-64 Helmholtz operator through FEEvaluation(values + gradients) + apply_quad_point_operations
    h = O.apply_helmholtz_cells(pr.mesh, pr.N, pr.D, pr.w, s)
    c = pr.mesh.constrained.astype(np.int64)
    h[c] = s[c]
    got = np.fromfile(prefix + "_helmholtz.bin")
    assert np.linalg.norm(got - h) < 1e-13 * np.linalg.norm(h)
    x, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 10)
    got = np.fromfile(prefix + "_cg.bin")
    assert np.linalg.norm(got - x) < 1e-11 * np.linalg.norm(x)


def test_reference_shaped_benchmark_lines():
    """the bench mode prints the reference's result lines (bp5/step-64.cu:470-474,512-516,543-547)"""
    r = subprocess.run([EXE, "bench", "4", "16", "20", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for tag in ("pcg-standard", "pcg-merged", "vmult"):
        m = re.search(rf"^{tag} (\d+) ([0-9.e+]+)$", r.stdout, flags=re.M)
        assert m and int(m.group(1)) == 65 ** 3 and float(m.group(2)) > 0


def test_reference_main_program_cycles():
    """`run` is PoissonProblem::run of the reference's main program (bp5/step-64.cu:619-700,724-730): its mesh family (a brick of
    (1|2|3) x (1|2) x (1|2) cells by cycle mod 6, refined cycle / 6 times), its lines per cycle.  Degree 5 as there, cycles 7...12:
    the DoF counts are the reference's (1 936 for cycle 7), both solvers print the same solution norm."""
    r = subprocess.run([EXE, "run", "5", "7", "12", "30", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    cycles = [int(c) for c in re.findall(r"^Cycle (\d+)$", r.stdout, flags=re.M)]
    assert cycles == [7, 8, 9, 10, 11, 12]
    want = {7: (3, 2, 2), 8: (2, 1, 1), 9: (3, 1, 1), 10: (2, 2, 1), 11: (3, 2, 1), 12: (1, 1, 1)}
    dofs = [int(n) for n in re.findall(r"^pcg-merged (\d+) ", r.stdout, flags=re.M)]
    for c, n in zip(cycles, dofs):
        ref = c // 6 - (1 if c % 6 == 1 else 0)
        sub = [s << ref for s in want[c]]
        assert n == (5 * sub[0] + 1) * (5 * sub[1] + 1) * (5 * sub[2] + 1)
    assert dofs[0] == 1936
    norms = re.findall(r"norm ([0-9.e+-]+)$", r.stdout, flags=re.M)
    assert len(norms) == 12                                          # one line per solve: two solvers x six cycles
    for a, b in zip(norms[0::2], norms[1::2]):
        assert abs(float(a) - float(b)) < 1e-9 * float(a)


@pytest.mark.parametrize("p,n", [(3, 2), (2, 3)])
def test_step64_helmholtz_solve_through_the_operator_agnostic_solvers(tmp_path, p, n, mode="helmholtz"):
    """HelmholtzProblem::solve (step-64/step-64.cu:505-530): the reference's solvers take ANY operator with vmult
    (bp5/solver.h:25-30,377,475).  The example's HelmholtzOperator is a user device functor behind the facade; SolverCG and
    SolverCGFullMerge reach it through bp5_cg_solve_operator.  Checked against the oracle's CG on apply_helmholtz_cells.
    p = 3 on 2^3 cells (343 DoFs) is the first cycle of the step-64 tutorial, whose printed `solution norm` is remembered
    as ~0.0205439 (a soft cross-check from outside /root/reference, reported, not asserted tightly)."""
    prefix = str(tmp_path / "h")
    r = subprocess.run([EXE, mode, str(p), str(n), prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vals = {k: float(v) for k, v in re.findall(r"^(\w+) ([0-9.e+-]+)", r.stdout, flags=re.M)}
    pr = O.Problem(p, (n, n, n), O.QUAD_GAUSS, h=1.0 / n)
    m = pr.mesh
    c = m.constrained.astype(np.int64)

    def A(s):                                   # HelmholtzOperator::vmult, step-64/step-64.cu:283-300
        d = O.apply_helmholtz_cells(m, pr.N, pr.D, pr.w, s)
        d[c] = s[c]
        return d

    b = pr.rhs()
    assert np.linalg.norm(np.fromfile(prefix + "_helmholtz_b.bin") - b) < 1e-13 * np.linalg.norm(b)
    xr, kr, _ = O.cg_plain(A, b, m.n_dofs, tol=1e-12 * np.linalg.norm(b))
    assert np.linalg.norm(A(xr) - b) < 1e-11 * np.linalg.norm(b)
    norm_ref = O.l2_norm_solution(m, xr)
    for tag in ("plain", "merged"):
        x = np.fromfile(prefix + f"_helmholtz_x_{tag}.bin")
        assert np.linalg.norm(x - xr) < 1e-11 * np.linalg.norm(xr), tag
        assert abs(vals[f"helmholtz_{tag}_iterations"] - kr) <= 2, (tag, vals, kr)
        assert abs(vals[f"helmholtz_{tag}_norm"] - norm_ref) < 1e-11 * norm_ref
    if (p, n) == (3, 2):
        soft = 0.0205439
        print(f"step-64 cycle 0 (p=3, 8 cells, 343 DoFs): solution norm {norm_ref:.7f} (GPU {vals['helmholtz_plain_norm']:.7f}); "
              f"remembered tutorial value {soft}; difference {norm_ref - soft:+.2e}")
        assert m.n_dofs == 343 and abs(norm_ref - soft) < 1e-7               # agrees to the six digits remembered


@pytest.mark.parametrize("p,n", [(3, 2), (2, 4), (4, 3)])
def test_step64_helmholtz_solve_on_the_native_kernel(tmp_path, p, n):
    """The same program with the example's HelmholtzOperatorNative (bp5_mf_set_operator(BP5_OP_HELMHOLTZ): the library's fused Helmholtz
    kernel behind the same class surface; it exposes coef(), so SolverCG / SolverCGFullMerge run it as the library's own operator): the
    same right-hand side, iteration counts, solution vectors and norms as the oracle's CG."""
    test_step64_helmholtz_solve_through_the_operator_agnostic_solvers(tmp_path, p, n, mode="helmholtz_native")


@pytest.mark.parametrize("p,general", [(2, False), (3, False), (2, True), (3, True)])
def test_facade_resolves_hanging_nodes(tmp_path, p, general):
    """FEEvaluation::read_dof_values / distribute_local_to_global of the facade honour Data::constraint_mask
    (resolve_hanging_nodes, bp5/fe_evaluation_gl.h:150-151,167-168): a user functor written like the reference's
    LocalPoissonOperator, on a mesh with one 2:1 refined interface handed over as flat arrays, against the library's
    hanging-node kernel and the oracle."""
    if general:      # staircase-shaped refined region: cells with one, two and three constrained faces, and constrained edges
        r = np.zeros((2, 2, 3), bool)
        r[0, 0, 0] = r[0, 0, 1] = r[0, 1, 0] = r[1, 0, 0] = r[1, 1, 2] = True
        m = O.RefinedBrickMesh(p, (3, 2, 2), r, H=0.5, deform_amp=0.03)
        assert any(int(k) & (512 | 1024 | 2048) for k in m.constraint_mask)
    else:
        m = O.HangingBrickMesh(p, 2, 2, 1, 3, H=0.5, deform_amp=0.03)
    prefix = str(tmp_path / "hang")
    s = O.deterministic_src(m.n_dofs, seed=81)
    m.l2g.astype(np.uint32).tofile(prefix + "_l2g.bin")
    m.coords.astype(np.float64).tofile(prefix + "_coords.bin")
    m.constrained.astype(np.uint32).tofile(prefix + "_constrained.bin")
    m.constraint_mask.astype(np.uint32).tofile(prefix + "_mask.bin")
    s.tofile(prefix + "_src.bin")
    r = subprocess.run([EXE, "hanging", str(p), prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    vals = {k: float(v) for k, v in re.findall(r"^(\w+) ([0-9.e+-]+)", r.stdout, flags=re.M)}
    assert vals["functor_vs_library"] < 1e-13 and vals["unmerged_functor_vs_library"] < 1e-13
    _, _, w, N, D = O.shape_tables(p, O.QUAD_GAUSS)
    ref = O.apply_cells(m, O.merged_metric(m, N, D, w), N, D, s)
    c = m.constrained.astype(np.int64)
    ref[c] = s[c]
    for tag in ("library", "functor", "functor_unmerged"):
        got = np.fromfile(prefix + f"_out_{tag}.bin")
        assert np.linalg.norm(got - ref) < 1e-13 * np.linalg.norm(ref), tag
