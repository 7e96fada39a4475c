"""Parity of the HIP path (through the C ABI) against the oracle on the same seeded inputs,
against the golden fixtures, and -- at full BASELINE sizes -- through size-independent
properties.  FP64 tolerances are stated per test; the north-star bound is 1e-11 relative l2 on
the CG solution vector."""
import os

import numpy as np
import pytest

import bp5_oracle as O
import c_oracle as CO
import bp5_pkg
from make_golden import CASES_APPLY

pytestmark = pytest.mark.gpu
pkg = bp5_pkg.load()
G = os.path.join(os.path.dirname(__file__), "golden")
TOL_OP = 1e-13     # one operator application (rounding + atomic summation order)
TOL_CG = 1e-11     # CG solution vector at a fixed iteration count (BASELINE north_star)


def _t():
    import torch
    return torch


def dev(x):
    return _t().from_numpy(np.ascontiguousarray(x)).to("cuda:0")


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def kappa_of(km):
    return O.kappa_step64 if km else O.kappa_none


# ------------------------------------------------------------------ geometry (a11, a3)
@pytest.mark.parametrize("p,quad,cells,amp,km", [(1, 0, (3, 2, 2), 0.05, 0), (2, 1, (2, 3, 2), 0.05, 1), (4, 0, (3, 2, 2), 0.04, 1),
                                                 (6, 0, (2, 2, 1), 0.05, 0), (8, 1, (2, 1, 1), 0.03, 1)])
def test_merged_metric(p, quad, cells, amp, km):
    mesh = pkg.BrickMesh(p, cells, deform_amp=amp)
    mf = pkg.MatrixFree().reinit(mesh, quad, km)
    coef = mf.evaluate_coefficients()
    got = mf.coef_reference_layout(coef).cpu().numpy().reshape(6, mesh.n_cells, -1)
    pr = O.Problem(p, cells, quad, deform_amp=amp, kappa=kappa_of(km))
    assert np.abs(got - pr.coef).max() < 1e-13 * np.abs(pr.coef).max()
    # MatrixFree::Data mirror: inv_jacobian / JxW with deal.II padding
    d = mf.get_data()
    n3 = (p + 1) ** 3
    pad = d.padding_length
    assert pad >= n3 and (pad & (pad - 1)) == 0
    import ctypes as C
    gp = mesh.n_cells * pad
    K = np.zeros(9 * gp)
    JxW = np.zeros(gp)
    pkg.lib().bp5_copy_d2h(K.ctypes.data, d.inv_jacobian, K.nbytes)
    pkg.lib().bp5_copy_d2h(JxW.ctypes.data, d.JxW, JxW.nbytes)
    Ko, JxWo, _ = O.jacobians(pr.mesh, pr.N, pr.D, pr.w)
    K = K.reshape(3, 3, mesh.n_cells, pad)[:, :, :, :n3].transpose(2, 3, 0, 1)
    assert np.abs(K - Ko).max() < 1e-12
    assert np.abs(JxW.reshape(mesh.n_cells, pad)[:, :n3] - JxWo).max() < 1e-14


# ------------------------------------------------------------------ operator (a1-a10)
@pytest.mark.parametrize("case", CASES_APPLY)
def test_vmult_golden(case):
    p, quad, cells, amp, km = case
    ref = np.load(os.path.join(G, "vmult_cases.npz"))[f"vmult_p{p}_q{quad}_{cells[0]}x{cells[1]}x{cells[2]}_a{amp}_k{km}"]
    mesh = pkg.BrickMesh(p, cells, deform_amp=amp)
    op = pkg.PoissonOperator(mesh, quad, km)
    s = O.deterministic_src(mesh.n_owned, seed=100 + p)
    dst = op.initialize_dof_vector()
    op.vmult(dst, dev(s))
    assert rel(dst.cpu().numpy(), ref) < TOL_OP


@pytest.mark.parametrize("p", range(1, 9))
@pytest.mark.parametrize("quad", [0, 1])
def test_cell_loop_all_degrees(p, quad):
    cells = (3, 3, 2) if p <= 4 else (3, 2, 1)        # cell counts not divisible by the team size
    pr = O.Problem(p, cells, quad, deform_amp=0.04, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.04)
    mf = pkg.MatrixFree().reinit(mesh, quad, pkg.COEF_STEP64)
    coef = mf.evaluate_coefficients()
    s = O.deterministic_src(mesh.n_owned, seed=3)    # non-zero boundary values too
    dst = mf.initialize_dof_vector()
    mf.cell_loop(coef, dev(s), dst)
    ref = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)
    assert rel(dst.cpu().numpy(), ref) < TOL_OP


@pytest.mark.parametrize("p,bricks", [(1, False), (2, False), (3, False), (4, False), (4, True), (6, True), (8, False)])
def test_cell_loop_against_closed_form_element_matrices(p, bricks):
    """The HIP path against matrices that do not come from the oracle: on affine cubes of edge h (coefficient 1, Gauss(p+1) quadrature) the
    cell operator is h (K x M x M + M x K x M + M x M x K) with the textbook 1-D stiffness / mass matrices of the GLL Lagrange basis
    (tests/test_oracle_known_answers.py::_textbook_1d: rational matrices for p = 1, 2, 40-digit polynomial algebra above) -- geometry kernel,
    merged metric, tables and the fused operator kernel (pencil kernel, or the block kernel on bricks) end to end, through the mesh's own
    local_to_global"""
    from test_oracle_known_answers import _textbook_1d
    K, M = _textbook_1d(p)
    cells, h = ((4, 4, 2), 0.25) if bricks else ((3, 2, 2), 0.5)
    kw = dict(cell_block=(4, 4, 2) if p == 4 else (4, 4, 2), dof_numbering=1, cell_block_order=1) if bricks else {}
    mesh = pkg.BrickMesh(p, cells, h=h, **kw)
    mf = pkg.MatrixFree().reinit(mesh, pkg.QUAD_GAUSS, pkg.COEF_ONE)
    if bricks:
        mf.set_apply_variant(56)
    coef = mf.evaluate_coefficients()
    s = O.deterministic_src(mesh.n_owned, seed=11)
    dst = mf.initialize_dof_vector()
    mf.cell_loop(coef, dev(s), dst)
    Ae = h * (np.kron(M, np.kron(M, K)) + np.kron(M, np.kron(K, M)) + np.kron(K, np.kron(M, M)))
    ref = np.zeros(mesh.n_owned)
    l2g = np.asarray(mesh.l2g).reshape(mesh.n_cells, -1).astype(np.int64)
    for c in range(mesh.n_cells):
        np.add.at(ref, l2g[c], Ae @ s[l2g[c]])
    assert rel(dst.cpu().numpy(), ref) < 1e-12


@pytest.mark.parametrize("p,bricks", [(1, False), (2, False), (3, False), (4, False), (4, True), (6, False)])
def test_cell_loop_on_sheared_cells_against_closed_form(p, bricks):
    """All six planes of the merged metric (order 00, 11, 22, 01, 02, 12, bp5/step-64.cu:107-113; K = d xi / d x) on the HIP path against the
    closed form of tests/test_oracle_known_answers.py::closed_form_cell_matrix_affine: the mesh's nodes are mapped by one affine map (shear,
    stretch, rotation), so every cell is the same parallelepiped and Gauss(p+1) quadrature is exact -- geometry kernel, merged metric and the
    fused operator kernels (pencil kernel; block kernel on bricks) against matrices that do not come from the oracle"""
    from test_oracle_known_answers import AFFINE_MAP, closed_form_cell_matrix_affine
    cells = (4, 4, 2) if bricks else (3, 2, 2)
    kw = dict(cell_block=(4, 4, 2), dof_numbering=1, cell_block_order=1) if bricks else {}
    mesh = pkg.BrickMesh(p, cells, h=0.5, **kw)
    mesh.coords = np.ascontiguousarray(np.asarray(mesh.coords) @ AFFINE_MAP.T)      # (the host arrays are handed over by MatrixFree.reinit)
    mf = pkg.MatrixFree().reinit(mesh, pkg.QUAD_GAUSS, pkg.COEF_ONE)
    if bricks:
        mf.set_apply_variant(56)
    coef = mf.evaluate_coefficients()
    s = O.deterministic_src(mesh.n_owned, seed=13)
    dst = mf.initialize_dof_vector()
    mf.cell_loop(coef, dev(s), dst)
    Ae = closed_form_cell_matrix_affine(p, 0.5 * AFFINE_MAP)          # cell edge 0.5: x = (0.5 A) xi + const
    ref = np.zeros(mesh.n_owned)
    l2g = np.asarray(mesh.l2g).reshape(mesh.n_cells, -1).astype(np.int64)
    for c in range(mesh.n_cells):
        np.add.at(ref, l2g[c], Ae @ s[l2g[c]])
    assert rel(dst.cpu().numpy(), ref) < 1e-12


@pytest.mark.parametrize("p", [2, 3, 4])
def test_rhs_diagonal_l2_norm_and_helmholtz_against_closed_forms(p):
    """The steps either side of the operator on affine cube cells against closed forms built from the textbook 1-D matrices (no oracle):
    b_i = int phi_i (bp5/step-64.cu:401-405) = row sums of the mass matrix, Dirichlet rows zero; diag(A) = sum over cells of the element
    matrix's diagonal; ||u_h||_L2 of a linear field (bp5/step-64.cu:602-616) by exact integration; step-64's Helmholtz cell operator with
    a = 1: Laplace + h^3 M x M x M (native kernel, accumulate mode)."""
    from test_oracle_known_answers import _textbook_1d
    torch = _t()
    K, M = _textbook_1d(p)
    cells, h = (3, 2, 2), 0.5
    mesh = pkg.BrickMesh(p, cells, h=h)
    l2g = np.asarray(mesh.l2g).reshape(mesh.n_cells, -1).astype(np.int64)
    con = np.zeros(mesh.n_owned, bool)
    con[np.asarray(mesh.constrained).astype(np.int64)] = True
    Ae = h * (np.kron(M, np.kron(M, K)) + np.kron(M, np.kron(K, M)) + np.kron(K, np.kron(M, M)))
    M3 = h ** 3 * np.kron(M, np.kron(M, M))
    op = pkg.PoissonOperator(mesh, pkg.QUAD_GAUSS, pkg.COEF_ONE)
    # RHS
    b_ref = np.zeros(mesh.n_owned)
    for c in range(mesh.n_cells):
        np.add.at(b_ref, l2g[c], M3.sum(axis=1))
    b_ref[con] = 0.0
    assert rel(op.assemble_rhs().cpu().numpy(), b_ref) < 1e-12
    # diagonal (unconstrained rows)
    d_ref = np.zeros(mesh.n_owned)
    for c in range(mesh.n_cells):
        np.add.at(d_ref, l2g[c], np.diag(Ae))
    d = op.compute_diagonal().cpu().numpy()
    assert np.linalg.norm((d - d_ref)[~con]) < 1e-12 * np.linalg.norm(d_ref[~con])
    # L2 norm of u = 2x - y + z/2 + 1 on [0, 1.5] x [0, 1] x [0, 1]: exact integral of a quadratic
    X = np.asarray(mesh.coords)[:mesh.n_owned]
    u = 2.0 * X[:, 0] - X[:, 1] + 0.5 * X[:, 2] + 1.0
    import sympy as sy
    x, y, z = sy.symbols("x y z")
    exact = float(sy.sqrt(sy.integrate((2 * x - y + z / 2 + 1) ** 2, (x, 0, sy.Rational(3, 2)), (y, 0, 1), (z, 0, 1))))
    assert abs(op.l2_norm_solution(dev(u)) - exact) < 1e-12 * exact
    # Helmholtz cell operator, a = 1
    hop = pkg.HelmholtzOperator(mesh, pkg.QUAD_GAUSS, pkg.COEF_ONE)
    s = O.deterministic_src(mesh.n_owned, seed=17)
    acc = hop.initialize_dof_vector()
    hop.mf_data.cell_loop(hop.coef, dev(s), acc)
    h_ref = np.zeros(mesh.n_owned)
    for c in range(mesh.n_cells):
        np.add.at(h_ref, l2g[c], (Ae + M3) @ s[l2g[c]])
    assert rel(acc.cpu().numpy(), h_ref) < 1e-12


@pytest.mark.parametrize("p,variant", [(4, 0), (4, 1), (4, 2), (4, 3), (4, 4), (4, 5), (5, 0), (5, 1), (6, 0), (6, 1), (8, 0), (8, 1),
                                       (1, 10), (2, 10), (3, 10), (4, 10), (4, 11), (4, 12), (4, 13), (5, 10), (6, 10), (7, 10), (8, 10),
                                       (1, 50), (2, 50), (3, 50), (4, 50), (4, 51), (5, 50), (6, 50), (7, 50), (8, 50), (4, 110), (4, 52), (4, 53), (4, 54), (4, 55)])
@pytest.mark.parametrize("quad", [0, 1])
def test_kernel_variants(p, variant, quad):
    cells = (7, 3, 1) if p <= 5 else (5, 1, 1)
    pr = O.Problem(p, cells, quad, deform_amp=0.03)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.03)
    mf = pkg.MatrixFree().reinit(mesh, quad)
    mf.set_apply_variant(variant)
    coef = mf.evaluate_coefficients()
    s = O.deterministic_src(mesh.n_owned, seed=5)
    dst = mf.initialize_dof_vector()
    mf.cell_loop(coef, dev(s), dst)
    ref = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)
    assert rel(dst.cpu().numpy(), ref) < TOL_OP


@pytest.mark.parametrize("p,cells,n_ranks,kw,variant", [
    (4, (5, 4, 7), 2, {}, 0), (4, (5, 4, 7), 3, {}, 10),
    (4, (9, 10, 17), 2, dict(cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1), 0),
    (4, (9, 6, 13), 3, dict(cell_block=(4, 4, 2)), 50),
    (2, (4, 3, 5), 2, {}, 0), (3, (3, 3, 4), 2, {}, 0), (6, (2, 2, 4), 2, {}, 0), (4, (3, 3, 6), 2, {}, 70)])
def test_rank_local_kernels_with_ghosts(p, cells, n_ranks, kw, variant):
    """The z-slab meshes of a multi-rank run, one after the other on this one GPU: every rank applies its
    cells to owned + ghost values and leaves partial sums in owned + ghost entries; summed over the ranks
    through global_ids they must reproduce the global operator.  (The exchange itself is RCCL and needs
    one GPU per rank; this covers every kernel family on meshes WITH ghost DoFs, interior cells first.)"""
    pr = O.Problem(p, cells, 0, deform_amp=0.03, kappa=O.kappa_step64)
    s = O.deterministic_src(pr.mesh.n_dofs, seed=31)
    ref = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)
    total = np.zeros_like(ref)
    for r in range(n_ranks):
        mesh = pkg.BrickMesh(p, cells, deform_amp=0.03, rank=r, n_ranks=n_ranks, **kw)
        assert (mesh.n_ghost > 0) == (r > 0)
        g = mesh.global_ids.astype(np.int64)
        # the library's step64 coefficient is a function of the physical point, so rank-local == global
        op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
        op.mf_data.set_apply_variant(56 if kw.get("cell_block_order") == 1 else variant)   # (0 picks 56 only on large meshes)
        op.mf_data.set_block_workgroups(8)                # several bricks per persistent workgroup
        if kw.get("cell_block_order") == 1:               # brick-major numbering: every rank gets the packed-index kernel,
            nb, max_runs, packed = op.mf_data.block_plan_info()   # also the ranks whose boundary bricks carry ghost rows
            assert packed and max_runs <= 128
        dst = op.initialize_dof_vector()
        assert dst.numel() == mesh.n_owned + mesh.n_ghost
        op.mf_data.cell_loop(op.coef, dev(s[g]), dst)
        np.add.at(total, g, dst.cpu().numpy())
        if mesh.n_interior_cells < mesh.n_cells:          # the two ranges of an overlapped schedule compose
            d2 = op.initialize_dof_vector()
            op.mf_data.set_apply_variant(3 if p == 4 else variant if variant < 50 else 0)
            op.mf_data.cell_loop(op.coef, dev(s[g]), d2, 0, mesh.n_interior_cells)
            assert float(d2[mesh.n_owned:].abs().max()) == 0.0      # interior cells touch no ghost
            op.mf_data.cell_loop(op.coef, dev(s[g]), d2, mesh.n_interior_cells, mesh.n_cells)
            assert rel(d2.cpu().numpy(), dst.cpu().numpy()) < TOL_OP
    assert rel(total, ref) < TOL_OP


@pytest.mark.parametrize("variant", [0, 3, 10, 11, 50])
def test_cell_ranges_and_accumulation(variant):
    """cell_loop accumulates (do_zero_out = false semantics) and ranges compose (ranges that cut
    through a team exercise the masked-cell path of the team kernel)."""
    p, cells = 4, (4, 3, 2)
    pr = O.Problem(p, cells, 0)
    mesh = pkg.BrickMesh(p, cells)
    mf = pkg.MatrixFree().reinit(mesh, 0)
    mf.set_apply_variant(variant)
    coef = mf.evaluate_coefficients()
    s = O.deterministic_src(mesh.n_owned, seed=11)
    dst = mf.initialize_dof_vector()
    src = dev(s)
    mf.cell_loop(coef, src, dst, 0, 7)
    mf.cell_loop(coef, src, dst, 7, 7)               # empty range
    mf.cell_loop(coef, src, dst, 7, mesh.n_cells)
    ref = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)
    assert rel(dst.cpu().numpy(), ref) < TOL_OP
    mf.cell_loop(coef, src, dst)
    assert rel(dst.cpu().numpy(), 2 * ref) < TOL_OP
    with pytest.raises(pkg.BP5Error):
        mf.cell_loop(coef, src, dst, 0, mesh.n_cells + 1)
    with pytest.raises(pkg.BP5Error):
        mf.cell_loop(coef, src, src)


@pytest.mark.parametrize("numbering", [0, 1])
@pytest.mark.parametrize("p,cells,block,quad", [(4, (6, 5, 4), (4, 4, 2), 0), (4, (8, 8, 8), (4, 4, 4), 1), (2, (7, 6, 5), (4, 4, 4), 0),
                                                (6, (3, 3, 2), (2, 2, 2), 0), (8, (2, 2, 2), (2, 2, 2), 1)])
def test_block_kernel_on_blocked_mesh(p, cells, block, quad, numbering):
    """Block-assembled kernel on a mesh whose cells are ordered brick by brick (partial bricks at
    the domain edges), with lexicographic and block-major DoF numbering: overwrite and accumulate
    modes, bitwise run-to-run reproducibility, CG."""
    torch = _t()
    pr = O.Problem(p, cells, quad, deform_amp=0.03, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.03, cell_block=block, dof_numbering=numbering)
    assert mesh.cell_block_offsets is not None and int(mesh.cell_block_offsets[-1]) == mesh.n_cells
    perm = mesh.global_ids.astype(np.int64)            # local index -> lexicographic id of the oracle
    assert {tuple(perm[r]) for r in mesh.l2g.astype(np.int64)} == {tuple(r) for r in pr.mesh.l2g.astype(np.int64)}
    op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64)
    op.mf_data.set_apply_variant(50)
    s = O.deterministic_src(mesh.n_owned, seed=13)      # lexicographic
    src = dev(s[perm])
    d1 = op.initialize_dof_vector()
    d1.fill_(float("nan"))                              # overwrite mode must define every entry
    op.vmult(d1, src)
    ref = pr.vmult(s)[perm]
    assert rel(d1.cpu().numpy(), ref) < TOL_OP
    d2 = op.initialize_dof_vector()
    d2.fill_(float("nan"))
    op.vmult(d2, src)
    assert torch.equal(d1, d2)                          # no atomics anywhere: bitwise reproducible
    acc = op.initialize_dof_vector()
    op.mf_data.cell_loop(op.coef, src, acc)
    op.mf_data.cell_loop(op.coef, src, acc)             # accumulate mode
    ref2 = 2 * O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)[perm]
    assert rel(acc.cpu().numpy(), ref2) < TOL_OP
    # the other kernels on the permuted numbering
    for v in (3 if p == 4 else 0, 10):                  # (p = 4: 0 would resolve to the block kernel on this mesh)
        op.mf_data.set_apply_variant(v)
        d3 = op.initialize_dof_vector()
        op.vmult(d3, src)
        assert rel(d3.cpu().numpy(), ref) < TOL_OP
    op.mf_data.set_apply_variant(50)
    # CG through the block kernel
    b = op.assemble_rhs()
    assert rel(b.cpu().numpy(), pr.rhs()[perm]) < 1e-13
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 6)
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(6, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG


@pytest.mark.parametrize("variant,order", [(52, 0), (53, 1), (56, 0), (56, 1), (57, 1), (58, 1), (59, 1), (49, 1), (48, 1), (3, 1)])
@pytest.mark.parametrize("quad", [0, 1])
def test_block_kernel_shapes_p4(variant, order, quad):
    """The other p = 4 block-kernel shapes (32 lanes per cell, single/double buffered, three transpose
    tiles or one tile used field after field, non-temporal metric loads, list or run-length write-out) on a
    deformed mesh with partial bricks, cells inside a brick in lexicographic or parity-class order; 3 is the
    pencil kernel on the same ordering."""
    torch = _t()
    p, cells = 4, (9, 6, 5)
    pr = O.Problem(p, cells, quad, deform_amp=0.03, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.03, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=order)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64)
    op.mf_data.set_apply_variant(variant)
    s = O.deterministic_src(mesh.n_owned, seed=29)
    d1 = op.initialize_dof_vector()
    d1.fill_(float("nan"))
    op.vmult(d1, dev(s[perm]))
    assert rel(d1.cpu().numpy(), pr.vmult(s)[perm]) < TOL_OP
    d2 = op.initialize_dof_vector()
    op.vmult(d2, dev(s[perm]))
    if variant != 3:
        assert torch.equal(d1, d2)                      # block kernels: no atomics, bitwise reproducible
    assert op.mf_data.get_apply_variant() == variant
    b = op.assemble_rhs()
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 6)
    x = op.initialize_dof_vector()
    pkg.SolverCGFullMerge(pkg.IterationNumberControl(6, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
    assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG


@pytest.mark.parametrize("variant", [50, 52, 56, 49, 59, 57])
@pytest.mark.parametrize("cells", [(6, 9, 10), (10, 9, 7)])
def test_block_kernel_persistent_loop_over_unequal_blocks(variant, cells):
    """A persistent workgroup walks several bricks whose DoF lists differ in length (partial bricks at the
    mesh edge come first in some ranges): the grid is capped at 8 workgroups so that the small mesh exercises
    the loop, and a kernel that leaves non-zero LDS behind runs right before (regression: the accumulator
    was only cleared up to the length of the workgroup's FIRST block)."""
    p = 4
    pr = O.Problem(p, cells, 0, deform_amp=0.03, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.03, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
    assert len(mesh.cell_block_offsets) - 1 >= 18
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    op.mf_data.set_block_workgroups(8)
    s = O.deterministic_src(mesh.n_owned, seed=41)
    ref = pr.vmult(s)[perm]
    refc = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)[perm]
    src = dev(s[perm])
    for rep in range(3):
        op.mf_data.set_apply_variant(3)                 # pencil kernel: fills LDS tiles with data
        scratch = op.initialize_dof_vector()
        op.vmult(scratch, src)
        op.mf_data.set_apply_variant(variant)
        d = op.initialize_dof_vector()
        d.fill_(float("nan"))
        op.vmult(d, src)
        assert rel(d.cpu().numpy(), ref) < TOL_OP
        c = op.initialize_dof_vector()
        op.mf_data.cell_loop(op.coef, src, c)
        assert rel(c.cpu().numpy(), refc) < TOL_OP


@pytest.mark.parametrize("block,cells,numbering,order", [((4, 4, 2), (9, 6, 5), 1, 1), ((2, 2, 2), (5, 4, 3), 1, 0), ((5, 3, 4), (11, 7, 9), 1, 1),
                                                         ((3, 3, 3), (7, 7, 4), 1, 1), ((4, 4, 4), (9, 5, 6), 0, 1), ((8, 2, 2), (17, 5, 3), 1, 0)])
def test_default_block_kernel_on_other_brick_shapes(block, cells, numbering, order):
    """The p = 4 default shape (packed indices when the lists have <= 128 runs, else run-length or list write-out) on
    bricks that are not 4x4x4: odd edge lengths (uneven parity classes -> under-full and multi-round passes), flat and
    long bricks, lexicographic numbering (hundreds of short runs -> unpacked fallback); several bricks per workgroup."""
    torch = _t()
    p = 4
    pr = O.Problem(p, cells, 0, deform_amp=0.03, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.03, cell_block=block, dof_numbering=numbering, cell_block_order=order)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    nb, max_runs, packed = op.mf_data.block_plan_info()
    assert packed == (max_runs <= 64) and (numbering == 1 or not packed)
    s = O.deterministic_src(mesh.n_owned, seed=53)
    ref = pr.vmult(s)[perm]
    refc = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)[perm]
    first = None
    for v in (56, 49, 48):
        op.mf_data.set_apply_variant(v)
        op.mf_data.set_block_workgroups(8)
        d1 = op.initialize_dof_vector()
        d1.fill_(float("nan"))
        op.vmult(d1, dev(s[perm]))
        assert rel(d1.cpu().numpy(), ref) < TOL_OP
        d2 = op.initialize_dof_vector()
        op.vmult(d2, dev(s[perm]))
        assert torch.equal(d1, d2)
        first = d1 if first is None else first
        assert torch.equal(d1, first)                    # packed / unpacked / CSR-combine builds: bitwise the same sums
        c = op.initialize_dof_vector()
        op.mf_data.cell_loop(op.coef, dev(s[perm]), c)
        assert rel(c.cpu().numpy(), refc) < TOL_OP


@pytest.mark.parametrize("p,cells,block", [(1, (17, 9, 10), (8, 8, 8)), (2, (9, 8, 5), (8, 8, 4)), (3, (9, 5, 6), (8, 4, 4)), (3, (5, 5, 5), (4, 4, 4)), (5, (5, 6, 3), (4, 4, 2)),
                                           (6, (5, 4, 3), (4, 4, 2)), (6, (4, 3, 3), (4, 2, 2)), (7, (5, 3, 3), (4, 2, 2)), (8, (3, 3, 3), (2, 2, 2))])
@pytest.mark.parametrize("quad", [0, 1])
def test_deterministic_block_kernel_for_the_other_degrees(p, cells, block, quad):
    """The non-atomic default shape (sequential tiles, run-length write-out, packed indices: variant 56) on p = 1, 2, 3, 5, 6, 7, 8 --
    the reference's scatter is an FP64 atomicAdd (bp5/fe_evaluation_gl.h:176-180: non-deterministic).  Bricks with partial
    ones at the mesh edge, several bricks per persistent workgroup, deformed cells, variable coefficient: equal to the oracle,
    bitwise reproducible; and the merged CG with the dot products fused into it matches the oracle's plain CG."""
    torch = _t()
    pr = O.Problem(p, cells, quad, h=0.2, deform_amp=0.03, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, h=0.2, deform_amp=0.03, cell_block=block, dof_numbering=1, cell_block_order=1)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64)
    mf = op.mf_data
    mf.set_apply_variant(56)
    mf.set_block_workgroups(8)
    assert mf.block_plan_info()[2]
    s = O.deterministic_src(pr.mesh.n_dofs, seed=61)
    ref = pr.vmult(s)[perm]
    outs = []
    for _ in range(3):
        d = op.initialize_dof_vector()
        d.fill_(float("nan"))
        op.vmult(d, dev(s[perm]))
        outs.append(d)
    assert rel(outs[0].cpu().numpy(), ref) < TOL_OP
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    acc = torch.full_like(outs[0], 0.5)                   # accumulate mode: cell_loop adds to dst
    mf.cell_loop(op.coef, dev(s[perm]), acc)
    refc = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)[perm]
    assert rel(acc.cpu().numpy() - 0.5, refc) < TOL_OP
    b = op.assemble_rhs()
    its = 8
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), its)
    sols = []
    for fused in (True, False):
        mf.set_cg_fusion(fused)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.dot_products_fused == fused
        assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG
        sols.append(x)
    assert float((sols[0] - sols[1]).abs().max()) < 1e-11 * float(sols[1].abs().max())   # (only the summation order differs)


def test_streaming_policy_changes_no_bit():
    """bp5_mf_set_streaming: non-temporal accesses to once-used data (metric planes in the lattice block kernel, v and x in the update kernel: the
    choice up to 2.4e7 local DoFs) and ordinary ones (beyond) are separate kernel builds -- same bits from the operator and from the merged CG
    solve (an odd and an even iteration count: both update kernels and the epilogue), the kernel name tells which block kernel ran."""
    torch = _t()
    mesh = pkg.BrickMesh(4, (9, 8, 6), h=0.2, deform_amp=0.03, cell_block=(4, 4, 2), dof_numbering=1, cell_block_order=1)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    mf = op.mf_data
    mf.set_apply_variant(56)
    mf.set_block_workgroups(8)
    g = torch.Generator(device="cuda:0").manual_seed(4)
    src = torch.rand(mesh.n_owned, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    b = op.assemble_rhs()
    got = {}
    for policy in (-1, 0, 1):
        mf.set_streaming(policy)
        d = op.initialize_dof_vector()
        op.vmult(d, src)
        xs = []
        for its in (6, 7):
            x = op.initialize_dof_vector()
            ctl = pkg.IterationNumberControl(its, 0.0)
            pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
            xs.append(x)
        got[policy] = (d, xs[0], int(ctl.apply_kernel.split(",")[-1].rstrip(">")), xs[1])
    assert got[1][2] & 32768 and got[-1][2] & 32768 and not got[0][2] & 32768 and got[0][2] & 16777216
    for policy in (-1, 1):
        assert all(torch.equal(got[policy][i], got[0][i]) for i in (0, 1, 3))
    with pytest.raises(pkg.BP5Error):
        mf.set_streaming(2)


def test_tuning_knobs_are_fields_of_the_handle_and_change_no_bit():
    """bp5_mf_set_tuning (round 4; ADVICE r3: the A/B knobs used to be getenv calls on the solve path): the update kernels launched flat (one trip per
    workgroup, the default) or on the capped grid-stride grid of rounds 1-3, 1 / 2 / 4 accesses per lane, v and x non-temporal or not: the same bits from
    the merged solve (odd and even iteration counts: both update kernels and the epilogue) and from the vector updates; two handles of one process differ
    in a knob; bad values are refused."""
    torch = _t()
    mesh = pkg.BrickMesh(4, (9, 8, 6), h=0.2, deform_amp=0.03, cell_block=(4, 4, 2), dof_numbering=1, cell_block_order=1)
    op, op2 = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64), pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    mf = op.mf_data
    mf.set_apply_variant(56)        # the deterministic block kernel (a mesh this small would get the atomic pencil kernel: not reproducible bit for bit)
    mf.set_block_workgroups(8)
    assert mf.get_tuning("update_flat") == 1 and mf.get_tuning("update_unroll") == 1 and mf.get_tuning("lattice_indices") == 1
    op2.mf_data.set_tuning("update_flat", 0)
    assert op2.mf_data.get_tuning("update_flat") == 0 and mf.get_tuning("update_flat") == 1       # per handle, not per process
    b = op.assemble_rhs()
    ref = None
    for flat, unroll, nt in ((1, 1, -1), (0, 4, 0), (1, 2, 1), (1, 4, 0), (0, 1, 1), (0, 2, -1)):
        mf.set_tuning("update_flat", flat); mf.set_tuning("update_unroll", unroll); mf.set_tuning("update_nt", nt)
        xs = []
        for its in (6, 7):
            x = op.initialize_dof_vector()
            pkg.SolverCGFullMerge(pkg.IterationNumberControl(its, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
            xs.append(x)
        y = torch.arange(mesh.n_owned, dtype=torch.float64, device="cuda:0") * 1e-3
        v, vb = pkg.Vector(mf), pkg.Vector(mf)
        v.values.copy_(y); vb.values.copy_(b); v.sadd(0.5, 2.0, vb)
        if ref is None:
            ref = (xs, v.values.clone())
        else:
            assert torch.equal(xs[0], ref[0][0]) and torch.equal(xs[1], ref[0][1]) and torch.equal(v.values, ref[1])
    for knob, bad in (("update_unroll", 3), ("update_nt", 2), ("update_flat", 2), ("early_gather", -1), (99, 0)):
        with pytest.raises(pkg.BP5Error):
            mf.set_tuning(knob, bad)
    assert isinstance(mf.wait_value_available(), bool)   # (creates the communication stream and runs the producer / consumer self-check)


@pytest.mark.parametrize("cells,block,kw", [((13, 8, 6), (4, 4, 4), {}), ((48, 4, 4), (4, 4, 4), {}), ((8, 8, 12), (4, 4, 2), dict(rank=1, n_ranks=2)),
                                            ((13, 9, 9), (4, 4, 4), dict(rank=1, n_ranks=3))])
def test_face_carry_keeps_shared_faces_in_lds_and_changes_no_bit_of_v(cells, block, kw):
    """BP5_TUNE_FACE_CARRY (round 4): the interior of the face two consecutive bricks of one workgroup's range share stays in LDS from
    the first brick's write-out to the second's, which stores p1 + p0 as an owner store; the combine pass of that launch walks tables without those
    DoFs.  v = A u is bitwise the result without the carry (p0 + p1), in overwrite and in accumulate mode, for every workgroup count (the carried set
    depends on the partition, the sums do not); the merged solver (its dot products take the carried DoFs from another kernel: rounding-level
    differences) is reproducible run to run and agrees to 1e-12.  Full and partial bricks, a slab with its ghost plane."""
    torch = _t()
    mesh = pkg.BrickMesh(4, cells, h=0.2, deform_amp=0.03, cell_block=block, dof_numbering=1, cell_block_order=1, **kw)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    mf = op.mf_data
    mf.set_apply_variant(56)
    assert mf.get_tuning("face_carry") in (0, 1)
    faces, n_shared, _ = mf.block_plan_carry()
    assert faces > 0 and n_shared > 0
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    u = torch.rand(mf.n_local, dtype=torch.float64, device="cuda:0", generator=g)
    w0 = torch.rand(mf.n_local, dtype=torch.float64, device="cuda:0", generator=g)
    ref = None
    for n_wg in (8, 16, 0):
        mf.set_block_workgroups(n_wg)
        for carry in (0, 1):
            mf.set_tuning("face_carry", carry)
            v = op.initialize_dof_vector(); op.vmult(v, u)
            in_tables = mf.block_plan_carry()[2]
            assert in_tables == n_shared if not carry else in_tables < n_shared if n_wg == 8 else in_tables <= n_shared   # (8 workgroups: several bricks each)
            op.do_zero_out = False
            w = w0.clone(); op.vmult(w, u)
            op.do_zero_out = True
            if ref is None:
                ref = (v.clone(), w.clone())
            assert torch.equal(v, ref[0]) and torch.equal(w, ref[1]), (n_wg, carry)
    if kw:
        return      # (a slab without its neighbours: operator only)
    mf.set_block_workgroups(8)
    b = op.assemble_rhs()
    xs = {}
    for carry in (0, 1, 1):
        mf.set_tuning("face_carry", carry)
        x = op.initialize_dof_vector()
        pkg.SolverCGFullMerge(pkg.IterationNumberControl(9, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
        if carry in xs:
            assert torch.equal(xs[carry], x)            # same bits run to run
        xs[carry] = x
    assert mf.block_plan_carry()[2] < n_shared              # the fused solve carried
    assert torch.allclose(xs[0], xs[1], rtol=1e-12, atol=1e-14 * float(xs[0].abs().max()))
    xp = op.initialize_dof_vector()
    pkg.SolverCG(pkg.IterationNumberControl(9, 0.0)).solve(op, xp, b, pkg.DiagonalMatrix())
    assert torch.allclose(xp, xs[1], rtol=1e-9, atol=1e-11 * float(xp.abs().max()))


def test_set_operator_is_refused_once_the_metric_array_is_sized():
    """bp5_mf_set_operator(BP5_OP_HELMHOLTZ) after the caller sized or filled a six-plane metric array would make every later kernel read a
    seventh plane out of bounds -- the handle remembers the plane count it handed out and refuses the switch (either direction)."""
    _t()
    mesh = pkg.BrickMesh(3, (3, 2, 2), h=0.5)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)         # sized and filled: six planes
    with pytest.raises(pkg.BP5Error):
        op.mf_data.set_operator(pkg.OP_HELMHOLTZ)
    op.mf_data.set_operator(pkg.OP_POISSON)                    # (no change of the plane count: fine)
    hop = pkg.HelmholtzOperator(mesh, 0, pkg.COEF_STEP64)       # operator first, then seven planes
    with pytest.raises(pkg.BP5Error):
        hop.mf_data.set_operator(pkg.OP_POISSON)


@pytest.mark.parametrize("p,cells,block,kw", [(4, (9, 8, 6), (4, 4, 4), {}), (4, (8, 8, 12), (4, 4, 2), dict(rank=1, n_ranks=2)), (4, (8, 8, 13), (4, 4, 4), dict(rank=1, n_ranks=3)),
                                            (1, (17, 9, 10), (8, 8, 8), {}), (2, (9, 8, 5), (8, 8, 4), {}), (3, (9, 5, 6), (8, 4, 4), dict(rank=1, n_ranks=2)),
                                            (5, (7, 5, 3), (6, 4, 2), {}), (6, (5, 4, 3), (4, 4, 2), {}), (7, (5, 3, 3), (4, 2, 2), {}), (8, (3, 3, 3), (2, 2, 2), {})])
def test_lattice_blocks_need_no_index_stream(p, cells, block, kw):
    """Structured bricks with brick-major numbering (the library's own generator; partial bricks at the mesh edges; the ghost-touching layer
    of a slab with its block-major ghost plane): every block is recognised as a LATTICE block (topologically, then verified entry by entry on
    the host), and the kernel build that computes list slots and DoFs in closed form -- no per-DoF index stream -- gives bitwise the result
    of the packed-index build (BP5_TUNE_LATTICE_INDICES = 0) and matches the atomic pencil kernel."""
    torch = _t()
    mesh = pkg.BrickMesh(p, cells, h=0.2, deform_amp=0.03, cell_block=block, dof_numbering=1, cell_block_order=1, **kw)
    ops = []
    for lattice in (1, 0):
        op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
        op.mf_data.set_tuning("lattice_indices", lattice)          # a knob of the HANDLE (two handles of one process differ here), fixed once the plan is built
        op.mf_data.set_tuning("face_carry", 0)                     # (p = 4 lattice build: the carried faces' dot-product terms would be summed by another kernel -- v is the same bits, the solve is not)
        op.mf_data.set_apply_variant(56)
        op.mf_data.set_block_workgroups(8)
        nb, _, packed = op.mf_data.block_plan_info()
        assert packed and op.mf_data.block_plan_lattice() == (nb if lattice else 0)
        assert op.mf_data.get_tuning("lattice_indices") == lattice
        with pytest.raises(pkg.BP5Error):
            op.mf_data.set_tuning("lattice_indices", 1 - lattice)  # the plan exists: refused instead of silently ignored
        ops.append(op)
    n = mesh.n_owned + mesh.n_ghost
    g = torch.Generator(device="cuda:0").manual_seed(9)
    src = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    outs, names = [], []
    b = ops[0].assemble_rhs()                                  # (one right-hand side for both: its assembly scatters with atomics)
    for op in ops:
        d = op.initialize_dof_vector()
        d.fill_(float("nan"))
        pkg.lib().bp5_apply(op.mf_data.handle, _cptr(op.coef), _cptr(src), _cptr(d), 1)    # (single-rank entry point: ghosts are plain entries here)
        outs.append(d)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(5, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix()) if not kw else None
        names.append(ctl.apply_kernel)
        outs.append(x)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    if not kw:
        assert int(names[0].split(",")[-1].rstrip(">")) & 16777216 and not int(names[1].split(",")[-1].rstrip(">")) & 16777216
    ops[0].mf_data.set_apply_variant(1 if p in (1, 3) else 0 if p != 4 else 3)
    ref = ops[0].initialize_dof_vector()
    pkg.lib().bp5_apply(ops[0].mf_data.handle, _cptr(ops[0].coef), _cptr(src), _cptr(ref), 1)
    assert float((ref - outs[0]).abs().max()) < 1e-12 * float(ref.abs().max())


def _cptr(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("quad", [0, 1])
def test_block_kernel_on_block_aligned_cell_ranges(quad):
    """bp5_apply_cells with the block kernel on cell ranges that are unions of whole bricks (what a host that
    overlaps the halo exchange itself calls: interior bricks, then boundary bricks): the ranges compose to the
    whole cell loop; DoFs shared with bricks outside the range are added atomically.  Unaligned ranges are
    refused for the explicit variant."""
    p, cells = 4, (9, 8, 6)
    pr = O.Problem(p, cells, quad, deform_amp=0.03, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.03, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64)
    mf = op.mf_data
    mf.set_apply_variant(56)
    mf.set_block_workgroups(8)
    off = mesh.cell_block_offsets
    s = O.deterministic_src(mesh.n_owned, seed=47)
    src = dev(s[perm])
    ref = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)[perm]
    for cuts in ([5], [1, 2, 9], list(range(1, len(off) - 1))):
        d = mf.initialize_dof_vector()
        edges = [0] + [int(off[k]) for k in cuts] + [mesh.n_cells]
        for a, b in zip(edges[:-1], edges[1:]):
            mf.cell_loop(op.coef, src, d, a, b)
        assert rel(d.cpu().numpy(), ref) < TOL_OP
    with pytest.raises(pkg.BP5Error):
        mf.cell_loop(op.coef, src, mf.initialize_dof_vector(), 0, int(off[1]) + 1)
    # slab mesh of rank 1 of 2: the interior / boundary split of the overlapped schedule is brick-aligned
    m1 = pkg.BrickMesh(p, cells, deform_amp=0.03, rank=1, n_ranks=2, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
    assert m1.n_interior_cells in set(int(x) for x in m1.cell_block_offsets)
    op1 = pkg.PoissonOperator(m1, quad, pkg.COEF_STEP64)
    g1 = m1.global_ids.astype(np.int64)
    whole, split = op1.initialize_dof_vector(), op1.initialize_dof_vector()
    op1.mf_data.set_apply_variant(3)
    op1.mf_data.cell_loop(op1.coef, dev(s_ := O.deterministic_src(pr.mesh.n_dofs, seed=48)[g1]), whole)
    op1.mf_data.set_apply_variant(56)
    op1.mf_data.cell_loop(op1.coef, dev(s_), split, 0, m1.n_interior_cells)
    assert float(split[m1.n_owned:].abs().max()) == 0.0
    op1.mf_data.cell_loop(op1.coef, dev(s_), split, m1.n_interior_cells, m1.n_cells)
    assert rel(split.cpu().numpy(), whole.cpu().numpy()) < TOL_OP


def test_default_variant_resolution():
    """What variant 0 means: p = 1, 3 team kernel; p = 4: block kernel when the mesh comes in cell blocks
    that fit, team kernel in affine mode, pencil kernel otherwise; other degrees pencil kernel."""
    def ev(p, cells, quad=0, geometry=None, **kw):
        op = pkg.PoissonOperator(pkg.BrickMesh(p, cells, **kw), quad, pkg.COEF_ONE, **({"geometry": geometry} if geometry is not None else {}))
        return op.mf_data.get_apply_variant()
    assert ev(4, (4, 4, 4)) == 0
    assert ev(4, (5, 4, 4), cell_block=(4, 4, 4)) == 0          # too few bricks per persistent workgroup (see the
    assert ev(4, (5, 4, 4), cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1) == 0   # full-size test for 56)
    assert ev(4, (8, 8, 8), cell_block=(8, 8, 8)) == 0          # 33^3 accumulator does not fit in LDS
    assert ev(4, (3, 3, 3), geometry=pkg.GEOM_AFFINE) == 10
    assert ev(3, (3, 3, 3)) == 10 and ev(1, (3, 3, 3)) == 10
    assert ev(2, (3, 3, 3)) == 0 and ev(5, (2, 2, 2)) == 0 and ev(6, (2, 2, 2), cell_block=(2, 2, 2)) == 0


@pytest.mark.parametrize("p", range(1, 9))
@pytest.mark.parametrize("quad", [0, 1])
def test_affine_geometry_mode(p, quad):
    """BP5_GEOM_AFFINE (per-cell K K^T + one scalar plane) gives the same operator as the six stored
    planes on affine meshes (the reference's meshes: cubes), with a variable coefficient."""
    cells = (3, 3, 2) if p <= 4 else (3, 2, 1)
    pr = O.Problem(p, cells, quad, h=0.5, kappa=O.kappa_step64)
    op = pkg.PoissonOperator(pkg.BrickMesh(p, cells, h=0.5), quad, pkg.COEF_STEP64, geometry=pkg.GEOM_AFFINE)
    assert op.coef is None
    s = O.deterministic_src(pr.mesh.n_dofs, seed=17)
    dst = op.initialize_dof_vector()
    op.vmult(dst, dev(s))
    assert rel(dst.cpu().numpy(), pr.vmult(s)) < TOL_OP
    b = op.assemble_rhs()
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 8)
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(8, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert rel(x.cpu().numpy(), xr) < TOL_CG


@pytest.mark.parametrize("variant,block,numbering", [(10, (0, 0, 0), 0), (110, (0, 0, 0), 0), (50, (4, 4, 4), 1), (51, (4, 4, 2), 0), (54, (4, 4, 4), 1), (55, (4, 4, 4), 0), (56, (4, 4, 4), 1), (57, (4, 4, 4), 0)])
@pytest.mark.parametrize("quad", [0, 1])
def test_affine_mode_team_and_block_kernels(variant, block, numbering, quad):
    p, cells = 4, (8, 5, 4)
    pr = O.Problem(p, cells, quad, h=0.25, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, h=0.25, cell_block=block, dof_numbering=numbering)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64, geometry=pkg.GEOM_AFFINE)
    op.mf_data.set_apply_variant(variant)
    s = O.deterministic_src(pr.mesh.n_dofs, seed=19)
    dst = op.initialize_dof_vector()
    dst.fill_(float("nan"))
    op.vmult(dst, dev(s[perm]))
    assert rel(dst.cpu().numpy(), pr.vmult(s)[perm]) < TOL_OP
    b = op.assemble_rhs()
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 8)
    x = op.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(8, 0.0)
    pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
    assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG


def test_affine_mode_rejects_deformed_mesh():
    mf = pkg.MatrixFree().reinit(pkg.BrickMesh(3, (2, 2, 2), deform_amp=0.03), 0)
    with pytest.raises(pkg.BP5Error) as e:
        mf.set_geometry_mode(pkg.GEOM_AFFINE)
    assert e.value.status == 5                      # BP5_ERR_UNSUPPORTED
    # sheared but affine cells are fine: handled by the same per-cell 3x3 metric (checked on a
    # stretched brick: h differs from 1)
    mf2 = pkg.MatrixFree().reinit(pkg.BrickMesh(2, (2, 3, 2), h=0.37), 1)
    mf2.set_geometry_mode(pkg.GEOM_AFFINE)


@pytest.mark.parametrize("p,cells,variant", [(1, (3, 2, 40), 70), (2, (5, 3, 6), 70), (3, (4, 3, 5), 70), (4, (7, 3, 5), 70), (4, (3, 3, 4), 71),
                                             (4, (5, 2, 3), 72), (5, (3, 2, 4), 70), (6, (3, 2, 3), 70), (7, (2, 2, 3), 70), (8, (2, 1, 3), 70)])
@pytest.mark.parametrize("quad", [0, 1])
def test_march_kernel(p, cells, variant, quad):
    """z-marching kernel: chains longer than the segment cap (p = 1: 40 cells in z, cap 32), ragged teams,
    deformed mesh, block-major numbering on one case; operator, accumulate mode and CG."""
    numbering, block = (1, (2, 2, 2)) if (p == 3) else (0, (0, 0, 0))
    pr = O.Problem(p, cells, quad, deform_amp=0.03, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, deform_amp=0.03, cell_block=block, dof_numbering=numbering)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64)
    op.mf_data.set_apply_variant(variant)
    s = O.deterministic_src(pr.mesh.n_dofs, seed=23)
    src = dev(s[perm])
    dst = op.initialize_dof_vector()
    dst.fill_(float("nan"))
    op.vmult(dst, src)
    assert rel(dst.cpu().numpy(), pr.vmult(s)[perm]) < TOL_OP
    acc = op.initialize_dof_vector()
    op.mf_data.cell_loop(op.coef, src, acc)
    op.mf_data.cell_loop(op.coef, src, acc)
    assert rel(acc.cpu().numpy(), 2 * O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)[perm]) < TOL_OP
    b = op.assemble_rhs()
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 6)
    x = op.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(6, 0.0)
    pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
    assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG


def test_vmult_dirichlet_and_zero_out_flag():
    p, cells = 3, (3, 3, 3)
    pr = O.Problem(p, cells, 0, deform_amp=0.05)
    op = pkg.PoissonOperator(pkg.BrickMesh(p, cells, deform_amp=0.05), 0)
    s = O.deterministic_src(pr.mesh.n_dofs, seed=3)
    dst = op.initialize_dof_vector()
    dst.fill_(123.0)
    op.vmult(dst, dev(s))
    ref = pr.vmult(s)
    got = dst.cpu().numpy()
    assert rel(got, ref) < TOL_OP
    c = pr.mesh.constrained.astype(np.int64)
    assert np.array_equal(got[c], s[c])
    op.do_zero_out = False                            # bp5/step-64.cu:483: accumulate
    op.vmult(dst, dev(s))
    ref2 = 2 * O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)
    ref2[c] = s[c]
    assert rel(dst.cpu().numpy(), ref2) < TOL_OP


def test_rhs_and_l2_norm():
    for p, cells, amp in [(2, (8, 8, 8), 0.0), (4, (3, 2, 2), 0.05)]:
        pr = O.Problem(p, cells, 0, deform_amp=amp)
        op = pkg.PoissonOperator(pkg.BrickMesh(p, cells, deform_amp=amp), 0)
        b = op.assemble_rhs().cpu().numpy()
        assert rel(b, pr.rhs()) < 1e-13
        u = O.deterministic_src(pr.mesh.n_dofs, pr.mesh.constrained, seed=9)
        assert abs(op.l2_norm_solution(dev(u)) - O.l2_norm_solution(pr.mesh, u)) < 1e-13 * O.l2_norm_solution(pr.mesh, u)


def test_blas1():
    torch = _t()
    import ctypes as C
    mesh = pkg.BrickMesh(1, (2, 2, 2))
    mf = pkg.MatrixFree().reinit(mesh, 0)
    L, h = pkg.lib(), mf.handle
    for n in (1, 2, 255, 100001):
        rng = np.random.default_rng(n)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        X, Y = dev(x), dev(y)
        p = lambda t: C.c_void_p(t.data_ptr())
        L.bp5_vec_axpy(h, p(Y), 0.5, p(X), n); y = y + 0.5 * x
        assert np.allclose(Y.cpu().numpy(), y, rtol=1e-15, atol=0)
        L.bp5_vec_sadd(h, p(Y), -2.0, 3.0, p(X), n); y = -2.0 * y + 3.0 * x
        assert np.allclose(Y.cpu().numpy(), y, rtol=1e-15, atol=1e-15)
        r = C.c_double()
        L.bp5_vec_dot(h, p(X), p(Y), n, C.byref(r))
        assert abs(r.value - x @ y) < 1e-12 * (np.abs(x) @ np.abs(y))
        L.bp5_vec_equ(h, p(Y), -1.0, p(X), n)
        assert np.array_equal(Y.cpu().numpy(), -x)
        L.bp5_vec_fill(h, p(Y), 2.5, n)
        assert np.all(Y.cpu().numpy() == 2.5)
        L.bp5_vec_l2_norm(h, p(X), n, C.byref(r))
        assert abs(r.value - np.linalg.norm(x)) < 1e-13 * np.linalg.norm(x)
        z = C.c_int(-1)
        L.bp5_vec_all_zero(h, p(X), n, C.byref(z))
        assert z.value == 0
        X.zero_()
        L.bp5_vec_all_zero(h, p(X), n, C.byref(z))
        assert z.value == 1
        X[n - 1] = float("nan")                      # a NaN is not zero
        L.bp5_vec_all_zero(h, p(X), n, C.byref(z))
        assert z.value == 0
    mf.synchronize()


@pytest.mark.parametrize("p,cells,quad,kw,geom", [(1, (3, 3, 2), 0, {}, None), (2, (3, 2, 2), 1, {}, None), (3, (3, 2, 2), 0, {}, None),
                                                  (4, (5, 4, 3), 0, dict(cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1), None),
                                                  (4, (3, 2, 2), 1, {}, None), (6, (2, 2, 1), 0, {}, None), (8, (2, 1, 1), 1, {}, None),
                                                  (4, (3, 2, 2), 0, {}, "affine"), (5, (2, 2, 2), 1, {}, "affine")])
def test_operator_diagonal_and_jacobi_pcg(p, cells, quad, kw, geom):
    """bp5_compute_diagonal == diag(A_eff) of the oracle (sum-factorised, matrix-free), its reciprocal as the
    Jacobi vector of DiagonalMatrix: preconditioned CG matches the oracle's PCG iterate for iterate."""
    amp = 0.0 if geom else 0.04
    pr = O.Problem(p, cells, quad, h=0.5, deform_amp=amp, kappa=O.kappa_step64)
    mesh = pkg.BrickMesh(p, cells, h=0.5, deform_amp=amp, **kw)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.PoissonOperator(mesh, quad, pkg.COEF_STEP64, **({"geometry": pkg.GEOM_AFFINE} if geom else {}))
    dref = O.operator_diagonal(pr.mesh, pr.coef, pr.N, pr.D)
    d = op.compute_diagonal().cpu().numpy()
    assert np.abs(d - dref[perm]).max() < 1e-13 * np.abs(dref).max()
    dinv = op.compute_diagonal(invert=True)
    assert np.abs(dinv.cpu().numpy() * dref[perm] - 1.0).max() < 1e-13
    b = op.assemble_rhs()
    its = 8
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), its, diag=1.0 / dref)
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix(dinv))
        assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG


def test_jacobi_preconditioner_pays_off_on_a_variable_coefficient():
    """step-64's coefficient varies by a factor ~100 over the domain: Jacobi PCG needs fewer iterations than
    the identity-preconditioned CG of the benchmark to reach the same relative residual."""
    p, cells = 3, (6, 6, 6)
    op = pkg.PoissonOperator(pkg.BrickMesh(p, cells, h=1.0 / 3, deform_amp=0.05), 0, pkg.COEF_STEP64)
    b = pkg.Vector(op.mf_data)
    b.values.copy_(op.assemble_rhs())
    tol = 1e-8 * b.l2_norm()
    its = []
    for precond in (pkg.DiagonalMatrix(), pkg.DiagonalMatrix(op.compute_diagonal(invert=True))):
        x = pkg.Vector(op.mf_data)
        ctl = pkg.SolverControl(2000, tol)
        pkg.SolverCG(ctl).solve(op, x, b, precond)
        r = pkg.Vector().reinit(x)
        op.vmult(r, x)
        r.add(-1.0, b)
        assert r.l2_norm() < 2.0 * tol                # the true residual, not only the recurrence
        its.append(ctl.last_step())
    assert its[1] < 0.8 * its[0], its


def test_vector_class_mirrors_the_reference_solve():
    """PoissonProblem::solve written against the Vector class (bp5/step-64.cu:428-453,467): reinit from the
    operator, = 0, import of host values, l2_norm in the tolerance, all_zero, add/equ/sadd, cg.solve on vectors."""
    p_, cells = 3, (3, 3, 2)
    pr = O.Problem(p_, cells, 0, deform_amp=0.02)
    op = pkg.PoissonOperator(pkg.BrickMesh(p_, cells, deform_amp=0.02), 0, pkg.COEF_ONE)
    solution, rhs = pkg.Vector(op.mf_data), pkg.Vector(op.mf_data)
    assert solution.all_zero() and solution.local_size() == pr.mesh.n_dofs == solution.size()
    rhs.import_(pr.rhs())
    assert not rhs.all_zero() and abs(rhs.l2_norm() - np.linalg.norm(pr.rhs())) < 1e-13 * np.linalg.norm(pr.rhs())
    ctl = pkg.IterationNumberControl(9, 1e-6 * rhs.l2_norm())
    solution.assign(0.0)
    pkg.SolverCGFullMerge(ctl).solve(op, solution, rhs, pkg.DiagonalMatrix())
    xr, k, _ = O.cg_plain(pr.vmult, pr.rhs(), 9, tol=1e-6 * np.linalg.norm(pr.rhs()))
    assert ctl.last_step() == k and rel(solution.values.cpu().numpy(), xr) < TOL_CG
    tmp = pkg.Vector().reinit(solution)
    op.vmult(tmp, solution)
    tmp.add(-1.0, rhs)                                # true residual == recurrence residual
    assert abs(tmp.l2_norm() - ctl.last_value()) < 1e-9 * rhs.l2_norm()
    tmp.equ(2.0, rhs)
    tmp.sadd(0.5, -1.0, rhs)
    assert tmp.l2_norm() < 1e-14 * rhs.l2_norm()
    tmp.reinit(solution)                              # same layout: zeroes
    assert tmp.all_zero()
    solution.update_ghost_values(); solution.compress_add(); solution.zero_out_ghosts()   # one rank: no-ops


# ------------------------------------------------------------------ CG (a12-a14)
def test_config1_cg_plain_and_merged():
    """BASELINE config 1: p=2, 8^3 cells (4913 DoFs), 10 CG iterations."""
    z = np.load(os.path.join(G, "config1_cg.npz"))
    mesh = pkg.BrickMesh(2, (8, 8, 8))
    op = pkg.PoissonOperator(mesh, 0)
    b = op.assemble_rhs()
    assert rel(b.cpu().numpy(), z["b"]) < 1e-13
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        x.fill_(7.0)                                      # solve() starts from x0 = 0 regardless
        ctl = pkg.IterationNumberControl(10, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == 10
        assert rel(x.cpu().numpy(), z["x"]) < TOL_CG
        assert abs(ctl.last_value() - z["residuals"][-1]) < 1e-9 * z["residuals"][-1]
        assert abs(ctl.initial_value() - np.linalg.norm(z["b"])) < 1e-12 * np.linalg.norm(z["b"])


def test_config1_solution_against_textbook_matrices():
    """BASELINE config 1 on the HIP path against a solve that shares no code with the oracle (global operator from the rational quadratic
    element matrices of every FEM text, CG written out in numpy: tests/test_oracle_known_answers.py::config1_from_textbook_matrices):
    solution vector within the north-star tolerance 1e-11, plain and merged solver, matched DoF by DoF through the node coordinates."""
    from test_oracle_known_answers import config1_from_textbook_matrices
    x_ref, res = config1_from_textbook_matrices(10)
    mesh = pkg.BrickMesh(2, (8, 8, 8))
    ijk = np.rint(np.asarray(mesh.coords).reshape(-1, 3)[:mesh.n_owned] * 2.0).astype(int)
    ref = x_ref[ijk[:, 2], ijk[:, 1], ijk[:, 0]]
    op = pkg.PoissonOperator(mesh, pkg.QUAD_GAUSS)
    b = op.assemble_rhs()
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(10, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == 10 and rel(x.cpu().numpy(), ref) < TOL_CG
        assert abs(ctl.initial_value() - res[0]) < 1e-12 * res[0] and abs(ctl.last_value() - res[-1]) < 1e-8 * res[-1]


@pytest.mark.parametrize("iters", [1, 2, 3, 4, 7, 8])
def test_merged_cg_epilogue_parity(iters):
    """The merged solver's deferred x update must reproduce plain CG at every stopping parity
    (the reference's schedule does not: SURVEY 0.4)."""
    pr = O.Problem(3, (3, 3, 3), 0, deform_amp=0.04)
    op = pkg.PoissonOperator(pkg.BrickMesh(3, (3, 3, 3), deform_amp=0.04), 0)
    b = op.assemble_rhs()
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), iters)
    x = op.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(iters, 0.0)
    pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
    assert ctl.last_step() == iters
    assert rel(x.cpu().numpy(), xr) < TOL_CG


@pytest.mark.parametrize("cells,block,wgs", [((9, 6, 7), (4, 4, 4), 8), ((8, 8, 5), (4, 4, 2), 16), ((4, 4, 4), (4, 4, 4), 0)])
def test_merged_cg_with_dot_products_fused_into_the_block_kernel(cells, block, wgs):
    """SolverCGFullMerge on the packed block kernel: the operator's write-out and combine pass also form the v-dependent dot
    products of update_b (bp5/solver.h:142-311) and write the Dirichlet rows (copy_constrained_values, bp5/step-64.cu:275).
    Same iteration as with the separate kernels (solution to rounding: only the summation order differs), equal to the
    oracle's plain CG within the north-star tolerance, bitwise reproducible, and the tolerance stop fires at the same step."""
    torch = _t()
    p = 4
    # (h = 1/8: on a domain of ~unit size step-64's coefficient varies by ~50; with h = 1 it varies by 6000 and CG amplifies
    # any change of summation order to 1e-7 within 14 iterations -- fused or not)
    mesh = pkg.BrickMesh(p, cells, h=0.125, deform_amp=0.03, cell_block=block, dof_numbering=1, cell_block_order=1)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    op.mf_data.set_apply_variant(56)
    if wgs:
        op.mf_data.set_block_workgroups(wgs)          # several bricks per persistent workgroup, some workgroups idle
    assert op.mf_data.block_plan_info()[2]             # packed indices: the fused path is taken
    pr = O.Problem(p, cells, 0, h=0.125, deform_amp=0.03, kappa=O.kappa_step64)
    perm = mesh.global_ids.astype(np.int64)
    b = op.assemble_rhs()
    iters = 14
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), iters)
    sols, ress = [], []
    for fused in (True, False, True):
        op.mf_data.set_cg_fusion(fused)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(iters, 0.0)
        pkg.SolverCGFullMerge(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == iters and ctl.apply_launches == iters
        assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG
        sols.append(x)
        ress.append(ctl.last_value())
    assert torch.equal(sols[0], sols[2])                                      # fixed summation order
    assert float((sols[0] - sols[1]).abs().max()) < 1e-12 * float(sols[1].abs().max())
    assert abs(ress[0] - ress[1]) < 1e-10 * ress[1]
    # right-hand side with NON-zero entries on the Dirichlet rows (inhomogeneous boundary values through b): then p is non-zero there and
    # the quadrature-point form of p.v needs its Dirichlet correction p (p - sum) -- same iterate as the separate kernels and as the
    # oracle's MERGED recurrence (A_eff is not symmetric for such vectors, so plain and merged CG are different iterations here)
    cdofs = torch.from_numpy(mesh.constrained.astype(np.int64)).cuda()
    bi = b.clone()
    bi[cdofs] = 0.3 + 0.1 * torch.cos(torch.arange(cdofs.numel(), dtype=torch.float64, device="cuda:0"))
    bi_lex = np.zeros(pr.mesh.n_dofs)
    bi_lex[perm] = bi.cpu().numpy()
    xi_ref, _, _ = O.cg_merged(pr.vmult, bi_lex, 8)
    for fused in (True, False):
        op.mf_data.set_cg_fusion(fused)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(8, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, bi, pkg.DiagonalMatrix())
        assert ctl.dot_products_fused == fused and rel(x.cpu().numpy(), xi_ref[perm]) < TOL_CG
    # tolerance stop (constant coefficient, undeformed: converges in tens of iterations): the device-side convergence flag
    # turns the fused kernels into no-ops at the same iteration, and the recurrence residual is the true one
    op2 = pkg.PoissonOperator(pkg.BrickMesh(p, cells, h=0.125, cell_block=block, dof_numbering=1, cell_block_order=1), 0)
    op2.mf_data.set_apply_variant(56)
    if wgs:
        op2.mf_data.set_block_workgroups(wgs)
    b2 = op2.assemble_rhs()
    bn = float(torch.linalg.norm(b2))
    steps = []
    for fused in (True, False):
        op2.mf_data.set_cg_fusion(fused)
        x = op2.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(400, 1e-8 * bn)
        pkg.SolverCGFullMerge(ctl, check_every=7).solve(op2, x, b2, pkg.DiagonalMatrix())
        steps.append(ctl.last_step())
        Ax = op2.initialize_dof_vector()
        op2.vmult(Ax, x)
        assert float(torch.linalg.norm(Ax - b2)) < 1.0001e-8 * bn and ctl.last_value() <= 1e-8 * bn
    assert steps[0] == steps[1] and 5 < steps[0] < 400
    # a Jacobi-preconditioned solve keeps the separate dot-product kernel (the fused sums assume D == 1)
    op.mf_data.set_cg_fusion(True)
    dinv = op.compute_diagonal(invert=True)
    x = op.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(iters, 0.0)
    pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix(dinv))
    xj, _, _ = O.cg_plain(pr.vmult, pr.rhs(), iters, diag=1.0 / O.operator_diagonal(pr.mesh, pr.coef, pr.N, pr.D))
    assert rel(x.cpu().numpy(), xj[perm]) < TOL_CG and not ctl.dot_products_fused
    # the standard SolverCG takes d.h (its only dot product with the operator's result) from the kernel as the quadrature-point energy:
    # with and without a preconditioner (D does not enter d.h), inhomogeneous Dirichlet rows included (h = d there)
    bi_plain, _, _ = O.cg_plain(pr.vmult, bi_lex, 8)
    for rhs, precond, ref, its in ((b, None, xr, iters), (b, dinv, xj, iters), (bi, None, bi_plain, 8)):
        outs = []
        for fused in (True, False, True):
            op.mf_data.set_cg_fusion(fused)
            x = op.initialize_dof_vector()
            ctl = pkg.IterationNumberControl(its, 0.0)
            pkg.SolverCG(ctl).solve(op, x, rhs, pkg.DiagonalMatrix(precond) if precond is not None else pkg.DiagonalMatrix())
            assert ctl.dot_products_fused == fused and ctl.last_step() == its and rel(x.cpu().numpy(), ref[perm]) < TOL_CG
            outs.append(x)
        assert torch.equal(outs[0], outs[2])
        assert float((outs[0] - outs[1]).abs().max()) < 1e-12 * float(outs[1].abs().max())


@pytest.mark.parametrize("solver_name", ["SolverCG", "SolverCGFullMerge"])
def test_solvers_take_any_operator_with_vmult(solver_name):
    """cg.solve(A, x, b, preconditioner) needs nothing of A but A.vmult (bp5/solver.h:25-30,377,475): a foreign operator
    (here: the library's Poisson operator plus a torch-side mass-like shift, enqueued on the same stream) is solved by the
    same solver kernels through bp5_cg_solve_operator; iterate for iterate equal to the oracle's CG on the same operator."""
    p, cells, sigma = 3, (4, 3, 3), 7.5
    mesh = pkg.BrickMesh(p, cells, h=0.25, deform_amp=0.03)
    inner = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    pr = O.Problem(p, cells, 0, h=0.25, deform_amp=0.03, kappa=O.kappa_step64)

    class Shifted:
        mf_data = inner.mf_data
        calls = 0

        def vmult(self, dst, src):
            inner.vmult(dst, src)
            dst.add_(src, alpha=sigma)
            Shifted.calls += 1

    b = inner.assemble_rhs()
    its = 9
    xr, _, _ = O.cg_plain(lambda s: pr.vmult(s) + sigma * s, pr.rhs(), its)
    x = inner.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(its, 0.0)
    getattr(pkg, solver_name)(ctl, profile=True).solve(Shifted(), x, b, pkg.DiagonalMatrix())
    assert ctl.last_step() == its and Shifted.calls == its and ctl.apply_launches == its
    assert rel(x.cpu().numpy(), xr) < TOL_CG

    class Broken:
        mf_data = inner.mf_data

        def vmult(self, dst, src):
            raise ValueError("operator failure")

    with pytest.raises(ValueError):            # a failing callback aborts the solve; the error reaches the caller
        getattr(pkg, solver_name)(pkg.IterationNumberControl(3, 0.0)).solve(Broken(), x, b, pkg.DiagonalMatrix())


def test_p4_variable_coefficient_deformed_cg_golden():
    z = np.load(os.path.join(G, "p4_kappa_deformed_cg.npz"))
    op = pkg.PoissonOperator(pkg.BrickMesh(4, (4, 3, 3), h=0.25, deform_amp=0.04), 0, pkg.COEF_STEP64)
    b = op.assemble_rhs()
    assert rel(b.cpu().numpy(), z["b"]) < 1e-13
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(12, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == 12 and rel(x.cpu().numpy(), z["x"]) < TOL_CG


@pytest.mark.parametrize("check_every", [0, 1, 5])
def test_cg_tolerance_stop_and_frozen_iterate(check_every):
    """IterationNumberControl(max, tol): stops at the first iteration with res <= tol; the
    iterate is frozen on device from then on whatever the host polling period."""
    pr = O.Problem(2, (4, 4, 4), 0)
    op = pkg.PoissonOperator(pkg.BrickMesh(2, (4, 4, 4)), 0)
    b = op.assemble_rhs()
    bn = np.linalg.norm(pr.rhs())
    xr, kr, resr = O.cg_plain(pr.vmult, pr.rhs(), 500, tol=1e-6 * bn)
    assert 3 < kr < 100
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(kr + 13, 1e-6 * bn)
        solver(ctl, check_every=check_every).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == kr
        assert rel(x.cpu().numpy(), xr) < 1e-10
        assert ctl.last_value() <= 1e-6 * bn


def test_cg_with_diagonal_preconditioner_vector():
    """A non-trivial DiagonalMatrix goes through the same kernels (bp5/solver.h:68,100,131,170)."""
    pr = O.Problem(2, (3, 3, 3), 0, deform_amp=0.05)
    op = pkg.PoissonOperator(pkg.BrickMesh(2, (3, 3, 3), deform_amp=0.05), 0)
    rng = np.random.default_rng(4)
    diag = rng.uniform(0.5, 2.0, pr.mesh.n_dofs)
    b = op.assemble_rhs()
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 12, diag=diag)
    xm, _, _ = O.cg_merged(pr.vmult, pr.rhs(), 12, diag=diag)
    assert rel(xm, xr) < 1e-12
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(12, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix(dev(diag)))
        assert rel(x.cpu().numpy(), xr) < TOL_CG


def test_medium_size_against_c_oracle():
    """p=4, 12^3 cells (117 649 DoFs) on the unit cube, variable coefficient, deformed: 20 CG
    iterations vs the C restatement (rounding-perturbation drift of this case: 2e-15), plus
    profile counters."""
    p, cells = 4, (12, 12, 12)
    mesh = pkg.BrickMesh(p, cells, h=1.0 / 12, deform_amp=0.04)
    cp = CO.CProblem(p, 0, mesh.l2g, mesh.coords, mesh.constrained, 1)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    b = op.assemble_rhs()
    bref = cp.rhs()
    assert rel(b.cpu().numpy(), bref) < 1e-13
    xr, kr, resr = cp.cg_plain(bref, 20)
    x = op.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(20, 0.0)
    pkg.SolverCG(ctl, profile=True).solve(op, x, b, pkg.DiagonalMatrix())
    assert ctl.last_step() == 20 and ctl.apply_launches == 20 and ctl.apply_ms_avg > 0
    assert rel(x.cpu().numpy(), xr) < TOL_CG
    assert abs(ctl.last_value() - resr) < 1e-9 * resr


# ------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("p,cells,quad,amp,km,kw,variant", [
    (4, (54, 54, 54), 0, 0.0, 1, {}, 0),                                                              # BASELINE config 2
    (6, (30, 30, 30), 0, 0.05, 0, {}, 0),                                                             # config 5 shape, reduced
    (4, (86, 84, 82), 0, 0.03, 1, dict(cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1), 56),   # bench mesh order -> block kernel
    (4, (116, 116, 120), 0, 0.0, 1, dict(cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1), 56),  # the bench workload itself (config 3 size on one GPU; rounds 1-3: 116^3)
    (1, (150, 140, 130), 0, 0.03, 1, {}, 10), (3, (61, 60, 59), 1, 0.03, 1, {}, 10),                  # team-kernel defaults at scale
    (2, (81, 80, 79), 0, 0.03, 0, {}, 0), (5, (33, 32, 31), 1, 0.03, 1, {}, 0), (7, (23, 22, 21), 0, 0.03, 1, {}, 0),
    (8, (20, 19, 18), 1, 0.03, 1, {}, 0),
    (4, (60, 59, 58), 0, 0.0, 1, dict(geometry="affine"), 10), (6, (30, 29, 28), 0, 0.0, 1, dict(geometry="affine"), 0),
    (4, (86, 84, 82), 1, 0.0, 1, dict(geometry="affine", cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1), 56),
    # the deterministic block kernel on the other degrees at the BASELINE sizes: config 5 (p = 6, 61^3, deformed, 49 430 863 DoFs) and
    # config 4 (~5e7 DoFs) for p = 2, 3, 5, 7 -- bricks sized for the LDS accumulator, partial bricks at the mesh edges
    (6, (61, 61, 61), 0, 0.05, 1, dict(cell_block=(4, 4, 2), dof_numbering=1, cell_block_order=1), 56),
    (5, (73, 73, 73), 0, 0.03, 1, dict(cell_block=(6, 4, 2), dof_numbering=1, cell_block_order=1), 56),
    (7, (52, 52, 52), 1, 0.03, 1, dict(cell_block=(4, 2, 2), dof_numbering=1, cell_block_order=1), 56),
    (3, (122, 122, 122), 0, 0.03, 1, dict(cell_block=(8, 4, 4), dof_numbering=1, cell_block_order=1), 56),
    (2, (184, 184, 184), 0, 0.0, 1, dict(cell_block=(8, 8, 4), dof_numbering=1, cell_block_order=1), 56),
    # ... and its end points at full size: p = 1 (367^3 cells, 49 836 032 DoFs) and p = 8 (46^3 cells, 50 243 409 DoFs; 81 lanes per cell:
    # three cells per pass, cells that span waves)
    (1, (367, 367, 367), 0, 0.0, 1, dict(cell_block=(8, 8, 8), dof_numbering=1, cell_block_order=1), 56),
    (8, (46, 46, 46), 0, 0.03, 1, dict(cell_block=(2, 2, 2), dof_numbering=1, cell_block_order=1), 56),
    # maximum sizes: five times the headline problem on one GPU (513 922 401 DoFs, half the 2^30 range of the local 32-bit indices; 64-bit
    # offsets into the 49 GB metric array); profiles/r2 holds a bench line at 1 003 003 001 DoFs
    (4, (200, 200, 200), 0, 0.0, 1, dict(cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1), 56)])
def test_full_size_properties(p, cells, quad, amp, km, kw, variant):
    """Size-independent properties at BASELINE scale: constants in the null space of the cell
    loop, symmetry, linearity; CG residual consistency.  The third case is the bench's mesh ordering
    (bricks, partial bricks at the edges, deformed): the library default must resolve to the block kernel."""
    torch = _t()
    import ctypes as C
    kw = dict(kw)
    affine = kw.pop("geometry", None) == "affine"
    mesh = pkg.BrickMesh(p, cells, h=1.0 / cells[0], deform_amp=amp, **kw)
    op = pkg.PoissonOperator(mesh, quad, km, **({"geometry": pkg.GEOM_AFFINE} if affine else {}))
    mf = op.mf_data
    assert mf.get_apply_variant() == variant
    n = mesh.n_owned
    L, h = pkg.lib(), mf.handle
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def dot(a, b):
        r = C.c_double()
        L.bp5_vec_dot(h, ptr(a), ptr(b), n, C.byref(r))
        return r.value

    one = torch.ones(n, dtype=torch.float64, device="cuda:0")
    y = mf.initialize_dof_vector()
    mf.cell_loop(op.coef, one, y)
    scale = float(torch.abs(op.coef[: mesh.n_cells * (p + 1) ** 3]).max()) if op.coef is not None else float(1.0 / cells[0])
    assert float(torch.abs(y).max()) < 1e-11 * scale * (p + 1) ** 3
    g = torch.Generator(device="cuda:0").manual_seed(1)
    u = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    v = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    mf.set_constrained_values(0.0, u)                  # A_eff = P A P + (I - P) is symmetric on
    mf.set_constrained_values(0.0, v)                  # vectors that vanish on Dirichlet DoFs
    Au, Av, Auv = (mf.initialize_dof_vector() for _ in range(3))
    op.vmult(Au, u)
    op.vmult(Av, v)
    vAu, uAv = dot(v, Au), dot(u, Av)
    assert abs(vAu - uAv) < 1e-11 * max(abs(vAu), dot(u, Au))
    assert dot(u, Au) > 0
    if variant != 0:
        # the non-atomic default kernels at scale (block kernel: partial bricks, multi-round passes, lists of
        # unequal length in one workgroup's range; team kernel: LDS-staged scatter): entry-wise agreement with the
        # atomic pencil kernel over repeated launches, bitwise reproducible
        # (regressions: accumulator cleared only up to the first block's length; LDS write in flight at a
        # loop-header barrier, tests/test_isa_checks.py)
        mf.set_apply_variant(3 if p == 4 else 110 if p == 2 else 1)   # an atomic kernel as reference
        ref = mf.initialize_dof_vector()
        op.vmult(ref, u)
        mf.set_apply_variant(0)
        tol = 1e-12 * float(ref.abs().max())
        for rep in range(6):
            d = mf.initialize_dof_vector()
            d.fill_(float("nan"))
            op.vmult(d, u)
            assert bool(((d - ref).abs() < tol).all())
            assert torch.equal(d, Au)
    w = 2.0 * u - 3.0 * v
    op.vmult(Auv, w)
    lin = 2.0 * Au - 3.0 * Av
    assert float(torch.linalg.norm(Auv - lin)) < 1e-12 * float(torch.linalg.norm(lin))
    # CG: recomputed residual ||A x - b|| equals the recurrence residual
    b = op.assemble_rhs()
    x = mf.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(15, 0.0)
    pkg.SolverCG(ctl).solve(op, x, b, pkg.DiagonalMatrix())
    op.vmult(Au, x)
    true_res = float(torch.linalg.norm(Au - b))
    assert abs(true_res - ctl.last_value()) < 1e-9 * ctl.initial_value()
    x2 = mf.initialize_dof_vector()
    ctl2 = pkg.IterationNumberControl(15, 0.0)
    pkg.SolverCGFullMerge(ctl2).solve(op, x2, b, pkg.DiagonalMatrix())
    assert float(torch.linalg.norm(x2 - x)) < TOL_CG * float(torch.linalg.norm(x))


def test_single_rank_communicator_and_distributed_entry_points():
    """RCCL bootstrap with one rank: allreduce / halo calls are no-ops but exercise the symbols."""
    comm = pkg.Communicator(0, 1)
    pr = O.Problem(2, (3, 3, 3), 0)
    op = pkg.PoissonOperator(pkg.BrickMesh(2, (3, 3, 3)), 0, comm=comm)
    b = op.assemble_rhs()
    x = op.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(8, 0.0)
    pkg.SolverCG(ctl).solve(op, x, b, pkg.DiagonalMatrix())
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 8)
    assert rel(x.cpu().numpy(), xr) < TOL_CG
    import ctypes as C
    L = pkg.lib()
    s = dev(O.deterministic_src(pr.mesh.n_dofs, seed=2))
    d = op.initialize_dof_vector()
    assert L.bp5_apply_distributed(op.mf_data.handle, C.c_void_p(op.coef.data_ptr()), C.c_void_p(s.data_ptr()),
                                   C.c_void_p(d.data_ptr()), 1) == 0
    assert rel(d.cpu().numpy(), pr.vmult(s.cpu().numpy())) < TOL_OP
    op.mf_data.synchronize()
    comm.close()


@pytest.mark.parametrize("p,cells,block,kw", [(8, (3, 3, 2), (0, 0, 0), {}), (8, (4, 4, 4), (2, 2, 2), {}), (5, (4, 3, 5), (2, 2, 2), dict(rank=1, n_ranks=2)),
                                            (6, (3, 4, 3), (0, 0, 0), {}), (7, (3, 2, 3), (2, 2, 2), {}), (3, (4, 3, 3), (0, 0, 0), {})])
def test_cell_interior_dofs_numbered_first_are_stored_plainly(p, cells, block, kw):
    """bp5_mesh_desc.dof_numbering = 2 (round 4): the DoFs strictly inside a cell come first, cell after cell.  bp5_mf_create recognises the property and the default
    pencil kernel of p >= 5 stores the (p-1)^3 entries a cell owns alone with plain stores instead of memory-side atomics (kernel id 32: half the atomics at p = 8,
    none of them in a cache line a store touches).  The cell loop of every rank's slab summed through the global ids against the oracle's, the merged solve against
    the oracle's; the knob switches the build off; the same mesh numbered otherwise never takes it."""
    torch = _t()
    blocked = all(b > 0 for b in block)
    n_ranks = kw.get("n_ranks", 1)
    pr = O.Problem(p, cells, 0, deform_amp=0.03, kappa=O.kappa_step64, h=0.25)
    s = O.deterministic_src(pr.mesh.n_dofs, seed=3)
    ref = O.apply_cells(pr.mesh, pr.coef, pr.N, pr.D, s)            # cell loop: no Dirichlet copy
    per = (p - 1) ** 3
    total = np.zeros(pr.mesh.n_dofs)
    for r in range(n_ranks):
        mesh = pkg.BrickMesh(p, cells, h=0.25, deform_amp=0.03, cell_block=block, dof_numbering=2, cell_block_order=1 if blocked else 0, rank=r, n_ranks=n_ranks)
        inner = np.asarray(mesh.l2g).reshape(mesh.n_cells, p + 1, p + 1, p + 1)[:, 1:p, 1:p, 1:p].reshape(mesh.n_cells, per)
        assert np.array_equal(inner, np.arange(mesh.n_cells * per, dtype=np.int64).reshape(mesh.n_cells, per))     # cell after cell, x fastest
        gid = np.asarray(mesh.global_ids).astype(np.int64)
        op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
        d = op.initialize_dof_vector()
        op.mf_data.cell_loop(op.coef, dev(s[gid]), d)
        np.add.at(total, gid, d.cpu().numpy())                       # owned and ghost entries land on their global ids
    assert rel(total, ref) < TOL_OP
    if n_ranks == 1:
        no = mesh.n_owned
        ctl = pkg.IterationNumberControl(3, 0.0)
        x = op.initialize_dof_vector()
        pkg.SolverCGFullMerge(ctl).solve(op, x, op.assemble_rhs(), pkg.DiagonalMatrix())
        xr, _, _ = O.cg_merged(pr.vmult, pr.rhs(), 3)
        assert rel(x.cpu().numpy(), xr[gid[:no]]) < 1e-10
        assert (ctl.apply_kernel.startswith("apply_pencil_kernel") and ctl.apply_kernel.endswith(",32>")) == (p >= 5), ctl.apply_kernel   # (p = 3: the team kernel)
        op.mf_data.set_tuning("interior_stores", 0)
        c0 = pkg.IterationNumberControl(3, 0.0)
        x0 = op.initialize_dof_vector()
        pkg.SolverCGFullMerge(c0).solve(op, x0, op.assemble_rhs(), pkg.DiagonalMatrix())
        assert not c0.apply_kernel.endswith(",32>") and rel(x0.cpu().numpy(), xr[gid[:no]]) < 1e-10
        # the same mesh numbered block-major / lexicographically is not recognised
        mesh1 = pkg.BrickMesh(p, cells, h=0.25, deform_amp=0.03, cell_block=block, dof_numbering=1 if blocked else 0, cell_block_order=1 if blocked else 0)
        op1 = pkg.PoissonOperator(mesh1, 0, pkg.COEF_STEP64)
        op1.mf_data.set_apply_variant(0)
        c1 = pkg.IterationNumberControl(2, 0.0)
        pkg.SolverCGFullMerge(c1).solve(op1, op1.initialize_dof_vector(), op1.assemble_rhs(), pkg.DiagonalMatrix())
        assert not c1.apply_kernel.endswith(",32>"), c1.apply_kernel


@pytest.mark.parametrize("p,cells,seed", [(2, (4, 3, 3), 1), (4, (4, 3, 2), 2), (5, (3, 2, 2), 3)])
def test_externally_numbered_mesh(p, cells, seed):
    """nothing in the library assumes the structured generator.  The flat arrays of bp5_mf_desc come
    from an 'external' mesh here: cells in random order, DoFs in a random numbering, cell groups of random sizes
    (1..12 cells, not bricks: their cells conflict, so the block kernel needs accumulation rounds) -- every kernel
    family against the oracle through the permutations."""
    from types import SimpleNamespace
    rng = np.random.default_rng(seed)
    pr = O.Problem(p, cells, 0, deform_amp=0.04, kappa=O.kappa_step64)
    om = pr.mesh
    n, nc = om.n_dofs, om.n_cells
    cperm = rng.permutation(nc)                       # new cell k = old cell cperm[k]
    new_of_old = rng.permutation(n).astype(np.int64)  # DoF numbering: old id -> new id
    old_of_new = np.argsort(new_of_old)
    l2g = new_of_old[om.l2g.astype(np.int64)[cperm]].astype(np.uint32)
    off = [0]
    while off[-1] < nc:
        off.append(min(nc, off[-1] + int(rng.integers(1, 13))))
    mesh = SimpleNamespace(degree=p, n=p + 1, cells=cells, n_cells=nc, n_interior_cells=nc, n_owned=n, n_ghost=0, n_local=n, n_global_dofs=n,
                           l2g=l2g, coords=np.ascontiguousarray(om.coords[old_of_new]), global_ids=old_of_new.astype(np.uint64),
                           constrained=np.sort(new_of_old[om.constrained.astype(np.int64)]).astype(np.uint32), n_neighbors=0,
                           neighbor_rank=np.zeros(0, np.int32), send_offsets=np.zeros(1, np.uint32), send_indices=np.zeros(0, np.uint32),
                           recv_offsets=np.zeros(1, np.uint32), cell_block_offsets=np.asarray(off, np.uint32), rank=0, n_ranks=1, h=1.0, deform_amp=0.04)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64)
    s = O.deterministic_src(n, seed=seed + 10)        # in the oracle's numbering
    ref = pr.vmult(s)[old_of_new]
    src = dev(s[old_of_new])
    variants = [0, 10, 50] + ([3, 56, 49, 59, 52] if p == 4 else [])
    for v in variants:
        op.mf_data.set_apply_variant(v)
        op.mf_data.set_block_workgroups(8)
        d = op.initialize_dof_vector()
        d.fill_(float("nan"))
        op.vmult(d, src)
        assert rel(d.cpu().numpy(), ref) < TOL_OP, v
    op.mf_data.set_apply_variant(0)
    assert rel(op.assemble_rhs().cpu().numpy(), pr.rhs()[old_of_new]) < 1e-13
    assert rel(op.compute_diagonal().cpu().numpy(), O.operator_diagonal(om, pr.coef, pr.N, pr.D)[old_of_new]) < 1e-13
    xr, _, _ = O.cg_plain(pr.vmult, pr.rhs(), 8)
    for v in ([0, 50] + ([56] if p == 4 else [])):
        op.mf_data.set_apply_variant(v)
        x = op.initialize_dof_vector()
        pkg.SolverCGFullMerge(pkg.IterationNumberControl(8, 0.0)).solve(op, x, op.assemble_rhs(), pkg.DiagonalMatrix())
        assert rel(x.cpu().numpy(), xr[old_of_new]) < TOL_CG, v


def test_halo_exchange_through_rccl_with_a_self_neighbour():
    """The halo path with REAL RCCL traffic on one GPU: a one-rank communicator whose only neighbour is the rank
    itself (RCCL supports send/recv to self inside a group).  The slab mesh of rank 1 of 2 supplies cells that read
    ghost DoFs; the 'owner' of the ghost plane is faked to be this rank's own top DoF plane.  Checked against the
    same steps done with torch indexing: gather (pack kernel, ncclSend/ncclRecv into the ghost range), all cells,
    scatter-add (ncclSend/ncclRecv, unpack-add kernel in a fixed order), ghost zeroing, Dirichlet copy."""
    from types import SimpleNamespace
    torch = _t()
    import ctypes as C
    p, cells = 3, (4, 3, 5)
    m1 = pkg.BrickMesh(p, cells, deform_amp=0.03, rank=1, n_ranks=2)
    ng, no = m1.n_ghost, m1.n_owned
    assert ng > 0
    send_idx = np.arange(no - ng, no, dtype=np.uint32)          # this rank's last owned DoFs play the neighbour's plane
    mesh = SimpleNamespace(degree=p, n=p + 1, cells=cells, n_cells=m1.n_cells, n_interior_cells=m1.n_interior_cells, n_owned=no, n_ghost=ng,
                           n_local=no + ng, n_global_dofs=no, l2g=m1.l2g, coords=m1.coords, global_ids=m1.global_ids, constrained=m1.constrained,
                           n_neighbors=1, neighbor_rank=np.zeros(1, np.int32), send_offsets=np.asarray([0, ng], np.uint32), send_indices=send_idx,
                           recv_offsets=np.asarray([0, ng], np.uint32), cell_block_offsets=None, rank=0, n_ranks=1, h=1.0, deform_amp=0.03)
    comm = pkg.Communicator(0, 1)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64, comm=comm)
    L, h = pkg.lib(), op.mf_data.handle
    ptr = lambda t: C.c_void_p(t.data_ptr())
    g = torch.Generator(device="cuda:0").manual_seed(5)
    v = torch.rand(no + ng, dtype=torch.float64, device="cuda:0", generator=g)
    ref = v.clone()
    ref[no:] = ref[torch.from_numpy(send_idx.astype(np.int64)).cuda()]
    assert L.bp5_halo_gather(h, ptr(v)) == 0
    assert torch.equal(v, ref)
    w = torch.rand(no + ng, dtype=torch.float64, device="cuda:0", generator=g)
    refw = w.clone()
    refw[torch.from_numpy(send_idx.astype(np.int64)).cuda()] += refw[no:]
    refw[no:] = 0.0
    assert L.bp5_halo_scatter_add(h, ptr(w)) == 0
    assert torch.equal(w, refw)
    # the start / finish split (update_ghost_values_start/finish, compress_start/finish): compute enqueued between the
    # two halves overlaps the transfer on the handle's communication stream and must not disturb it
    v2, w2 = ref.clone(), w.clone()
    v2[no:] = -1.0
    w2[no:] = refw[torch.from_numpy(send_idx.astype(np.int64)).cuda()]
    refw2 = w2.clone()
    refw2[torch.from_numpy(send_idx.astype(np.int64)).cuda()] += refw2[no:]
    refw2[no:] = 0.0
    busy = torch.zeros(1 << 22, dtype=torch.float64, device="cuda:0")
    assert L.bp5_mf_set_overlap(h, 1) == 0                    # (the default, 2, overlaps on large slabs only)
    assert L.bp5_halo_gather_start(h, ptr(v2)) == 0
    for _ in range(4):
        assert L.bp5_vec_fill(h, ptr(busy), 1.0, busy.numel()) == 0
    assert L.bp5_halo_gather_finish(h, ptr(v2)) == 0
    assert L.bp5_halo_scatter_add_start(h, ptr(w2)) == 0
    for _ in range(4):
        assert L.bp5_vec_fill(h, ptr(busy), 2.0, busy.numel()) == 0
    assert L.bp5_halo_scatter_add_finish(h, ptr(w2)) == 0
    assert torch.equal(v2, ref) and torch.equal(w2, refw2)
    assert L.bp5_mf_set_overlap(h, 0) == 0                    # same calls with the exchange on the compute stream
    v3 = ref.clone()
    v3[no:] = -1.0
    assert L.bp5_halo_gather_start(h, ptr(v3)) == 0 and L.bp5_halo_gather_finish(h, ptr(v3)) == 0
    assert torch.equal(v3, ref)
    assert L.bp5_mf_set_overlap(h, 1) == 0
    # the whole distributed application against the same steps by hand
    src = torch.rand(no + ng, dtype=torch.float64, device="cuda:0", generator=g)
    src[no:] = 0.0
    s2 = src.clone()
    s2[no:] = s2[torch.from_numpy(send_idx.astype(np.int64)).cuda()]
    want = op.initialize_dof_vector()
    op.mf_data.cell_loop(op.coef, s2, want)
    want[torch.from_numpy(send_idx.astype(np.int64)).cuda()] += want[no:]
    want[no:] = 0.0
    c = torch.from_numpy(m1.constrained.astype(np.int64)).cuda()
    want[c] = src[c]
    got = op.initialize_dof_vector()
    got.fill_(float("nan"))
    src_in = src.clone()
    assert L.bp5_apply_distributed(h, ptr(op.coef), ptr(src_in), ptr(got), 1) == 0
    assert float((got - want).abs().max()) < 1e-12 * float(want.abs().max())
    assert float(src_in[no:].abs().max()) == 0.0        # ghosts of src zeroed again
    # CG with the exchange inside every operator application (solver path of an N > 1 run): the glued mesh is still
    # SPD on the free DoFs, so the recurrence residual must equal the true residual computed through the same operator
    b = op.assemble_rhs()                                # ghost contributions travel to their (faked) owners
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(12, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        Ax = op.initialize_dof_vector()
        xin = x.clone()
        assert L.bp5_apply_distributed(h, ptr(op.coef), ptr(xin), ptr(Ax), 1) == 0
        true_res = float(torch.linalg.norm((Ax - b)[:no]))
        assert ctl.last_step() == 12 and abs(true_res - ctl.last_value()) < 1e-9 * ctl.initial_value()
        assert ctl.last_value() < 0.5 * ctl.initial_value()
    op.mf_data.synchronize()
    op.mf_data.close()
    comm.close()


def _consistent_self_glue(m1):
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


@pytest.mark.parametrize("variant,neighbours", [(56, 1), (3, 1), (56, 2)])
def test_block_kernel_behind_the_halo_exchange(variant, neighbours):
    """The bench's rank-local configuration for ranks > 0 (brick-ordered slab mesh with a ghost plane, block kernel with
    packed indices, overwrite mode) inside bp5_apply_distributed with real RCCL traffic (self neighbour): equals the
    atomic pencil kernel through the same exchange.  neighbours = 2: the neighbour tables of a MIDDLE rank as the mesh
    generator writes them (rank 1 of 3: ghosts received from the rank below, own top plane sent to the rank above: one
    receive-only and one send-only neighbour, zero-length messages skipped), both neighbours mapped to the rank itself."""
    from types import SimpleNamespace
    torch = _t()
    import ctypes as C
    p, cells = 4, (9, 8, 10)
    m1 = pkg.BrickMesh(p, cells, deform_amp=0.03, rank=1, n_ranks=neighbours + 1, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
    ng, no = m1.n_ghost, m1.n_owned
    if neighbours == 1:
        tables = dict(n_neighbors=1, neighbor_rank=np.zeros(1, np.int32), send_offsets=np.asarray([0, ng], np.uint32),
                      send_indices=_consistent_self_glue(m1), recv_offsets=np.asarray([0, ng], np.uint32))
    else:
        assert m1.n_neighbors == 2 and list(m1.send_offsets) == [0, 0, ng] and list(m1.recv_offsets) == [0, ng, ng]
        tables = dict(n_neighbors=2, neighbor_rank=np.zeros(2, np.int32), send_offsets=m1.send_offsets, send_indices=m1.send_indices,
                      recv_offsets=m1.recv_offsets)
    mesh = SimpleNamespace(degree=p, n=p + 1, cells=cells, n_cells=m1.n_cells, n_interior_cells=m1.n_interior_cells, n_owned=no, n_ghost=ng,
                           n_local=no + ng, n_global_dofs=no, l2g=m1.l2g, coords=m1.coords, global_ids=m1.global_ids, constrained=m1.constrained,
                           cell_block_offsets=m1.cell_block_offsets, rank=0, n_ranks=1, h=1.0, deform_amp=0.03, **tables)
    comm = pkg.Communicator(0, 1)
    op = pkg.PoissonOperator(mesh, 0, pkg.COEF_STEP64, comm=comm)
    L, h = pkg.lib(), op.mf_data.handle
    ptr = lambda t: C.c_void_p(t.data_ptr())
    g = torch.Generator(device="cuda:0").manual_seed(6)
    src = torch.rand(no + ng, dtype=torch.float64, device="cuda:0", generator=g)
    src[no:] = 0.0
    outs = []
    assert L.bp5_mf_set_overlap(h, 1) == 0                # force the overlapped schedule (the default decides by slab size)
    for v in (3, variant):
        op.mf_data.set_apply_variant(v)
        op.mf_data.set_block_workgroups(8)
        d = op.initialize_dof_vector()
        d.fill_(float("nan"))
        s_in = src.clone()
        assert L.bp5_apply_distributed(h, ptr(op.coef), ptr(s_in), ptr(d), 1) == 0
        outs.append(d)
    assert float((outs[1] - outs[0]).abs().max()) < 1e-12 * float(outs[0].abs().max())
    # the overlapped 3-phase schedule (default, the reference's overlap_communication_computation, bp5/step-64.cu:241:
    # exchange on the communication stream under the interior bricks, ghost-touching bricks after it) against the
    # sequential one (overlap off: exchange and one unsplit launch on the compute stream).  Same kernels, same
    # per-brick order, ONE combine pass after the last range: the block kernel's result is bitwise identical.
    assert L.bp5_mf_set_overlap(h, 0) == 0
    d_seq = op.initialize_dof_vector()
    d_seq.fill_(float("nan"))
    s_in = src.clone()
    assert L.bp5_apply_distributed(h, ptr(op.coef), ptr(s_in), ptr(d_seq), 1) == 0
    if variant == 56:
        assert torch.equal(d_seq, outs[1])
    else:
        assert float((d_seq - outs[1]).abs().max()) < 1e-13 * float(outs[1].abs().max())
    assert L.bp5_mf_set_overlap(h, 1) == 0
    d_acc = torch.full_like(d_seq, 0.25)                  # accumulate mode (zero_dst = 0) through the phases
    d_acc[no:] = 0.0                                      # (ghost entries of dst hold contributions only: zero on entry)
    s_in = src.clone()
    assert L.bp5_apply_distributed(h, ptr(op.coef), ptr(s_in), ptr(d_acc), 0) == 0
    c = torch.from_numpy(m1.constrained.astype(np.int64)).cuda()
    want_acc = d_seq + 0.25
    want_acc[no:] = 0.0
    want_acc[c] = src[c]
    assert float((d_acc - want_acc).abs().max()) < 1e-12 * float(want_acc.abs().max())
    b = op.assemble_rhs()
    xs, scheds = [], []
    for overlap in (1, 0, 1, 2):
        assert L.bp5_mf_set_overlap(h, overlap) == 0
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(10, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        Ax, xin = op.initialize_dof_vector(), x.clone()
        assert L.bp5_apply_distributed(h, ptr(op.coef), ptr(xin), ptr(Ax), 1) == 0
        assert abs(float(torch.linalg.norm((Ax - b)[:no])) - ctl.last_value()) < 1e-9 * ctl.initial_value()
        xs.append(x)
        scheds.append((ctl.exchange_schedule, ctl.dot_products_fused))
    if variant == 56:
        # the dot products are formed inside the block kernel on every rank's cells in BOTH exchange schedules (the owners' unpack kernel
        # corrects v.v and r.v for the contributions it adds): overlap on = boundary-first (ghost-touching bricks, their combine rows, the
        # exchange on the communication stream under the interior bricks), overlap off = one launch, exchange on the compute stream
        assert scheds == [(2, True), (1, True), (2, True), (4, True)]
        assert ctl.apply_kernel.startswith("apply_block_kernel<4,false,32,1,") and int(ctl.apply_kernel.split(",")[-1].rstrip(">")) & 1048576
        assert torch.equal(xs[0], xs[2])                  # fixed summation order: the boundary-first schedule is bitwise reproducible
        # ... and all three schedules run the same kernels over the same workgroup ranges and columns: bitwise the same solution
        assert torch.equal(xs[0], xs[1]) and torch.equal(xs[3], xs[1])
        assert L.bp5_mf_set_overlap(h, 0) == 0
        sols = []
        for fused in (True, False, True):
            op.mf_data.set_cg_fusion(fused)
            x = op.initialize_dof_vector()
            ctl = pkg.IterationNumberControl(10, 0.0)
            pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
            assert ctl.dot_products_fused == fused
            sols.append(x)
        assert torch.equal(sols[0], sols[2])              # fixed summation order: bitwise reproducible
        assert torch.equal(sols[0], xs[1])                # (the unsplit solve above took the fused path too)
        assert float((sols[0] - sols[1]).abs().max()) < 1e-11 * float(sols[1].abs().max())
        # separate dot products: bitwise independent of the exchange schedule (3-phase with combine windows against unsplit)
        assert L.bp5_mf_set_overlap(h, 1) == 0
        op.mf_data.set_cg_fusion(False)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(10, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert not ctl.dot_products_fused and ctl.exchange_schedule == 3 and torch.equal(x, sols[1])
        op.mf_data.set_cg_fusion(True)
        # phase stamps (profile = 2): every phase of the boundary-first iteration is stamped; the stamps do not change the result
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(10, 0.0)
        pkg.SolverCGFullMerge(ctl, profile=2).solve(op, x, b, pkg.DiagonalMatrix())
        assert torch.equal(x, xs[0]) and ctl.exchange_schedule == 2
        ph = ctl.phase_ms
        assert all(v >= 0.0 for v in ph) and ph[0] > 0 and ph[2] > 0 and abs(sum(ph[:7]) - ph[7]) < 0.25 * ph[7] + 0.05
    else:
        assert float((xs[0] - xs[1]).abs().max()) < 1e-11 * float(xs[1].abs().max())
    op.mf_data.synchronize()
    op.mf_data.close()
    comm.close()

# ------------------------------------------------------------------ native Helmholtz operator
def _helmholtz_oracle(pr):
    c = pr.mesh.constrained.astype(np.int64)

    def A(s):                                   # HelmholtzOperator::vmult, step-64/step-64.cu:283-300
        d = O.apply_helmholtz_cells(pr.mesh, pr.N, pr.D, pr.w, s)
        d[c] = s[c]
        return d
    return A


@pytest.mark.parametrize("p,quad,cells,amp", [(1, 0, (5, 4, 3), 0.03), (2, 0, (4, 3, 3), 0.03), (2, 1, (3, 3, 2), 0.0), (3, 0, (3, 3, 2), 0.04), (3, 1, (3, 2, 2), 0.04),
                                              (4, 0, (3, 2, 2), 0.04), (4, 1, (3, 2, 2), 0.0), (5, 0, (2, 2, 2), 0.03), (6, 1, (2, 2, 2), 0.03), (7, 0, (2, 2, 1), 0.03),
                                              (8, 0, (2, 1, 2), 0.02), (8, 1, (2, 2, 1), 0.0)])
def test_native_helmholtz_operator_pencil_kernel(p, quad, cells, amp):
    """bp5_mf_set_operator(BP5_OP_HELMHOLTZ): step-64's (grad v, grad u) + (v, a u) (LocalHelmholtzOperator + HelmholtzOperatorQuad,
    step-64/step-64.cu:154-160,201-219) as a build of the fused pencil kernel -- evaluate(true, true) / integrate(true, true) cost one more
    contraction each way, the coefficient a(x_q) JxW is the seventh plane -- against the oracle's restatement, every degree, both
    quadratures, deformed cells; accumulate mode; both solvers."""
    torch = _t()
    pr = O.Problem(p, cells, quad, h=0.25, deform_amp=amp)
    mesh = pkg.BrickMesh(p, cells, h=0.25, deform_amp=amp)
    op = pkg.HelmholtzOperator(mesh, quad, pkg.COEF_STEP64)
    mf = op.mf_data
    assert mf.coef_size() == 7 * mesh.n_cells * (p + 1) ** 3 and mf.get_apply_variant() == 0
    # the seven planes: six merged Laplace planes with coefficient 1, then a JxW
    K, JxW, xq = O.jacobians(pr.mesh, pr.N, pr.D, pr.w)
    got = mf.coef_reference_layout(op.coef).cpu().numpy().reshape(7, mesh.n_cells, -1)
    ref6 = O.merged_metric(pr.mesh, pr.N, pr.D, pr.w, O.kappa_none)
    assert np.abs(got[:6] - ref6).max() < 1e-12 * np.abs(ref6).max()
    mass = O.kappa_step64(xq) * JxW
    assert np.abs(got[6] - mass.reshape(mesh.n_cells, -1)).max() < 1e-12 * np.abs(mass).max()
    A = _helmholtz_oracle(pr)
    s = O.deterministic_src(pr.mesh.n_dofs, seed=81)
    d = op.initialize_dof_vector()
    d.fill_(float("nan"))
    op.vmult(d, dev(s))
    assert rel(d.cpu().numpy(), A(s)) < TOL_OP
    acc = torch.full_like(d, 0.25)
    mf.cell_loop(op.coef, dev(s), acc)
    assert rel(acc.cpu().numpy() - 0.25, O.apply_helmholtz_cells(pr.mesh, pr.N, pr.D, pr.w, s)) < TOL_OP
    b = op.assemble_rhs()
    its = 6
    xr, _, _ = O.cg_plain(A, pr.rhs(), its)
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == its and "8388608" in ctl.apply_kernel and rel(x.cpu().numpy(), xr) < TOL_CG
    with pytest.raises(pkg.BP5Error):
        mf.set_apply_variant(10)                          # pencil kernel (0) and block kernel (56) only
    # Jacobi for the Helmholtz operator: diag(A) with the mass term (sum-factorised: N*N in every direction on the seventh plane) against
    # the definition (A e_g)_g on the smallest meshes; Jacobi-preconditioned merged CG against the oracle's
    if pr.mesh.n_dofs <= 400:
        d_ref = np.array([A(np.eye(1, pr.mesh.n_dofs, g).ravel())[g] for g in range(pr.mesh.n_dofs)])
        assert rel(op.compute_diagonal().cpu().numpy(), d_ref) < 1e-13
        xj, _, _ = O.cg_merged(A, pr.rhs(), its, diag=1.0 / d_ref)
        x = op.initialize_dof_vector()
        pkg.SolverCGFullMerge(pkg.IterationNumberControl(its, 0.0)).solve(op, x, b, pkg.DiagonalMatrix(op.compute_diagonal(invert=True)))
        assert rel(x.cpu().numpy(), xj) < TOL_CG


@pytest.mark.parametrize("p,cells,block", [(1, (17, 9, 10), (8, 8, 8)), (2, (9, 8, 5), (8, 8, 4)), (3, (9, 5, 6), (8, 4, 4)), (4, (9, 8, 6), (4, 4, 4)), (4, (6, 5, 5), (4, 4, 2)),
                                           (5, (7, 5, 3), (6, 4, 2)), (6, (5, 4, 3), (4, 4, 2)), (7, (5, 3, 3), (4, 2, 2)), (8, (3, 3, 3), (2, 2, 2))])
@pytest.mark.parametrize("quad", [0, 1])
def test_native_helmholtz_operator_block_kernel(p, cells, block, quad):
    """The same operator on the deterministic block kernel (variant 56: bricks, packed indices, run-length write-out), with the CG dot
    products fused into it: p.(A p) is the quadrature-point energy ghat^T S ghat + a JxW u_q^2, taken from registers."""
    torch = _t()
    pr = O.Problem(p, cells, quad, h=0.2, deform_amp=0.03)
    mesh = pkg.BrickMesh(p, cells, h=0.2, deform_amp=0.03, cell_block=block, dof_numbering=1, cell_block_order=1)
    perm = mesh.global_ids.astype(np.int64)
    op = pkg.HelmholtzOperator(mesh, quad, pkg.COEF_STEP64)
    mf = op.mf_data
    mf.set_apply_variant(56)
    mf.set_block_workgroups(8)
    A = _helmholtz_oracle(pr)
    s = O.deterministic_src(pr.mesh.n_dofs, seed=82)
    ref = A(s)[perm]
    outs = []
    for _ in range(3):
        d = op.initialize_dof_vector()
        d.fill_(float("nan"))
        op.vmult(d, dev(s[perm]))
        outs.append(d)
    assert rel(outs[0].cpu().numpy(), ref) < TOL_OP
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    b = op.assemble_rhs()
    its = 8
    xr, _, _ = O.cg_plain(A, pr.rhs(), its)
    sols = []
    for fused in (True, False, True):
        mf.set_cg_fusion(fused)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.dot_products_fused == fused and ctl.apply_kernel.startswith(f"apply_block_kernel<{p},")
        assert rel(x.cpu().numpy(), xr[perm]) < TOL_CG
        sols.append(x)
    assert torch.equal(sols[0], sols[2])
    x = op.initialize_dof_vector()
    ctl = pkg.IterationNumberControl(its, 0.0)
    pkg.SolverCG(ctl).solve(op, x, b, pkg.DiagonalMatrix())          # standard CG: d.h from the kernel's energy
    assert ctl.dot_products_fused and rel(x.cpu().numpy(), xr[perm]) < TOL_CG


def test_native_helmholtz_reproduces_the_step64_tutorial_norms():
    """HelmholtzProblem::run of step-64 (step-64/step-64.cu:602-616,619-650): FE_Q(3), unit cube, -Laplace u + a u = 1, zero Dirichlet values,
    CG to 1e-12 |b|: the tutorial prints `solution norm` 0.0205439 / 0.0205269 for 343 / 2197 DoFs (remembered output of the upstream
    tutorial, SURVEY 8c: a sanity signal, the oracle reproduces the same digits).  Here through the library's NATIVE Helmholtz kernel."""
    for n, want in ((2, 0.0205439), (4, 0.0205269)):
        mesh = pkg.BrickMesh(3, (n, n, n), h=1.0 / n)
        op = pkg.HelmholtzOperator(mesh, pkg.QUAD_GAUSS, pkg.COEF_STEP64)
        b = op.assemble_rhs()
        x = op.initialize_dof_vector()
        ctl = pkg.SolverControl(1000, 1e-12 * float(_t().linalg.norm(b)))
        pkg.SolverCG(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert abs(op.l2_norm_solution(x) - want) < 6e-8, (n, op.l2_norm_solution(x))


def test_native_helmholtz_behind_the_halo_exchange():
    """The Helmholtz build inside the distributed entry points: a slab mesh with a ghost plane exchanged with itself through RCCL (as
    test_block_kernel_behind_the_halo_exchange), block kernel with fused dot products in every exchange schedule: the same bits."""
    from types import SimpleNamespace
    torch = _t()
    p, cells = 4, (9, 8, 10)
    m1 = pkg.BrickMesh(p, cells, deform_amp=0.03, rank=1, n_ranks=2, cell_block=(4, 4, 4), dof_numbering=1, cell_block_order=1)
    ng, no = m1.n_ghost, m1.n_owned
    mesh = SimpleNamespace(degree=p, n=p + 1, cells=cells, n_cells=m1.n_cells, n_interior_cells=m1.n_interior_cells, n_owned=no, n_ghost=ng,
                           n_local=no + ng, n_global_dofs=no, l2g=m1.l2g, coords=m1.coords, global_ids=m1.global_ids, constrained=m1.constrained,
                           cell_block_offsets=m1.cell_block_offsets, rank=0, n_ranks=1, h=1.0, deform_amp=0.03, n_neighbors=1,
                           neighbor_rank=np.zeros(1, np.int32), send_offsets=np.asarray([0, ng], np.uint32), send_indices=_consistent_self_glue(m1),
                           recv_offsets=np.asarray([0, ng], np.uint32))
    comm = pkg.Communicator(0, 1)
    op = pkg.HelmholtzOperator(mesh, 0, pkg.COEF_STEP64, comm=comm)
    op.mf_data.set_apply_variant(56)
    op.mf_data.set_block_workgroups(8)
    b = op.assemble_rhs()
    xs = []
    for overlap, sched in ((0, 1), (1, 2), (2, 4)):
        op.mf_data.set_overlap(overlap)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(10, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.dot_products_fused and ctl.exchange_schedule == sched and "9725952" in ctl.apply_kernel
        xs.append(x)
    assert torch.equal(xs[0], xs[1]) and torch.equal(xs[0], xs[2])
    op.mf_data.set_cg_fusion(False)
    op.mf_data.set_overlap(0)
    x = op.initialize_dof_vector()
    pkg.SolverCGFullMerge(pkg.IterationNumberControl(10, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
    assert float((x - xs[0]).abs().max()) < 1e-11 * float(x.abs().max())
    op.mf_data.synchronize()
    op.mf_data.close()
    comm.close()


# ------------------------------------------------------------------ hanging nodes
def _hanging_namespace(m):
    from types import SimpleNamespace
    return SimpleNamespace(degree=m.p, n=m.n, n_cells=m.n_cells, n_interior_cells=m.n_cells, n_owned=m.n_dofs, n_ghost=0, n_local=m.n_dofs,
                           n_global_dofs=m.n_dofs, l2g=m.l2g, coords=m.coords, constrained=m.constrained, n_neighbors=0,
                           neighbor_rank=np.zeros(0, np.int32), send_offsets=np.zeros(1, np.uint32), send_indices=np.zeros(0, np.uint32),
                           recv_offsets=np.zeros(1, np.uint32), cell_block_offsets=None, constraint_mask=m.constraint_mask, rank=0, n_ranks=1)


@pytest.mark.parametrize("p,quad,amp", [(1, 0, 0.0), (2, 0, 0.0), (2, 1, 0.03), (3, 0, 0.03), (3, 1, 0.0), (4, 0, 0.02), (5, 0, 0.0)])
def test_hanging_nodes_on_a_2_to_1_refined_mesh(p, quad, amp):
    """constraint_mask path of read_dof_values / distribute_local_to_global (resolve_hanging_nodes,
    bp5/fe_evaluation_gl.h:150-151,167-168): a mesh with one planar 2:1 interface (2x2x1 coarse cubes, then 3x4x2 cubes of
    half the size).  The fine cells at the interface name the coarse face's DoFs and carry BP5_HANG_* masks; geometry,
    operator, RHS, both CG solvers and the L2 norm against the oracle, whose hanging-node path is pinned by the known-answer
    tests of tests/test_oracle_known_answers.py (null space, symmetry, energy of polynomials across the interface)."""
    torch = _t()
    m = O.HangingBrickMesh(p, 2, 2, 1, 3, H=0.5, deform_amp=amp)
    assert (m.constraint_mask != 0).sum() == 8
    _, _, w, N, D = O.shape_tables(p, quad)
    coef_ref = O.merged_metric(m, N, D, w, O.kappa_step64)
    op = pkg.PoissonOperator(_hanging_namespace(m), quad, pkg.COEF_STEP64)
    assert op.mf_data.get_apply_variant() == 90
    got = op.mf_data.coef_reference_layout(op.coef).cpu().numpy().reshape(6, m.n_cells, -1)
    assert np.abs(got - coef_ref).max() < 1e-12 * np.abs(coef_ref).max()
    c = m.constrained.astype(np.int64)

    def A(s):
        d = O.apply_cells(m, coef_ref, N, D, s)
        d[c] = s[c]
        return d

    s = O.deterministic_src(m.n_dofs, seed=71)
    dst = op.initialize_dof_vector()
    dst.fill_(float("nan"))
    op.vmult(dst, dev(s))
    assert rel(dst.cpu().numpy(), A(s)) < TOL_OP
    acc = torch.full_like(dst, 0.25)                      # cell_loop accumulates
    op.mf_data.cell_loop(op.coef, dev(s), acc)
    assert rel(acc.cpu().numpy() - 0.25, O.apply_cells(m, coef_ref, N, D, s)) < TOL_OP
    b = op.assemble_rhs()
    b_ref = O.assemble_rhs(m)
    assert rel(b.cpu().numpy(), b_ref) < TOL_OP
    its = 3 if p == 1 else 6                              # (p = 1 has only a handful of free DoFs: CG is exact after a few steps)
    xr, _, _ = O.cg_plain(A, b_ref, its)
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == its and rel(x.cpu().numpy(), xr) < TOL_CG
    assert abs(op.l2_norm_solution(x) - O.l2_norm_solution(m, xr)) < 1e-11 * O.l2_norm_solution(m, xr)
    d = op.mf_data.get_data()
    import ctypes as C
    masks = np.zeros(m.n_cells, np.uint32)
    pkg.lib().bp5_copy_d2h(masks.ctypes.data, C.c_void_p(d.constraint_mask), masks.nbytes)
    assert np.array_equal(masks, m.constraint_mask)       # MatrixFree::Data::constraint_mask mirrors the input
    with pytest.raises(pkg.BP5Error):
        op.mf_data.set_apply_variant(3)                   # no other kernel honours the masks
    d_ref = O.operator_diagonal(m, coef_ref, N, D)        # coarse DoFs on the interface: diagonal of R^T A_e R
    assert rel(op.compute_diagonal().cpu().numpy(), d_ref) < 1e-13
    # malformed masks are refused at create time
    bad = _hanging_namespace(m)
    for wrong in (1 << 12,                                # unknown bit
                  8 | 64,                                 # position bits without a constrained face or edge
                  1 | 2 | 8 | 16 | 128):                  # faces x and y: SIDE_X (where face x sits) and HALF_X (face y's interpolation along x) disagree
        bad.constraint_mask = m.constraint_mask.copy()
        bad.constraint_mask[-1] = wrong
        with pytest.raises(pkg.BP5Error):
            pkg.PoissonOperator(bad, quad)


def _refined(pattern, p, amp):
    if pattern == "L":                # re-entrant edge along z: constrained EDGES
        coarse, r = (2, 2, 2), np.zeros((2, 2, 2), bool)
        r[:, 0, 0] = r[:, 0, 1] = r[:, 1, 0] = True
    elif pattern == "core":           # one refined cube inside 3^3: three constrained faces on every child
        coarse, r = (3, 3, 3), np.zeros((3, 3, 3), bool)
        r[1, 1, 1] = True
    else:                             # staircase: one, two, three faces and edges in one mesh
        coarse, r = (3, 2, 2), np.zeros((2, 2, 3), bool)
        r[0, 0, 0] = r[0, 0, 1] = r[0, 1, 0] = r[1, 0, 0] = r[1, 1, 2] = True
    return O.RefinedBrickMesh(p, coarse, r, H=0.5, deform_amp=amp)


@pytest.mark.parametrize("pattern,p,quad,amp", [("L", 1, 0, 0.0), ("L", 2, 0, 0.03), ("L", 3, 1, 0.0), ("L", 4, 0, 0.02), ("core", 2, 0, 0.0),
                                                ("core", 3, 0, 0.03), ("stairs", 1, 1, 0.0), ("stairs", 2, 0, 0.03), ("stairs", 3, 0, 0.0),
                                                ("stairs", 5, 0, 0.0), ("L", 6, 0, 0.0), ("L", 8, 0, 0.02)])
def test_hanging_nodes_on_general_two_level_meshes(pattern, p, quad, amp):
    """General 2:1 meshes (oracle: RefinedBrickMesh, pinned by its known-answer tests): cells on the rim of a refined region with
    one, two (its edges) and three (its corners) constrained faces, and constrained edges at re-entrant corners (BP5_HANG_EDGE_*).
    Geometry (the node coordinates are interpolated like any FE function), operator, RHS, both solvers, L2 norm."""
    torch = _t()
    m = _refined(pattern, p, amp)
    kinds = set(int(k) for k in m.constraint_mask)
    if pattern == "L":
        assert any(k & (512 | 1024 | 2048) for k in kinds)
    if pattern == "stairs":
        assert {bin(k & 7).count("1") for k in kinds} >= {0, 1, 2, 3} and any(k & (512 | 1024 | 2048) for k in kinds)
    _, _, w, N, D = O.shape_tables(p, quad)
    coef_ref = O.merged_metric(m, N, D, w, O.kappa_step64)
    op = pkg.PoissonOperator(_hanging_namespace(m), quad, pkg.COEF_STEP64)
    assert op.mf_data.get_apply_variant() == 90
    got = op.mf_data.coef_reference_layout(op.coef).cpu().numpy().reshape(6, m.n_cells, -1)
    assert np.abs(got - coef_ref).max() < 1e-12 * np.abs(coef_ref).max()
    c = m.constrained.astype(np.int64)

    def A(s):
        d = O.apply_cells(m, coef_ref, N, D, s)
        d[c] = s[c]
        return d

    s = O.deterministic_src(m.n_dofs, seed=72)
    dst = op.initialize_dof_vector()
    dst.fill_(float("nan"))
    op.vmult(dst, dev(s))
    assert rel(dst.cpu().numpy(), A(s)) < TOL_OP
    b = op.assemble_rhs()
    b_ref = O.assemble_rhs(m)
    assert rel(b.cpu().numpy(), b_ref) < TOL_OP
    its = 3 if p == 1 else 6
    xr, _, _ = O.cg_plain(A, b_ref, its)
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == its and rel(x.cpu().numpy(), xr) < TOL_CG
    assert abs(op.l2_norm_solution(x) - O.l2_norm_solution(m, xr)) < 1e-11 * O.l2_norm_solution(m, xr)
    # Jacobi preconditioner on the refined mesh: the diagonal entries of coarse DoFs named on constrained faces / edges come from
    # R^T A_e R (R e_s spreads into two faces for a DoF on their common edge: one cell-operator application per such entry)
    d_ref = O.operator_diagonal(m, coef_ref, N, D)
    assert rel(op.compute_diagonal().cpu().numpy(), d_ref) < 1e-13
    xj, _, _ = O.cg_merged(A, b_ref, its, diag=1.0 / d_ref)
    x = op.initialize_dof_vector()
    pkg.SolverCGFullMerge(pkg.IterationNumberControl(its, 0.0)).solve(op, x, b, pkg.DiagonalMatrix(op.compute_diagonal(invert=True)))
    assert rel(x.cpu().numpy(), xj) < TOL_CG


def _with_cell_blocks(m, group):
    """The mesh handed over the way a host that cares for the block kernel does it: cells in groups of `group` consecutive cells, DoFs
    renumbered block-major -- all DoFs touched by the same SET of cell groups consecutively (the brick-major numbering of the library's own
    mesh generator, for an arbitrary mesh): a group's DoF list is then a few dozen runs (the oracle numbers by first appearance: a run per line
    of a shared face).  Returns the mesh namespace and new_of_old (vectors: new[new_of_old] = old)."""
    # unrefined cubes in groups of `group` consecutive cells, refined ones parent by parent (the generators emit the children of one cube
    # consecutively: compact 2 x 2 x 2 groups)
    levels = getattr(m, "levels", None)
    starts = [0, m.n_coarse_cells] if levels is None else [int(np.argmax(levels == lv)) for lv in sorted(set(levels))]
    starts.append(m.n_cells)
    offsets = []
    for k in range(len(starts) - 1):
        offsets += list(range(starts[k], starts[k + 1], group if k == 0 else 8))
    offsets = np.asarray(offsets + [m.n_cells], np.uint32)
    l2g = m.l2g.astype(np.int64)
    sig = [set() for _ in range(m.n_dofs)]
    for b in range(len(offsets) - 1):
        for g in np.unique(l2g[offsets[b]:offsets[b + 1]]):
            sig[g].add(b)
    con = np.zeros(m.n_dofs, bool)
    con[m.constrained.astype(np.int64)] = True          # (Dirichlet DoFs apart: the run tables carry the flag per run)
    order = sorted(range(m.n_dofs), key=lambda g: (min(sig[g]) if sig[g] else 1 << 30, tuple(sorted(sig[g])), bool(con[g]), g))
    new_of_old = np.empty(m.n_dofs, np.int64)
    new_of_old[np.asarray(order)] = np.arange(m.n_dofs)
    ns = _hanging_namespace(m)
    ns.l2g = new_of_old[l2g].astype(np.uint32)
    ns.coords = m.coords[np.asarray(order)]
    ns.constrained = np.sort(new_of_old[m.constrained.astype(np.int64)]).astype(np.uint32)
    ns.cell_block_offsets = offsets
    return ns, new_of_old


def _three_level(p, amp):
    r0 = np.zeros((3, 3, 4), bool)
    r0[:, :, :2] = True
    r1 = np.zeros((6, 6, 8), bool)
    r1[:2, :2, :2] = True
    return O.OctreeBrickMesh(p, (4, 3, 3), [r0, r1], H=0.5, deform_amp=amp)


@pytest.mark.parametrize("mesh_kind,p,quad,amp,group", [("stairs", 1, 0, 0.0, 3), ("stairs", 2, 0, 0.03, 3), ("L", 3, 1, 0.0, 2), ("L", 4, 0, 0.02, 2), ("core", 2, 1, 0.0, 3),
                                                        ("stairs", 5, 0, 0.0, 3), ("L", 6, 0, 0.02, 2), ("L", 7, 1, 0.0, 2), ("L", 8, 0, 0.02, 1),
                                                        ("three", 2, 0, 0.0, 4), ("three", 3, 0, 0.03, 4), ("three", 4, 0, 0.0, 2)])
def test_hanging_nodes_in_the_deterministic_block_kernel(mesh_kind, p, quad, amp, group):
    """resolve_hanging_nodes inside the block kernel (bp5/fe_evaluation_gl.h:150-151,167-168; round 2 ran refined meshes on the atomic pencil
    kernel only): fix-up after the packed gather, adjoint before the accumulation into the brick vector, only in passes that hold a flagged
    cell.  Cells handed over in groups of consecutive cells (neighbours share DoFs: multi-round passes), every degree; two-level meshes
    (faces in every number, constrained edges) and a THREE-level octree mesh.  Against the oracle and the pencil kernel; bitwise
    reproducible; merged CG with the dot products fused into it (the energy is taken from the resolved values); Jacobi preconditioner."""
    torch = _t()
    m = _three_level(p, amp) if mesh_kind == "three" else _refined(mesh_kind, p, amp)
    if mesh_kind == "three":
        assert set(m.levels) == {0, 1, 2}
    _, _, w, N, D = O.shape_tables(p, quad)
    coef_ref = O.merged_metric(m, N, D, w, O.kappa_step64)
    c = m.constrained.astype(np.int64)

    def A(s):
        d = O.apply_cells(m, coef_ref, N, D, s)
        d[c] = s[c]
        return d

    ns, new_of_old = _with_cell_blocks(m, group)
    to_new = lambda v: dev(v[np.argsort(new_of_old)])      # oracle order -> the renumbered mesh
    to_old = lambda t: t.cpu().numpy()[new_of_old]
    op = pkg.PoissonOperator(ns, quad, pkg.COEF_STEP64)
    mf = op.mf_data
    mf.set_apply_variant(56)
    mf.set_block_workgroups(8)
    assert mf.block_plan_info()[2], mf.block_plan_info()   # packed indices available
    s = O.deterministic_src(m.n_dofs, seed=73)
    outs = []
    for _ in range(3):
        d = op.initialize_dof_vector()
        d.fill_(float("nan"))
        op.vmult(d, to_new(s))
        outs.append(d)
    assert rel(to_old(outs[0]), A(s)) < TOL_OP
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    mf.set_apply_variant(90)                                # the atomic pencil kernel on the same handle
    d90 = op.initialize_dof_vector()
    op.vmult(d90, to_new(s))
    assert float((d90 - outs[0]).abs().max()) < 1e-12 * float(d90.abs().max())
    mf.set_apply_variant(56)
    b = op.assemble_rhs()
    b_ref = O.assemble_rhs(m)
    assert rel(to_old(b), b_ref) < TOL_OP
    its = 3 if p == 1 else 6
    xr, _, _ = O.cg_plain(A, b_ref, its)
    sols = []
    for fused in (True, False, True):
        mf.set_cg_fusion(fused)
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(its, 0.0)
        pkg.SolverCGFullMerge(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.dot_products_fused == fused and ctl.apply_kernel.startswith(f"apply_block_kernel<{p},")
        assert int(ctl.apply_kernel.split(",")[-1].rstrip(">")) & 2097152
        assert rel(to_old(x), xr) < TOL_CG
        sols.append(x)
    assert torch.equal(sols[0], sols[2])
    x = op.initialize_dof_vector()
    pkg.SolverCG(pkg.IterationNumberControl(its, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
    assert rel(to_old(x), xr) < TOL_CG
    d_ref = O.operator_diagonal(m, coef_ref, N, D)
    assert rel(to_old(op.compute_diagonal()), d_ref) < 1e-13
    assert abs(op.l2_norm_solution(x) - O.l2_norm_solution(m, xr)) < 1e-11 * O.l2_norm_solution(m, xr)


@pytest.mark.parametrize("mesh_kind,p,quad", [("stairs", 2, 0), ("L", 3, 1), ("three", 2, 0), ("L", 5, 0)])
def test_hanging_nodes_in_the_affine_geometry_mode(mesh_kind, p, quad):
    """Undeformed 2:1 meshes are affine cell by cell: BP5_GEOM_AFFINE (per-cell K K^T + one scalar plane, G = 1) with hanging nodes -- operator
    (pencil kernel: affine build + hanging-node fix-up), solvers and the Jacobi diagonal (round 2 refused the diagonal in this mode)."""
    m = _three_level(p, 0.0) if mesh_kind == "three" else _refined(mesh_kind, p, 0.0)
    _, _, w, N, D = O.shape_tables(p, quad)
    coef_ref = O.merged_metric(m, N, D, w, O.kappa_step64)
    c = m.constrained.astype(np.int64)

    def A(s):
        d = O.apply_cells(m, coef_ref, N, D, s)
        d[c] = s[c]
        return d

    op = pkg.PoissonOperator(_hanging_namespace(m), quad, pkg.COEF_STEP64, geometry=pkg.GEOM_AFFINE)
    assert op.coef is None and op.mf_data.get_apply_variant() == 90
    s = O.deterministic_src(m.n_dofs, seed=74)
    d = op.initialize_dof_vector()
    d.fill_(float("nan"))
    op.vmult(d, dev(s))
    assert rel(d.cpu().numpy(), A(s)) < TOL_OP
    d_ref = O.operator_diagonal(m, coef_ref, N, D)
    assert rel(op.compute_diagonal().cpu().numpy(), d_ref) < 1e-13
    b = op.assemble_rhs()
    its = 6
    xj, _, _ = O.cg_merged(A, O.assemble_rhs(m), its, diag=1.0 / d_ref)
    x = op.initialize_dof_vector()
    pkg.SolverCGFullMerge(pkg.IterationNumberControl(its, 0.0)).solve(op, x, b, pkg.DiagonalMatrix(op.compute_diagonal(invert=True)))
    assert rel(x.cpu().numpy(), xj) < TOL_CG
    with pytest.raises(pkg.BP5Error):
        op.mf_data.set_apply_variant(56)                  # (accepted as a request, refused at the launch: the affine block build has no hanging-node path)
        op.vmult(d, dev(s))


def test_hanging_node_golden_fixtures():
    """the committed outputs of tests/golden/hanging_cases.npz (operator, RHS, diagonal, 6 CG iterations on a staircase-refined mesh)
    straight against the HIP path, without the oracle in between"""
    from make_golden import HANGING_CASES, hanging_mesh
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hanging_cases.npz"))
    for p, amp in HANGING_CASES:
        m = hanging_mesh(p, amp)                       # mesh arrays only (inputs); expected outputs come from the file
        k = f"p{p}_a{amp}"
        assert np.array_equal(m.constraint_mask, z[k + "_masks"])
        op = pkg.PoissonOperator(_hanging_namespace(m), 0, pkg.COEF_STEP64)
        dst = op.initialize_dof_vector()
        op.vmult(dst, dev(O.deterministic_src(m.n_dofs, seed=300 + p)))
        assert rel(dst.cpu().numpy(), z[k + "_vmult"]) < TOL_OP
        b = op.assemble_rhs()
        assert rel(b.cpu().numpy(), z[k + "_rhs"]) < TOL_OP
        assert rel(op.compute_diagonal().cpu().numpy(), z[k + "_diag"]) < 1e-13
        x = op.initialize_dof_vector()
        pkg.SolverCG(pkg.IterationNumberControl(6, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
        assert rel(x.cpu().numpy(), z[k + "_x"]) < TOL_CG


# ------------------------------------------------------------------ edge cases
def test_single_cell_and_tiny_meshes():
    """one cell (every DoF on the Dirichlet boundary except the interior ones), p = 1 and p = 8"""
    for p in (1, 4, 8):
        pr = O.Problem(p, (1, 1, 1), 0, deform_amp=0.0)
        op = pkg.PoissonOperator(pkg.BrickMesh(p, (1, 1, 1)), 0)
        s = O.deterministic_src(pr.mesh.n_dofs, seed=1)
        d = op.initialize_dof_vector()
        op.vmult(d, dev(s))
        assert rel(d.cpu().numpy(), pr.vmult(s)) < TOL_OP
        b = op.assemble_rhs()
        x = op.initialize_dof_vector()
        ctl = pkg.IterationNumberControl(3, 0.0)
        if p == 1:                                   # no interior DoF: b == 0 -> converged at step 0
            pkg.SolverCG(ctl).solve(op, x, b, pkg.DiagonalMatrix())
            assert ctl.last_step() == 0 and float(x.abs().max()) == 0.0
        else:
            pkg.SolverCG(ctl).solve(op, x, b, pkg.DiagonalMatrix())
            xr, k, _ = O.cg_plain(pr.vmult, pr.rhs(), 3)
            assert rel(x.cpu().numpy(), xr) < TOL_CG


def test_zero_rhs_and_zero_iterations():
    op = pkg.PoissonOperator(pkg.BrickMesh(2, (3, 3, 3)), 0)
    b = op.initialize_dof_vector()                   # b = 0: residual 0 <= tol -> 0 iterations, x = 0
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        x.fill_(3.0)
        ctl = pkg.IterationNumberControl(5, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == 0 and float(x.abs().max()) == 0.0
    b = op.assemble_rhs()
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):   # max_iter = 0: returns the initial guess 0
        x = op.initialize_dof_vector()
        x.fill_(3.0)
        ctl = pkg.IterationNumberControl(0, 0.0)
        solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
        assert ctl.last_step() == 0 and float(x.abs().max()) == 0.0
        assert abs(ctl.last_value() - float(b.norm())) < 1e-12 * float(b.norm())


def test_cg_breakdown_is_reported_not_thrown_across_the_abi():
    """p.Ap == 0 (here: an all-zero metric) must surface as BP5_ERR_BREAKDOWN (ExcDivideByZero,
    bp5/solver.h:501), through the status code."""
    torch = _t()
    op = pkg.PoissonOperator(pkg.BrickMesh(2, (2, 2, 2)), 0)
    op.coef.zero_()
    b = op.assemble_rhs()
    for solver in (pkg.SolverCG, pkg.SolverCGFullMerge):
        x = op.initialize_dof_vector()
        with pytest.raises(pkg.BP5Error) as e:
            solver(pkg.IterationNumberControl(5, 0.0)).solve(op, x, b, pkg.DiagonalMatrix())
        assert e.value.status == 6


def test_argument_validation():
    torch = _t()
    import ctypes as C
    mesh = pkg.BrickMesh(2, (2, 2, 2))
    op = pkg.PoissonOperator(mesh, 0)
    x = op.initialize_dof_vector()
    short = torch.zeros(5, dtype=torch.float64, device="cuda:0")
    with pytest.raises(pkg.BP5Error):
        op.vmult(short, x)                           # too short
    with pytest.raises(pkg.BP5Error):
        op.vmult(x, x.float())                       # wrong dtype
    L, h = pkg.lib(), op.mf_data.handle
    p = lambda t: C.c_void_p(t.data_ptr())
    odd = torch.zeros(mesh.n_owned + 1, dtype=torch.float64, device="cuda:0")[1:]   # 8-byte but not 16-byte aligned
    assert L.bp5_vec_axpy(h, p(odd), 1.0, p(x), 4) == 1
    assert L.bp5_apply(h, None, p(x), p(x), 1) == 1
    # unknown variants and the timing-only ablation builds (wrong results by construction; they exist only in
    # libbp5_timing.so) are refused by the product library when they are SET, not at the first apply
    for bad_variant in (77, 21, 23, 41, 61, 80, 81, 85, 91, 93, 95, 97, 99, 73):
        assert L.bp5_mf_set_apply_variant(h, bad_variant) == 1
    assert L.bp5_mf_set_apply_variant(h, 0) == 0
    # a corrupt local_to_global is rejected on the host, before anything reaches the GPU
    from deal_and_ceed_on_gpu_amd import _lib
    bad = mesh.l2g.copy()
    bad[0, 0] = mesh.n_local + 5
    d = _lib.MFDesc()
    d.dim, d.degree, d.quadrature = 3, 2, 0
    d.n_cells, d.n_interior_cells, d.n_owned, d.n_ghost = mesh.n_cells, mesh.n_cells, mesh.n_owned, 0
    xyz, cst = np.ascontiguousarray(mesh.coords), np.ascontiguousarray(mesh.constrained)
    d.local_to_global_host, d.node_coords_host, d.constrained_host = bad.ctypes.data, xyz.ctypes.data, cst.ctypes.data
    d.n_constrained = cst.size
    hh = C.c_void_p()
    assert L.bp5_mf_create(C.byref(d), C.byref(hh)) == 1
