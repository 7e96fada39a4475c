"""CPU ORACLE (test infrastructure, NOT product code) -- numpy restatement of the BP5 hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
file.  The product path (deal-and-ceed-on-gpu_amd/) never does.

PARITY UNPINNED: the reference (peterrum/deal-and-ceed-on-gpu) holds no golden vectors,
known-answer tests or fixtures for this path (its tests/ directory holds one MPI smoke test,
tests/cuda_aware_mpi.cc) and its arithmetic lives in deal.II 9.2.0-pre (fork
peterrum/dealii, branch dealii-on-gpu; scripts/daint-gcc/make_dealii.sh:91), which is neither
vendored in /root/reference nor installed here.  The oracle is therefore pinned by the
mathematical known-answer tests of SURVEY.md Appendix A.7 (tests/test_oracle_known_answers.py)
and restates the published algorithm, following these reference call sites:

  * 1-D tables / local numbering ....... bp5/fe_evaluation_gl.h:139-142, bp5/step-64.cu:243-247
  * merged metric (6 planes, order) .... bp5/step-64.cu:84-114
  * per-cell operator .................. bp5/step-64.cu:147-194   (cell offset FIXED)
  * unmerged metric K, JxW ............. bp5/fe_evaluation_gl.h:318-369
  * vmult + Dirichlet copy ............. bp5/step-64.cu:263-276
  * RHS b_i = int phi_i ................ bp5/step-64.cu:372-418
  * plain PCG (parity target) .......... bp5/step-64.cu:428-453 (+ deal.II SolverCG, App. A.5)
  * fused PCG scalar formulas .......... bp5/solver.h:343-542    (x schedule FIXED)
  * mesh family ........................ bp5/step-64.cu:629-663
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import legendre as _L

QUAD_GAUSS = 0  # QGauss<1>(p+1), reference default (bp5/step-64.cu:246)
QUAD_GLL = 1    # QGaussLobatto<1>(p+1), "COLLOCATION" (bp5/step-64.cu:244)

# plane order of the merged symmetric metric, bp5/step-64.cu:107-113
PLANE_PAIRS = ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))


# ----------------------------------------------------------------------------- 1-D data (A.1)
def gauss_01(n):
    x, w = _L.leggauss(n)
    return (x + 1.0) / 2.0, w / 2.0


def gll_01(n):
    """Gauss-Lobatto-Legendre points/weights on [0,1] (FE_Q(p) support points, n = p+1)."""
    assert n >= 2
    c = np.zeros(n)
    c[-1] = 1.0                        # P_{n-1}
    if n == 2:
        x = np.array([-1.0, 1.0])
    else:
        xi = _L.legroots(_L.legder(c))
        for _ in range(3):             # Newton polish of P'_{n-1}
            xi = xi - _L.legval(xi, _L.legder(c)) / _L.legval(xi, _L.legder(c, 2))
        x = np.concatenate([[-1.0], np.sort(xi), [1.0]])
    x = 0.5 * (x - x[::-1])            # enforce exact antisymmetry
    w = 2.0 / (n * (n - 1) * _L.legval(x, c) ** 2)
    return (x + 1.0) / 2.0, w / 2.0


def lagrange_tables(nodes, points):
    """N[q][i] = phi_i(x_q), D[q][i] = phi_i'(x_q); phi_i Lagrange on `nodes` (both on [0,1])."""
    nodes = np.asarray(nodes, dtype=np.float64)
    points = np.asarray(points, dtype=np.float64)
    n = len(nodes)
    N = np.zeros((len(points), n))
    D = np.zeros((len(points), n))
    for q, x in enumerate(points):
        for i in range(n):
            den = 1.0
            for m in range(n):
                if m != i:
                    den *= nodes[i] - nodes[m]
            num = 1.0
            for m in range(n):
                if m != i:
                    num *= x - nodes[m]
            N[q, i] = num / den
            s = 0.0
            for l in range(n):
                if l == i:
                    continue
                t = 1.0
                for m in range(n):
                    if m != i and m != l:
                        t *= x - nodes[m]
                s += t
            D[q, i] = s / den
    return N, D


def shape_tables(p, quadrature):
    """(nodes, points, weights, N, D) for FE_Q(p) on GLL nodes with Gauss(p+1) or GLL(p+1)."""
    n = p + 1
    nodes, _ = gll_01(n)
    if quadrature == QUAD_GLL:
        pts, w = gll_01(n)
    elif quadrature == QUAD_GAUSS:
        pts, w = gauss_01(n)
    else:
        raise ValueError("quadrature")
    N, D = lagrange_tables(nodes, pts)
    if quadrature == QUAD_GLL:
        N = np.eye(n)                  # collocation: identity interpolation, exactly
    return nodes, pts, w, N, D


# ----------------------------------------------------------------------------- mesh (A.2, bp5/step-64.cu:629-663)
def kappa_none(X):
    return np.ones(X.shape[:-1])


def kappa_step64(X):
    """step-64/step-64.cu:117 coefficient 10/(0.05+2|x|^2), used as kappa for 'variable coefficient'."""
    return 10.0 / (0.05 + 2.0 * np.sum(X * X, axis=-1))


def deform_sine(X, L, amp):
    """Boundary-preserving smooth displacement (the build's definition)."""
    s = (np.sin(2 * np.pi * X[..., 0] / L[0]) * np.sin(2 * np.pi * X[..., 1] / L[1])
         * np.sin(2 * np.pi * X[..., 2] / L[2]))
    out = X.copy()
    for c, sc in enumerate((1.0, -0.8, 0.6)):
        out[..., c] += amp * sc * L[c] * s
    return out


class BrickMesh:
    """n0 x n1 x n2 cubes of side h on [0,n0 h]x[0,n1 h]x[0,n2 h], FE_Q(p), lexicographic numbering.

    local index i + n(j + n k)   (bp5/fe_evaluation_gl.h:139-142)
    cell  index cx + n0 (cy + n1 cz)
    global DoF  I + NX (J + NY K), I = p cx + i ...
    Zero Dirichlet on the whole boundary (bp5/step-64.cu:354-357).
    """

    def __init__(self, p, cells, h=1.0, deform_amp=0.0):
        self.p = p
        self.n = n = p + 1
        self.cells = tuple(int(c) for c in cells)
        n0, n1, n2 = self.cells
        self.h = h
        self.L = (n0 * h, n1 * h, n2 * h)
        self.NX, self.NY, self.NZ = p * n0 + 1, p * n1 + 1, p * n2 + 1
        self.n_dofs = self.NX * self.NY * self.NZ
        self.n_cells = n0 * n1 * n2
        nodes, _ = gll_01(n)
        cx, cy, cz = np.meshgrid(np.arange(n0), np.arange(n1), np.arange(n2), indexing="ij")
        # cell id = cx + n0*(cy + n1*cz): order arrays accordingly
        cid = (cx + n0 * (cy + n1 * cz)).ravel()
        order = np.argsort(cid)
        cx, cy, cz = cx.ravel()[order], cy.ravel()[order], cz.ravel()[order]
        i = np.arange(n)
        I = p * cx[:, None, None, None] + i[None, None, None, :]
        J = p * cy[:, None, None, None] + i[None, None, :, None]
        K = p * cz[:, None, None, None] + i[None, :, None, None]
        gid = I + self.NX * (J + self.NY * K)          # [cell][k][j][i]
        self.l2g = gid.reshape(self.n_cells, n ** 3).astype(np.uint32)
        # coordinates of global DoFs
        gx = (np.arange(self.NX) // p + nodes[np.arange(self.NX) % p]) * h
        gx[-1] = n0 * h
        gy = (np.arange(self.NY) // p + nodes[np.arange(self.NY) % p]) * h
        gy[-1] = n1 * h
        gz = (np.arange(self.NZ) // p + nodes[np.arange(self.NZ) % p]) * h
        gz[-1] = n2 * h
        Z, Y, X = np.meshgrid(gz, gy, gx, indexing="ij")
        coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=-1)
        if deform_amp != 0.0:
            coords = deform_sine(coords, self.L, deform_amp)
        self.coords = coords                            # [n_dofs][3]
        Ig, Jg, Kg = np.meshgrid(np.arange(self.NX), np.arange(self.NY), np.arange(self.NZ), indexing="ij")
        bnd = ((Ig == 0) | (Ig == self.NX - 1) | (Jg == 0) | (Jg == self.NY - 1)
               | (Kg == 0) | (Kg == self.NZ - 1))
        g = (Ig + self.NX * (Jg + self.NY * Kg))[bnd]
        self.constrained = np.sort(g.ravel()).astype(np.uint32)


# ----------------------------------------------------------------------------- hanging nodes (2:1 refinement)
# constraint_mask bits of a cell (include/bp5.h BP5_HANG_*; role of MatrixFree::Data::constraint_mask consumed by
# resolve_hanging_nodes at bp5/fe_evaluation_gl.h:150-151,167-168 -- the bit layout of deal.II's own header is not in the
# reference, this is the library's documented one).  A fine cell whose face normal to direction d lies on a coarser neighbour:
#   FACE_d   the face normal to d is constrained;  SIDE_d  it is the face at xi_d = 1 (else xi_d = 0);
#   HALF_t   for the two directions t tangential to that face: the fine face covers the upper half [1/2,1] of the coarse one.
# local_to_global of the entries ON that face names the COARSE face's DoFs (same orientation); read_dof_values interpolates
# them to the fine face nodes, distribute_local_to_global applies the transpose.
HANG_FACE = (1, 2, 4)
HANG_SIDE = (8, 16, 32)
HANG_HALF = (64, 128, 256)
# general 2:1 meshes (cells at edges / corners of a refined region, re-entrant corners): any subset of FACE bits, plus
#   EDGE_d   the edge along d at the corner (SIDE_e1, SIDE_e2) of the two other directions lies on a coarser cell's edge although
#            neither face through it is constrained; its n entries name the COARSE edge's DoFs.
# SIDE_e = HALF_e = the cell's position (0 / 1) inside its parent along e: SIDE_e locates constrained faces / edges, HALF_d picks
# the half of the coarse entity that the interpolation ALONG d covers.
HANG_EDGE = (512, 1024, 2048)


def hanging_interpolation(p):
    """I[h][a][b] = phi_b(xi_a / 2 + h / 2): values at the fine face's nodes of the coarse 1-D basis, h = lower / upper half."""
    nodes, _ = gll_01(p + 1)
    out = []
    for h in (0, 1):
        N, _ = lagrange_tables(nodes, 0.5 * nodes + 0.5 * h)
        out.append(N)
    return np.stack(out)


def hanging_lines(mask, n):
    """For one cell's mask: per direction d a boolean [k][j][i] array (constant along d) marking the local lines ALONG d whose
    entries hold coarse values and are interpolated along d with I[HALF_d]: the lines of every constrained face that is
    tangential to d, and the constrained edge along d.  None where a direction has no such line."""
    last = n - 1
    ax = np.arange(n)
    coord = [ax[None, None, :], ax[None, :, None], ax[:, None, None]]           # direction e lives on array axis 2 - e of [k][j][i]
    side = [last if (mask & HANG_SIDE[e]) else 0 for e in range(3)]
    out = []
    for d in range(3):
        e1, e2 = [e for e in range(3) if e != d]
        on = np.zeros((n, n, n), bool)
        for e in (e1, e2):
            if mask & HANG_FACE[e]:
                on |= np.broadcast_to(coord[e] == side[e], (n, n, n))
        if mask & HANG_EDGE[d]:
            on |= np.broadcast_to((coord[e1] == side[e1]) & (coord[e2] == side[e2]), (n, n, n))
        out.append(on if on.any() else None)
    return out


def resolve_hanging(mesh, u, c0=0, c1=None, transpose=False):
    """In place on u[c0:c1] viewed as [cell][k][j][i][...]: coarse values on constrained faces / edges -> the cell's own nodal
    values, one 1-D interpolation per direction on the flagged lines (transpose: the adjoint, used before the scatter; the
    per-direction operators commute).  Cells without mask bits are untouched."""
    mask = getattr(mesh, "constraint_mask", None)
    if mask is None:
        return u
    c1 = mesh.n_cells if c1 is None else c1
    m = np.asarray(mask[c0:c1])
    if not m.any():
        return u
    I = hanging_interpolation(mesh.p)
    n = mesh.n
    extra = (None,) * (u.ndim - 4)
    for mval in np.unique(m[m != 0]):
        ids = np.nonzero(m == mval)[0]
        blk = u[ids]
        for d, on in enumerate(hanging_lines(int(mval), n)):
            if on is None:
                continue
            M = I[1 if (mval & HANG_HALF[d]) else 0]
            axis = 1 + (2 - d)
            new = np.moveaxis(np.tensordot(M.T if transpose else M, blk, axes=([1], [axis])), 0, axis)
            blk = np.where(on[(None,) + (slice(None),) * 3 + extra], new, blk)
        u[ids] = blk
    return u


class HangingBrickMesh:
    """Two refinement levels with one planar 2:1 interface: ncx x ny x nz coarse cubes of side H on x in [0, ncx H], then
    nfx x 2ny x 2nz cubes of side H/2 up to x = ncx H + nfx H/2.  FE_Q(p); the interface plane carries the COARSE face DoFs only;
    the fine cells next to it are flagged (FACE_X, xi = 0 side, HALF_Y / HALF_Z) and their i = 0 entries name the coarse face's
    DoFs.  Zero Dirichlet on the whole boundary.  Cells: coarse first, then fine, x fastest."""

    def __init__(self, p, ncx, ny, nz, nfx, H=1.0, deform_amp=0.0):
        self.p, self.n = p, p + 1
        n = self.n
        nodes, _ = gll_01(n)
        NXc, NYc, NZc = p * ncx + 1, p * ny + 1, p * nz + 1
        NXf, NYf, NZf = p * nfx, 2 * p * ny + 1, 2 * p * nz + 1       # fine lattice WITHOUT the interface plane
        n_coarse = NXc * NYc * NZc
        self.n_dofs = n_coarse + NXf * NYf * NZf
        self.L = (ncx * H + nfx * H / 2, ny * H, nz * H)

        def line(ncell, h):
            N1 = p * ncell + 1
            g = (np.arange(N1) // p + nodes[np.arange(N1) % p]) * h
            g[-1] = ncell * h
            return g
        gxc, gyc, gzc = line(ncx, H), line(ny, H), line(nz, H)
        gxf, gyf, gzf = ncx * H + line(nfx, H / 2)[1:], line(2 * ny, H / 2), line(2 * nz, H / 2)
        Zc, Yc, Xc = np.meshgrid(gzc, gyc, gxc, indexing="ij")
        Zf, Yf, Xf = np.meshgrid(gzf, gyf, gxf, indexing="ij")
        coords = np.concatenate([np.stack([Xc.ravel(), Yc.ravel(), Zc.ravel()], -1), np.stack([Xf.ravel(), Yf.ravel(), Zf.ravel()], -1)])
        i = np.arange(n)
        cells, masks = [], []
        for cz in range(nz):
            for cy in range(ny):
                for cx in range(ncx):
                    Ig = p * cx + i[None, None, :]
                    Jg = p * cy + i[None, :, None]
                    Kg = p * cz + i[:, None, None]
                    cells.append((Ig + NXc * (Jg + NYc * Kg)).ravel())
                    masks.append(0)
        for cz in range(2 * nz):
            for cy in range(2 * ny):
                for cx in range(nfx):
                    Ig = p * cx + i[None, None, :] - 1                 # fine lattice index (-1: the interface plane)
                    Jg = p * cy + i[None, :, None]
                    Kg = p * cz + i[:, None, None]
                    gid = n_coarse + Ig + NXf * (Jg + NYf * Kg) + 0 * (Jg + Kg)
                    msk = 0
                    if cx == 0:                                        # i = 0 entries: DoFs of the coarse face, coarse (j, k) order
                        Jc = p * (cy // 2) + i[None, :, None]
                        Kc = p * (cz // 2) + i[:, None, None]
                        face = (NXc - 1) + NXc * (Jc + NYc * Kc) + 0 * i[None, None, :]
                        gid = np.where(i[None, None, :] == 0, face, gid)
                        msk = HANG_FACE[0] | (HANG_HALF[1] if cy % 2 else 0) | (HANG_HALF[2] if cz % 2 else 0)
                    cells.append(gid.ravel())
                    masks.append(msk)
        self.l2g = np.asarray(cells, dtype=np.uint32)
        self.constraint_mask = np.asarray(masks, dtype=np.uint32)
        self.n_cells = self.l2g.shape[0]
        self.n_coarse_cells = ncx * ny * nz
        if deform_amp != 0.0:
            coords = deform_sine(coords, self.L, deform_amp)
        self.coords = coords
        x, y, z = coords[:, 0], coords[:, 1], coords[:, 2]
        if deform_amp == 0.0:
            eps = 1e-12
            bnd = (x < eps) | (x > self.L[0] - eps) | (y < eps) | (y > self.L[1] - eps) | (z < eps) | (z > self.L[2] - eps)
        else:                                                          # the sine map keeps boundary points on the boundary
            c0 = np.concatenate([np.stack([Xc.ravel(), Yc.ravel(), Zc.ravel()], -1), np.stack([Xf.ravel(), Yf.ravel(), Zf.ravel()], -1)])
            eps = 1e-12
            bnd = np.zeros(self.n_dofs, bool)
            for e in range(3):
                bnd |= (c0[:, e] < eps) | (c0[:, e] > self.L[e] - eps)
        self.constrained = np.nonzero(bnd)[0].astype(np.uint32)

    def cell_node_coords(self):
        """positions of every cell's OWN nodes through the hanging-node interpolation of the coordinate field"""
        n = self.n
        X = self.coords[self.l2g.astype(np.int64)].reshape(self.n_cells, n, n, n, 3).copy()
        return resolve_hanging(self, X)


class RefinedBrickMesh:
    """General two-level 2:1 mesh: a brick of Nx x Ny x Nz cubes of side H, the cubes flagged in `refine` ([z][y][x] bool) split
    into 8.  Fine cells on the rim of the refined region carry constrained faces (up to three: corners of the region) and,
    at re-entrant corners, constrained edges (HANG_EDGE).  Global DoFs: every node of an unrefined cube and every fine node
    that does not lie on a coarser cell; constrained faces / edges name the coarse neighbour's DoFs.  Zero Dirichlet on
    the whole boundary.  Cells: unrefined cubes first (x fastest), then the children (parent x fastest, child x fastest)."""

    def __init__(self, p, coarse, refine, H=1.0, deform_amp=0.0):
        self.p, self.n = p, p + 1
        n, last = self.n, p
        nodes, _ = gll_01(n)
        Nx, Ny, Nz = coarse
        refine = np.asarray(refine, bool).reshape(Nz, Ny, Nx)
        self.L = (Nx * H, Ny * H, Nz * H)
        keys, coords = {}, []

        def node_ids(origin, size):
            X = np.stack(np.meshgrid(origin[2] + size * nodes, origin[1] + size * nodes, origin[0] + size * nodes, indexing="ij")[::-1], -1)
            ids = np.empty((n, n, n), np.int64)
            for idx in np.ndindex(n, n, n):
                k = tuple(np.round(X[idx] / H * 2 ** 30).astype(np.int64))
                if k not in keys:
                    keys[k] = len(coords)
                    coords.append(X[idx])
                ids[idx] = keys[k]
            return ids, X

        inside = lambda c: 0 <= c[0] < Nx and 0 <= c[1] < Ny and 0 <= c[2] < Nz
        is_coarse = lambda c: inside(c) and not refine[c[2], c[1], c[0]]
        is_fine = lambda c: inside(c) and refine[c[2], c[1], c[0]]
        ax = np.arange(n)
        coord = [ax[None, None, :], ax[None, :, None], ax[:, None, None]]
        cells, masks, coarse_l2g = [], [], {}
        for cz in range(Nz):
            for cy in range(Ny):
                for cx in range(Nx):
                    if not refine[cz, cy, cx]:
                        ids, _ = node_ids((cx * H, cy * H, cz * H), H)
                        coarse_l2g[(cx, cy, cz)] = ids
                        cells.append(ids.ravel())
                        masks.append(0)
        self.n_coarse_cells = len(cells)
        unit = np.eye(3, dtype=int)
        fine_nodes = []                                        # (cell index, coordinates, replaced) for the conformity audit below
        for cz in range(Nz):
            for cy in range(Ny):
                for cx in range(Nx):
                    if not refine[cz, cy, cx]:
                        continue
                    P = np.array([cx, cy, cz])
                    for ch in np.ndindex(2, 2, 2):
                        c = ch[::-1]                            # child position (x, y, z)
                        sgn = [1 if c[e] else -1 for e in range(3)]
                        msk = 0
                        replaced = np.zeros((n, n, n), bool)
                        src = np.full((n, n, n), -1, np.int64)
                        for e in range(3):
                            Q = tuple(P + sgn[e] * unit[e])
                            if is_coarse(Q):
                                msk |= HANG_FACE[e]
                                sel = np.broadcast_to(coord[e] == c[e] * last, (n, n, n))
                                opp = np.broadcast_to(coord[e] == (1 - c[e]) * last, (n, n, n))
                                src[sel] = coarse_l2g[Q][opp]
                                replaced |= sel
                        for d in range(3):
                            e1, e2 = [e for e in range(3) if e != d]
                            Q1, Q2 = tuple(P + sgn[e1] * unit[e1]), tuple(P + sgn[e2] * unit[e2])
                            Qd = tuple(P + sgn[e1] * unit[e1] + sgn[e2] * unit[e2])
                            if is_coarse(Qd) and is_fine(Q1) and is_fine(Q2):
                                msk |= HANG_EDGE[d]
                                sel = np.broadcast_to((coord[e1] == c[e1] * last) & (coord[e2] == c[e2] * last), (n, n, n))
                                opp = np.broadcast_to((coord[e1] == (1 - c[e1]) * last) & (coord[e2] == (1 - c[e2]) * last), (n, n, n))
                                src[sel] = coarse_l2g[Qd][opp]
                                replaced |= sel
                        if msk:
                            for e in range(3):
                                if c[e]:
                                    msk |= HANG_SIDE[e] | HANG_HALF[e]
                        origin = ((cx + c[0] / 2) * H, (cy + c[1] / 2) * H, (cz + c[2] / 2) * H)
                        X = np.stack(np.meshgrid(origin[2] + H / 2 * nodes, origin[1] + H / 2 * nodes, origin[0] + H / 2 * nodes, indexing="ij")[::-1], -1)
                        ids = src.copy()
                        for idx in np.ndindex(n, n, n):
                            if not replaced[idx]:
                                k = tuple(np.round(X[idx] / H * 2 ** 30).astype(np.int64))
                                if k not in keys:
                                    keys[k] = len(coords)
                                    coords.append(X[idx])
                                ids[idx] = keys[k]
                        fine_nodes.append((X, replaced))
                        cells.append(ids.ravel())
                        masks.append(msk)
        # audit: a fine node that keeps its own DoF must not lie on an unrefined cube unless it IS one of that cube's nodes
        cube_nodes = {c: set(map(int, ids.ravel())) for c, ids in coarse_l2g.items()}
        for (X, replaced), ids in zip(fine_nodes, cells[self.n_coarse_cells:]):
            ids = ids.reshape(n, n, n)
            for idx in np.ndindex(n, n, n):
                if replaced[idx]:
                    continue
                x = X[idx] / H
                for c, own in cube_nodes.items():
                    if all(c[e] - 1e-12 <= x[e] <= c[e] + 1 + 1e-12 for e in range(3)):
                        assert int(ids[idx]) in own, "hanging node without a constraint"
        self.l2g = np.asarray(cells, dtype=np.uint32)
        self.constraint_mask = np.asarray(masks, dtype=np.uint32)
        self.n_cells = self.l2g.shape[0]
        self.n_dofs = len(coords)
        c0 = np.asarray(coords)
        eps = 1e-12
        bnd = np.zeros(self.n_dofs, bool)
        for e in range(3):
            bnd |= (c0[:, e] < eps) | (c0[:, e] > self.L[e] - eps)
        self.constrained = np.nonzero(bnd)[0].astype(np.uint32)
        self.coords = deform_sine(c0, self.L, deform_amp) if deform_amp != 0.0 else c0

    def cell_node_coords(self):
        """positions of every cell's OWN nodes through the hanging-node interpolation of the coordinate field"""
        n = self.n
        X = self.coords[self.l2g.astype(np.int64)].reshape(self.n_cells, n, n, n, 3).copy()
        return resolve_hanging(self, X)


class OctreeBrickMesh:
    """2:1 balanced mesh with ANY number of refinement levels: a brick of Nx x Ny x Nz cubes of side H (level 0); `refine[l]` is a bool array
    on the level-l grid ([z][y][x], 2^l Nz x 2^l Ny x 2^l Nx) flagging the level-l cubes that are split into 8 (a flag only counts where the
    cube exists, i.e. where its parent was split).  Same per-cell description as RefinedBrickMesh (which it reproduces for one level): a leaf
    whose neighbour across a face / edge is a leaf ONE level coarser carries HANG_FACE / HANG_EDGE bits and names the coarse entity's DoFs.
    What the per-cell masks cannot express is refused here: a level difference of more than one across a face or edge (2:1 balance), and
    chained constraints (a coarse entity whose own DoFs hang on a still coarser cell) -- refined regions of successive levels must keep one
    unconstrained cell between their rims.  Cells: level by level; level 0 x fastest, finer levels parent by parent (the children of one
    cube are consecutive: compact 2 x 2 x 2 groups).  Zero Dirichlet on the whole boundary."""

    def __init__(self, p, coarse, refine, H=1.0, deform_amp=0.0):
        self.p, self.n = p, p + 1
        n, last = self.n, p
        nodes, _ = gll_01(n)
        Nx, Ny, Nz = coarse
        self.L = (Nx * H, Ny * H, Nz * H)
        n_levels = len(refine) + 1
        split = []                                              # split[l][(x, y, z)]: the level-l cube exists and is refined
        exists = [{(x, y, z) for z in range(Nz) for y in range(Ny) for x in range(Nx)}]
        for l, r in enumerate(refine):
            r = np.asarray(r, bool).reshape(2 ** l * Nz, 2 ** l * Ny, 2 ** l * Nx)
            sp = {c for c in exists[l] if r[c[2], c[1], c[0]]}
            split.append(sp)
            exists.append({(2 * c[0] + dx, 2 * c[1] + dy, 2 * c[2] + dz) for c in sp for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)})
        split.append(set())
        leaf = lambda l, c: 0 <= l < n_levels and c in exists[l] and c not in split[l]
        fine = lambda l, c: 0 <= l < n_levels and c in split[l]
        keys, coords = {}, []
        ax = np.arange(n)
        coord = [ax[None, None, :], ax[None, :, None], ax[:, None, None]]
        unit = np.eye(3, dtype=int)

        def own_nodes(l, c):
            h = H / 2 ** l
            return np.stack(np.meshgrid(c[2] * h + h * nodes, c[1] * h + h * nodes, c[0] * h + h * nodes, indexing="ij")[::-1], -1)

        def key(x):
            return tuple(np.round(x / H * 2 ** 30).astype(np.int64))

        # pass 1: masks and the entries that name a coarser leaf's DoFs (by (level, cube, local index) until those are numbered)
        leaves = []
        for l in range(n_levels):
            for c in sorted(exists[l] - split[l], key=lambda c: (c[2], c[1], c[0]) if l == 0 else (c[2] // 2, c[1] // 2, c[0] // 2, c[2], c[1], c[0])):
                msk, src = 0, {}
                if l > 0:
                    P = np.array([c[0] // 2, c[1] // 2, c[2] // 2])
                    ch = [c[0] % 2, c[1] % 2, c[2] % 2]
                    sgn = [1 if ch[e] else -1 for e in range(3)]
                    for e in range(3):
                        Q = tuple(P + sgn[e] * unit[e])
                        if leaf(l - 1, Q):
                            msk |= HANG_FACE[e]
                            sel = np.broadcast_to(coord[e] == ch[e] * last, (n, n, n))
                            opp = np.broadcast_to(coord[e] == (1 - ch[e]) * last, (n, n, n))
                            for a, b in zip(np.argwhere(sel), np.argwhere(opp)):
                                src[tuple(a)] = (l - 1, Q, tuple(b))
                        elif not fine(l - 1, Q) and all(0 <= Q[k] < 2 ** (l - 1) * coarse[k] for k in range(3)):
                            raise ValueError("not 2:1 balanced across a face")
                    for d in range(3):
                        e1, e2 = [e for e in range(3) if e != d]
                        Q1, Q2 = tuple(P + sgn[e1] * unit[e1]), tuple(P + sgn[e2] * unit[e2])
                        Qd = tuple(P + sgn[e1] * unit[e1] + sgn[e2] * unit[e2])
                        if leaf(l - 1, Qd) and fine(l - 1, Q1) and fine(l - 1, Q2):
                            msk |= HANG_EDGE[d]
                            sel = np.broadcast_to((coord[e1] == ch[e1] * last) & (coord[e2] == ch[e2] * last), (n, n, n))
                            opp = np.broadcast_to((coord[e1] == (1 - ch[e1]) * last) & (coord[e2] == (1 - ch[e2]) * last), (n, n, n))
                            for a, b in zip(np.argwhere(sel), np.argwhere(opp)):
                                src[tuple(a)] = (l - 1, Qd, tuple(b))
                    if msk:
                        for e in range(3):
                            if ch[e]:
                                msk |= HANG_SIDE[e] | HANG_HALF[e]
                leaves.append((l, c, msk, src))
        # pass 2: number the DoFs level by level (a coarse leaf's entries exist before a finer leaf names them)
        ids_of, cells, masks = {}, [], []
        for l, c, msk, src in leaves:
            X = own_nodes(l, c)
            ids = np.empty((n, n, n), np.int64)
            for idx in np.ndindex(n, n, n):
                if idx in src:
                    continue
                k = key(X[idx])
                if k not in keys:
                    keys[k] = len(coords)
                    coords.append(X[idx])
                ids[idx] = keys[k]
            for idx, (lq, Q, j) in src.items():
                srcs = [lv for lv in leaves if lv[0] == lq and lv[1] == Q][0][3]
                if j in srcs:
                    raise ValueError("chained constraint: the coarse entity's DoFs hang on a still coarser cell")
                ids[idx] = ids_of[(lq, Q)][j]
            ids_of[(l, c)] = ids
            cells.append(ids.ravel())
            masks.append(msk)
        # audit: a node that keeps its own DoF must not lie on a COARSER leaf unless it is one of that leaf's nodes
        own = {(l, c): set(map(int, ids_of[(l, c)].ravel())) for l, c, _, _ in leaves}
        for l, c, msk, src in leaves:
            X, ids = own_nodes(l, c), ids_of[(l, c)]
            for idx in np.ndindex(n, n, n):
                if idx in src:
                    continue
                for (lq, Q), dofs in own.items():
                    if lq >= l:
                        continue
                    hq = H / 2 ** lq
                    if all(Q[e] * hq - 1e-12 <= X[idx][e] <= (Q[e] + 1) * hq + 1e-12 for e in range(3)):
                        assert int(ids[idx]) in dofs, "hanging node without a constraint"
        self.levels = np.asarray([l for l, _, _, _ in leaves])
        self.l2g = np.asarray(cells, dtype=np.uint32)
        self.constraint_mask = np.asarray(masks, dtype=np.uint32)
        self.n_cells = self.l2g.shape[0]
        self.n_coarse_cells = int((self.levels == 0).sum())
        self.n_dofs = len(coords)
        c0 = np.asarray(coords)
        eps = 1e-12
        bnd = np.zeros(self.n_dofs, bool)
        for e in range(3):
            bnd |= (c0[:, e] < eps) | (c0[:, e] > self.L[e] - eps)
        self.constrained = np.nonzero(bnd)[0].astype(np.uint32)
        self.coords = deform_sine(c0, self.L, deform_amp) if deform_amp != 0.0 else c0

    def cell_node_coords(self):
        n = self.n
        X = self.coords[self.l2g.astype(np.int64)].reshape(self.n_cells, n, n, n, 3).copy()
        return resolve_hanging(self, X)


# ----------------------------------------------------------------------------- geometry (A.3)
def _grad_ref(u, N, D):
    """u: [..., k, j, i] nodal values -> (g0,g1,g2) reference gradients at q-points [..., qk, qj, qi].

    g0 = (N (x) N (x) D) u differentiates along x (fastest index), A.4.
    """
    g0 = np.einsum("ck,bj,ai,...kji->...cba", N, N, D, u, optimize=True)
    g1 = np.einsum("ck,bj,ai,...kji->...cba", N, D, N, u, optimize=True)
    g2 = np.einsum("ck,bj,ai,...kji->...cba", D, N, N, u, optimize=True)
    return g0, g1, g2


def _interp(u, N):
    return np.einsum("ck,bj,ai,...kji->...cba", N, N, N, u, optimize=True)


def jacobians(mesh, N, D, w):
    """K = J^{-1} (K[d][e] = d xi_d / d x_e, bp5/fe_evaluation_gl.h:334-343), JxW, q-point coords."""
    n = mesh.n
    Xc = mesh.coords[mesh.l2g.astype(np.int64)].reshape(mesh.n_cells, n, n, n, 3)
    if getattr(mesh, "constraint_mask", None) is not None:   # the coordinate field is interpolated like any FE function
        Xc = resolve_hanging(mesh, Xc.copy())
    J = np.empty((mesh.n_cells, n, n, n, 3, 3))
    for e in range(3):
        g = _grad_ref(Xc[..., e], N, D)
        for d in range(3):
            J[..., e, d] = g[d]                        # J[e][d] = d x_e / d xi_d
    K = np.linalg.inv(J)
    det = np.linalg.det(J)
    W = w[:, None, None] * w[None, :, None] * w[None, None, :]
    JxW = np.abs(det) * W[None]
    xq = np.stack([_interp(Xc[..., e], N) for e in range(3)], axis=-1)
    nq = n ** 3
    return K.reshape(mesh.n_cells, nq, 3, 3), JxW.reshape(mesh.n_cells, nq), xq.reshape(mesh.n_cells, nq, 3)


def merged_metric(mesh, N, D, w, kappa=kappa_none):
    """coef[c][cell][q] = kappa(x_q) * JxW * (K K^T)_c, six planes (bp5/step-64.cu:98-113)."""
    K, JxW, xq = jacobians(mesh, N, D, w)
    G = np.einsum("cqdf,cqef->cqde", K, K)
    s = JxW * kappa(xq)
    coef = np.stack([s * G[:, :, d, e] for (d, e) in PLANE_PAIRS], axis=0)
    return np.ascontiguousarray(coef)


# ----------------------------------------------------------------------------- operator (A.4)
def apply_cells(mesh, coef, N, D, src, chunk=4096, cell_range=None, dst=None):
    """dst = sum_cells P^T B^T S B P src  (no Dirichlet step), bp5/step-64.cu:147-194.
    cell_range = (begin, end) restricts the loop to those cells and dst (if given) is accumulated into: one colour of
    MatrixFree::cell_loop's overlapped schedule (bp5/step-64.cu:241,274)."""
    n = mesh.n
    if dst is None:
        dst = np.zeros(mesh.n_dofs)
    lo, hi = (0, mesh.n_cells) if cell_range is None else cell_range
    for c0 in range(lo, hi, chunk):
        c1 = min(hi, c0 + chunk)
        idx = mesh.l2g[c0:c1].astype(np.int64)
        u = resolve_hanging(mesh, src[idx].reshape(c1 - c0, n, n, n), c0, c1)     # read_dof_values incl. hanging-node fix-up
        g0, g1, g2 = _grad_ref(u, N, D)
        S = coef[:, c0:c1].reshape(6, c1 - c0, n, n, n)
        t0 = S[0] * g0 + S[3] * g1 + S[4] * g2
        t1 = S[3] * g0 + S[1] * g1 + S[5] * g2
        t2 = S[4] * g0 + S[5] * g1 + S[2] * g2
        y = (np.einsum("ck,bj,ai,...cba->...kji", N, N, D, t0, optimize=True)
             + np.einsum("ck,bj,ai,...cba->...kji", N, D, N, t1, optimize=True)
             + np.einsum("ck,bj,ai,...cba->...kji", D, N, N, t2, optimize=True))
        y = resolve_hanging(mesh, y, c0, c1, transpose=True)                      # distribute_local_to_global's counterpart
        np.add.at(dst, idx.ravel(), y.reshape(-1))
    return dst


def apply_cells_unmerged(mesh, K, JxW, N, D, src, kappa_q=None):
    """Same operator through submit_gradient(get_gradient()) (bp5/step-64.cu:190,
    bp5/fe_evaluation_gl.h:318-369): t = JxW * K (K^T ghat)."""
    n = mesh.n
    dst = np.zeros(mesh.n_dofs)
    idx = mesh.l2g.astype(np.int64)
    u = src[idx].reshape(mesh.n_cells, n, n, n)
    g = np.stack([x.reshape(mesh.n_cells, -1) for x in _grad_ref(u, N, D)], axis=-1)  # [c][q][d]
    phys = np.einsum("cqde,cqd->cqe", K, g)           # grad_x = K^T ghat
    s = JxW if kappa_q is None else JxW * kappa_q
    t = np.einsum("cqde,cqe->cqd", K, phys) * s[..., None]
    t = t.reshape(mesh.n_cells, n, n, n, 3)
    y = (np.einsum("ck,bj,ai,...cba->...kji", N, N, D, t[..., 0], optimize=True)
         + np.einsum("ck,bj,ai,...cba->...kji", N, D, N, t[..., 1], optimize=True)
         + np.einsum("ck,bj,ai,...cba->...kji", D, N, N, t[..., 2], optimize=True))
    np.add.at(dst, idx.ravel(), y.reshape(-1))
    return dst


def apply_helmholtz_cells(mesh, N, D, w, src, coefficient=kappa_step64):
    """step-64 Helmholtz cell loop (grad v, grad u) + (v, a(x) u), a = 10/(0.05+2|x|^2)
    (step-64/step-64.cu:99-118,154-160,201-219): evaluate(true,true); per q-point
    submit_value(a * u) and submit_gradient(get_gradient()); integrate(true,true)."""
    n = mesh.n
    K, JxW, xq = jacobians(mesh, N, D, w)
    idx = mesh.l2g.astype(np.int64)
    u = src[idx].reshape(mesh.n_cells, n, n, n)
    uq = _interp(u, N).reshape(mesh.n_cells, -1)
    g = np.stack([x.reshape(mesh.n_cells, -1) for x in _grad_ref(u, N, D)], axis=-1)
    phys = np.einsum("cqde,cqd->cqe", K, g)
    t = (np.einsum("cqde,cqe->cqd", K, phys) * JxW[..., None]).reshape(mesh.n_cells, n, n, n, 3)
    tv = (coefficient(xq) * uq * JxW).reshape(mesh.n_cells, n, n, n)
    y = (np.einsum("ck,bj,ai,...cba->...kji", N, N, N, tv, optimize=True)
         + np.einsum("ck,bj,ai,...cba->...kji", N, N, D, t[..., 0], optimize=True)
         + np.einsum("ck,bj,ai,...cba->...kji", N, D, N, t[..., 1], optimize=True)
         + np.einsum("ck,bj,ai,...cba->...kji", D, N, N, t[..., 2], optimize=True))
    dst = np.zeros(mesh.n_dofs)
    np.add.at(dst, idx.ravel(), y.reshape(-1))
    return dst


def vmult(mesh, coef, N, D, src):
    """PoissonOperator::vmult (bp5/step-64.cu:263-276): cell loop on unmodified src, then
    dst[c] = src[c] on Dirichlet DoFs."""
    dst = apply_cells(mesh, coef, N, D, src)
    c = mesh.constrained.astype(np.int64)
    dst[c] = src[c]
    return dst


def element_matrix(coef_cell, N, D):
    """Dense B^T S B for one cell (A.7-6). coef_cell: [6][nq^3]."""
    n = N.shape[1]
    nq = N.shape[0]
    B = np.zeros((3, nq ** 3, n ** 3))
    B[0] = np.kron(N, np.kron(N, D))
    B[1] = np.kron(N, np.kron(D, N))
    B[2] = np.kron(D, np.kron(N, N))
    A = np.zeros((n ** 3, n ** 3))
    for c, (d, e) in enumerate(PLANE_PAIRS):
        A += B[d].T @ (coef_cell[c][:, None] * B[e])
        if d != e:
            A += B[e].T @ (coef_cell[c][:, None] * B[d])
    return A


def operator_diagonal(mesh, coef, N, D, chunk=4096):
    """diag(A_eff): sum over cells of diag(B^T S B) scattered through l2g, 1 on Dirichlet DoFs
    (A_eff = P A P + (I - P), Appendix A.4).  The Jacobi preconditioner the reference's solver kernels
    already thread through as `diag` (bp5/solver.h:68,100,131,170) is its reciprocal.
    Sum-factorised: diag_ijk = sum_abc S_de(a,b,c) X_de[a,i] Y_de[b,j] Z_de[c,k] with the entrywise
    products N*N, D*D, N*D as 1-D factors."""
    n = mesh.n
    NN, DD, ND = N * N, D * D, N * D
    fac = [(DD, NN, NN, 1.0), (NN, DD, NN, 1.0), (NN, NN, DD, 1.0), (ND, ND, NN, 2.0), (ND, NN, ND, 2.0), (NN, ND, ND, 2.0)]
    diag = np.zeros(mesh.n_dofs)
    for c0 in range(0, mesh.n_cells, chunk):
        c1 = min(mesh.n_cells, c0 + chunk)
        S = coef[:, c0:c1].reshape(6, c1 - c0, n, n, n)
        y = np.zeros((c1 - c0, n, n, n))
        for c, (X, Y, Z, f) in enumerate(fac):
            y += f * np.einsum("ck,bj,ai,...cba->...kji", Z, Y, X, S[c], optimize=True)
        mask = getattr(mesh, "constraint_mask", None)
        if mask is not None and np.asarray(mask[c0:c1]).any():
            # cells with hanging nodes: the entries on constrained faces / edges stand for COARSE DoFs, whose diagonal entry
            # is that of R^T A_e R (R = the cell's hanging-node interpolation): dense element matrix for those cells
            n3 = n ** 3
            for cl in np.nonzero(np.asarray(mask[c0:c1]))[0]:
                ids = np.arange(c0 + cl, c0 + cl + 1)
                R = np.eye(n3).reshape(1, n, n, n, n3).copy()
                sub = type("M", (), dict(p=mesh.p, n=n, n_cells=1, constraint_mask=np.asarray(mask[ids])))
                R = resolve_hanging(sub, R).reshape(n3, n3)
                A = element_matrix(coef[:, c0 + cl].reshape(6, -1), N, D)
                y[cl] = np.einsum("as,ab,bs->s", R, A, R).reshape(n, n, n)
        np.add.at(diag, mesh.l2g[c0:c1].astype(np.int64).ravel(), y.reshape(-1))
    diag[mesh.constrained.astype(np.int64)] = 1.0
    return diag


# ----------------------------------------------------------------------------- RHS (A.6)
def assemble_rhs(mesh, w_unused=None):
    """b_i = sum_cells sum_q phi_i(x_q) JxW(q) with Gauss(p+1) (bp5/step-64.cu:380,401-405),
    constrained rows 0 (bp5/step-64.cu:409-411)."""
    _, _, w, N, D = shape_tables(mesh.p, QUAD_GAUSS)
    _, JxW, _ = jacobians(mesh, N, D, w)
    n = mesh.n
    y = np.einsum("ck,bj,ai,...cba->...kji", N, N, N, JxW.reshape(mesh.n_cells, n, n, n), optimize=True)
    y = resolve_hanging(mesh, y, transpose=True)
    b = np.zeros(mesh.n_dofs)
    np.add.at(b, mesh.l2g.astype(np.int64).ravel(), y.reshape(-1))
    b[mesh.constrained.astype(np.int64)] = 0.0
    return b


# ----------------------------------------------------------------------------- CG (A.5)
def cg_plain(A, b, max_iter, tol=0.0, diag=None, x0=None, dtype=np.float64, history=None):
    """deal.II SolverCG recurrence (bp5/step-64.cu:446-453, Appendix A.5), IterationNumberControl:
    stop at res <= tol or k == max_iter.  Returns (x, iterations, last residual)."""
    b = b.astype(dtype)
    x = np.zeros_like(b) if x0 is None else x0.astype(dtype).copy()
    one = np.ones_like(b) if diag is None else diag.astype(dtype)
    g = -b.copy() if x0 is None else (A(x) - b)
    res = np.sqrt(g @ g)
    if res <= tol:
        return x, 0, res
    h = one * g
    d = -h
    gh = g @ h
    k = 0
    while True:
        k += 1
        h = A(d).astype(dtype)
        alpha = gh / (d @ h)
        x = x + alpha * d
        g = g + alpha * h
        res = np.sqrt(g @ g)
        if history is not None:
            history.append(float(res))
        if res <= tol or k == max_iter:
            return x, k, res
        h = one * g
        gh_old = gh
        gh = g @ h
        beta = gh / gh_old
        d = beta * d - h


def cg_merged(A, b, max_iter, tol=0.0, diag=None, history=None):
    """SolverCGFullMerge (bp5/solver.h:343-542) with the x-update schedule FIXED:
    it==1 update_a0, even it update_a<false>, odd it>=3 update_a1; epilogue as solver.h:510-526."""
    x = np.zeros_like(b)
    D_ = np.ones_like(b) if diag is None else diag
    r = -b.copy()
    res = np.sqrt(r @ r)
    if res <= tol:
        return x, 0, res
    p = np.zeros_like(b)
    v = np.zeros_like(b)
    alpha = beta = alpha_old = beta_old = 0.0
    it = 0
    while True:
        it += 1
        if it == 1:                                     # update_a0, solver.h:48-72
            p = -D_ * r
        elif it % 2 == 0:                               # update_a<false>, solver.h:74-104
            r = r + alpha * v
            p = beta * p - D_ * r
        else:                                           # update_a1, solver.h:106-140
            r_old = r
            x = x + (alpha + alpha_old / beta_old) * p + (alpha_old / beta_old) * (D_ * r_old)
            r = r + alpha * v
            p = beta * p - D_ * r
        v = A(p)
        # update_b: 7 dots, solver.h:142-311
        R = np.array([p @ v, v @ v, r @ v, r @ r, r @ (D_ * v), v @ (D_ * v), r @ (D_ * r)])
        alpha_old, beta_old = alpha, beta
        alpha = R[6] / R[0]                              # solver.h:502
        res = np.sqrt(max(R[3] + 2 * alpha * R[2] + alpha * alpha * R[1], 0.0))  # solver.h:504-505
        if history is not None:
            history.append(float(res))
        if res <= tol or it == max_iter:
            if it % 2 == 1:
                x = x + alpha * p                        # solver.h:511
            else:                                        # update_c, solver.h:315-336,513-525
                x = x + (alpha + alpha_old / beta_old) * p + (alpha_old / beta_old) * (D_ * r)
            return x, it, res
        beta = alpha * (R[4] + alpha * R[5]) / R[6]      # solver.h:533


# ----------------------------------------------------------------------------- post-processing
def l2_norm_solution(mesh, u):
    """||u_h||_L2 by Gauss(p+1) quadrature (bp5/step-64.cu:602-616)."""
    _, _, w, N, D = shape_tables(mesh.p, QUAD_GAUSS)
    _, JxW, _ = jacobians(mesh, N, D, w)
    n = mesh.n
    uq = _interp(resolve_hanging(mesh, u[mesh.l2g.astype(np.int64)].reshape(mesh.n_cells, n, n, n)), N).reshape(mesh.n_cells, -1)
    return float(np.sqrt(np.sum(uq * uq * JxW)))


def deterministic_src(n_dofs, constrained=None, seed=20190930):
    """deterministic pseudo-random f64 in [-1,1], zero on Dirichlet DoFs."""
    rng = np.random.default_rng(seed)
    s = rng.uniform(-1.0, 1.0, size=n_dofs)
    if constrained is not None:
        s[constrained.astype(np.int64)] = 0.0
    return s


class Problem:
    """Convenience bundle: mesh + tables + metric + RHS, mirrors PoissonProblem::setup_system."""

    def __init__(self, p, cells, quadrature=QUAD_GAUSS, h=1.0, deform_amp=0.0, kappa=kappa_none):
        self.mesh = BrickMesh(p, cells, h=h, deform_amp=deform_amp)
        self.nodes, self.pts, self.w, self.N, self.D = shape_tables(p, quadrature)
        self.coef = merged_metric(self.mesh, self.N, self.D, self.w, kappa)
        self.quadrature = quadrature

    def vmult(self, src):
        return vmult(self.mesh, self.coef, self.N, self.D, src)

    def rhs(self):
        return assemble_rhs(self.mesh)
