"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/bp5.h declares,
its host-only entry points agree with the oracle, and compute fails loudly without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import bp5_oracle as O
import bp5_pkg

pkg = bp5_pkg.load()


def test_library_exports_every_header_symbol():
    L = pkg.lib()
    assert len(pkg.HEADER_SYMBOLS) >= 40
    for s in pkg.HEADER_SYMBOLS:
        assert hasattr(L, s), s
    # and the binding declares a prototype for each of them
    assert set(pkg.HEADER_SYMBOLS) == set(L._protos)


@pytest.mark.parametrize("p", range(1, 9))
@pytest.mark.parametrize("quad", [0, 1])
def test_shape_tables_match_oracle(p, quad):
    a = pkg.shape_tables(p, quad)
    b = O.shape_tables(p, quad)
    for x, y in zip(a, b):
        assert np.allclose(x, y, atol=1e-13, rtol=1e-13)
    # bitwise (anti)symmetry the kernels rely on
    N, D = a[3], a[4]
    assert np.array_equal(N, N[::-1, ::-1]) and np.array_equal(D, -D[::-1, ::-1])


def test_shape_tables_rejects_bad_degree():
    from deal_and_ceed_on_gpu_amd import _lib
    assert pkg.lib().bp5_shape_tables(0, 0, None, None, None, None, None) == 1
    assert pkg.lib().bp5_shape_tables(9, 0, None, None, None, None, None) == 1
    assert pkg.lib().bp5_shape_tables(3, 7, None, None, None, None, None) == 1
    assert b"quadrature" in pkg.lib().bp5_last_error()


@pytest.mark.parametrize("p,cells,amp", [(1, (2, 3, 2), 0.0), (2, (8, 8, 8), 0.0), (4, (3, 2, 4), 0.05), (7, (2, 2, 2), 0.03)])
def test_mesh_matches_oracle_mesh(p, cells, amp):
    m = pkg.BrickMesh(p, cells, h=0.5, deform_amp=amp)
    o = O.BrickMesh(p, cells, h=0.5, deform_amp=amp)
    assert m.n_cells == o.n_cells and m.n_owned == o.n_dofs and m.n_ghost == 0
    assert m.n_interior_cells == m.n_cells
    assert np.array_equal(m.l2g, o.l2g)
    assert np.array_equal(m.constrained, o.constrained)
    assert np.array_equal(m.global_ids, np.arange(o.n_dofs, dtype=np.uint64))
    assert np.abs(m.coords - o.coords).max() < 1e-14


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_mesh_partition_consistency(n_ranks):
    p, cells = 3, (2, 3, 5)
    o = O.BrickMesh(p, cells, deform_amp=0.04)
    owned = []
    total_cells = 0
    for r in range(n_ranks):
        m = pkg.BrickMesh(p, cells, deform_amp=0.04, rank=r, n_ranks=n_ranks)
        total_cells += m.n_cells
        g = m.global_ids.astype(np.int64)
        owned.append(g[:m.n_owned])
        # coordinates / constraints agree with the global mesh
        assert np.abs(m.coords - o.coords[g]).max() < 1e-14
        assert set(g[m.constrained.astype(np.int64)]) == set(o.constrained.astype(np.int64)) & set(g)
        # l2g maps onto the global l2g of some cell
        cell_sets = {tuple(row) for row in o.l2g.astype(np.int64)}
        for row in m.l2g.astype(np.int64):
            assert tuple(g[row]) in cell_sets
        # interior cells never touch ghosts; the others do
        touches = (m.l2g >= m.n_owned).any(axis=1)
        assert not touches[:m.n_interior_cells].any() and touches[m.n_interior_cells:].all()
        # halo plan shape
        assert m.n_neighbors == (1 if r in (0, n_ranks - 1) else 2)
        assert int(m.recv_offsets[-1]) == m.n_ghost
        assert (m.send_indices < m.n_owned).all()
    allowned = np.concatenate(owned)
    assert len(allowned) == o.n_dofs and len(np.unique(allowned)) == o.n_dofs
    assert total_cells == o.n_cells


def test_compute_fails_loudly_without_gpu():
    """No CPU fallback: without a device bp5_mf_create must return BP5_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = pkg.BrickMesh(2, (2, 2, 2))
    with pytest.raises(pkg.BP5Error) as e:
        from deal_and_ceed_on_gpu_amd import _lib
        d = _lib.MFDesc()
        d.dim, d.degree, d.quadrature = 3, 2, 0
        d.n_cells, d.n_interior_cells, d.n_owned, d.n_ghost = m.n_cells, m.n_cells, m.n_owned, 0
        l2g = np.ascontiguousarray(m.l2g)
        xyz = np.ascontiguousarray(m.coords)
        cst = np.ascontiguousarray(m.constrained)
        d.local_to_global_host, d.node_coords_host, d.constrained_host = l2g.ctypes.data, xyz.ctypes.data, cst.ctypes.data
        d.n_constrained = cst.size
        h = C.c_void_p()
        _lib.check(pkg.lib().bp5_mf_create(C.byref(d), C.byref(h)))
    assert e.value.status == 3


def test_product_never_imports_oracle():
    """The product path must not import, link or call anything under oracle/."""
    root = os.path.join(bp5_pkg.ROOT, "deal-and-ceed-on-gpu_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dp, f)).read()
                assert "bp5_oracle" not in text and "c_oracle" not in text and "orc_" not in text, os.path.join(dp, f)


@pytest.mark.parametrize("p,cells,block,n_ranks", [(4, (8, 8, 8), (4, 4, 4), 1), (2, (7, 6, 5), (4, 4, 2), 1), (3, (5, 4, 6), (2, 2, 2), 2),
                                                   (1, (4, 4, 9), (2, 2, 2), 3)])
def test_block_major_numbering(p, cells, block, n_ranks):
    """dof_numbering = 1: a permutation of the lexicographic numbering in which the DoFs strictly
    inside a cell block are contiguous; everything else (coordinates, constraints, halo plan,
    cell connectivity) is consistent with the lexicographic oracle mesh through global_ids."""
    o = O.BrickMesh(p, cells, deform_amp=0.03)
    cell_sets = {tuple(sorted(row)) for row in o.l2g.astype(np.int64)}
    all_owned = []
    for r in range(n_ranks):
        m = pkg.BrickMesh(p, cells, deform_amp=0.03, rank=r, n_ranks=n_ranks, cell_block=block, dof_numbering=1)
        g = m.global_ids.astype(np.int64)
        assert len(np.unique(g)) == m.n_local                      # numbering is injective
        all_owned.append(g[:m.n_owned])
        assert np.abs(m.coords - o.coords[g]).max() < 1e-14
        assert set(g[m.constrained.astype(np.int64)]) == set(o.constrained.astype(np.int64)) & set(g)
        n3 = (p + 1) ** 3
        for row in m.l2g.astype(np.int64):
            assert tuple(sorted(g[row])) in cell_sets
        # in-cell lexicographic order is preserved: x fastest
        row = m.l2g[0].astype(np.int64)
        G = g[row]
        NX = p * cells[0] + 1
        assert G[1] - G[0] == 1 and G[p + 1] - G[0] == NX
        # block interiors are contiguous index ranges
        off = m.cell_block_offsets
        for b in range(min(len(off) - 1, 6)):
            dofs = np.unique(m.l2g[off[b]:off[b + 1]].ravel())
            others = np.unique(np.concatenate([m.l2g[:off[b]].ravel(), m.l2g[off[b + 1]:].ravel()])) if m.n_cells > off[b + 1] - off[b] else np.zeros(0, np.uint32)
            excl = np.setdiff1d(dofs, others)
            excl = np.sort(excl[excl < m.n_owned]).astype(np.int64)
            if off[b + 1] - off[b] == block[0] * block[1] * block[2] and len(excl):   # a full brick
                runs = np.split(excl, np.where(np.diff(excl) != 1)[0] + 1)
                want = (p * block[0] - 1) * (p * block[1] - 1) * (p * block[2] - 1)
                assert max(len(r_) for r_ in runs) >= want
        assert (m.send_indices < m.n_owned).all() and int(m.recv_offsets[-1]) == m.n_ghost
        if r < n_ranks - 1:                                       # the sent plane is this rank's top DoF plane
            NY, NZ = p * cells[1] + 1, p * cells[2] + 1
            sent = g[m.send_indices.astype(np.int64)]
            assert len(sent) == NX * NY and len(set(sent // (NX * NY))) == 1
    allowned = np.concatenate(all_owned)
    assert len(allowned) == o.n_dofs and len(np.unique(allowned)) == o.n_dofs


@pytest.mark.parametrize("p,cells,block,n_ranks", [(4, (8, 8, 8), (4, 4, 4), 1), (2, (7, 6, 5), (4, 4, 2), 2), (3, (5, 4, 3), (2, 2, 2), 1)])
def test_parity_class_cell_order(p, cells, block, n_ranks):
    """cell_block_order = 1: the same blocks with the same cells as the lexicographic order, but inside a
    block the cells come parity class by parity class, so the cells of one class (which are consecutive)
    share no DoF."""
    for r in range(n_ranks):
        a = pkg.BrickMesh(p, cells, rank=r, n_ranks=n_ranks, cell_block=block, dof_numbering=1)
        b = pkg.BrickMesh(p, cells, rank=r, n_ranks=n_ranks, cell_block=block, dof_numbering=1, cell_block_order=1)
        assert np.array_equal(a.cell_block_offsets, b.cell_block_offsets) and np.array_equal(a.global_ids, b.global_ids)
        off = a.cell_block_offsets
        saw_reorder = False
        for k in range(len(off) - 1):
            ra = {tuple(row) for row in a.l2g[off[k]:off[k + 1]]}
            rb = [tuple(row) for row in b.l2g[off[k]:off[k + 1]]]
            assert ra == set(rb)
            saw_reorder |= [tuple(row) for row in a.l2g[off[k]:off[k + 1]]] != rb
            # walk the block: a new class starts whenever the next cell touches a DoF of the running class
            seen, n_classes = set(), 1
            for row in rb:
                if seen & set(row):
                    n_classes += 1
                    seen = set()
                seen |= set(row)
            assert n_classes <= 8
        assert saw_reorder == (max(block) > 2)                    # a 2x2x2 block in class order IS lexicographic
    with pytest.raises(pkg.BP5Error):
        pkg.BrickMesh(2, (2, 2, 2), cell_block_order=1)            # needs cell_block
    with pytest.raises(pkg.BP5Error):
        pkg.BrickMesh(2, (2, 2, 2), cell_block=(2, 2, 2), cell_block_order=2)


def test_pure_c_consumer_of_the_abi():
    """include/bp5.h is valid C11 and libbp5.so links from a C program (tests/c/abi_smoke.c)."""
    import subprocess
    cdir = os.path.join(bp5_pkg.ROOT, "tests", "c")
    subprocess.check_call(["make", "-s", "-C", cdir])
    r = subprocess.run([os.path.join(cdir, "abi_smoke")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abi_smoke ok" in r.stdout, r.stdout + r.stderr


def test_bench_launches_its_own_ranks_and_strong_scales_by_default():
    """`python bench.py --gpus N` without a launcher starts N ranks itself (child torch.distributed.run, before any GPU
    call) and, by default, splits ONE problem into N z-slabs (BASELINE config 3: strong scaling); `--scaling weak` gives
    every rank a slab of the N = 1 size.  Host-only dry run: no GPU, no process group."""
    import json
    import subprocess
    import sys
    bench = os.path.join(bp5_pkg.ROOT, "bench.py")
    for mode, n_global in (("strong", (4 * 8 + 1) ** 2 * (4 * 10 + 1)), ("weak", (4 * 8 + 1) ** 2 * (4 * 20 + 1))):
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--cells", "8", "8", "10", "--scaling", mode, "--dry-run"],
                           capture_output=True, text=True, timeout=600, cwd=bp5_pkg.ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        parts = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")), key=lambda d: d["rank"])
        assert [d["rank"] for d in parts] == [0, 1] and all(d["world"] == 2 and d["scaling"] == mode for d in parts)
        assert all(d["n_global_dofs"] == n_global for d in parts)
        assert sum(d["n_owned"] for d in parts) == n_global                  # every DoF owned exactly once
        assert sum(d["n_cells"] for d in parts) == 8 * 8 * (10 if mode == "strong" else 20)
        assert parts[0]["neighbors"] == [1] and parts[1]["neighbors"] == [0] and parts[1]["n_ghost"] == 33 * 33


@pytest.mark.parametrize("p,cells,block,kw", [(8, (3, 3, 2), (0, 0, 0), {}), (5, (4, 3, 5), (2, 2, 2), dict(rank=1, n_ranks=2)), (3, (4, 3, 3), (0, 0, 0), {}),
                                            (2, (3, 3, 3), (2, 2, 2), dict(rank=0, n_ranks=2)), (4, (5, 4, 6), (4, 4, 2), dict(rank=2, n_ranks=3))])
def test_mesh_numbering_cell_interiors_first(p, cells, block, kw):
    """bp5_mesh_desc.dof_numbering = 2 (round 4): the owned DoFs strictly inside a cell come first, (p-1)^3 consecutive DoFs per cell in the order the cells are handed
    over; everything else is the same mesh: the same global DoFs per cell entry, the same Dirichlet set, coordinates, ghosts and halo plan as the numbering it renumbers."""
    blocked = all(b > 0 for b in block)
    m2 = pkg.BrickMesh(p, cells, h=0.25, deform_amp=0.03, cell_block=block, dof_numbering=2, cell_block_order=1 if blocked else 0, **kw)
    m1 = pkg.BrickMesh(p, cells, h=0.25, deform_amp=0.03, cell_block=block, dof_numbering=1 if blocked else 0, cell_block_order=1 if blocked else 0, **kw)
    per = (p - 1) ** 3
    inner = np.asarray(m2.l2g).reshape(m2.n_cells, p + 1, p + 1, p + 1)[:, 1:p, 1:p, 1:p].reshape(m2.n_cells, per)
    assert np.array_equal(inner, np.arange(m2.n_cells * per).reshape(m2.n_cells, per))
    g1, g2 = np.asarray(m1.global_ids).astype(np.int64), np.asarray(m2.global_ids).astype(np.int64)
    assert (m1.n_owned, m1.n_ghost, m1.n_cells, m1.n_interior_cells) == (m2.n_owned, m2.n_ghost, m2.n_cells, m2.n_interior_cells)
    assert np.array_equal(g1[np.asarray(m1.l2g).astype(np.int64)], g2[np.asarray(m2.l2g).astype(np.int64)])
    assert np.array_equal(np.sort(g1[:m1.n_owned]), np.sort(g2[:m2.n_owned])) and np.array_equal(g1[m1.n_owned:], g2[m2.n_owned:])
    assert np.array_equal(np.sort(g1[np.asarray(m1.constrained).astype(np.int64)]), np.sort(g2[np.asarray(m2.constrained).astype(np.int64)]))
    assert np.all(np.diff(np.asarray(m2.constrained).astype(np.int64)) > 0)
    c1, c2 = np.asarray(m1.coords).reshape(-1, 3), np.asarray(m2.coords).reshape(-1, 3)
    assert np.array_equal(c1[np.argsort(g1)], c2[np.argsort(g2)])
    assert np.array_equal(g1[np.asarray(m1.send_indices).astype(np.int64)], g2[np.asarray(m2.send_indices).astype(np.int64)])
    with pytest.raises(pkg.BP5Error):
        pkg.BrickMesh(p, cells, dof_numbering=3)


@pytest.mark.parametrize("probe", ["hbm_sweep", "metric_stream", "wait_value_probe"])
def test_probes_compile_for_gfx950(tmp_path, probe):
    """tools/probes/*.hip (the standalone measurements profiles/r3, r4 quote) cross-compile for gfx950: they stay runnable as the toolchain moves."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(bp5_pkg.ROOT, "tools", "probes", probe + ".hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-c", src, "-o", str(tmp_path / (probe + ".o"))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_bench_default_mesh_splits_into_equal_slabs_at_2_4_8_ranks():
    """BASELINE config 3 under strong scaling: the default mesh has 120 cell layers, so that 2, 4 and 8 z-slabs hold the same number of
    cells (116 layers gave one of eight ranks 15 against a mean of 14.5: +3.4 %); owned DoFs differ only by the interface plane the lower rank owns (< 2 %).
    Host-only dry run of all eight ranks."""
    import json
    import subprocess
    import sys
    bench = os.path.join(bp5_pkg.ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--dry-run"], capture_output=True, text=True, timeout=900, cwd=bp5_pkg.ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    parts = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in parts] == list(range(8)) and all(d["cells"] == [116, 116, 120] for d in parts)
    cells = [d["n_cells"] for d in parts]
    assert max(cells) == min(cells) == 116 * 116 * 15
    owned = [d["n_owned"] for d in parts]
    assert sum(owned) == parts[0]["n_global_dofs"] == 465 * 465 * 481 and max(owned) <= 1.02 * sum(owned) / 8
    assert [d["neighbors"] for d in parts] == [[1]] + [[k - 1, k + 1] for k in range(1, 7)] + [[6]]


def test_round2_entry_points_validate_their_arguments_without_a_gpu():
    """The entry points added in round 2 reject NULL handles / callbacks before anything touches a device (status codes, no crash)."""
    L = pkg.lib()
    null = C.c_void_p()
    from deal_and_ceed_on_gpu_amd import _lib
    assert L.bp5_cg_solve_operator(null, _lib.VMULT_FN(0), None, None, None, None, None, None) == 1     # NULL callback
    for fn in (L.bp5_halo_gather_start, L.bp5_halo_gather_finish, L.bp5_halo_scatter_add_start, L.bp5_halo_scatter_add_finish):
        assert fn(null, null) == 1
    assert L.bp5_mf_set_overlap(null, 1) == 1 and L.bp5_mf_set_cg_fusion(null, 1) == 1
    assert b"null" in L.bp5_last_error().lower() or b"bad" in L.bp5_last_error().lower()
    # the hanging-node mask bits are part of the ABI (include/bp5.h BP5_HANG_*): the oracle uses the same values
    assert (O.HANG_FACE, O.HANG_SIDE, O.HANG_HALF, O.HANG_EDGE) == ((1, 2, 4), (8, 16, 32), (64, 128, 256), (512, 1024, 2048))
    text = open(os.path.join(bp5_pkg.ROOT, "include", "bp5.h")).read()
    for name, val in (("BP5_HANG_FACE_X", 1), ("BP5_HANG_FACE_Z", 4), ("BP5_HANG_SIDE_X", 8), ("BP5_HANG_HALF_X", 64), ("BP5_HANG_HALF_Z", 256),
                      ("BP5_HANG_EDGE_X", 512), ("BP5_HANG_EDGE_Z", 2048)):
        assert f"{name} = {val}" in text
