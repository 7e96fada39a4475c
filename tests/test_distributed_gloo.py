"""world_size-2/3 rehearsal of the multi-GPU path on CPU (gloo).

What runs here is the library's own slab partition + halo plan (bp5_mesh_create_brick) and the
exact exchange sequence of bp5_apply_distributed / bp5_cg_solve (ghost gather started -> first part
of the interior cells -> gather finished -> ghost-touching cells -> scatter-add started -> rest of
the interior cells -> scatter-add finished -> zero ghosts -> Dirichlet copy: the reference's
overlap_communication_computation schedule, bp5/step-64.cu:241,274; dot products over OWNED entries
with one all-reduce), with the oracle standing in for the HIP kernels and gloo for RCCL.  It pins the
partition, ownership and halo-plan semantics the C++ RCCL path relies on; the RCCL calls
themselves are exercised on the GPU box with a 1-rank communicator (test_gpu_parity.py) and by the
driver's multi-GPU bench."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bp5_oracle as O
import bp5_pkg


class _LocalMesh:
    """adapter: a rank's piece of the mesh in the oracle's mesh interface"""

    def __init__(self, m):
        self.p, self.n = m.degree, m.degree + 1
        self.l2g = m.l2g
        self.coords = m.coords
        self.n_cells = m.n_cells
        self.n_dofs = m.n_local
        self.constrained = m.constrained


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _halo_gather_start(m, v):
    """owners send send_indices[...] values; ghosts are received into the ghost range (bp5_halo_gather_start)"""
    reqs, bufs = [], []
    for k in range(m.n_neighbors):
        nb = int(m.neighbor_rank[k])
        s0, s1 = int(m.send_offsets[k]), int(m.send_offsets[k + 1])
        r0, r1 = int(m.recv_offsets[k]), int(m.recv_offsets[k + 1])
        if s1 > s0:
            t = torch.from_numpy(v[m.send_indices[s0:s1].astype(np.int64)].copy())
            bufs.append(t)
            reqs.append(dist.isend(t, nb))
        if r1 > r0:
            t = torch.empty(r1 - r0, dtype=torch.float64)
            bufs.append((t, r0, r1))
            reqs.append(dist.irecv(t, nb))
    return reqs, bufs


def _halo_gather_finish(m, v, pending):
    reqs, bufs = pending
    for q in reqs:
        q.wait()
    for b in bufs:
        if isinstance(b, tuple):
            t, r0, r1 = b
            v[m.n_owned + r0:m.n_owned + r1] = t.numpy()


def _halo_gather(m, v):
    _halo_gather_finish(m, v, _halo_gather_start(m, v))


def _halo_scatter_add_start(m, v):
    """ghost contributions go back to the owner (bp5_halo_scatter_add_start: the ghost entries are final here)"""
    reqs, bufs = [], []
    for k in range(m.n_neighbors):
        nb = int(m.neighbor_rank[k])
        s0, s1 = int(m.send_offsets[k]), int(m.send_offsets[k + 1])
        r0, r1 = int(m.recv_offsets[k]), int(m.recv_offsets[k + 1])
        if r1 > r0:
            t = torch.from_numpy(v[m.n_owned + r0:m.n_owned + r1].copy())
            bufs.append(t)
            reqs.append(dist.isend(t, nb))
        if s1 > s0:
            t = torch.empty(s1 - s0, dtype=torch.float64)
            bufs.append((t, s0, s1))
            reqs.append(dist.irecv(t, nb))
    return reqs, bufs


def _halo_scatter_add_finish(m, v, pending):
    """... and are added (compress(add)); ghosts zeroed"""
    reqs, bufs = pending
    for q in reqs:
        q.wait()
    for b in bufs:
        if isinstance(b, tuple):
            t, s0, s1 = b
            np.add.at(v, m.send_indices[s0:s1].astype(np.int64), t.numpy())
    v[m.n_owned:] = 0.0


def _halo_scatter_add(m, v):
    _halo_scatter_add_finish(m, v, _halo_scatter_add_start(m, v))


def _interior_split(m):
    """== interior_split() of csrc/bp5_device.hip: half of the interior cells, rounded up to a brick boundary"""
    half = m.n_interior_cells // 2
    if m.cell_block_offsets is None:
        return half
    off = np.asarray(m.cell_block_offsets)
    i = int(np.searchsorted(off, half, side="left"))
    return int(off[i]) if i < off.size and off[i] <= m.n_interior_cells else m.n_interior_cells


def _allreduce(x):
    t = torch.tensor(x, dtype=torch.float64).reshape(-1)
    dist.all_reduce(t)
    return t.numpy() if t.numel() > 1 else float(t[0])


def _worker(rank, world, port, p, cells, quad, amp, numbering, block, iters, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = bp5_pkg.load()
        m = pkg.BrickMesh(p, cells, deform_amp=amp, rank=rank, n_ranks=world, cell_block=block, dof_numbering=numbering,
                          cell_block_order=1 if (numbering == 1 and block[0] > 2) else 0)
        lm = _LocalMesh(m)
        _, _, w, N, D = O.shape_tables(p, quad)
        coef = O.merged_metric(lm, N, D, w, O.kappa_step64)
        no, c = m.n_owned, m.constrained.astype(np.int64)

        def vmult(src):                       # == bp5_apply_distributed(zero_dst = 1): apply_overlapped() of csrc/bp5_device.hip
            split, n_int = _interior_split(m), m.n_interior_cells
            pending = _halo_gather_start(m, src)
            dst = O.apply_cells(lm, coef, N, D, src, cell_range=(0, split))                     # under the ghost gather
            _halo_gather_finish(m, src, pending)
            O.apply_cells(lm, coef, N, D, src, cell_range=(n_int, m.n_cells), dst=dst)          # the cells that touch ghosts
            pending = _halo_scatter_add_start(m, dst)
            O.apply_cells(lm, coef, N, D, src, cell_range=(split, n_int), dst=dst)              # under the scatter-add
            _halo_scatter_add_finish(m, dst, pending)
            src[no:] = 0.0
            dst[c] = src[c]
            dst[no:] = 0.0
            return dst

        # rhs: local assembly + compress(add) + constrained rows zero (bp5_assemble_rhs)
        _, _, wg, Ng, Dg = O.shape_tables(p, O.QUAD_GAUSS)
        _, JxW, _ = O.jacobians(lm, Ng, Dg, wg)
        n = p + 1
        y = np.einsum("ck,bj,ai,...cba->...kji", Ng, Ng, Ng, JxW.reshape(m.n_cells, n, n, n))
        b = np.zeros(m.n_local)
        np.add.at(b, m.l2g.astype(np.int64).ravel(), y.ravel())
        _halo_scatter_add(m, b)
        b[c] = 0.0
        # plain CG, dots over owned entries only (Appendix A.5)
        x = np.zeros(m.n_local)
        g = -b.copy()
        d = b.copy()
        gh = _allreduce(g[:no] @ g[:no])
        for _ in range(iters):
            h = vmult(d)
            alpha = gh / _allreduce(d[:no] @ h[:no])
            x[:no] += alpha * d[:no]
            g[:no] += alpha * h[:no]
            gg = _allreduce(g[:no] @ g[:no])
            beta, gh = gg / gh, gg
            d[:no] = beta * d[:no] - g[:no]
        # the merged solver's FUSED dot products across ranks (csrc/bp5_device.hip: solver_vmult with fuse_r, unpack_add_dots_kernel):
        # p.v as the sum over every rank's OWN cells of u_e . (A_e u_e) (no owner bookkeeping), v.v / r.v over owned DoFs formed
        # from the LOCAL sums first and corrected by the owners for the ghost contributions they add (Dirichlet owners keep v = p)
        pv_, rv_ = d.copy(), g.copy()
        _halo_gather(m, pv_)
        n3 = (p + 1) ** 3
        idx = m.l2g.astype(np.int64)
        v_loc = O.apply_cells(lm, coef, N, D, pv_)                       # local sums, ghost entries included
        ye = np.zeros((m.n_cells, n3))
        for c0 in range(m.n_cells):                                      # per-cell energies u_e . (A_e u_e)
            one = np.zeros(m.n_local)
            one[idx[c0]] = pv_[idx[c0]]
            ye[c0] = O.apply_cells(lm, coef, N, D, one, cell_range=(c0, c0 + 1))[idx[c0]]
        energy = float(np.sum(pv_[idx] * ye))
        con = np.zeros(m.n_local, bool)
        con[c] = True
        v_own = np.where(con[:no], pv_[:no], v_loc[:no])                 # write-out: Dirichlet rows store p
        vv_loc, rv_loc = float(v_own @ v_own), float(rv_[:no] @ v_own)
        energy += float(np.sum(pv_[:no][con[:no]] * (pv_[:no][con[:no]] - v_loc[:no][con[:no]])))
        contrib = np.zeros(m.n_local)
        contrib[no:] = v_loc[no:]
        _halo_scatter_add(m, contrib)                                    # owners receive the ghost contributions
        add = np.where(con[:no], 0.0, contrib[:no])
        vv_corr = float(np.sum((v_own + add) ** 2 - v_own ** 2))
        rv_corr = float(rv_[:no] @ add)
        fused = _allreduce([energy, vv_loc + vv_corr, rv_loc + rv_corr])
        pv_[no:] = 0.0
        h_ref = vmult(d.copy())                                          # the assembled, exchanged product
        plain = _allreduce([float(d[:no] @ h_ref[:no]), float(h_ref[:no] @ h_ref[:no]), float(g[:no] @ h_ref[:no])])
        assert np.allclose(fused, plain, rtol=1e-12, atol=1e-14 * abs(plain[1])), (fused, plain)
        # one more vmult of a deterministic vector with non-zero boundary values
        s_lex = O.deterministic_src(int(m.n_global_dofs), seed=21)
        src = np.zeros(m.n_local)
        src[:no] = s_lex[m.global_ids[:no].astype(np.int64)]
        Asrc = vmult(src)
        np.savez(os.path.join(out, f"rank{rank}.npz"), gid=m.global_ids[:no], x=x[:no], b=b[:no], A=Asrc[:no])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,p,cells,numbering,block", [(2, 2, (3, 3, 4), 0, (0, 0, 0)), (2, 3, (4, 3, 5), 1, (2, 2, 2)),
                                                           (3, 1, (3, 2, 7), 0, (2, 2, 2)),
                                                           (2, 4, (5, 4, 6), 1, (4, 4, 4)),    # the bench's ordering (parity-class bricks)
                                                           (8, 4, (4, 4, 16), 1, (4, 4, 2)),   # EIGHT ranks, equal slabs of one brick layer: first, six middle, last rank (BASELINE config 3's split)
                                                           (8, 2, (3, 3, 11), 0, (0, 0, 0))])  # eight ranks, ragged slabs (1 or 2 cell layers)
def test_distributed_cg_matches_single_domain(tmp_path, world, p, cells, numbering, block):
    quad, amp, iters = O.QUAD_GAUSS, 0.03, 8
    port = _free_port()
    mp.spawn(_worker, args=(world, port, p, cells, quad, amp, numbering, block, iters, str(tmp_path)), nprocs=world, join=True)
    pr = O.Problem(p, cells, quad, deform_amp=amp, kappa=O.kappa_step64)
    b_ref = pr.rhs()
    x_ref, _, _ = O.cg_plain(pr.vmult, b_ref, iters)
    A_ref = pr.vmult(O.deterministic_src(pr.mesh.n_dofs, seed=21))
    x, b, A = (np.full(pr.mesh.n_dofs, np.nan) for _ in range(3))
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        gid = z["gid"].astype(np.int64)
        assert np.isnan(x[gid]).all()                   # every DoF owned exactly once
        x[gid], b[gid], A[gid] = z["x"], z["b"], z["A"]
    assert not np.isnan(x).any()
    assert np.linalg.norm(b - b_ref) < 1e-13 * np.linalg.norm(b_ref)
    assert np.linalg.norm(A - A_ref) < 1e-13 * np.linalg.norm(A_ref)
    assert np.linalg.norm(x - x_ref) < 1e-11 * np.linalg.norm(x_ref)
