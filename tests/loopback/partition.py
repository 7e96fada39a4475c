"""Test helper: cut an oracle mesh (any object with l2g / coords / constrained / constraint_mask) into rank-local pieces in the
library's convention -- owned DoF range then ghost range, every DoF shared between ranks owned by ONE of the ranks that touch it, the cells
that touch no ghost first (n_interior_cells), per-neighbour send / receive lists in ascending global-DoF order on both sides."""
import numpy as np


def partition(m, cell_rank, n_ranks, owner_rule="lowest"):
    """owner_rule "lowest": a shared DoF belongs to the lowest rank touching it (every neighbour relation is one-way, as in the library's
    slab generator); "alternate": even global ids to the lowest, odd ones to the highest toucher (neighbours that send AND receive)."""
    l2g = m.l2g.astype(np.int64)
    cell_rank = np.asarray(cell_rank)
    lo = np.full(m.n_dofs, n_ranks, np.int64)
    hi = np.full(m.n_dofs, -1, np.int64)
    for r in range(n_ranks):
        ids = np.unique(l2g[cell_rank == r])
        lo[ids] = np.minimum(lo[ids], r)
        hi[ids] = np.maximum(hi[ids], r)
    toucher = lo if owner_rule == "lowest" else np.where(np.arange(m.n_dofs) % 2 == 0, lo, hi)   # the owner of each DoF
    con = np.zeros(m.n_dofs, bool)
    con[m.constrained.astype(np.int64)] = True
    mask = getattr(m, "constraint_mask", None)
    pieces = []
    for r in range(n_ranks):
        cells = np.nonzero(cell_rank == r)[0]
        used = np.unique(l2g[cells])
        owned = np.nonzero(toucher == r)[0]
        ghosts = used[toucher[used] != r]
        ghosts = ghosts[np.lexsort((ghosts, toucher[ghosts]))]    # grouped by owner, ascending global id inside a group
        loc = np.full(m.n_dofs, -1, np.int64)
        loc[owned] = np.arange(owned.size)
        loc[ghosts] = owned.size + np.arange(ghosts.size)
        touches_ghost = (loc[l2g[cells]] >= owned.size).any(axis=1)
        order = np.concatenate([cells[~touches_ghost], cells[touches_ghost]])
        gids = np.concatenate([owned, ghosts])
        neighbors, send_idx, send_off, recv_off = [], [], [0], [0]
        for q in range(n_ranks):
            if q == r:
                continue
            used_q = np.unique(l2g[cell_rank == q])
            send = used_q[toucher[used_q] == r]                   # my DoFs that q's cells reference (ascending global id = q's ghost order)
            recv = ghosts[toucher[ghosts] == q]
            if send.size or recv.size:
                neighbors.append(q)
                send_idx.append(loc[send])
                send_off.append(send_off[-1] + send.size)
                recv_off.append(recv_off[-1] + recv.size)
        pieces.append(dict(
            n_cells=order.size, n_interior_cells=int((~touches_ghost).sum()), n_owned=owned.size, n_ghost=ghosts.size, n_global_dofs=m.n_dofs,
            l2g=loc[l2g[order]].astype(np.uint32), coords=np.ascontiguousarray(m.coords[gids]), global_ids=gids.astype(np.uint64),
            constrained=np.nonzero(con[gids])[0].astype(np.uint32),
            constraint_mask=(np.asarray(mask)[order].astype(np.uint32) if mask is not None else np.zeros(order.size, np.uint32)),
            n_neighbors=len(neighbors), neighbor_rank=np.asarray(neighbors, np.int32), send_offsets=np.asarray(send_off, np.uint32),
            send_indices=(np.concatenate(send_idx) if send_idx else np.zeros(0)).astype(np.uint32), recv_offsets=np.asarray(recv_off, np.uint32)))
    return pieces
