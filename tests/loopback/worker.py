"""One rank of an N-rank run of the library ON ONE GPU (test infrastructure; started by tests/test_gpu_multirank_loopback.py
with BP5_LIB = libbp5_loopback.so, whose nine RCCL calls go through the shared-memory transport of loopback_rccl.cpp).
Everything else is the product: slab mesh + halo plan of rank r, RHS assembly with compress(add), the distributed operator
(every exchange schedule), plain and merged CG with the per-iteration all-reduce, the ghosted L2 norm.  The rank's owned
entries go to rank<r>.npz; the parent test compares the union over ranks with the oracle on the undivided mesh.

  python tests/loopback/worker.py RANK WORLD PORT OUTDIR P NX NY NZ BX BY BZ NUMBERING ITERS VARIANT
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out = sys.argv[4]
    p, nx, ny, nz, bx, by, bz, numbering, iters, variant = (int(a) for a in sys.argv[5:15])
    stop_tol = float(sys.argv[15]) if len(sys.argv) > 15 else 0.0   # > 0: additional solves that stop on this absolute residual tolerance
    assert os.environ.get("BP5_LIB", "").endswith("libbp5_loopback.so"), "this worker must run on the loopback build"
    import torch
    import torch.distributed as dist
    import bp5_oracle as O          # deterministic input vectors only
    import bp5_pkg
    pkg = bp5_pkg.load()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        comm = pkg.Communicator.from_torch_distributed()
        block = (bx, by, bz)
        blocked = numbering == 1 and bx > 2
        mesh = pkg.BrickMesh(p, (nx, ny, nz), deform_amp=0.03, rank=rank, n_ranks=world, cell_block=block, dof_numbering=numbering,
                             cell_block_order=1 if blocked else 0)
        op = pkg.PoissonOperator(mesh, pkg.QUAD_GAUSS, pkg.COEF_STEP64, comm=comm)
        assert op.distributed
        op.mf_data.set_apply_variant(variant)      # 0 = the library's choice (meshes this small: the pencil kernel); 56 = block kernel
        no = mesh.n_owned
        res = {"gid": mesh.global_ids[:no], "variant": np.asarray(op.mf_data.get_apply_variant())}
        b = op.assemble_rhs()
        res["b"] = b[:no].cpu().numpy()
        # operator on a vector with non-zero boundary values, every exchange schedule
        s_lex = O.deterministic_src(int(mesh.n_global_dofs), seed=21)
        src = op.initialize_dof_vector()
        src[:no] = torch.from_numpy(s_lex[mesh.global_ids[:no].astype(np.int64)]).cuda()
        for mode in (0, 1, 2):
            op.mf_data.set_overlap(mode)
            dst = op.initialize_dof_vector()
            dst.fill_(float("nan"))
            s_in = src.clone()
            op.vmult(dst, s_in)
            if mesh.n_ghost:
                assert float(s_in[no:].abs().max()) == 0.0           # ghosts of src zeroed again
            res[f"A{mode}"] = dst[:no].cpu().numpy()
        # solvers: plain CG; merged CG unsplit exchange (dot products fused when the block kernel runs), overlapped (block kernel:
        # boundary-first, dot products still fused; atomic kernels: 3-phase, separate dot products), library default, fusion switched off
        norms = []

        def solve(Solver, overlap, fusion, key):
            op.mf_data.set_overlap(overlap)
            op.mf_data.set_cg_fusion(fusion)
            x = op.initialize_dof_vector()
            ctl = pkg.IterationNumberControl(iters, 0.0)
            Solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
            res["x_" + key] = x[:no].cpu().numpy()
            res["fused_" + key] = np.asarray(bool(ctl.dot_products_fused))
            res["sched_" + key] = np.asarray(int(ctl.exchange_schedule))
            norms.append(ctl.last_value())
            return x

        solve(pkg.SolverCG, 2, True, "plain")
        solve(pkg.SolverCGFullMerge, 0, True, "merged_unsplit")
        solve(pkg.SolverCGFullMerge, 1, True, "merged_overlapped")
        x = solve(pkg.SolverCGFullMerge, 2, True, "merged_default")
        solve(pkg.SolverCGFullMerge, 0, False, "merged_unfused")
        solve(pkg.SolverCGFullMerge, 0, True, "merged_unsplit_again")
        solve(pkg.SolverCGFullMerge, 1, True, "merged_overlapped_again")
        solve(pkg.SolverCG, 1, True, "plain_overlapped")
        op.mf_data.set_tuning("early_gather", 0)   # the ghost gather of p AFTER the update kernel instead of underneath it: the same bits
        solve(pkg.SolverCGFullMerge, 0, True, "merged_unsplit_late_gather")
        solve(pkg.SolverCGFullMerge, 1, True, "merged_overlapped_late_gather")
        op.mf_data.set_tuning("early_gather", 1)
        op.mf_data.set_tuning("combine_signal", 1)  # the default schedule with ONE combine launch (ghost rows first, stream wait-value): the same bits
        solve(pkg.SolverCGFullMerge, 2, True, "merged_default_one_combine_launch")
        op.mf_data.set_tuning("combine_signal", 0)
        op.mf_data.set_tuning("ghost_combine_on_comm", 1)   # the default schedule with the ghost-row combine on the communication stream (round 4): the same bits
        solve(pkg.SolverCGFullMerge, 2, True, "merged_default_ghost_combine_on_comm")
        op.mf_data.set_tuning("ghost_combine_on_comm", 0)
        res["norms"] = np.asarray(norms)
        if stop_tol > 0.0:
            # tolerance stop across ranks: every rank sees the same all-reduced residual, the device-side convergence flag fires on all of them in
            # the same iteration and freezes the iterate (the host only enqueues; check_every = 3 polls now and then)
            # (constant coefficient: the step-64 coefficient of the other solves makes CG wander for hundreds of iterations on these meshes)
            op1 = pkg.PoissonOperator(mesh, pkg.QUAD_GAUSS, pkg.COEF_ONE, comm=comm)
            op1.mf_data.set_apply_variant(variant)
            b1 = op1.assemble_rhs()
            for Solver, overlap, key in ((pkg.SolverCG, 2, "stop_plain"), (pkg.SolverCGFullMerge, 2, "stop_merged_default"),
                                         (pkg.SolverCGFullMerge, 1, "stop_merged_overlapped"), (pkg.SolverCGFullMerge, 0, "stop_merged_unsplit")):
                op1.mf_data.set_overlap(overlap)
                xs_ = op1.initialize_dof_vector()
                ctl_ = pkg.SolverControl(400, stop_tol)
                Solver(ctl_, check_every=3).solve(op1, xs_, b1, pkg.DiagonalMatrix())
                res["x_" + key] = xs_[:no].cpu().numpy()
                res["its_" + key] = np.asarray(int(ctl_.last_step()))
                res["res_" + key] = np.asarray(float(ctl_.last_value()))
            op1.mf_data.synchronize()
            op1.mf_data.close()
        # Jacobi-preconditioned merged CG (diagonal assembled across ranks)
        op.mf_data.set_overlap(2)
        inv_diag = op.compute_diagonal(invert=True)
        res["inv_diag"] = inv_diag[:no].cpu().numpy()
        xj = op.initialize_dof_vector()
        pkg.SolverCGFullMerge(pkg.IterationNumberControl(iters, 0.0)).solve(op, xj, b, pkg.DiagonalMatrix(inv_diag))
        res["x_jacobi"] = xj[:no].cpu().numpy()
        res["l2"] = np.asarray(op.l2_norm_solution(x))       # ghosts of x refreshed inside (bp5_l2_norm_solution)
        # step-64's Helmholtz operator on the native kernel across the ranks (same mesh, same halo plan): operator and merged CG
        hop = pkg.HelmholtzOperator(mesh, pkg.QUAD_GAUSS, pkg.COEF_STEP64, comm=comm)
        hop.mf_data.set_apply_variant(56 if variant == 56 else 0)
        hdst = hop.initialize_dof_vector()
        hdst.fill_(float("nan"))
        hop.vmult(hdst, src.clone())
        res["Ah"] = hdst[:no].cpu().numpy()
        xh = hop.initialize_dof_vector()
        hctl = pkg.IterationNumberControl(iters, 0.0)
        pkg.SolverCGFullMerge(hctl).solve(hop, xh, b, pkg.DiagonalMatrix())
        res["x_helmholtz"] = xh[:no].cpu().numpy()
        res["fused_helmholtz"] = np.asarray(bool(hctl.dot_products_fused))
        hop.mf_data.synchronize()
        hop.mf_data.close()
        np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
        op.mf_data.synchronize()
        op.mf_data.close()
        comm.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
