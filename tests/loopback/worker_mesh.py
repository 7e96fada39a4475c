"""One rank of an N-rank run ON ONE GPU for a mesh handed over as flat arrays (tests/loopback/partition.py; test infrastructure, loopback
build of the library -- see worker.py): here the pieces of a 2:1 refined mesh with hanging-node masks.

  python tests/loopback/worker_mesh.py RANK WORLD PORT DIR P ITERS      (reads DIR/mesh<RANK>.npz, writes DIR/rank<RANK>.npz)
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    rank, world, port, out, p, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
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
        z = np.load(os.path.join(out, f"mesh{rank}.npz"))
        mesh = SimpleNamespace(degree=p, n=p + 1, n_local=int(z["n_owned"]) + int(z["n_ghost"]), cell_block_offsets=None, rank=rank, n_ranks=world,
                               **{k: (int(z[k]) if z[k].ndim == 0 else z[k]) for k in z.files})
        op = pkg.PoissonOperator(mesh, pkg.QUAD_GAUSS, pkg.COEF_STEP64, comm=comm)
        assert op.distributed
        no = mesh.n_owned
        gid = mesh.global_ids[:no].astype(np.int64)
        res = {"gid": mesh.global_ids[:no], "variant": np.asarray(op.mf_data.get_apply_variant())}
        b = op.assemble_rhs()
        res["b"] = b[:no].cpu().numpy()
        s_lex = O.deterministic_src(int(mesh.n_global_dofs), seed=23)
        src = op.initialize_dof_vector()
        src[:no] = torch.from_numpy(s_lex[gid]).cuda()
        for mode in (0, 1):
            op.mf_data.set_overlap(mode)
            dst = op.initialize_dof_vector()
            dst.fill_(float("nan"))
            op.vmult(dst, src.clone())
            res[f"A{mode}"] = dst[:no].cpu().numpy()
        op.mf_data.set_overlap(2)
        norms = []
        for name, Solver in (("plain", pkg.SolverCG), ("merged", pkg.SolverCGFullMerge)):
            x = op.initialize_dof_vector()
            ctl = pkg.IterationNumberControl(iters, 0.0)
            Solver(ctl).solve(op, x, b, pkg.DiagonalMatrix())
            res["x_" + name] = x[:no].cpu().numpy()
            norms.append(ctl.last_value())
        res["norms"] = np.asarray(norms)
        inv_diag = op.compute_diagonal(invert=True)
        res["inv_diag"] = inv_diag[:no].cpu().numpy()
        xj = op.initialize_dof_vector()
        pkg.SolverCGFullMerge(pkg.IterationNumberControl(iters, 0.0)).solve(op, xj, b, pkg.DiagonalMatrix(inv_diag))
        res["x_jacobi"] = xj[:no].cpu().numpy()
        res["l2"] = np.asarray(op.l2_norm_solution(x))
        np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
        op.mf_data.synchronize()
        op.mf_data.close()
        comm.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
