"""Generates tests/golden/*.npz from the numpy oracle (oracle/bp5_oracle.py).

The reference repository holds no golden vectors for this path and cannot be built here (it needs
deal.II), so these fixtures are outputs of this repo's own oracle, which is pinned by the
known-answer tests of SURVEY.md Appendix A.7.  Inputs are regenerated from seeds; only expected
outputs are stored.  Run:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bp5_oracle as O  # noqa: E402

CASES_APPLY = [  # (p, quad, cells, amp, kappa_mode)
    (1, 0, (3, 3, 2), 0.05, 0), (2, 0, (3, 2, 2), 0.05, 1), (3, 1, (3, 2, 2), 0.05, 0), (4, 0, (3, 2, 2), 0.04, 1),
    (4, 1, (3, 2, 2), 0.04, 0), (5, 0, (2, 2, 2), 0.05, 0), (6, 0, (2, 2, 1), 0.05, 1), (7, 1, (2, 1, 2), 0.03, 0),
    (8, 0, (2, 1, 1), 0.03, 0),
]


def main():
    out = {}
    for (p, quad, cells, amp, km) in CASES_APPLY:
        pr = O.Problem(p, cells, quad, deform_amp=amp, kappa=O.kappa_step64 if km else O.kappa_none)
        s = O.deterministic_src(pr.mesh.n_dofs, seed=100 + p)
        key = f"vmult_p{p}_q{quad}_{cells[0]}x{cells[1]}x{cells[2]}_a{amp}_k{km}"
        out[key] = pr.vmult(s)
    np.savez_compressed(os.path.join(HERE, "vmult_cases.npz"), **out)
    # BASELINE config 1: p=2, 8^3, 10 CG iterations
    pr = O.Problem(2, (8, 8, 8), O.QUAD_GAUSS)
    b = pr.rhs()
    hist = []
    x, k, res = O.cg_plain(pr.vmult, b, 10, history=hist)
    np.savez_compressed(os.path.join(HERE, "config1_cg.npz"), b=b, x=x, residuals=np.array(hist), iterations=k)
    # p=4 CG, variable coefficient, deformed, 12 iterations (small twin of config 2).  CG amplifies
    # rounding differences; a fixture is only a valid 1e-11 pin while that drift is far below it,
    # so the drift against an extended-precision recurrence is asserted here.
    pr = O.Problem(4, (4, 3, 3), O.QUAD_GAUSS, h=0.25, deform_amp=0.04, kappa=O.kappa_step64)
    b = pr.rhs()
    hist = []
    x, k, res = O.cg_plain(pr.vmult, b, 12, history=hist)
    xl, _, _ = O.cg_plain(lambda v: pr.vmult(v.astype(np.float64)).astype(np.longdouble), b, 12, dtype=np.longdouble)
    drift = np.linalg.norm(x - xl.astype(np.float64)) / np.linalg.norm(x)
    assert drift < 1e-13, drift
    np.savez_compressed(os.path.join(HERE, "p4_kappa_deformed_cg.npz"), b=b, x=x, residuals=np.array(hist), iterations=k)

if __name__ == "__main__":
    main()
