"""C restatement (oracle/bp5_oracle.c) against the numpy oracle."""
import numpy as np
import pytest

import bp5_oracle as O
import c_oracle as CO


@pytest.mark.parametrize("p", range(1, 9))
@pytest.mark.parametrize("quad", [0, 1])
def test_tables_match(p, quad):
    a = O.shape_tables(p, quad)
    b = CO.tables(p, quad)
    for x, y in zip(a, b):
        assert np.allclose(x, y, atol=1e-13, rtol=1e-13)


@pytest.mark.parametrize("p,quad,amp,kmode", [(1, 0, 0.0, 0), (2, 0, 0.05, 0), (3, 1, 0.05, 1), (4, 0, 0.04, 1),
                                              (6, 0, 0.05, 0), (8, 1, 0.0, 0)])
def test_metric_apply_rhs(p, quad, amp, kmode):
    kappa = O.kappa_step64 if kmode else O.kappa_none
    pr = O.Problem(p, (2, 3, 2), quad, deform_amp=amp, kappa=kappa)
    m = pr.mesh
    cp = CO.CProblem(p, quad, m.l2g, m.coords, m.constrained, kmode)
    assert np.abs(cp.coef - pr.coef).max() < 1e-13 * np.abs(pr.coef).max()
    K, JxW, _ = O.jacobians(m, pr.N, pr.D, pr.w)
    Kc, JxWc = cp.geometry()
    assert np.abs(Kc - K).max() < 1e-12 and np.abs(JxWc - JxW).max() < 1e-14
    s = O.deterministic_src(m.n_dofs)
    ref = O.apply_cells(m, pr.coef, pr.N, pr.D, s)
    assert np.linalg.norm(cp.apply(s) - ref) < 1e-13 * np.linalg.norm(ref)
    s2 = O.deterministic_src(m.n_dofs, seed=3)           # nonzero boundary values
    assert np.linalg.norm(cp.vmult(s2) - pr.vmult(s2)) < 1e-13 * np.linalg.norm(pr.vmult(s2))
    assert np.linalg.norm(cp.rhs() - pr.rhs()) < 1e-13 * np.linalg.norm(pr.rhs())


def test_cg_config1():
    pr = O.Problem(2, (8, 8, 8), O.QUAD_GAUSS)
    m = pr.mesh
    cp = CO.CProblem(2, 0, m.l2g, m.coords, m.constrained)
    b = pr.rhs()
    x, k, res = cp.cg_plain(b, 10)
    xr, kr, resr = O.cg_plain(pr.vmult, b, 10)
    assert k == kr == 10
    assert np.linalg.norm(x - xr) < 1e-13 * np.linalg.norm(xr)
    assert abs(res - resr) < 1e-12 * resr
