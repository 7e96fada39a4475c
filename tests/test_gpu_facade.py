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
