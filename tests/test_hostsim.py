"""CPU-only: the kernels' arithmetic (cpi_amd/csrc/cpi_math.hpp compiled for the host, driven
lane-by-lane exactly as the HIP kernels drive it) against the golden vectors from the compiled
reference.  This validates the kernel DESIGN where no GPU exists: the order-preserving segment
composition of the mean kernel (any lanes-per-window L), the column-lane covariance recursion with
its transpose exchange, the state-transition columns of model 2 and the factor blocks.
The GPU tests (-m gpu) then validate the actual HIP launch path through the C-ABI."""
import os

import numpy as np
import pytest

from oracle import oracle_py as op
from tests import hostsim_py as hs
from tests.tol import check_pre


def _gold(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _mode_out(d, m):
    key = "m%d_avg%d_stj%d__" % m
    return {k[len(key):]: v for k, v in d.items() if k.startswith(key)}


@pytest.mark.parametrize("fname", ["pre_cfg1.npz", "pre_w48.npz"])
@pytest.mark.parametrize("model,avg,jac,L", [(1, 0, 0, 1), (1, 0, 0, 16), (1, 1, 0, 64), (1, 0, 1, 1), (1, 0, 1, 8),
                                              (1, 1, 1, 32), (2, 0, 0, 1), (2, 1, 0, 1), (2, 0, 1, 1), (2, 1, 1, 1),
                                              (2, 0, 0, 2), (2, 0, 0, 6), (2, 1, 0, 5), (2, 1, 0, 16), (2, 0, 0, 64)])
def test_mean_kernel_math(golden_dir, fname, model, avg, jac, L):
    d = _gold(golden_dir, fname)
    out = op.split_out(hs.mean(model, jac, avg, L, d["knots"], d["lin"], d["q_k_lin"]))
    ref = _mode_out(d, (model, avg, 0 if (model == 2 and jac) else 1))
    check_pre(out, ref, what=("mean", "jac") if jac else ("mean",), v2=(model == 2))


@pytest.mark.parametrize("fname", ["pre_cfg1.npz", "pre_w48.npz"])
@pytest.mark.parametrize("model,avg", [(1, 0), (1, 1), (2, 0), (2, 1)])
def test_cov_kernel_math(golden_dir, fname, model, avg):
    d = _gold(golden_dir, fname)
    out = op.split_out(hs.cov(model, avg, d["knots"], d["lin"], d["q_k_lin"]))
    ref = _mode_out(d, (model, avg, 1))
    check_pre(out, ref, what=("mean", "cov", "jac") if model == 2 else ("mean", "cov"), v2=(model == 2))


@pytest.mark.parametrize("model", [1, 2])
def test_factor_kernel_math(golden_dir, model):
    d = _gold(golden_dir, "factor_256.npz")
    err, H1, H2 = hs.factor(model, d["v%d_rec" % model], d["v%d_xi" % model], d["v%d_xj" % model])
    assert np.abs(err - d["v%d_err" % model]).max() < 1e-12
    assert np.abs(H1 - d["v%d_H1" % model]).max() < 1e-12
    assert np.abs(H2 - d["v%d_H2" % model]).max() < 1e-12
    xj = hs.predict(model, d["v%d_rec" % model], d["v%d_xi" % model])
    assert np.abs(xj - op.oracle().predict(model, d["v%d_rec" % model], d["v%d_xi" % model])).max() < 1e-12


@pytest.mark.parametrize("fname", ["pre_cfg1.npz", "pre_w48.npz"])
def test_forster_kernel_math(golden_dir, fname):
    """The comparator kernel's lane algorithm (sparse F in [theta b_g v b_a p] order, row exchange, per-lane bias
    Jacobian columns) against the dense GTSAM-order restatement (oracle/forster_oracle.c, parity unpinned)."""
    d = _gold(golden_dir, fname)
    out = op.split_out(hs.forster(d["knots"], d["lin"]))
    ref = op.oracle().run(op.make_params(3), d["knots"], d["lin"])
    check_pre(out, ref, what=("mean", "jac", "cov"))
    for k in ("alpha", "beta", "q", "J_q", "J_a", "J_b", "H_a", "H_b"):
        assert np.abs(out[k] - ref[k]).max() < 1e-12, k


@pytest.mark.parametrize("model,avg", [(1, 0), (1, 1), (2, 0), (2, 1)])
def test_nan_separated_windows_kernel_math(model, avg):
    """Non-chained feed_IMU sequences as the facades encode them: NaN-time separator knots carrying zero readings (both
    intervals touching one are skipped).  Kernel logic of the mean (several lane splits), analytic-Jacobian and covariance
    paths on the host, against the reference (compiled, when present) driven by the same knot arrays."""
    from cpi_amd import synth
    W, N = 40, 37
    kn, lin, q = synth.make_windows(W, N, seed=9090 + model + 2 * avg, edge_cases=False)
    kn, lin, q = kn.numpy().copy(), lin.numpy(), q.numpy()
    rng = np.random.default_rng(3)
    for w in range(W):
        pos = rng.choice(np.arange(0, N + 1), size=int(rng.integers(0, 5)), replace=False)
        kn[w, pos, 0] = np.nan
        kn[w, pos, 1:] = 0.0
    lib = op.reference() or op.oracle()
    ref = lib.run(op.make_params(model, avg, 1), kn, lin, q)
    assert np.all(np.isfinite(ref["P"]))
    for L in ((1, 4, 16, 64) if model == 1 else (1, 6)):
        out = op.split_out(hs.mean(model, 0, avg, L, kn, lin, q))
        check_pre(out, ref, what=("mean",), label="hostsim mean L%d" % L, regression=True)
    out = op.split_out(hs.cov(model, avg, kn, lin, q))
    check_pre(out, ref, what=("mean", "cov", "jac") if model == 2 else ("mean", "cov"), v2=(model == 2), regression=True)
    if model == 1:
        out = op.split_out(hs.mean(1, 1, avg, 1, kn, lin, q))
        check_pre(out, ref, what=("mean", "jac"), regression=True)


@pytest.mark.parametrize("model", [1, 2])
def test_hessian_block_algebra_equals_the_dense_definition(model):
    """cpi_factor_hessian_kernel's arithmetic (cpi_math.hpp: hsn -- Lam = R^T R by row blocks, Z = Lam H1 through H1's 13
    non-zero 3x3 blocks, G11 = H1^T Z, G12 / G22 through H2's block diagonal) run lane by lane on the host, against the
    dense definition [R H1, R H2, -R e]^T [R H1, R H2, -R e] built from the column functions of the dense sweep: the block
    table (signs, transposes, the dt Rk block) and the packed indexing are pinned here, on the golden factor cases."""
    import os
    d = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "factor_256.npz")))
    rec, xi, xj = d["v%d_rec" % model], d["v%d_xi" % model], d["v%d_xj" % model]
    F = rec.shape[0]
    rng = np.random.default_rng(7 + model)
    A = rng.standard_normal((F, 15, 15))
    P = A @ A.transpose(0, 2, 1) + 15.0 * np.eye(15)
    R = np.stack([np.linalg.cholesky(np.linalg.inv(P[f])).T for f in range(F)])          # upper triangular, [row][col]
    Rcm = np.ascontiguousarray(R.transpose(0, 2, 1)).reshape(F, 225)                      # column-major
    err, H1, H2 = hs.factor(model, rec, xi, xj)
    H1 = H1.reshape(F, 15, 15).transpose(0, 2, 1); H2 = H2.reshape(F, 15, 15).transpose(0, 2, 1)
    Ab = np.concatenate([R @ H1, R @ H2, -(R @ err[:, :, None])], axis=2)                 # [F, 15, 31]
    M = np.einsum("fki,fkj->fij", Ab, Ab)
    want = np.stack([M[:, i, dd] for dd in range(31) for i in range(dd + 1)], axis=1)
    got = hs.hessian(model, rec, xi, xj, Rcm)
    assert got.shape == (F, 496) and np.all(np.isfinite(got))
    scale = np.abs(want).max(axis=1, keepdims=True)
    assert (np.abs(got - want) / scale).max() < 1e-13
