"""CPU-only: the evaluateError restatement (parity UNPINNED against the reference: its TUs need
GTSAM/Boost).  What we can check: golden regression, finite-difference consistency of H1/H2 with
JPLNavState::retract (JPLNavState.cpp:37-71: left-multiplicative dq (x) q, additive others), and
that the residual vanishes at the predicted state (GraphSolver_IMU.cpp:263-307)."""
import os

import numpy as np
import pytest

from oracle import oracle_py as op


def _gold(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "factor_256.npz")))


@pytest.mark.parametrize("model", [1, 2])
def test_factor_golden_regression(golden_dir, model):
    d = _gold(golden_dir)
    err, H1, H2 = op.oracle().factor(model, d["v%d_rec" % model], d["v%d_xi" % model], d["v%d_xj" % model])
    assert np.array_equal(err, d["v%d_err" % model])
    assert np.array_equal(H1, d["v%d_H1" % model])
    assert np.array_equal(H2, d["v%d_H2" % model])


@pytest.mark.parametrize("model", [1, 2])
def test_factor_zero_at_predicted_state(golden_dir, model):
    d = _gold(golden_dir)
    rec, xi = d["v%d_rec" % model][:32].copy(), d["v%d_xi" % model][:32].copy()
    # put the biases ON the linearisation point and (v2) the orientation on q_K_lin
    xi[:, 4:7] = rec[:, 13:16]      # bg_lin
    xi[:, 10:13] = rec[:, 10:13]    # ba_lin
    if model == 2:
        xi[:, 0:4] = rec[:, 65:69]
    xj = op.oracle().predict(model, rec, xi)
    err, _, _ = op.oracle().factor(model, rec, xi, xj, want_H=False)
    assert np.abs(err).max() < 1e-12


@pytest.mark.parametrize("model", [1, 2])
def test_factor_jacobians_finite_difference(golden_dir, model):
    d = _gold(golden_dir)
    o = op.oracle()
    rec, XI, XJ = d["v%d_rec" % model], d["v%d_xi" % model], d["v%d_xj" % model]
    h = 1e-6
    worst = 0.0
    for k in range(0, 24):
        e0, H1, H2 = o.factor(model, rec[k:k + 1], XI[k:k + 1], XJ[k:k + 1])
        H1 = H1[0].reshape(15, 15).T      # column-major -> [row][col]
        H2 = H2[0].reshape(15, 15).T
        for which, H in ((0, H1), (1, H2)):
            for c in range(15):
                dx = np.zeros(15); dx[c] = h
                xp = o.retract(XI[k] if which == 0 else XJ[k], dx)
                xm = o.retract(XI[k] if which == 0 else XJ[k], -dx)
                if which == 0:
                    ep, _, _ = o.factor(model, rec[k:k + 1], xp[None], XJ[k:k + 1], want_H=False)
                    em, _, _ = o.factor(model, rec[k:k + 1], xm[None], XJ[k:k + 1], want_H=False)
                else:
                    ep, _, _ = o.factor(model, rec[k:k + 1], XI[k:k + 1], xp[None], want_H=False)
                    em, _, _ = o.factor(model, rec[k:k + 1], XI[k:k + 1], xm[None], want_H=False)
                fd = (ep[0] - em[0]) / (2 * h)
                worst = max(worst, np.abs(fd - H[:, c]).max())
    # measured: <= 7e-8 (finite-difference noise) on every block of H1 and H2
    assert worst < 1e-6, worst
