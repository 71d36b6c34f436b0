"""CPU-only: what can be pinned about the Forster / GTSAM comparator restatement (oracle/forster_oracle.c) without
GTSAM (absent from the reference tree and this image -- PARITY UNPINNED):
  * A, B, C of NavState::update are the derivatives of the discrete update in NavState's own local coordinates,
  * the bias Jacobians are the derivatives of the preintegrated means with respect to the bias,
  * means and covariance converge (first order in dt) to the continuous CPI model-1 result, which IS pinned to the
    compiled reference -- in the coordinates the restatement claims (position / velocity errors in the body frame)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from cpi_amd import synth
from oracle import oracle_py as op


def _local(s0, s1):
    """gtsam::NavState::localCoordinates: [Log(R0^T R1), R0^T (t1 - t0), R0^T (v1 - v0)]"""
    R0, R1 = s0[:9].reshape(3, 3), s1[:9].reshape(3, 3)
    return np.concatenate([Rotation.from_matrix(R0.T @ R1).as_rotvec(), R0.T @ (s1[9:12] - s0[9:12]),
                           R0.T @ (s1[12:15] - s0[12:15])])


def _rand_state(rng):
    R = Rotation.from_rotvec(rng.normal(size=3) * 0.7).as_matrix()
    return np.concatenate([R.reshape(-1), rng.normal(size=3), rng.normal(size=3) * 2])


def test_update_jacobians_are_the_derivatives_of_the_discrete_update():
    o = op.oracle()
    rng = np.random.default_rng(5)
    for dt in (0.005, 0.05):
        X = _rand_state(rng)
        acc, om = rng.normal(size=3) * 3 + [0, 0, 9.8], rng.normal(size=3) * 0.8
        Y, A, B, Cm = o.navstate_update(X, acc, om, dt)
        h = 1e-6
        An, Bn, Cn = np.zeros((9, 9)), np.zeros((9, 3)), np.zeros((9, 3))
        for i in range(9):
            d = np.zeros(9); d[i] = h
            Yp = o.navstate_update(o.navstate_retract(X, d), acc, om, dt)[0]
            Ym = o.navstate_update(o.navstate_retract(X, -d), acc, om, dt)[0]
            An[:, i] = (_local(Y, Yp) - _local(Y, Ym)) / (2 * h)
        for i in range(3):
            d = np.zeros(3); d[i] = h
            Bn[:, i] = (_local(Y, o.navstate_update(X, acc + d, om, dt)[0]) -
                        _local(Y, o.navstate_update(X, acc - d, om, dt)[0])) / (2 * h)
            Cn[:, i] = (_local(Y, o.navstate_update(X, acc, om + d, dt)[0]) -
                        _local(Y, o.navstate_update(X, acc, om - d, dt)[0])) / (2 * h)
        assert np.abs(A - An).max() < 2e-8
        assert np.abs(B - Bn).max() < 2e-8
        assert np.abs(Cm - Cn).max() < 2e-8


def _m3(x):
    return x.reshape(-1, 3, 3).transpose(0, 2, 1)   # column-major flat -> [W,3,3]


def test_bias_jacobians_are_the_derivatives_of_the_means():
    kn, lin, _ = synth.make_windows(8, 50, seed=77, edge_cases=False)
    kn, lin = kn.numpy(), lin.numpy()
    prm = op.make_params(3)
    o = op.oracle()
    base = o.run(prm, kn, lin)
    h = 1e-6
    R0 = Rotation.from_quat(base["q"]).as_matrix()     # JPL q of R_k->k+1: scipy reads it as the Hamilton q of deltaRij
    for c in range(6):
        lp, lm = lin.copy(), lin.copy()
        lp[:, c] += h
        lm[:, c] -= h
        p, m = o.run(prm, kn, lp), o.run(prm, kn, lm)
        dalpha = (p["alpha"] - m["alpha"]) / (2 * h)
        dbeta = (p["beta"] - m["beta"]) / (2 * h)
        Rp, Rm = Rotation.from_quat(p["q"]).as_matrix(), Rotation.from_quat(m["q"]).as_matrix()
        # deltaRij(b + d) = deltaRij(b) Exp(delRdelBiasOmega d)
        dth = (Rotation.from_matrix(np.einsum("wji,wjk->wik", R0, Rp)).as_rotvec() -
               Rotation.from_matrix(np.einsum("wji,wjk->wik", R0, Rm)).as_rotvec()) / (2 * h)
        if c < 3:     # gyro bias
            assert np.abs(dalpha - _m3(base["J_a"])[:, :, c]).max() < 1e-7
            assert np.abs(dbeta - _m3(base["J_b"])[:, :, c]).max() < 1e-7
            assert np.abs(dth - (-_m3(base["J_q"]))[:, :, c]).max() < 1e-7
        else:         # accelerometer bias
            assert np.abs(dalpha - _m3(base["H_a"])[:, :, c - 3]).max() < 1e-7
            assert np.abs(dbeta - _m3(base["H_b"])[:, :, c - 3]).max() < 1e-7
            assert np.abs(dth).max() < 1e-9


def _smooth_windows(W, n, dt, seed):
    """IMU sampled from smooth signals, so that refining dt refines the same trajectory."""
    rng = np.random.default_rng(seed)
    t = np.arange(n + 1) * dt
    ph = rng.uniform(0, 6.28, size=(W, 6, 1))
    fr = rng.uniform(0.5, 3.0, size=(W, 6, 1))
    amp = np.concatenate([np.full((W, 3, 1), 0.6), np.full((W, 3, 1), 2.5)], axis=1)
    sig = amp * np.sin(fr * t[None, None, :] + ph)
    sig[:, 5, :] += 9.8
    kn = np.zeros((W, n + 1, 7))
    kn[:, :, 0] = t
    kn[:, :, 1:] = sig.transpose(0, 2, 1)
    return kn, rng.normal(size=(W, 6)) * 0.01


def _to_cpi_coords(out):
    """the comparator's covariance holds position / velocity errors in the END body frame (NavState::retract);
    rotate them into frame k, where CPI's alpha / beta errors live:  T = diag(I, I, deltaRij, I, deltaRij)"""
    R = Rotation.from_quat(out["q"]).as_matrix()        # deltaRij (see above)
    W = R.shape[0]
    T = np.zeros((W, 15, 15))
    for b in (0, 3, 9):
        T[:, b:b + 3, b:b + 3] = np.eye(3)
    T[:, 6:9, 6:9] = R
    T[:, 12:15, 12:15] = R
    P = out["P"].reshape(W, 15, 15).transpose(0, 2, 1)
    return T @ P @ T.transpose(0, 2, 1)


def test_converges_to_the_continuous_model_at_first_order():
    o = op.oracle()
    errs = []
    for n, dt in ((40, 0.005), (400, 0.0005)):
        kn, lin = _smooth_windows(6, n, dt, seed=3)
        f = o.run(op.make_params(3), kn, lin)
        c = o.run(op.make_params(1), kn, lin)
        Pc = c["P"].reshape(-1, 15, 15).transpose(0, 2, 1)
        Pf = _to_cpi_coords(f)
        d = np.sqrt(np.einsum("wii->wi", Pc))
        rel = np.abs(Pf - Pc) / (d[:, :, None] * d[:, None, :])
        errs.append((np.abs(f["alpha"] - c["alpha"]).max(), np.abs(f["beta"] - c["beta"]).max(),
                     np.abs(f["q"] - c["q"]).max(), rel.max()))
        assert np.allclose(f["DT"], c["DT"], atol=1e-12)
    coarse, fine = np.array(errs[0]), np.array(errs[1])
    assert coarse[3] < 0.05 and fine[3] < 0.005          # covariance, normalised by the CPI standard deviations
    assert np.all(fine[[0, 1, 3]] < 0.2 * coarse[[0, 1, 3]])   # ~10x smaller at dt / 10
    assert fine[2] < 1e-9 and coarse[2] < 1e-9           # same piecewise-constant-rate rotation in both models


def test_degenerate_windows():
    o = op.oracle()
    kn, lin = _smooth_windows(3, 10, 0.005, seed=9)
    kn[1, :, 0] = kn[1, 0, 0]            # all dt == 0: skipped
    kn[2, 5:, 0] = kn[2, 5, 0]           # tail of duplicates
    f = o.run(op.make_params(3), kn, lin)
    assert f["DT"][1] == 0 and np.all(f["P"][1] == 0) and np.allclose(f["q"][1], [0, 0, 0, 1])
    g = o.run(op.make_params(3), kn[2:3, :6], lin[2:3])
    assert np.array_equal(g["P"][0], f["P"][2]) and np.array_equal(g["alpha"][0], f["alpha"][2])
    assert np.all(np.isfinite(f["P"]))
    # symmetric PSD
    P = f["P"][0].reshape(15, 15)
    assert np.abs(P - P.T).max() < 1e-18 + 1e-12 * np.abs(P).max()
    assert np.linalg.eigvalsh(0.5 * (P + P.T)).min() > -1e-20
