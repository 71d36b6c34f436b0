"""CPU-only: pin the C restatement (oracle/) against the golden vectors the compiled reference
produced (tests/golden/, made by oracle/gen_golden.py) and, when oracle/_ref is present, against
the compiled reference itself on fresh seeded inputs."""
import os

import numpy as np
import pytest

from cpi_amd import synth
from oracle import oracle_py as op
from tests.tol import check_pre, cov_rel_err

MODES = [(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 1), (2, 0, 0), (2, 1, 0)]


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _mode_out(d, m):
    key = "m%d_avg%d_stj%d__" % m
    return {k[len(key):]: v for k, v in d.items() if k.startswith(key)}


@pytest.mark.parametrize("fname", ["pre_cfg1.npz", "pre_w48.npz"])
@pytest.mark.parametrize("mode", MODES)
def test_restatement_matches_reference_golden(golden_dir, fname, mode):
    d = _load(golden_dir, fname)
    out = op.oracle().run(op.make_params(*mode), d["knots"], d["lin"], d["q_k_lin"])
    ref = _mode_out(d, mode)
    # the restatement is the same algorithm in the same order: demand far tighter than the gates
    for k in ("DT", "alpha", "beta", "q", "R", "J_q", "J_a", "J_b", "H_a", "H_b", "O_a", "O_b"):
        assert np.abs(out[k] - ref[k]).max() < 1e-12, k
    assert cov_rel_err(out["P"], ref["P"]) < 1e-12
    check_pre(out, ref, v2=(mode[0] == 2))


@pytest.mark.parametrize("mode", MODES)
def test_restatement_matches_compiled_reference_live(mode):
    ref = op.reference()
    if ref is None:
        pytest.skip("oracle/_ref/libcpi_ref.so not present")
    kn, lin, q = synth.make_windows(96, 50, seed=4242 + mode[0])
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    prm = op.make_params(*mode)
    a = op.oracle().run(prm, kn, lin, q)
    b = ref.run(prm, kn, lin, q)
    for k in ("DT", "alpha", "beta", "q", "R", "J_q", "J_a", "J_b", "H_a", "H_b", "O_a", "O_b"):
        assert np.abs(a[k] - b[k]).max() < 1e-12, k
    assert cov_rel_err(a["P"], b["P"]) < 1e-12


def test_trace_end_state_equals_window(golden_dir):
    for model in (1, 2):
        d = _load(golden_dir, "trace_v%d.npz" % model)
        tr = op.oracle().trace(op.make_params(model, 0, 1), d["knots"], d["lin"], d["q_k_lin"])
        for k in ("alpha", "beta", "q", "P", "J_a"):
            assert np.array_equal(tr[k], d[k])
        w48 = _load(golden_dir, "pre_w48.npz")
        ref = _mode_out(w48, (model, 0, 1))
        assert np.abs(tr["alpha"][-1] - ref["alpha"][0]).max() < 1e-12
        assert cov_rel_err(tr["P"][-1], ref["P"][0]) < 1e-12


def test_structural_invariants():
    """P symmetric, zero blocks theta-ba / bw-ba, mean composition."""
    kn, lin, q = synth.make_windows(32, 50, seed=99)
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    for model in (1, 2):
        out = op.oracle().run(op.make_params(model, 0, 1), kn, lin, q)
        P = out["P"].reshape(-1, 15, 15)
        assert np.array_equal(P, P.transpose(0, 2, 1))
        assert np.abs(P[:, 0:3, 9:12]).max() == 0.0     # theta / b_a
        assert np.abs(P[:, 3:6, 9:12]).max() == 0.0     # b_w / b_a
        ev = np.linalg.eigvalsh(P)
        assert ev.min() > -1e-18


def test_mean_composition_v1():
    """Halves composed == whole (associativity used by the lane-parallel kernels):
    R_AB = R_B R_A, beta = beta_A + R_A^T beta_B, alpha = alpha_A + beta_A DT_B + R_A^T alpha_B."""
    kn, lin, q = synth.make_windows(16, 50, seed=7, edge_cases=False)
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    prm = op.make_params(1, 0, 1)
    whole = op.oracle().run(prm, kn, lin, q)
    A = op.oracle().run(prm, kn[:, :26], lin, q)
    B = op.oracle().run(prm, kn[:, 25:], lin, q)
    RA = A["R"].reshape(-1, 3, 3).transpose(0, 2, 1)   # column-major -> [i][j]
    RB = B["R"].reshape(-1, 3, 3).transpose(0, 2, 1)
    R = np.einsum("wij,wjk->wik", RB, RA)
    beta = A["beta"] + np.einsum("wji,wj->wi", RA, B["beta"])
    alpha = A["alpha"] + A["beta"] * B["DT"][:, None] + np.einsum("wji,wj->wi", RA, B["alpha"])
    Rw = whole["R"].reshape(-1, 3, 3).transpose(0, 2, 1)
    assert np.abs(R - Rw).max() < 1e-14
    assert np.abs(beta - whole["beta"]).max() < 1e-14
    assert np.abs(alpha - whole["alpha"]).max() < 1e-14


@pytest.mark.parametrize("dt_scale,w_scale,a_scale", [(4.0, 1.0, 1.0), (0.02, 1.0, 1.0), (1.0, 40.0, 1.0), (1.0, 1.0, 50.0),
                                                     (0.5, 10.0, 10.0), (1.0, 1e-4, 1.0)])
def test_restatement_matches_compiled_reference_over_the_dynamic_range(dt_scale, w_scale, a_scale):
    """The scales tests/test_gpu_parity.py::test_dynamic_range_stress checks the HIP path on (50 Hz - 10 kHz, rates up
    to ~20 rad/s and down to the Taylor branch below 0.0087 rad/s, specific forces up to ~500 m/s^2): the restatement
    the GPU is compared with there is itself pinned to the compiled reference on the same inputs."""
    ref = op.reference()
    if ref is None:
        pytest.skip("oracle/_ref/libcpi_ref.so not present")
    kn, lin, q = synth.make_windows(64, 50, seed=31337, edge_cases=False)
    kn, lin, q = kn.numpy().copy(), lin.numpy().copy(), q.numpy()
    t0 = kn[:, :1, 0].copy()
    kn[:, :, 0] = t0 + (kn[:, :, 0] - t0) * dt_scale
    kn[:, :, 1:4] *= w_scale
    kn[:, :, 4:7] *= a_scale
    lin[:, 0:3] *= w_scale
    lin[:, 3:6] *= a_scale
    for mode in [(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 0)]:
        prm = op.make_params(*mode)
        a, b = op.oracle().run(prm, kn, lin, q), ref.run(prm, kn, lin, q)
        for k in ("DT", "alpha", "beta", "q", "J_q", "J_a", "J_b", "H_a", "H_b", "O_a", "O_b"):
            scale = max(1.0, float(np.abs(b[k]).max()))
            assert np.abs(a[k] - b[k]).max() <= 1e-11 * scale, (mode, k)
        assert cov_rel_err(a["P"], b["P"]) < 1e-10, mode
