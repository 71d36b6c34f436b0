"""Device-side quaternion / SO(3) helpers (cpi_math.hpp, the shipped instruction sequences: v_rsq_f64 + Newton, Horner
sin / cos) against the REFERENCE'S OWN quat_ops.h functions, through the test hook cpi_test_quat_ops of libcpi_amd_test.so
(include/cpi_amd_test.h; the product sources built with -DCPI_TEST_HOOKS).  Expected values: tests/golden/quat_ops.npz, produced by the compiled reference
(oracle/gen_quat_ops.py).  With these primitives pinned, what remains unpinned of evaluateError / predict is the block
assembly -- covered by the finite-difference test of the DEVICE residual under JPLNavState::retract below."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

OPS = {"rot_2_quat": (0, 9, 4), "skew_x": (1, 3, 9), "quat_2_Rot": (2, 4, 9), "quat_multiply": (3, 8, 4), "Exp": (4, 3, 9),
       "Inv": (5, 4, 4)}
# measured on MI355X: worst 4.4e-16 (relative to max(1, |expected|)); gate = ~25x that floor
TOL_DEVICE = 1e-14


@pytest.fixture(scope="module")
def eng():
    import cpi_amd
    return cpi_amd.Engine(device=0)


@pytest.fixture(scope="module")
def hooks():
    """libcpi_amd_test.so (the product sources + the hooks of include/cpi_amd_test.h) with a context of its own on the
    default stream: the product library does not export cpi_test_quat_ops."""
    from tests import hooks_py
    h = hooks_py.lib()
    ctx = C.c_void_p()
    assert h.cpi_ctx_create(0, None, C.byref(ctx)) == 0, h.cpi_last_error(None)
    yield h, ctx
    h.cpi_ctx_destroy(ctx)


def run_device(hooks, name, x):
    h, ctx = hooks
    opcode, nin, nout = OPS[name]
    dev = torch.device("cuda", 0)
    xin = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).reshape(-1, nin)).to(dev)
    out = torch.full((xin.shape[0], nout), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    rc = h.cpi_test_quat_ops(ctx, opcode, xin.shape[0], xin.data_ptr(), out.data_ptr())
    assert rc == 0, h.cpi_last_error(ctx)
    assert h.cpi_ctx_synchronize(ctx) == 0
    return out.cpu().numpy()


@pytest.mark.parametrize("name", sorted(OPS))
def test_device_helpers_equal_the_compiled_reference(hooks, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "quat_ops.npz"))
    x, want = g[name + "__in"], g[name + "__out"]
    got = run_device(hooks, name, x)
    scale = np.maximum(1.0, np.abs(want).max())
    err = np.abs(got - want).max() / scale
    print("device %s: max err %.3e over %d cases" % (name, err, x.shape[0]))
    assert np.all(np.isfinite(got))
    assert err <= TOL_DEVICE, (name, err)


def test_device_helpers_vs_live_reference_large_sample(hooks):
    """When the compiled reference travelled to this box: 20 000 fresh random cases per helper."""
    from oracle import oracle_py as op
    ref = op.reference()
    if ref is None or not hasattr(ref.lib, "cpi_ref_quat_ops"):
        pytest.skip("oracle/_ref/libcpi_ref.so not present")
    fn = ref.lib.cpi_ref_quat_ops
    fn.restype = C.c_int

    def call_ref(name, x):
        opcode, nin, nout = OPS[name]
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, nin)
        out = np.zeros((x.shape[0], nout))
        assert fn(C.c_int(opcode), C.c_long(x.shape[0]), x.ctypes.data_as(C.POINTER(C.c_double)),
                  out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        return out
    rng = np.random.default_rng(4)
    n = 20000
    q = rng.standard_normal((n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    p = rng.standard_normal((n, 4)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    w = rng.standard_normal((n, 3)) * 10 ** rng.uniform(-9, 0.6, (n, 1))
    cases = {"quat_multiply": np.hstack([q, p]), "quat_2_Rot": q, "Inv": q, "Exp": w, "skew_x": w,
             "rot_2_quat": call_ref("quat_2_Rot", q)}
    for name, x in cases.items():
        want = call_ref(name, x)
        got = run_device(hooks, name, x)
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err <= TOL_DEVICE, (name, err)


# --------------------------------------------------------------------------- block assembly of evaluateError, on the device
def _golden_factor_inputs(golden_dir, model, eng):
    from oracle import oracle_py as op
    d = dict(np.load(os.path.join(golden_dir, "factor_256.npz")))
    rec, xi, xj = d["v%d_rec" % model], d["v%d_xi" % model], d["v%d_xj" % model]
    cols, o = {}, 0
    for name, n in op.FACTOR_FIELDS:
        cols[name] = rec[:, o:o + n]; o += n
    meas = dict(DT=cols["deltatime"][:, 0], alpha=cols["alpha"], beta=cols["beta"], q=cols["q_KtoK1"], J_q=cols["J_q"],
                J_b=cols["J_beta"], J_a=cols["J_alpha"], H_b=cols["H_beta"], H_a=cols["H_alpha"], O_b=cols["O_beta"],
                O_a=cols["O_alpha"])
    lin = np.concatenate([cols["bg_lin"], cols["ba_lin"]], axis=1)
    return rec, xi, xj, meas, lin, cols["q_K_lin"]


def _dev(a, eng):
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


@pytest.mark.parametrize("model", [1, 2])
def test_device_jacobians_are_the_derivative_of_the_device_residual_under_retract(eng, golden_dir, model):
    """H1 / H2 of the HIP evaluateError == central finite differences of the HIP residual along
    JPLNavState::retract (JPLNavState.cpp:37-71: dq(theta) (x) q on the quaternion, additive on b_g, v, b_a, p), for all
    15 + 15 tangent directions of 96 golden cases.  Together with the helper pins above this fixes the block layout,
    signs and scalings of ImuFactorCPIv1.cpp:109-143,169-185 / ImuFactorCPIv2.cpp:73-119 on the device itself."""
    from oracle import oracle_py as op
    rec, xi, xj, meas, lin, qk = _golden_factor_inputs(golden_dir, model, eng)
    K = 96
    h = 1e-6
    o = op.oracle()
    # states: [0, K) = xi, [K, 2K) = xj, then for every (case, side, direction, sign) one perturbed state
    pert = []
    idx_i, idx_j, src = [], [], []
    for k in range(K):
        for side in (0, 1):
            base = xi[k] if side == 0 else xj[k]
            for c in range(15):
                for sgn in (+1.0, -1.0):
                    dx = np.zeros(15); dx[c] = sgn * h
                    pert.append(o.retract(base, dx))
                    pid = 2 * K + len(pert) - 1
                    idx_i.append(pid if side == 0 else k)
                    idx_j.append(pid if side == 1 else K + k)
                    src.append(k)
    states = np.concatenate([xi[:K], xj[:K], np.array(pert)], axis=0)
    src = np.array(src)
    m_rep = {kk: _dev(v[src], eng) for kk, v in meas.items()}
    out = eng.factor_eval(model, m_rep, _dev(lin[src], eng), _dev(qk[src], eng), _dev(states, eng),
                          _dev(np.array(idx_i, dtype=np.int32), eng), _dev(np.array(idx_j, dtype=np.int32), eng), want_H=False)
    m0 = {kk: _dev(v[:K], eng) for kk, v in meas.items()}
    ana = eng.factor_eval(model, m0, _dev(lin[:K], eng), _dev(qk[:K], eng), _dev(states, eng),
                          _dev(np.arange(K, dtype=np.int32), eng), _dev((np.arange(K) + K).astype(np.int32), eng))
    torch.cuda.synchronize()
    e = out["err"].cpu().numpy().reshape(K, 2, 15, 2, 15)          # [case, side, direction, sign, residual row]
    fd = (e[:, :, :, 0, :] - e[:, :, :, 1, :]) / (2 * h)            # [case, side, direction(col), row]
    H1 = ana["H1"].cpu().numpy().reshape(K, 15, 15)                 # column-major flat -> [case, col, row]
    H2 = ana["H2"].cpu().numpy().reshape(K, 15, 15)
    w1 = np.abs(fd[:, 0] - H1).max()
    w2 = np.abs(fd[:, 1] - H2).max()
    print("device FD check model %d: H1 %.2e  H2 %.2e" % (model, w1, w2))
    # measured 7e-8 / 3e-8 (central differences with h = 1e-6 on O(10) entries: truncation + cancellation noise)
    assert w1 < 1e-6 and w2 < 1e-6, (w1, w2)
    # and no structural zero is filled: blocks that ImuFactorCPIv1.cpp leaves zero are exactly zero on the device
    Z1 = np.ones((15, 15), dtype=bool)
    for (r, c) in ((0, 0), (0, 3), (3, 3), (6, 0), (6, 3), (6, 6), (6, 9), (9, 9), (12, 0), (12, 3), (12, 6), (12, 9), (12, 12)):
        Z1[r:r + 3, c:c + 3] = False
    assert np.all(H1.transpose(0, 2, 1)[:, Z1] == 0.0)
    Z2 = np.ones((15, 15), dtype=bool)
    for r in (0, 3, 6, 9, 12):
        Z2[r:r + 3, r:r + 3] = False
    assert np.all(H2.transpose(0, 2, 1)[:, Z2] == 0.0)


@pytest.mark.parametrize("model", [1, 2])
def test_device_residual_vanishes_at_the_device_prediction(eng, golden_dir, model):
    """cpi_predict_batch (GraphSolver_IMU.cpp:263-307) and cpi_factor_eval_batch are mutually consistent on the device:
    with the biases on the linearisation point (and, model 2, q_GtoK on q_K_lin) the residual at the predicted state is 0."""
    rec, xi, xj, meas, lin, qk = _golden_factor_inputs(golden_dir, model, eng)
    xi = xi.copy()
    xi[:, 4:7] = lin[:, 0:3]
    xi[:, 10:13] = lin[:, 3:6]
    if model == 2:
        xi[:, 0:4] = qk
    m = {kk: _dev(v, eng) for kk, v in meas.items()}
    xjp = eng.predict(model, m, _dev(xi, eng))
    states = torch.cat([_dev(xi, eng), xjp], dim=0).contiguous()
    F = xi.shape[0]
    out = eng.factor_eval(model, m, _dev(lin, eng), _dev(qk, eng), states, _dev(np.arange(F, dtype=np.int32), eng),
                          _dev((np.arange(F) + F).astype(np.int32), eng), want_H=False)
    torch.cuda.synchronize()
    assert out["err"].abs().max().item() < 1e-12
