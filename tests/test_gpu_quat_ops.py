"""Device-side quaternion / SO(3) helpers (cpi_math.hpp, the shipped instruction sequences: v_rsq_f64 + Newton, Horner
sin / cos) against the REFERENCE'S OWN quat_ops.h functions, through the test hook cpi_test_quat_ops of libcpi_amd.so
(include/cpi_amd_test.h).  Expected values: tests/golden/quat_ops.npz, produced by the compiled reference
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


def run_device(eng, name, x):
    opcode, nin, nout = OPS[name]
    xin = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).reshape(-1, nin)).to(eng.device)
    out = torch.full((xin.shape[0], nout), float("nan"), dtype=torch.float64, device=eng.device)
    fn = eng.lib.cpi_test_quat_ops
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    eng._check(fn(eng.ctx, opcode, xin.shape[0], xin.data_ptr(), out.data_ptr()))
    eng.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", sorted(OPS))
def test_device_helpers_equal_the_compiled_reference(eng, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "quat_ops.npz"))
    x, want = g[name + "__in"], g[name + "__out"]
    got = run_device(eng, name, x)
    scale = np.maximum(1.0, np.abs(want).max())
    err = np.abs(got - want).max() / scale
    print("device %s: max err %.3e over %d cases" % (name, err, x.shape[0]))
    assert np.all(np.isfinite(got))
    assert err <= TOL_DEVICE, (name, err)


def test_device_helpers_vs_live_reference_large_sample(eng):
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
        got = run_device(eng, name, x)
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err <= TOL_DEVICE, (name, err)
