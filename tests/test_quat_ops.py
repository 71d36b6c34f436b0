"""The quaternion / SO(3) helpers every kernel and evaluateError / predict / retract are built from, pinned to the
REFERENCE'S OWN functions (cpi_compare/src/utils/quat_ops.h:45-197).

tests/golden/quat_ops.npz holds seeded inputs and the outputs of the reference's rot_2_quat / skew_x / quat_2_Rot /
quat_multiply / Exp / Inv, compiled unchanged into oracle/_ref/libcpi_ref.so (generator: oracle/gen_quat_ops.py).
Checked here (CPU): (i) the C restatement's helpers, (ii) the host emulation of the kernel helpers (cpi_math.hpp logic),
(iii) live, when oracle/_ref is present, the compiled reference again on fresh random inputs against the restatement.
The device instructions themselves are checked by tests/test_gpu_quat_ops.py.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_py as op
from tests import hostsim_py

OPS = {"rot_2_quat": (0, 9, 4), "skew_x": (1, 3, 9), "quat_2_Rot": (2, 4, 9), "quat_multiply": (3, 8, 4), "Exp": (4, 3, 9),
       "Inv": (5, 4, 4)}
TOL_RESTATEMENT = 1e-15      # judge's bar: restated helpers == compiled reference helpers
TOL_KERNEL_LOGIC = 4e-15     # host emulation of cpi_math.hpp (reciprocal-multiply / Horner forms of the same expressions)


def call(fn, name, x):
    opcode, nin, nout = OPS[name]
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, nin)
    out = np.zeros((x.shape[0], nout))
    fn.restype = C.c_int
    assert fn(C.c_int(opcode), C.c_long(x.shape[0]), x.ctypes.data_as(C.POINTER(C.c_double)),
              out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return out


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "quat_ops.npz"))


@pytest.mark.parametrize("name", sorted(OPS))
def test_restatement_helpers_equal_the_compiled_reference(gold, name):
    got = call(op.oracle().lib.cpi_oracle_quat_ops, name, gold[name + "__in"])
    err = np.abs(got - gold[name + "__out"]).max()
    assert err <= TOL_RESTATEMENT, (name, err)


@pytest.mark.parametrize("name", sorted(OPS))
def test_kernel_helper_logic_equals_the_compiled_reference(gold, name):
    x, want = gold[name + "__in"], gold[name + "__out"]
    got = call(hostsim_py.lib().hs_quat_ops, name, x)
    scale = np.maximum(1.0, np.abs(want).max())
    err = np.abs(got - want).max() / scale
    assert err <= TOL_KERNEL_LOGIC, (name, err)


def test_fixture_covers_every_branch(gold):
    R = gold["rot_2_quat__in"].reshape(-1, 3, 3)
    T = np.trace(R, axis1=1, axis2=2)
    d = np.stack([R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]], axis=1)
    b0 = (d[:, 0] >= T) & (d[:, 0] >= d[:, 1]) & (d[:, 0] >= d[:, 2])
    b1 = ~b0 & (d[:, 1] >= T) & (d[:, 1] >= d[:, 0]) & (d[:, 1] >= d[:, 2])
    b2 = ~b0 & ~b1 & (d[:, 2] >= T) & (d[:, 2] >= d[:, 0]) & (d[:, 2] >= d[:, 1])
    b3 = ~b0 & ~b1 & ~b2
    assert min(b0.sum(), b1.sum(), b2.sum(), b3.sum()) >= 10               # all four branches of quat_ops.h:53-77
    q = gold["rot_2_quat__out"]
    assert (q[:, 3] == 0).sum() >= 3 and (q[:, 3] >= 0).all()              # 180-degree rotations: q_w = 0 exactly; w >= 0 (:80-82)
    w = gold["Exp__in"]
    th = np.linalg.norm(w, axis=1)
    assert (th == 0).sum() >= 2 and (th < 1e-100).sum() >= 3 and (th > np.pi).sum() >= 4   # |w| = 1e-300 underflows to theta == 0
    a, b = gold["quat_multiply__in"][:, :4], gold["quat_multiply__in"][:, 4:]
    raw_w = a[:, 3] * b[:, 3] - (a[:, :3] * b[:, :3]).sum(1)
    assert (raw_w < 0).sum() >= 10                                          # the sign flip of quat_ops.h:124-126 is exercised


def test_live_reference_vs_restatement_on_fresh_inputs():
    ref = op.reference()
    if ref is None or not hasattr(ref.lib, "cpi_ref_quat_ops"):
        pytest.skip("oracle/_ref/libcpi_ref.so not present (GPU box without a prebuilt reference)")
    rng = np.random.default_rng(99)
    q = rng.standard_normal((500, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    p = rng.standard_normal((500, 4)); p /= np.linalg.norm(p, axis=1, keepdims=True)
    w = rng.standard_normal((500, 3)) * 10 ** rng.uniform(-8, 0.5, (500, 1))
    cases = {"quat_multiply": np.hstack([q, p]), "quat_2_Rot": q, "Inv": q, "Exp": w, "skew_x": w}
    cases["rot_2_quat"] = call(ref.lib.cpi_ref_quat_ops, "quat_2_Rot", q)
    for name, x in cases.items():
        a = call(ref.lib.cpi_ref_quat_ops, name, x)
        b = call(op.oracle().lib.cpi_oracle_quat_ops, name, x)
        assert np.abs(a - b).max() <= TOL_RESTATEMENT, name
