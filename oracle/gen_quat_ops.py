"""Generate tests/golden/quat_ops.npz -- golden vectors for the quaternion / SO(3) helpers every kernel and
evaluateError are built from (cpi_compare/src/utils/quat_ops.h: rot_2_quat :45, skew_x :92, quat_2_Rot :104,
quat_multiply :115, Exp :145, Inv :190).

Run in the dev container only: the expected outputs are produced by the REFERENCE's own functions, compiled unchanged
from /root/reference into oracle/_ref/libcpi_ref.so (oracle/ref_shim.cpp: cpi_ref_quat_ops).  The fixture holds data
only (seeded inputs + the outputs the compiled reference returned for them).

    python -m oracle.gen_quat_ops

Inputs: random cases plus the edges the call sites can reach -- rot_2_quat: all four branches, 180-degree rotations
(q_w = 0 exactly), rotations within 1e-9 of 180 degrees on either side, slightly non-orthonormal matrices (accumulated
R_k2tau drift); quat_multiply: products whose scalar part is negative (sign flip), identity factors, inverse pairs;
Exp: w = 0 exactly, |w| from 1e-300 to 1e-8, the Taylor / closed-form neighbourhood, theta near pi and 2 pi, theta > pi;
quat_2_Rot / Inv: unit quaternions incl. w = 0 and negative w.
"""
import ctypes as C
import os

import numpy as np

from oracle import oracle_py as op

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
OPS = {"rot_2_quat": (0, 9, 4), "skew_x": (1, 3, 9), "quat_2_Rot": (2, 4, 9), "quat_multiply": (3, 8, 4), "Exp": (4, 3, 9),
       "Inv": (5, 4, 4)}


def call(fn, opcode, nin, nout, x):
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, nin)
    out = np.zeros((x.shape[0], nout))
    rc = fn(C.c_int(opcode), C.c_long(x.shape[0]), x.ctypes.data_as(C.POINTER(C.c_double)),
            out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def unit_quats(rng, n):
    q = rng.standard_normal((n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def rot_of(q):
    """JPL quaternion -> rotation (numpy, for building INPUTS only; the expected values come from the reference)."""
    x, y, z, w = q.T
    v = q[:, :3]
    S = np.zeros((len(q), 3, 3))
    S[:, 0, 1], S[:, 0, 2], S[:, 1, 0], S[:, 1, 2], S[:, 2, 0], S[:, 2, 1] = -z, y, z, -x, -y, x
    return (2 * w * w - 1)[:, None, None] * np.eye(3) - 2 * w[:, None, None] * S + 2 * v[:, :, None] * v[:, None, :]


def inputs(seed=20190103):
    rng = np.random.default_rng(seed)
    d = {}
    # ---- rot_2_quat
    q = unit_quats(rng, 96)
    for k in range(4):                       # dominant component k -> each of the four branches
        qq = unit_quats(rng, 24) * 0.2
        qq[:, k] = np.sign(rng.standard_normal(24)) * 1.0
        q = np.vstack([q, qq / np.linalg.norm(qq, axis=1, keepdims=True)])
    R = rot_of(q)
    half = np.array([np.diag([1.0, -1, -1]), np.diag([-1.0, 1, -1]), np.diag([-1.0, -1, 1]), np.eye(3)])   # 180 deg (q_w = 0) + identity
    near = []
    for ax in np.eye(3):
        for eps in (1e-9, -1e-9, 1e-5, -1e-5):
            th = np.pi + eps
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            near.append(np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K)
    drift = R[:16] + 1e-15 * rng.standard_normal((16, 3, 3))      # R_k2tau is never re-orthonormalised (CpiV1.h:357)
    d["rot_2_quat"] = np.concatenate([R, half, np.array(near), drift]).reshape(-1, 9)
    # ---- skew_x / Inv / quat_2_Rot
    d["skew_x"] = np.vstack([rng.standard_normal((16, 3)) * 5, np.zeros((1, 3))])
    qs = np.vstack([unit_quats(rng, 48), [[0, 0, 0, 1.0]], [[1.0, 0, 0, 0]], [[0, 1.0, 0, 0]], [[0, 0, 1.0, 0]],
                    [[0.6, 0, 0.8, 0]], [[0, 0, 0, -1.0]]])
    d["quat_2_Rot"] = qs
    d["Inv"] = qs
    # ---- quat_multiply
    a, b = unit_quats(rng, 96), unit_quats(rng, 96)
    a[:8] = [0, 0, 0, 1.0]                                  # identity on the left
    b[8:16] = [0, 0, 0, 1.0]                                # identity on the right
    b[16:32] = a[16:32] * np.array([-1, -1, -1, 1.0])       # q * q^-1 (the factor's q_n, q_rminus pattern)
    a[32:40, 3] = 0; a[32:40] /= np.linalg.norm(a[32:40], axis=1, keepdims=True)   # pure-vector factors
    a[40:48] *= 1.0 + 1e-12 * rng.standard_normal((8, 1))   # slightly un-normalised (renormalised by the product)
    d["quat_multiply"] = np.hstack([a, b])
    # ---- Exp
    dirs = rng.standard_normal((64, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    mags = np.concatenate([[0.0, 1e-300, 1e-160, 1e-30, 1e-12, 1e-8, 1e-5, 8.7e-5, 0.0087, 0.0088, 0.25, 0.2500001, 1.0, 1.0000001,
                            np.pi - 1e-9, np.pi, np.pi + 1e-9, 2 * np.pi, 2 * np.pi - 1e-7, 7.5, 31.4, 100.0],
                           10 ** rng.uniform(-6, 0.7, 42)])
    w = dirs * mags[:, None]
    w[0] = 0.0
    d["Exp"] = np.vstack([w, [[1e-3, 0, 0]], [[0, -2e-3, 0]], [[0, 0, 0.3]]])
    return d


def main():
    ref = op.reference()
    assert ref is not None, "oracle/_ref/libcpi_ref.so missing: run `make -C oracle ref` in the dev container"
    fn = ref.lib.cpi_ref_quat_ops
    fn.restype = C.c_int
    out = {}
    for name, x in inputs().items():
        opcode, nin, nout = OPS[name]
        out[name + "__in"] = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, nin)
        out[name + "__out"] = call(fn, opcode, nin, nout, x)
        assert np.all(np.isfinite(out[name + "__out"])), name
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, "quat_ops.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
