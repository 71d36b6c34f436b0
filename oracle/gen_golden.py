"""Generate tests/golden/*.npz -- golden input/output vectors for the CPI hot path.

Run in the dev container only (needs oracle/_ref/libcpi_ref.so = the reference's own CpiV1/CpiV2
headers compiled unchanged from /root/reference).  The fixtures hold DATA only: seeded synthetic
inputs and the outputs the compiled reference produced for them.

  python -m oracle.gen_golden

Files
  pre_cfg1.npz     1 window x 100 samples @200 Hz (BASELINE.json configs[0]); V1 and V2 outputs
  pre_w48.npz      48 windows x 50 samples, edge cases forced in (Taylor branch, dt==0, 5x gap,
                   non-uniform dt, negative dt, tail interval with a repeated reading); outputs for
                   {V1,V2} x {imu_avg 0,1} x {V2 state_transition_jacobians 0,1}  -- from the REFERENCE
  trace_v1.npz / trace_v2.npz   per-sample snapshots of window 0 of pre_w48 -- from the C restatement
                   (validated above against the reference end states)
  factor_256.npz   256 evaluateError cases per model.  PARITY UNPINNED: produced by the C restatement
                   (the reference factor TUs need GTSAM/Boost and cannot be built here).
"""
import os

import numpy as np

from cpi_amd import synth
from oracle import oracle_py as op

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
MODES = [(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 1), (2, 0, 0), (2, 1, 0)]  # (model, imu_avg, stj)


def mode_key(m):
    return "m%d_avg%d_stj%d" % m


def edge_windows(W=48, N=50, seed=synth.BASE_SEED):
    kn, lin, q = synth.make_windows(W, N, seed=seed, edge_cases=False)
    kn, lin, q = kn.numpy().copy(), lin.numpy().copy(), q.numpy().copy()
    rng = np.random.default_rng(seed)
    # windows 0-5: quiet (|w| below / straddling the 0.008726646 rad/s threshold), noise-free
    for w in range(6):
        amp = [0.001, 0.004, 0.0087, 0.009, 0.012, 0.02][w]
        t = kn[w, :, 0:1]
        kn[w, :, 1:4] = lin[w, 0:3] + amp * np.sin(2 * np.pi * 0.7 * t + rng.uniform(0, 6.28, (1, 3)))
    kn[5, 10, 1:4] = lin[5, 0:3]                   # exactly zero w_hat in one sample
    # 6: one repeated timestamp (dt == 0)
    kn[6, 20:, 0] -= (kn[6, 20, 0] - kn[6, 19, 0])
    # 7: one 5x gap
    kn[7, 30:, 0] += 4 * 0.005
    # 8: non-uniform dt (jitter)
    kn[8, :, 0] += np.sort(rng.uniform(0, 0.004, 51))
    # 9: a negative dt (interval skipped by the caller's dt >= 0 guard)
    kn[9, 25, 0] = kn[9, 24, 0] - 0.001
    # 10: tail interval -- last knot repeats the previous reading at a non-grid time
    kn[10, 50, 1:7] = kn[10, 49, 1:7]
    kn[10, 50, 0] = kn[10, 49, 0] + 0.0031
    # 11: fast rotation (|w| ~ 4 rad/s)
    kn[11, :, 1:4] *= 2.0
    # 12: identity linearisation orientation
    q[12] = [0, 0, 0, 1]
    return kn, lin, q


def main():
    ref = op.reference()
    orc = op.oracle()
    assert ref is not None, "oracle/_ref/libcpi_ref.so missing: run `make -C oracle` in the dev container"
    os.makedirs(GOLD, exist_ok=True)

    # ---- config 1: 1 window x 100 samples @ 200 Hz
    kn, lin, q = synth.make_windows(1, 100, seed=synth.BASE_SEED + 1, edge_cases=False)
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    d = dict(knots=kn, lin=lin, q_k_lin=q)
    for m in MODES:
        out = ref.run(op.make_params(*m), kn, lin, q)
        for k, v in out.items():
            d["%s__%s" % (mode_key(m), k)] = v
    np.savez_compressed(os.path.join(GOLD, "pre_cfg1.npz"), **d)

    # ---- 48 x 50 with edge cases
    kn, lin, q = edge_windows()
    d = dict(knots=kn, lin=lin, q_k_lin=q)
    for m in MODES:
        out = ref.run(op.make_params(*m), kn, lin, q)
        for k, v in out.items():
            d["%s__%s" % (mode_key(m), k)] = v
    np.savez_compressed(os.path.join(GOLD, "pre_w48.npz"), **d)

    # ---- per-sample traces (restatement; end state equals the reference's above)
    for model in (1, 2):
        tr = orc.trace(op.make_params(model, 0, 1), kn[0], lin[0], q[0])
        np.savez_compressed(os.path.join(GOLD, "trace_v%d.npz" % model), knots=kn[0], lin=lin[0],
                            q_k_lin=q[0], **tr)

    # ---- factor cases (UNPINNED: restatement outputs)
    import torch
    knf, linf, qf = synth.make_windows(256, 50, seed=synth.BASE_SEED + 2)
    d = {}
    for model in (1, 2):
        out = ref.run(op.make_params(model, 0, 1), knf.numpy(), linf.numpy(), qf.numpy())
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        # state_i orientation near q_k_lin for v2 so that dth_k is small but non-zero
        xi, xj = synth.make_states(T(out["alpha"]), T(out["beta"]), T(out["q"]), T(out["DT"]), linf, model)
        xi, xj = xi.numpy().copy(), xj.numpy().copy()
        if model == 2:
            # q_GtoK = small rotation * q_k_lin
            rng = np.random.default_rng(5)
            dth = 1e-2 * rng.standard_normal((256, 3))
            for k in range(256):
                xi[k] = orc.retract(np.concatenate([qf.numpy()[k], xi[k, 4:]]),
                                    np.concatenate([dth[k], np.zeros(12)]))
            # re-predict state_j from the new orientation
            rec0 = op.factor_records(out, linf.numpy(), qf.numpy())
            xjp = orc.predict(2, rec0, xi)
            pert = rng.standard_normal((256, 15)) * np.array([1e-3] * 3 + [1e-4] * 3 + [1e-2] * 3 + [1e-3] * 3 + [1e-2] * 3)
            xj = np.stack([orc.retract(xjp[k], pert[k]) for k in range(256)])
        rec = op.factor_records(out, linf.numpy(), qf.numpy() if model == 2 else None)
        err, H1, H2 = orc.factor(model, rec, xi, xj)
        d.update({"v%d_rec" % model: rec, "v%d_xi" % model: xi, "v%d_xj" % model: xj,
                  "v%d_err" % model: err, "v%d_H1" % model: H1, "v%d_H2" % model: H2})
    np.savez_compressed(os.path.join(GOLD, "factor_256.npz"), **d)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
