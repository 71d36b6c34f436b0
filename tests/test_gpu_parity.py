"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against
  (1) the golden vectors the compiled reference produced (tests/golden/),
  (2) the CPU oracle on fresh seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.
Contractual gates (tests/tol.py): 1e-9 on DT/alpha/beta/q, 1e-6 * sqrt(P_ii P_jj) on the covariance,
1e-8 on the bias Jacobians, 1e-9 on evaluateError.  Wherever the expected values come from the COMPILED REFERENCE
(golden fixtures; oracle/_ref/libcpi_ref.so, which travels to the GPU box) on realistic inputs, the regression gates
apply instead: 100 x the measured floor (2e-13 / 2e-13 relative / 3e-11).  Seeded tests compare with the compiled
reference directly when it is present (_cpu below) and fall back to the pinned restatement otherwise."""
import os

import numpy as np
import pytest
import torch

from cpi_amd import synth
from tests.tol import REG_FACTOR, TOL_COV, TOL_FACTOR, TOL_MEAN, check_pre, cov_rel_err

pytestmark = pytest.mark.gpu

MODES = [(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 1), (2, 0, 0), (2, 1, 0)]


@pytest.fixture(scope="module")
def eng():
    import cpi_amd
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return cpi_amd.Engine()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle_py as op
    return op


def _dev(a, eng):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


def _host(out):
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def _cpu(orc, mode, kn, lin, q, nthreads=None):
    """CPU expectation for seeded inputs: the reference's own CpiV1 / CpiV2 (oracle/_ref) when present, else the
    restatement (pinned to it).  Returns (outputs, from_reference)."""
    ref = orc.reference()
    lib = ref if ref is not None else orc.oracle()
    return lib.run(orc.make_params(*mode), kn, lin, q, nthreads=nthreads or min(16, os.cpu_count() or 1)), ref is not None


def _cpu_lib(orc):
    """The compiled reference when present, else the restatement (same run() interface, models 1 and 2)."""
    ref = orc.reference()
    return ref if ref is not None else orc.oracle()


def _mode_out(d, m):
    key = "m%d_avg%d_stj%d__" % m
    return {k[len(key):]: v for k, v in d.items() if k.startswith(key)}


def _run(eng, m, kn, lin, q, want=("mean", "jac", "cov"), lanes=0, **kw):
    prm = eng.make_params(m[0], m[1], m[2], lanes_per_window=lanes)
    return _host(eng.preintegrate(_dev(kn, eng), _dev(lin, eng), _dev(q, eng), prm, want=want, **kw))


# --------------------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("fname", ["pre_cfg1.npz", "pre_w48.npz"])
@pytest.mark.parametrize("mode", MODES)
def test_full_outputs_vs_reference_golden(eng, golden_dir, fname, mode):
    d = dict(np.load(os.path.join(golden_dir, fname)))
    out = _run(eng, mode, d["knots"], d["lin"], d["q_k_lin"])
    check_pre(out, _mode_out(d, mode), v2=(mode[0] == 2), label="%s %s" % (fname, mode), regression=True)


@pytest.mark.parametrize("lanes", [0, 1, 2, 3, 5, 6, 8, 12, 16, 64])
@pytest.mark.parametrize("avg", [0, 1])
def test_model2_mean_only_segment_form_vs_reference_golden(eng, golden_dir, lanes, avg):
    """Model 2 mean-only with several lanes per window: segments integrate the raw specific force plus the gravity
    response matrices and are composed afterwards (cpi_math.hpp: mean_step_v2seg / grav_combine / grav_apply)."""
    d = dict(np.load(os.path.join(golden_dir, "pre_w48.npz")))
    out = _run(eng, (2, avg, 1), d["knots"], d["lin"], d["q_k_lin"], want=("mean",), lanes=lanes)
    check_pre(out, _mode_out(d, (2, avg, 1)), what=("mean",), regression=True)


@pytest.mark.parametrize("lanes", [0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64])
@pytest.mark.parametrize("avg", [0, 1])
def test_mean_only_vs_reference_golden_all_lane_splits(eng, golden_dir, lanes, avg):
    d = dict(np.load(os.path.join(golden_dir, "pre_w48.npz")))
    out = _run(eng, (1, avg, 1), d["knots"], d["lin"], d["q_k_lin"], want=("mean",), lanes=lanes)
    assert set(out) == {"DT", "alpha", "beta", "q"}
    check_pre(out, _mode_out(d, (1, avg, 1)), what=("mean",), regression=True)
    out = _run(eng, (1, avg, 1), d["knots"], d["lin"], d["q_k_lin"], want=("mean", "jac"), lanes=lanes)
    check_pre(out, _mode_out(d, (1, avg, 1)), what=("mean", "jac"), regression=True)


def test_mean_only_model2_vs_golden(eng, golden_dir):
    d = dict(np.load(os.path.join(golden_dir, "pre_w48.npz")))
    for avg in (0, 1):
        out = _run(eng, (2, avg, 1), d["knots"], d["lin"], d["q_k_lin"], want=("mean",))
        check_pre(out, _mode_out(d, (2, avg, 1)), what=("mean",))


def test_cov_only_and_jac_only_requests(eng, golden_dir):
    d = dict(np.load(os.path.join(golden_dir, "pre_w48.npz")))
    for m in [(1, 0, 1), (2, 0, 1), (2, 0, 0)]:
        ref = _mode_out(d, m)
        out = _run(eng, m, d["knots"], d["lin"], d["q_k_lin"], want=("cov",))
        assert set(out) == {"P"}
        check_pre(out, ref, what=("cov",))
        out = _run(eng, m, d["knots"], d["lin"], d["q_k_lin"], want=("jac",))
        check_pre(out, ref, what=("jac",), v2=(m[0] == 2))


# --------------------------------------------------------------------------- vs oracle, seeded
@pytest.mark.parametrize("mode", MODES)
def test_vs_oracle_seeded(eng, orc, mode):
    W = 1500 if mode[0] == 1 else 700
    kn, lin, q = synth.make_windows(W, 50, seed=777 + 10 * mode[0] + mode[1])
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    ref, from_ref = _cpu(orc, mode, kn, lin, q)
    out = _run(eng, mode, kn, lin, q)
    check_pre(out, ref, v2=(mode[0] == 2), label=str(mode), regression=from_ref)


def test_window_of_100_samples(eng, orc):
    kn, lin, q = synth.make_windows(333, 100, seed=5)     # W not a multiple of any group size
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    for mode in [(1, 0, 1), (2, 0, 1)]:
        ref, from_ref = _cpu(orc, mode, kn, lin, q)
        check_pre(_run(eng, mode, kn, lin, q), ref, v2=(mode[0] == 2), regression=from_ref)


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("W,N", [(700003, 50), (830001, 7), (700000, 100)])
def test_three_knot_kernel_on_the_dense_layout_bitwise_vs_two_knot_and_vs_reference(eng, orc, model, W, N):
    """From 700 000 windows a one-lane mean-only launch on the dense layout runs cpi_mean_kernel<..., BIG> (three knots per
    chunk, 32-bit staging offsets from the wavefront's lowest knot, flat LDS tile; cpi_mean.hip), unless per-window counts are
    given.  Chunking does not touch the arithmetic: BIT FOR BIT the two-knot kernel's outputs (same batch with counts, all
    full), for a batch whose last block is ragged (W not a multiple of 64: clamped idle lanes, the per-element path of the last two
    blocks), for N not a multiple of three (padded steps of the last chunk) and N = 100; a strided sample against the
    compiled reference (CpiV1.h:67-154 / CpiV2.h:88-205) at the regression gates."""
    kn, lin, q = synth.make_windows(W, N, seed=4100 + N + model, device=eng.device)
    for avg in (0, 1):
        prm = eng.make_params(model, avg, lanes_per_window=1)
        a = eng.preintegrate(kn, lin, q, prm, want=("mean",))
        b = eng.preintegrate(kn, lin, q, prm, want=("mean",), count=torch.full((W,), N, dtype=torch.int32, device=eng.device))
        torch.cuda.synchronize()
        for k in ("DT", "alpha", "beta", "q"):
            assert torch.equal(a[k], b[k]), (model, avg, W, N, k)
        pick = torch.cat([torch.arange(0, W, 1013, device=eng.device), torch.arange(W - 70, W, device=eng.device)])
        ref, from_ref = _cpu(orc, (model, avg, 1), kn[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
        check_pre({k: v[pick].cpu().numpy() for k, v in a.items()}, ref, what=("mean",), v2=(model == 2),
                  label="three-knot dense kernel m%d avg%d W=%d N=%d" % (model, avg, W, N), regression=from_ref)


# --------------------------------------------------------------------------- ragged / shared-knot windows
def test_ragged_windows_cut_from_one_stream(eng, orc):
    """Windows cut from ONE IMU stream at irregular update times (GraphSolver_IMU.cpp:50-69):
    first[]/count[] index a shared knot array; lengths 0..73 incl. empty and single-interval windows."""
    rng = np.random.default_rng(3)
    lens = np.concatenate([[0, 1, 2, 73, 16, 17, 31, 32, 33, 64], rng.integers(1, 70, 150)]).astype(np.int32)
    K = int(lens.sum()) + 1
    kn1, _, _ = synth.make_windows(1, K - 1, seed=11, edge_cases=False)
    stream = kn1.numpy()[0]                                  # [K,7]
    first = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    W = len(lens)
    _, lin, q = synth.make_windows(W, 4, seed=12)
    lin, q = lin.numpy(), q.numpy()
    N = int(lens.max())
    for mode in [(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 0)]:
        prm = eng.make_params(*mode)
        out = _host(eng.preintegrate(_dev(stream, eng), _dev(lin, eng), _dev(q, eng), prm, first=_dev(first, eng),
                                     count=_dev(lens, eng), N=N))
        # oracle window by window (zero-length windows: identity state)
        ref = {k: np.zeros_like(v) for k, v in out.items()}
        oprm = orc.make_params(*mode)
        for w in range(W):
            n = int(lens[w])
            kn_w = stream[first[w]:first[w] + n + 1][None]
            r = _cpu_lib(orc).run(oprm, kn_w, lin[w:w + 1], q[w:w + 1])
            for k in ref:
                ref[k][w] = r[k][0]
        check_pre(out, ref, v2=(mode[0] == 2), label="ragged %s" % (mode,))
        assert out["DT"][0] == 0.0 and np.array_equal(out["q"][0], [0, 0, 0, 1]) and np.all(out["P"][0] == 0)
        # mean-only kernel on the same ragged layout, several lane splits
        if mode[0] == 1:
            for lanes in (1, 4, 6, 64):
                prm = eng.make_params(*mode, lanes_per_window=lanes)
                o2 = _host(eng.preintegrate(_dev(stream, eng), _dev(lin, eng), _dev(q, eng), prm, want=("mean",),
                                            first=_dev(first, eng), count=_dev(lens, eng), N=N))
                check_pre(o2, ref, what=("mean",))


def test_empty_batch_and_bad_arguments(eng):
    import cpi_amd
    z = torch.zeros((0, 51, 7), dtype=torch.float64, device=eng.device)
    out = eng.preintegrate(z, torch.zeros((0, 6), dtype=torch.float64, device=eng.device))
    assert out["alpha"].shape == (0, 3)
    kn, lin, q = synth.make_windows(4, 10, seed=1, device=eng.device)
    with pytest.raises(cpi_amd.CpiError):                    # model 2 without q_k_lin
        eng.preintegrate(kn, lin, None, eng.make_params(2))
    with pytest.raises(cpi_amd.CpiError):
        eng.preintegrate(kn, lin, q, eng.make_params(4))      # 3 = the Forster comparator (tests/test_gpu_forster.py)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(3))
    with pytest.raises(cpi_amd.CpiError):                    # ... whose measurement is evaluated as a model-1 factor
        eng.factor_eval(3, meas, lin, None, torch.zeros((5, 16), dtype=torch.float64, device=eng.device))
    with pytest.raises(cpi_amd.CpiError):
        eng.preintegrate(kn, lin, q, eng.make_params(1, lanes_per_window=7))


# --------------------------------------------------------------------------- factors
@pytest.mark.parametrize("model", [1, 2])
def test_factor_eval_vs_golden_and_oracle(eng, orc, golden_dir, model):
    d = dict(np.load(os.path.join(golden_dir, "factor_256.npz")))
    rec, xi, xj = d["v%d_rec" % model], d["v%d_xi" % model], d["v%d_xj" % model]
    F = rec.shape[0]
    cols, o = {}, 0
    for name, n in orc.FACTOR_FIELDS:
        cols[name] = rec[:, o:o + n]; o += n
    meas = dict(DT=cols["deltatime"][:, 0], alpha=cols["alpha"], beta=cols["beta"], q=cols["q_KtoK1"], J_q=cols["J_q"],
                J_b=cols["J_beta"], J_a=cols["J_alpha"], H_b=cols["H_beta"], H_a=cols["H_alpha"], O_b=cols["O_beta"],
                O_a=cols["O_alpha"])
    meas = {k: _dev(v, eng) for k, v in meas.items()}
    lin = np.concatenate([cols["bg_lin"], cols["ba_lin"]], axis=1)
    states = np.concatenate([xi, xj], axis=0)
    idx_i = np.arange(F, dtype=np.int32); idx_j = (np.arange(F) + F).astype(np.int32)
    out = eng.factor_eval(model, meas, _dev(lin, eng), _dev(cols["q_K_lin"], eng), _dev(states, eng), _dev(idx_i, eng),
                          _dev(idx_j, eng))
    out = _host(out)
    for k, key in (("err", "v%d_err"), ("H1", "v%d_H1"), ("H2", "v%d_H2")):
        e = np.abs(out[k] - d[key % model]).max()
        assert e <= TOL_FACTOR and e <= REG_FACTOR, (k, e)    # contractual gate, and "nothing moved"
    # error-only call (boost::optional none)
    o2 = _host(eng.factor_eval(model, meas, _dev(lin, eng), _dev(cols["q_K_lin"], eng), _dev(states, eng),
                               _dev(idx_i, eng), _dev(idx_j, eng), want_H=False))
    assert set(o2) == {"err"} and np.array_equal(o2["err"], out["err"])
    # prediction
    xjp = eng.predict(model, meas, _dev(xi, eng)); torch.cuda.synchronize()
    assert np.abs(xjp.cpu().numpy() - orc.oracle().predict(model, rec, xi)).max() <= TOL_FACTOR


@pytest.mark.parametrize("model", [1, 2])
def test_factor_chain_sweep_vs_oracle(eng, orc, model):
    """cfg-4 shape at oracle-checkable size: F chained factors over F+1 states (idx NULL -> f, f+1),
    measurements straight from the GPU preintegration outputs."""
    F = 3001
    kn, lin, q = synth.make_windows(F, 50, seed=21, device=eng.device)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(model))
    torch.cuda.synchronize()
    xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
    states = torch.cat([xi, xj[-1:]], dim=0).contiguous()    # chained: state f+1 is the next factor's state_i
    out = _host(eng.factor_eval(model, meas, lin, q if model == 2 else None, states))
    m = _host(meas)
    rec = orc.factor_records(m, lin.cpu().numpy(), q.cpu().numpy() if model == 2 else None)
    st = states.cpu().numpy()
    err, H1, H2 = orc.oracle().factor(model, rec, st[:-1], st[1:])
    assert np.abs(out["err"] - err).max() <= TOL_FACTOR * max(1.0, np.abs(err).max())
    assert np.abs(out["H1"] - H1).max() <= TOL_FACTOR * max(1.0, np.abs(H1).max())
    assert np.abs(out["H2"] - H2).max() <= TOL_FACTOR


# --------------------------------------------------------------------------- reference-shaped classes
def test_reference_shaped_classes_config1(eng, golden_dir):
    """BASELINE.json configs[0]: 1 window x 100 samples @200 Hz driven exactly like
    GraphSolver_IMU.cpp:43-75 drives the reference class."""
    import cpi_amd
    d = dict(np.load(os.path.join(golden_dir, "pre_cfg1.npz")))
    kn, lin, q = d["knots"][0], d["lin"][0], d["q_k_lin"][0]
    for cls, mode in ((cpi_amd.CpiV1, (1, 0, 1)), (cpi_amd.CpiV2, (2, 0, 1))):
        cpi = cls(0.005, 4e-6, 0.01, 2e-4, engine=eng)
        cpi.setLinearizationPoints(lin[0:3], lin[3:6], q, [0, 0, 9.8])
        cpi.imu_avg = False
        for i in range(kn.shape[0] - 1):
            cpi.feed_IMU(kn[i, 0], kn[i + 1, 0], kn[i, 1:4], kn[i, 4:7], kn[i + 1, 1:4], kn[i + 1, 4:7])
        ref = _mode_out(d, mode)
        assert abs(cpi.DT - ref["DT"][0]) <= TOL_MEAN
        assert np.abs(cpi.alpha_tau - ref["alpha"][0]).max() <= TOL_MEAN
        assert np.abs(cpi.beta_tau - ref["beta"][0]).max() <= TOL_MEAN
        assert np.abs(cpi.q_k2tau - ref["q"][0]).max() <= TOL_MEAN
        assert np.abs(cpi.J_a - ref["J_a"][0].reshape(3, 3).T).max() <= 1e-8
        assert cov_rel_err(cpi.P_meas.T.reshape(1, 225), ref["P"]) <= 1e-6
        # factor built with the reference's argument order (GraphSolver_IMU.cpp:74-75 / 129-130)
        if mode[0] == 1:
            fac = cpi_amd.ImuFactorCPIv1(cpi.P_meas, cpi.DT, cpi.grav, cpi.alpha_tau, cpi.beta_tau, cpi.q_k2tau,
                                         cpi.b_a_lin, cpi.b_w_lin, cpi.J_q, cpi.J_b, cpi.J_a, cpi.H_b, cpi.H_a, engine=eng)
        else:
            fac = cpi_amd.ImuFactorCPIv2(cpi.P_meas, cpi.DT, cpi.grav, cpi.alpha_tau, cpi.beta_tau, cpi.q_k2tau,
                                         cpi.q_k_lin, cpi.b_a_lin, cpi.b_w_lin, cpi.J_q, cpi.J_b, cpi.J_a, cpi.H_b,
                                         cpi.H_a, cpi.O_b, cpi.O_a, engine=eng)
        xi = np.concatenate([q, lin[0:3], [0.3, -0.2, 0.1], lin[3:6], [1.0, 2.0, 3.0]])
        e, H1, H2 = fac.evaluateError(xi, xi)
        assert e.shape == (15,) and H1.shape == (15, 15) and np.all(np.isfinite(H1))
        assert np.abs(e[3:6]).max() == 0 and np.abs(e[9:12]).max() == 0


# --------------------------------------------------------------------------- full-size properties
def test_config2_size_composition_property(eng):
    """BASELINE configs[1] (10k x 50, model-1 mean-only): halves composed == whole,
    R_AB = R_B R_A, beta = beta_A + R_A^T beta_B, alpha = alpha_A + beta_A DT_B + R_A^T alpha_B."""
    kn, lin, q = synth.make_windows(10000, 50, seed=31, device=eng.device)
    prm = eng.make_params(1)
    whole = eng.preintegrate(kn, lin, q, prm, want=("mean",))
    A = eng.preintegrate(kn[:, :26].contiguous(), lin, q, prm, want=("mean",))
    B = eng.preintegrate(kn[:, 25:].contiguous(), lin, q, prm, want=("mean",))
    torch.cuda.synchronize()

    def R_of(qt):  # quat_2_Rot, batched
        x, y, z, w = qt[:, 0], qt[:, 1], qt[:, 2], qt[:, 3]
        v = qt[:, :3]
        S = torch.zeros((qt.shape[0], 3, 3), dtype=qt.dtype, device=qt.device)
        S[:, 0, 1], S[:, 0, 2], S[:, 1, 0], S[:, 1, 2], S[:, 2, 0], S[:, 2, 1] = -z, y, z, -x, -y, x
        return (2 * w * w - 1)[:, None, None] * torch.eye(3, dtype=qt.dtype, device=qt.device) - 2 * w[:, None, None] * S \
            + 2 * v[:, :, None] * v[:, None, :]
    RA, RB, RW = R_of(A["q"]), R_of(B["q"]), R_of(whole["q"])
    assert (torch.bmm(RB, RA) - RW).abs().max().item() < 1e-12
    beta = A["beta"] + torch.bmm(RA.transpose(1, 2), B["beta"].unsqueeze(-1)).squeeze(-1)
    alpha = A["alpha"] + A["beta"] * B["DT"][:, None] + torch.bmm(RA.transpose(1, 2), B["alpha"].unsqueeze(-1)).squeeze(-1)
    assert (beta - whole["beta"]).abs().max().item() < 1e-12
    assert (alpha - whole["alpha"]).abs().max().item() < 1e-12
    assert (A["DT"] + B["DT"] - whole["DT"]).abs().max().item() < 1e-13
    # every lane split gives the same answer to round-off; launches are deterministic
    ref = whole
    for lanes in (1, 5, 8, 12, 64):
        o = eng.preintegrate(kn, lin, q, eng.make_params(1, lanes_per_window=lanes), want=("mean",))
        assert (o["alpha"] - ref["alpha"]).abs().max().item() < 1e-12
        assert (o["q"] - ref["q"]).abs().max().item() < 1e-13
    again = eng.preintegrate(kn, lin, q, prm, want=("mean",))
    for k in ref:
        assert torch.equal(again[k], ref[k])


@pytest.mark.parametrize("model", [1, 2])
def test_config2_size_launch_geometries_vs_reference_sample(eng, orc, model):
    """BASELINE configs[1] AT ITS OWN LAUNCH GEOMETRY (10 000 windows x 50 samples, mean-only; model 2 = bench row v2_mean):
    128 strided windows of the 10 000-window launch against the compiled reference (CpiV1.h:67-154 / CpiV2.h:88-205) at the
    regression gates, for every way the bench and the facades launch it -- (i) the auto lane split the headline runs (L = 6 at
    this size), (ii) one lane and 64 lanes per window, (iii) the tiled SPLIT kernel (157 tiles x 4 wavefronts) fed by the
    device assembler cpi_assemble_tiles, (iv) the zero-copy stream entry with a partial tail interval in every window.
    Until round 4 this size was checked through properties only (goldens hold 48 windows, seeded runs 1 500)."""
    from cpi_amd import stream as st
    W, N = 10000, 50
    mode = (model, 0, 1)
    pick = np.arange(0, W, 77)[:128]                       # every 77th window: all tiles / wavefronts, every lane-group position
    assert pick.size == 128 and pick[-1] < W
    # ---- dense layout: (i) auto, (ii) L = 1 and 64 (+ the neighbours of the auto choice)
    kn, lin, q = synth.make_windows(W, N, seed=53 + model, device=eng.device)
    ref, from_ref = _cpu(orc, mode, kn[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
    ref = {k: ref[k] for k in ("DT", "alpha", "beta", "q")}
    for lanes in (0, 1, 5, 6, 8, 64):
        out = eng.preintegrate(kn, lin, q, eng.make_params(model, lanes_per_window=lanes), want=("mean",))
        torch.cuda.synchronize()
        check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, ref, what=("mean",), regression=from_ref,
                  label="configs[1] size, model %d, dense, lanes %d" % (model, lanes))
    # ---- ONE stream of 500 001 readings + 10 000 update times: (iii) assembler + tiled SPLIT kernel, (iv) stream entry
    for phase in (0.0, 0.4):                               # on the IMU grid (N whole intervals) / a partial tail per window
        stream, upd, slin, sq = synth.make_stream(W, N, seed=59 + model, device=eng.device, phase=phase)
        knots, first, count = st.assemble_windows(stream.cpu().numpy(), upd.cpu().numpy())       # host assembler = the deque loop
        Nw = int(count.max())
        assert Nw == (N if phase == 0.0 else N + 1)
        dense = np.stack([knots[first[u] + np.minimum(np.arange(Nw + 1), count[u])] for u in pick])   # short windows: dt = 0 padding
        sref, from_ref2 = _cpu(orc, mode, dense, slin[pick].cpu().numpy(), sq[pick].cpu().numpy())
        sref = {k: sref[k] for k in ("DT", "alpha", "beta", "q")}
        tiles, cnt = eng.assemble_tiles(stream, upd, Nw)
        torch.cuda.synchronize()
        assert np.array_equal(cnt.cpu().numpy(), count)
        for S in (0, 1, 4):                                # 0 = auto: 157 tiles < 640 -> four wavefronts per tile (SPLIT)
            out = eng.preintegrate_tiled(tiles, W, slin, sq, eng.make_params(model, lanes_per_window=S), count=cnt)
            torch.cuda.synchronize()
            check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, sref, what=("mean",), regression=from_ref2,
                      label="configs[1] size, model %d, tiled S=%d, phase %.1f" % (model, S, phase))
        for lanes in (0, 1, 6):
            out = eng.preintegrate_stream(stream, upd, slin, sq, eng.make_params(model, lanes_per_window=lanes), want=("mean",), N=Nw)
            torch.cuda.synchronize()
            check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, sref, what=("mean",), regression=from_ref2,
                      label="configs[1] size, model %d, stream entry lanes %d, phase %.1f" % (model, lanes, phase))


def test_config3_size_structure_properties(eng, orc):
    """BASELINE configs[2] (100k x 50, model 2, covariance + bias Jacobians): structural invariants of
    the reference (P symmetric PSD, theta/b_a and b_w/b_a blocks exactly zero, b_w and b_a blocks
    sigma^2 * DT * I), agreement of the covariance kernel's means with the mean kernel, determinism -- and, at the full
    size, a strided sample of 128 windows against the compiled reference (all 299 outputs, regression gates) plus
    slice independence (a slice re-run on its own equals the big run bit for bit)."""
    W = 100000
    kn, lin, q = synth.make_windows(W, 50, seed=41, device=eng.device)
    prm = eng.make_params(2)
    out = eng.preintegrate(kn, lin, q, prm)
    m = eng.preintegrate(kn, lin, q, prm, want=("mean",))
    torch.cuda.synchronize()
    for k in ("DT", "alpha", "beta", "q"):
        assert (out[k] - m[k]).abs().max().item() < 1e-12, k
    P = out["P"].reshape(W, 15, 15)
    assert torch.equal(P, P.transpose(1, 2))
    assert P[:, 0:3, 9:12].abs().max().item() == 0.0 and P[:, 3:6, 9:12].abs().max().item() == 0.0
    eye = torch.eye(3, dtype=torch.float64, device=eng.device)
    DT = out["DT"][:, None, None]
    assert (P[:, 3:6, 3:6] - (4e-6) ** 2 * DT * eye).abs().max().item() < 1e-24
    assert (P[:, 9:12, 9:12] - (2e-4) ** 2 * DT * eye).abs().max().item() < 1e-20
    ev = torch.linalg.eigvalsh(P[:2000])
    assert ev.min().item() > -1e-18
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    # H_b = d beta / d b_a is -int R^T: its scale is DT; J_q ~ DT
    assert (out["H_b"].abs().max(dim=1).values <= out["DT"] * 1.0001 + 1e-12).all()
    again = eng.preintegrate(kn, lin, q, prm)
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(again[k], out[k]), k
    # model 1 on the same windows: identical rotation, covariance of the same magnitude
    o1 = eng.preintegrate(kn, lin, q, eng.make_params(1), want=("mean", "cov"))
    assert (o1["q"] - out["q"]).abs().max().item() < 1e-12
    # the full-size launch against the compiled reference on a strided sample (every 787th window: all tiles of the grid,
    # both halves of a wavefront), all 299 doubles of a window
    pick = torch.arange(0, W, 787, device=eng.device)[:128]
    assert pick.numel() == 128
    ref, from_ref = _cpu(orc, (2, 0, 1), kn[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
    check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, ref, v2=True, label="configs[2] full-size sample", regression=from_ref)
    # slice independence: the covariance kernel gives one lane group to a window whatever the batch, so a slice re-run on its
    # own -- from the middle of the batch, not aligned to a wavefront -- reproduces the big run bit for bit
    for lo, n in ((0, 4096), (50001, 4096), (W - 777, 777)):
        sl = slice(lo, lo + n)
        o = eng.preintegrate(kn[sl].contiguous(), lin[sl].contiguous(), q[sl].contiguous(), prm)
        torch.cuda.synchronize()
        for k in o:
            assert torch.equal(o[k], out[k][sl]), (k, lo)


@pytest.mark.parametrize("model", [1, 2])
def test_config4_full_size_factor_sweep_properties(eng, orc, model):
    """BASELINE configs[3] at full size (1 M factors): the large-sweep kernel (8 lanes per factor) must agree
    bit-for-bit with the small-sweep kernel (16 lanes) on any slice, vanish at the predicted state, keep the
    block structure of H2 (ImuFactorCPIv1.cpp:169-185), be deterministic, and match the oracle on a sample."""
    F = 1000000
    kn, lin, q = synth.make_windows(F, 50, seed=77, device=eng.device, edge_cases=False)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac"))
    torch.cuda.synchronize()
    del kn
    qq = q if model == 2 else None
    xi, _ = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
    # at the linearisation point (biases = b_lin, model 2: q_GtoK = q_K_lin) every first-order correction is zero,
    # so the residual at the exactly predicted state must vanish
    xi[:, 4:7] = lin[:, 0:3]
    xi[:, 10:13] = lin[:, 3:6]
    if model == 2:
        xi[:, 0:4] = q
    xj = eng.predict(model, meas, xi)
    states = torch.cat([xi, xj], dim=0).contiguous()
    idx_i = torch.arange(F, dtype=torch.int32, device=eng.device)
    idx_j = idx_i + F
    out = eng.factor_eval(model, meas, lin, qq, states, idx_i=idx_i, idx_j=idx_j)
    torch.cuda.synchronize()
    assert out["err"].abs().max().item() < 1e-9
    for k in ("err", "H1", "H2"):
        assert torch.isfinite(out[k]).all(), k
    H2 = out["H2"].reshape(F, 15, 15).transpose(1, 2)       # [row][col]
    mask = torch.ones((15, 15), dtype=torch.bool, device=eng.device)
    for b in range(5):
        mask[3 * b:3 * b + 3, 3 * b:3 * b + 3] = False
    assert H2[:, mask].abs().max().item() == 0.0             # block diagonal
    eye = torch.eye(3, dtype=torch.float64, device=eng.device)
    assert torch.equal(H2[:, 3:6, 3:6], eye.expand(F, 3, 3)) and torch.equal(H2[:, 9:12, 9:12], eye.expand(F, 3, 3))
    assert torch.equal(H2[:, 6:9, 6:9], H2[:, 12:15, 12:15])                       # both are R(q_GtoK)
    # slices through the small-sweep kernel
    for lo, n in ((0, 4099), (500000, 1000), (F - 777, 777)):
        sl = slice(lo, lo + n)
        sub = {k: v[sl] for k, v in meas.items()}
        o2 = eng.factor_eval(model, sub, lin[sl], None if qq is None else qq[sl], states,
                             idx_i=idx_i[sl].contiguous(), idx_j=idx_j[sl].contiguous())
        torch.cuda.synchronize()
        for k in ("err", "H1", "H2"):
            assert torch.equal(o2[k], out[k][sl]), (k, lo)
    again = eng.factor_eval(model, meas, lin, qq, states, idx_i=idx_i, idx_j=idx_j)
    torch.cuda.synchronize()
    for k in ("err", "H1", "H2"):
        assert torch.equal(again[k], out[k]), k
    # oracle on a strided sample, with perturbed state_j so that the residual is not trivially zero
    pick = torch.arange(0, F, 1999, device=eng.device)
    xs = states.clone()
    xs[F:, 4:] += 1e-3
    o3 = eng.factor_eval(model, meas, lin, qq, xs, idx_i=idx_i, idx_j=idx_j)
    torch.cuda.synchronize()
    m = {k: v[pick].cpu().numpy() for k, v in meas.items()}
    rec = orc.factor_records(m, lin[pick].cpu().numpy(), q[pick].cpu().numpy() if model == 2 else None)
    err, H1, H2o = orc.oracle().factor(model, rec, xs[pick].cpu().numpy(), xs[pick + F].cpu().numpy())
    assert np.abs(o3["err"][pick].cpu().numpy() - err).max() <= TOL_FACTOR * max(1.0, np.abs(err).max())
    assert np.abs(o3["H1"][pick].cpu().numpy() - H1).max() <= TOL_FACTOR * max(1.0, np.abs(H1).max())
    assert np.abs(o3["H2"][pick].cpu().numpy() - H2o).max() <= TOL_FACTOR


def test_config5_one_gpu_share_1M_windows_x_100_samples(eng, orc):
    """BASELINE configs[4] is 8 M windows x 100 samples over 8 GPUs: one GPU's share (1 M x 100, 5.7 GB of knots).
    Size-independent checks: DT telescopes, every slice re-run with a different lane split agrees to round-off,
    a strided sample matches the oracle, the launch is deterministic."""
    W, N = 1000000, 100
    kn, lin, q = synth.make_windows(W, N, seed=88, device=eng.device, edge_cases=False)
    prm = eng.make_params(1)
    out = eng.preintegrate(kn, lin, q, prm, want=("mean",))
    torch.cuda.synchronize()
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    assert (out["DT"] - (kn[:, -1, 0] - kn[:, 0, 0])).abs().max().item() < 1e-12
    assert (out["q"].norm(dim=1) - 1).abs().max().item() < 1e-14 and (out["q"][:, 3] >= 0).all()
    for lo, n, lanes in ((0, 10000, 6), (123456, 20000, 3), (W - 5000, 5000, 12)):
        sl = slice(lo, lo + n)
        o = eng.preintegrate(kn[sl], lin[sl], q[sl], eng.make_params(1, lanes_per_window=lanes), want=("mean",))
        torch.cuda.synchronize()
        for k in ("DT", "alpha", "beta", "q"):
            assert (o[k] - out[k][sl]).abs().max().item() < 1e-11, (k, lo)
    pick = torch.arange(0, W, 9973, device=eng.device)
    ref = _cpu_lib(orc).run(orc.make_params(1, 0, 1), kn[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
    check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, ref, what=("mean",), label="1M x 100 sample")
    again = eng.preintegrate(kn, lin, q, prm, want=("mean",))
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(again[k], out[k]), k


def test_config5_one_gpu_share_full_v1_1M_windows_x_100_samples(eng, orc):
    """The FULL-V1 variant of BASELINE configs[4] at one GPU's share (1 M windows x 100 samples: means + analytic bias
    Jacobians + 15x15 covariance, 2.25 GB of outputs).  Size-independent checks: the means equal the mean-only kernel's
    bit for bit over the whole batch, every covariance is exactly symmetric with a positive diagonal, a strided sample of
    128 windows matches the compiled reference at the regression gates, slices re-run on their own agree (bit for bit
    where one kernel configuration serves every batch size), and the launch is deterministic."""
    W, N = 1000000, 100
    kn, lin, q = synth.make_windows(W, N, seed=89, device=eng.device, edge_cases=True)
    prm = eng.make_params(1)
    out = eng.preintegrate(kn, lin, q, prm)
    torch.cuda.synchronize()
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    P = out["P"].view(W, 15, 15)
    assert torch.equal(P, P.transpose(1, 2))
    assert (torch.diagonal(P, dim1=1, dim2=2) > 0).all()
    pick = torch.arange(0, W, 7873, device=eng.device)[:128]
    ref, from_ref = _cpu(orc, (1, 0, 1), kn[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
    check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, ref, label="1M x 100 full V1 sample", regression=from_ref)
    for lo, n in ((0, 4096), (500001, 3001), (W - 1234, 1234)):
        sl = slice(lo, lo + n)
        o = eng.preintegrate(kn[sl].contiguous(), lin[sl].contiguous(), q[sl].contiguous(), prm)
        torch.cuda.synchronize()
        for k in o:
            if k in ("DT", "alpha", "beta", "q", "P"):      # covariance kernel: one lane group per window whatever the batch
                assert torch.equal(o[k], out[k][sl]), (k, lo)
            else:                                            # analytic Jacobians: the lane split follows the batch size
                assert (o[k] - out[k][sl]).abs().max().item() < 1e-11, (k, lo)
    mean_only = eng.preintegrate(kn, lin, q, eng.make_params(1, lanes_per_window=1), want=("mean",))
    torch.cuda.synchronize()
    assert torch.equal(mean_only["DT"], out["DT"])
    for k in ("alpha", "beta", "q"):     # two kernels, two operation orders (column rotation vs prefix products): round-off apart
        assert (mean_only[k] - out[k]).abs().max().item() < 1e-11, k
    del mean_only
    again = eng.preintegrate(kn, lin, q, prm)
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(again[k], out[k]), k


# --------------------------------------------------------------------------- edge sizes and rare branches
@pytest.mark.parametrize("W,N", [(1, 1), (1, 50), (3, 2), (63, 7), (65, 33), (130, 129), (17, 257)])
def test_edge_sizes(eng, orc, W, N):
    """Single windows / single intervals / sizes that straddle every tile, chunk and group boundary
    (mean-kernel chunks, phase-A passes of 8/16 intervals, 4- and 2-window wavefronts)."""
    kn, lin, q = synth.make_windows(W, N, seed=1000 + W + N)
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    for mode in [(1, 0, 1), (2, 0, 1), (2, 1, 0)]:
        ref, from_ref = _cpu(orc, mode, kn, lin, q)
        check_pre(_run(eng, mode, kn, lin, q), ref, v2=(mode[0] == 2), label="W%d N%d %s" % (W, N, mode), regression=from_ref)
    for lanes in (0, 1, 64):
        out = _run(eng, (1, 0, 1), kn, lin, q, want=("mean",), lanes=lanes)
        ref, from_ref = _cpu(orc, (1, 0, 1), kn, lin, q)
        check_pre(out, ref, what=("mean",), regression=from_ref)


def test_large_rotation_angles_take_the_reduced_sincos_path(eng, orc):
    """|w| dt far beyond anything physical (up to ~8 rad per interval): exercises the Cody-Waite range
    reduction of the device sin/cos (cpi_math.hpp: sincos_fast), which no realistic window reaches."""
    kn, lin, q = synth.make_windows(96, 50, seed=77, edge_cases=False)
    kn, lin, q = kn.numpy().copy(), lin.numpy(), q.numpy()
    kn[:, :, 1:4] *= np.linspace(5.0, 400.0, 96)[:, None, None]        # up to ~1000 rad/s
    for mode in [(1, 0, 1), (2, 0, 1)]:
        ref, _ = _cpu(orc, mode, kn, lin, q)
        wdt = np.linalg.norm(kn[:, :-1, 1:4] - lin[:, None, 0:3], axis=2) * np.diff(kn[:, :, 0], axis=1)
        assert wdt.max() > 3.0 and (wdt > 1.0).mean() > 0.3
        out = _run(eng, mode, kn, lin, q)
        # rotations this violent amplify round-off: gate the means at 1e-9 relative to their magnitude
        for k in ("alpha", "beta", "q", "DT"):
            assert np.abs(out[k] - ref[k]).max() <= 1e-9 * max(1.0, np.abs(ref[k]).max()), k
        # Classic RK4 of the Lyapunov equation (eigenvalues = pairwise sums, up to 2|w|) is only stable for
        # |w| dt < 1.39; beyond that the reference's own covariance explodes (up to 1e70) and round-off differences
        # are amplified without bound, so the covariance gate applies to the windows inside the stability region --
        # which still includes reduced-path angles in (1, 1.3).
        wmax = wdt.max(axis=1)
        ok = wmax < 1.3
        assert ok.sum() >= 8 and (wmax[ok] > 1.0).any()
        assert cov_rel_err(out["P"][ok], ref["P"][ok]) <= 1e-6
        m = _run(eng, mode, kn, lin, q, want=("mean",))
        assert np.abs(m["q"] - ref["q"]).max() <= 1e-9


# --------------------------------------------------------------------------- packed evaluateError
@pytest.mark.parametrize("model", [1, 2])
def test_packed_factor_eval_rebuilds_the_dense_pair(eng, model):
    """cpi_factor_eval_packed_batch: the 15 + 54 state-dependent doubles, expanded with the block table of
    include/cpi_amd.h, must reproduce the dense evaluateError output bit for bit (same device arithmetic)."""
    import cpi_amd
    F = 40003                                              # not a multiple of 8: ragged last wavefront
    kn, lin, q = synth.make_windows(F, 50, seed=123, device=eng.device)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac"))
    torch.cuda.synchronize()
    xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
    states = torch.cat([xi, xj[-1:]], dim=0).contiguous()
    qq = q if model == 2 else None
    dense = eng.factor_eval(model, meas, lin, qq, states)
    packed = eng.factor_eval_packed(model, meas, lin, qq, states)
    torch.cuda.synchronize()
    assert packed.shape == (F, 72) and (packed[:, 69:] == 0).all()
    err, H1, H2 = cpi_amd.unpack_factor(packed, meas)
    assert torch.equal(err, dense["err"])
    # -0.0 vs 0.0 in the constant blocks is not a difference
    assert (H1 - dense["H1"]).abs().max().item() == 0.0
    assert (H2 - dense["H2"]).abs().max().item() == 0.0
    # gathered states
    idx_i = torch.randint(0, F, (F,), dtype=torch.int32, device=eng.device)
    idx_j = torch.randint(0, F + 1, (F,), dtype=torch.int32, device=eng.device)
    d2 = eng.factor_eval(model, meas, lin, qq, states, idx_i=idx_i, idx_j=idx_j)
    p2 = eng.factor_eval_packed(model, meas, lin, qq, states, idx_i=idx_i, idx_j=idx_j)
    torch.cuda.synchronize()
    e2, H12, H22 = cpi_amd.unpack_factor(p2, meas)
    assert torch.equal(e2, d2["err"]) and (H12 - d2["H1"]).abs().max().item() == 0.0 and (H22 - d2["H2"]).abs().max().item() == 0.0


# --------------------------------------------------------------------------- randomized shapes
def test_fuzz_random_shapes_layouts_and_lane_splits(eng, orc):
    """40 random (W, N, model, imu_avg, Jacobian mode, dense | ragged, lanes per window) combinations against the
    oracle: every supported lane split meets ragged tails, empty windows, short windows (N < L) and both
    staging variants of the mean kernel (one / two knots per chunk)."""
    rng = np.random.default_rng(20240917)
    lanes_all = [0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64]
    for case in range(40):
        W = int(rng.integers(1, 260))
        N = int(rng.integers(1, 90))
        model = int(rng.integers(1, 3))
        avg = int(rng.integers(0, 2))
        stj = int(rng.integers(0, 2)) if model == 2 else 1
        ragged = bool(rng.integers(0, 2))
        lanes = int(rng.choice(lanes_all))      # model 2 honours it for mean-only requests (segment form), else ignores it
        label = "case %d W%d N%d m%d avg%d stj%d %s L%d" % (case, W, N, model, avg, stj, "ragged" if ragged else "dense", lanes)
        oprm = orc.make_params(model, avg, stj)
        prm = eng.make_params(model, avg, stj, lanes_per_window=lanes)
        want = ("mean",) if case % 3 != 1 else ("mean", "jac", "cov")
        if not ragged:
            kn, lin, q = synth.make_windows(W, N, seed=5000 + case)
            kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
            ref = _cpu_lib(orc).run(oprm, kn, lin, q)
            out = _host(eng.preintegrate(_dev(kn, eng), _dev(lin, eng), _dev(q, eng), prm, want=want))
        else:
            lens = rng.integers(0, N + 1, W).astype(np.int32)
            lens[rng.integers(0, W)] = N
            K = int(lens.sum()) + 1
            kn1, _, _ = synth.make_windows(1, max(K - 1, 1), seed=6000 + case, edge_cases=False)
            stream = kn1.numpy()[0][:K]
            first = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            _, lin, q = synth.make_windows(W, 4, seed=7000 + case)
            lin, q = lin.numpy(), q.numpy()
            out = _host(eng.preintegrate(_dev(stream, eng), _dev(lin, eng), _dev(q, eng), prm, want=want,
                                         first=_dev(first, eng), count=_dev(lens, eng), N=N))
            ref = None
            for w in range(W):
                n = int(lens[w])
                r = _cpu_lib(orc).run(oprm, stream[first[w]:first[w] + n + 1][None], lin[w:w + 1], q[w:w + 1])
                if ref is None:
                    ref = {k: np.zeros((W,) + v.shape[1:]) for k, v in r.items()}
                for k in ref:
                    ref[k][w] = r[k][0]
        check_pre(out, ref, what=want, v2=(model == 2), label=label)


@pytest.mark.parametrize("dt_scale,w_scale,a_scale", [(0.02, 1.0, 1.0), (1.0, 4.0, 1.0), (4.0, 1.0, 1.0), (1.0, 1.0, 40.0),
                                                      (0.2, 8.0, 10.0)])
def test_dynamic_range_stress(eng, orc, dt_scale, w_scale, a_scale):
    """Sampling rates from 50 Hz to 10 kHz, rates up to ~20 rad/s, specific forces up to ~500 m/s^2 (all inside the
    stability region |w| dt < 1.3 of the reference's RK4): relative parity must not depend on the scales."""
    W = 600
    kn, lin, q = synth.make_windows(W, 50, seed=31337, edge_cases=False)
    kn = kn.clone()
    t0 = kn[:, :1, 0].clone()
    kn[:, :, 0] = t0 + (kn[:, :, 0] - t0) * dt_scale
    kn[:, :, 1:4] *= w_scale
    kn[:, :, 4:7] *= a_scale
    lin = lin.clone()
    lin[:, 0:3] *= w_scale
    lin[:, 3:6] *= a_scale
    kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
    assert (np.abs(kn[:, :, 1:4]).max() * np.diff(kn[:, :, 0], axis=1).max()) < 1.3
    for mode in [(1, 0, 1), (2, 0, 1), (1, 1, 1)]:
        ref, _ = _cpu(orc, mode, kn, lin, q)
        out = _run(eng, mode, kn, lin, q)
        # means: relative to the magnitude of the quantity (alpha grows with a dt^2 N^2)
        for k in ("DT", "alpha", "beta", "q"):
            scale = max(1.0, float(np.abs(ref[k]).max()))
            assert np.abs(out[k] - ref[k]).max() <= TOL_MEAN * scale, (mode, k)
        assert cov_rel_err(out["P"], ref["P"]) <= TOL_COV, mode
        for k in ("J_q", "J_a", "J_b", "H_a", "H_b"):
            scale = max(1.0, float(np.abs(ref[k]).max()))
            assert np.abs(out[k] - ref[k]).max() <= 1e-8 * scale, (mode, k)


def test_engine_pool_overlapping_contexts_give_the_same_results(eng):
    """Independent batches issued round-robin through three contexts (cpi_amd.EnginePool): bit-identical to the single
    context, whatever the interleaving; allocations made on the pool's streams are handed back through result()."""
    import cpi_amd
    pool = cpi_amd.EnginePool(3)
    prm = eng.make_params(1)
    batches = [synth.make_windows(500 + 37 * b, 50, seed=4000 + b, device=eng.device) for b in range(7)]
    jobs = [pool.preintegrate(kn, lin, q, prm, want=("mean", "jac", "cov") if b % 2 else ("mean",))
            for b, (kn, lin, q) in enumerate(batches)]
    for b, (kn, lin, q) in enumerate(batches):
        got = jobs[b].result()
        ref = eng.preintegrate(kn, lin, q, prm, want=("mean", "jac", "cov") if b % 2 else ("mean",))
        torch.cuda.synchronize()
        assert set(got) == set(ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]), (b, k)
    pool.synchronize()
    pool.close()


@pytest.mark.parametrize("model", [1, 2, 3])
def test_maximum_window_length(eng, orc, model):
    """N = 65535 intervals per window is the documented maximum (16-bit segment lengths in the mean kernel): three
    such windows, mean-only with one and with 64 lanes per window, and the full recursion; N = 65536 is refused."""
    import cpi_amd
    N = 65535
    kn, lin, q = synth.make_windows(3, N, seed=99, edge_cases=False)
    kn = kn.clone()
    kn[:, :, 1:4] *= 0.05                      # keep the 5-minute windows' rotation and drift moderate
    knn, linn, qn = kn.numpy(), lin.numpy(), q.numpy()
    oprm = orc.make_params(model, 0, 1)
    ref = (orc.oracle() if model == 3 else _cpu_lib(orc)).run(oprm, knn, linn, qn, nthreads=3)
    # means grow with a t^2 / 2 over 328 s: gate relative to the magnitude, as the stress test does
    def close(out, what):
        for k in what:
            scale = max(1.0, float(np.abs(ref[k]).max()))
            assert np.abs(out[k] - ref[k]).max() <= 1e-9 * scale, (model, k, np.abs(out[k] - ref[k]).max(), scale)
    for lanes in ((1, 64) if model < 3 else (0,)):
        prm = eng.make_params(model, lanes_per_window=lanes)
        out = _host(eng.preintegrate(kn.to(eng.device), lin.to(eng.device), q.to(eng.device), prm, want=("mean",)))
        close(out, ("DT", "alpha", "beta", "q"))
    out = _host(eng.preintegrate(kn.to(eng.device), lin.to(eng.device), q.to(eng.device), eng.make_params(model)))
    close(out, ("DT", "alpha", "beta", "q"))
    assert cov_rel_err(out["P"], ref["P"]) <= TOL_COV
    big = torch.zeros((1, 65537, 7), dtype=torch.float64, device=eng.device)
    with pytest.raises(cpi_amd.CpiError):
        eng.preintegrate(big, lin[:1].to(eng.device), q[:1].to(eng.device), eng.make_params(1), want=("mean",))


# --------------------------------------------------------------------------- NaN-separated (non-chained) feed_IMU sequences
@pytest.mark.parametrize("mode", MODES)
def test_nan_separated_non_chained_intervals(eng, orc, mode):
    """The reference's feed_IMU only ever uses t_1 - t_0, so successive calls need not chain; the facades express such a
    window with NaN-time separator knots (both intervals touching a separator are skipped).  Mean, analytic-Jacobian and
    covariance kernels -- every lane split of the mean kernel -- against the reference driven by the same knot array
    (its caller's `dt >= 0` guard, GraphSolver_IMU.cpp:52, skips the NaN intervals as well)."""
    W, N = 300, 61
    kn, lin, q = synth.make_windows(W, N, seed=4321 + mode[0] + 2 * mode[1], edge_cases=False)
    kn, lin, q = kn.numpy().copy(), lin.numpy(), q.numpy()
    rng = np.random.default_rng(17)
    for w in range(W):
        nsep = int(rng.integers(0, 6))
        pos = rng.choice(np.arange(1, N), size=nsep, replace=False)
        kn[w, pos, 0] = np.nan
        kn[w, pos, 1:] = 0.0                              # separators carry zero readings (what the facades emit)
        if w % 7 == 0:
            kn[w, N, 0] = np.nan; kn[w, N, 1:] = 0.0       # a separator as the last knot
        if w % 11 == 0:
            kn[w, 0, 0] = np.nan; kn[w, 0, 1:] = 0.0       # ... and as the first
    ref, from_ref = _cpu(orc, mode, kn, lin, q)
    assert np.all(np.isfinite(ref["alpha"])) and np.all(np.isfinite(ref["P"]))
    out = _run(eng, mode, kn, lin, q)
    check_pre(out, ref, v2=(mode[0] == 2), label="nan-separated %s" % (mode,), regression=from_ref)
    if mode[2] == 1:
        for lanes in (1, 2, 5, 8, 16, 64):
            o2 = _run(eng, mode, kn, lin, q, want=("mean",), lanes=lanes)
            check_pre(o2, ref, what=("mean",), label="nan-separated mean L%d" % lanes, regression=from_ref)


@pytest.mark.parametrize("avg", [False, True])
def test_facade_feed_imu_with_gaps_between_calls(eng, orc, avg):
    """cpi_amd.CpiV1 / CpiV2.feed_IMU with calls that do not chain in time (a gap after every third call) and -- with
    imu_avg -- closing readings that differ from the next opening reading: NaN separators are inserted by the facade.
    Without imu_avg and with chaining times the closing readings are unused and every call costs exactly one knot."""
    import cpi_amd
    rng = np.random.default_rng(5)
    for cls, model in ((cpi_amd.CpiV1, 1), (cpi_amd.CpiV2, 2)):
        c = cls(0.005, 4e-6, 0.01, 2e-4, avg, engine=eng)
        bw, ba = 0.01 * rng.standard_normal(3), 0.05 * rng.standard_normal(3)
        qk = rng.standard_normal(4); qk /= np.linalg.norm(qk); qk *= np.sign(qk[3])
        c.setLinearizationPoints(bw, ba, qk, np.array([0, 0, 9.8]))
        t, calls = 100.0, []
        for k in range(40):
            w0, a0 = rng.standard_normal(3), np.array([0, 0, 9.8]) + rng.standard_normal(3)
            w1, a1 = (rng.standard_normal(3), np.array([0, 0, 9.8]) + rng.standard_normal(3)) if avg else (None, None)
            calls.append((t, t + 0.005, w0, a0, w1, a1))
            t += 0.005 + (0.1 if k % 3 == 2 else 0.0)
        for (t0, t1, w0, a0, w1, a1) in calls:
            c.feed_IMU(t0, t1, w0, a0, w1, a1)
        knots = c._knots()
        if not avg:
            assert knots.shape[0] == 40 + 1 + 2 * 13      # one knot per call + the first + (separator, opener) per gap
        ref = _cpu_lib(orc).run(orc.make_params(model, int(avg), 1), knots[None], np.concatenate([bw, ba])[None], qk[None])
        assert abs(c.DT - 40 * 0.005) < 1e-12
        assert np.abs(c.alpha_tau - ref["alpha"][0]).max() < 2e-13 and np.abs(c.beta_tau - ref["beta"][0]).max() < 2e-13
        assert np.abs(c.q_k2tau - ref["q"][0]).max() < 2e-13
        assert cov_rel_err(c.P_meas.T.reshape(1, 225), ref["P"][0:1]) < 1e-12
        assert np.abs(c.J_a.T.reshape(9) - ref["J_a"][0]).max() < 3e-11


def test_dense_layout_with_counts_ignores_the_unwritten_tail_knots(eng, orc):
    """Dense layout knots[W][N+1][7] with per-window counts: the knots behind a window's last interval are the caller's
    to leave unwritten.  NaN / Inf there must not reach any output of any kernel (mean kernel at every lane split, analytic
    Jacobians, covariance of both models, the Forster comparator)."""
    W, N = 517, 40
    kn, lin, q = synth.make_windows(W, N, seed=246, edge_cases=False)
    kn, lin, q = kn.numpy().copy(), lin.numpy(), q.numpy()
    rng = np.random.default_rng(8)
    lens = rng.integers(0, N + 1, W).astype(np.int32)
    lens[:4] = [0, 1, N, N - 1]
    for w in range(W):
        kn[w, lens[w] + 1:, :] = np.nan if w % 2 else np.inf
    for mode in [(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 0)]:
        ref = {}
        oprm = orc.make_params(*mode)
        rows = [_cpu_lib(orc).run(oprm, kn[w:w + 1, :lens[w] + 1], lin[w:w + 1], q[w:w + 1]) for w in range(W)]
        for k in rows[0]:
            ref[k] = np.concatenate([r[k] for r in rows], axis=0)
        prm = eng.make_params(*mode)
        out = _host(eng.preintegrate(_dev(kn, eng), _dev(lin, eng), _dev(q, eng), prm, count=_dev(lens, eng), N=N))
        for k, v in out.items():
            assert np.all(np.isfinite(v)), (mode, k)
        check_pre(out, ref, v2=(mode[0] == 2), label="dense+count %s" % (mode,), regression=orc.reference() is not None)
        for lanes in (0, 1, 2, 3, 6, 8, 16, 64):
            prm = eng.make_params(*mode, lanes_per_window=lanes)
            o2 = _host(eng.preintegrate(_dev(kn, eng), _dev(lin, eng), _dev(q, eng), prm, want=("mean",), count=_dev(lens, eng), N=N))
            assert all(np.all(np.isfinite(v)) for v in o2.values()), (mode, lanes)
            check_pre(o2, ref, what=("mean",), label="dense+count mean L%d %s" % (lanes, mode))
    fo = _host(eng.preintegrate(_dev(kn, eng), _dev(lin, eng), None, eng.make_params(3), count=_dev(lens, eng), N=N))
    assert all(np.all(np.isfinite(v)) for v in fo.values())


def test_calls_are_capturable_into_a_hip_graph(eng):
    """The device-pointer entries only enqueue kernels on the context's stream (no allocation, no synchronisation), and the
    Python engine follows torch's current stream: a caller can capture a re-linearisation round -- preintegration, square-root
    information, whitened evaluateError -- into ONE HIP graph and replay it.  Replays must reproduce the eager results bit
    for bit, also after the inputs changed in place."""
    W = 2000
    kn, lin, q = synth.make_windows(W, 50, seed=404, device=eng.device)
    prm = eng.make_params(2)
    meas = eng.alloc_outputs(W, ("mean", "jac", "cov"), 2)
    xi = torch.zeros((W + 1, 16), dtype=torch.float64, device=eng.device); xi[:, 3] = 1.0
    out = {"err": torch.empty((W, 15), dtype=torch.float64, device=eng.device),
           "H1": torch.empty((W, 225), dtype=torch.float64, device=eng.device),
           "H2": torch.empty((W, 225), dtype=torch.float64, device=eng.device)}
    R = torch.empty((W, 225), dtype=torch.float64, device=eng.device)

    prm1 = eng.make_params(1)
    meas1 = eng.alloc_outputs(W, ("mean", "jac", "cov"), 1)    # model 1, everything: two kernels forked onto a side stream
    out1 = {"err": torch.empty((W, 15), dtype=torch.float64, device=eng.device)}

    def round_():
        eng.preintegrate(kn, lin, q, prm, out=meas)
        eng.lib.cpi_sqrt_information_batch(eng.ctx, W, meas["P"].data_ptr(), R.data_ptr())
        eng.factor_eval(2, meas, lin, q, xi, out=out, sqrt_info=R)
        eng.preintegrate(kn, lin, q, prm1, out=meas1)
        eng.factor_eval(1, meas1, lin, None, xi, out=out1, want_H=False)
        out["err1"] = out1["err"]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        round_()                                            # warm-up on the side stream, as graph capture requires
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    eager = {k: v.clone() for k, v in out.items()}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        round_()
    for v in out.values():
        v.zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(out[k], eager[k]), k
    kn[:, :, 1:4] *= 1.01                                   # new measurements in the same buffers
    g.replay()
    torch.cuda.synchronize()
    replayed = {k: v.clone() for k, v in out.items()}
    round_()
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(out[k], replayed[k]) and not torch.equal(out[k], eager[k]), k
