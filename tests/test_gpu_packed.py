"""GPU: the packed triangles of ABI 3 (include/cpi_amd.h: CPI_TRI_INDEX) -- P_meas as its upper triangle out of the covariance /
Forster kernels (cpi_outputs.P_sym), R as its non-zero triangle through cpi_sqrt_information_packed_batch,
cpi_factor_eval_whitened_tri_batch and cpi_factor_hessian_tri_batch.  The bar: value for value the SAME BITS as the dense
drop-in forms (ImuFactorCPIv1.h:82 takes the dense covariance; CpiV1.h:352-353 is where the reference asserts P = P^T), on
the reference's golden vectors, on ragged grids F = 1 ... 700 and on a strided sample of a 1 M launch."""
import os

import numpy as np
import pytest
import torch

import cpi_amd
from cpi_amd import synth
from tests.tol import check_pre

pytestmark = pytest.mark.gpu
MODES = [(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 1), (2, 0, 0), (2, 1, 0)]


@pytest.fixture(scope="module")
def eng():
    return cpi_amd.Engine()


def _dev(a, eng):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


def test_pack_helpers_roundtrip(eng):
    """The Python mirrors of CPI_TRI_INDEX: entry (i, j), i <= j, at i + j (j + 1) / 2."""
    g = torch.Generator().manual_seed(3)
    A = torch.randn((5, 15, 15), dtype=torch.float64, generator=g)
    S = (A + A.transpose(1, 2)).transpose(1, 2).reshape(5, 225).to(eng.device)        # column-major flat (symmetric: either order)
    Ps = cpi_amd.pack_sym(S)
    assert Ps.shape == (5, 120)
    for i, j in ((0, 0), (0, 14), (3, 7), (14, 14), (6, 6), (1, 2)):
        assert torch.equal(Ps[:, i + j * (j + 1) // 2], S[:, j * 15 + i])
    assert torch.equal(cpi_amd.unpack_sym(Ps), S)
    U = torch.triu(A).transpose(1, 2).reshape(5, 225).to(eng.device)                  # upper triangular, column-major
    assert torch.equal(cpi_amd.unpack_tri(cpi_amd.pack_tri(U)), U)


@pytest.mark.parametrize("fname", ["pre_cfg1.npz", "pre_w48.npz"])
@pytest.mark.parametrize("mode", MODES)
def test_P_sym_on_the_reference_goldens(eng, golden_dir, fname, mode):
    """P_sym against the COMPILED REFERENCE's covariance (regression gate), bitwise against the dense P of the same call and of a
    call that asks for P alone, and alone (no dense P allocated at all)."""
    d = dict(np.load(os.path.join(golden_dir, fname)))
    key = "m%d_avg%d_stj%d__" % mode
    ref = {k[len(key):]: v for k, v in d.items() if k.startswith(key)}
    prm = eng.make_params(*mode)
    kn, lin, q = _dev(d["knots"], eng), _dev(d["lin"], eng), _dev(d["q_k_lin"], eng)
    both = eng.preintegrate(kn, lin, q, prm, want=("mean", "jac", "cov", "cov_sym"))
    dense = eng.preintegrate(kn, lin, q, prm, want=("mean", "jac", "cov"))
    alone = eng.preintegrate(kn, lin, q, prm, want=("cov_sym",))
    torch.cuda.synchronize()
    assert set(alone) == {"P_sym"} and both["P_sym"].shape == (kn.shape[0], 120)
    assert torch.equal(both["P"], dense["P"])
    assert torch.equal(both["P_sym"], cpi_amd.pack_sym(dense["P"]))
    assert torch.equal(alone["P_sym"], both["P_sym"])
    # the dense matrix the kernels write IS symmetric bit for bit, so unpacking reproduces it whole
    assert torch.equal(cpi_amd.unpack_sym(both["P_sym"]), dense["P"])
    out = {k: v.cpu().numpy() for k, v in both.items()}
    out["P"] = cpi_amd.unpack_sym(both["P_sym"]).cpu().numpy()
    check_pre(out, ref, v2=(mode[0] == 2), label="%s %s packed" % (fname, mode), regression=True)


@pytest.mark.parametrize("model", [1, 2, 3])
@pytest.mark.parametrize("W,N", [(1, 1), (3, 2), (5, 50), (63, 7), (130, 33), (1003, 50)])
def test_P_sym_ragged_grids_every_model(eng, model, W, N):
    """Partial wavefronts (4 windows per wavefront for models 1 / 3, 2 for model 2), per-window counts, nothing written past W."""
    kn, lin, q = synth.make_windows(W, N, seed=700 + W + N, device=eng.device)
    cnt = torch.randint(0, N + 1, (W,), dtype=torch.int32, device=eng.device)
    prm = eng.make_params(model)
    qq = q if model != 3 else None
    dense = eng.preintegrate(kn, lin, qq, prm, want=("mean", "jac", "cov"), count=cnt)
    big = torch.full((W + 2, 120), -7.0, dtype=torch.float64, device=eng.device)
    out = eng.alloc_outputs(W, ("mean", "cov_sym"), model)
    out["P_sym"] = big[:W]
    eng.preintegrate(kn, lin, qq, prm, want=("mean", "cov_sym"), count=cnt, out=out)
    torch.cuda.synchronize()
    assert torch.all(big[W:] == -7.0)
    assert torch.equal(big[:W], cpi_amd.pack_sym(dense["P"]))
    if model != 3:
        # models 1 / 2: the recursion forms k = M + M^T entry by entry (the reference's P = (P + P^T) / 2, CpiV1.h:352-353), so the
        # dense matrix IS symmetric bit for bit and unpacking reproduces all 225 entries
        assert torch.equal(cpi_amd.unpack_sym(big[:W]), dense["P"])
    else:
        # the Forster comparator (GTSAM's A P A^T + B Q B^T, never symmetrised there either) is symmetric to rounding only:
        # P_sym holds the upper triangle of the dense result, the mirrored half differs from the dense lower half by <= a few ulp
        from tests.tol import cov_rel_err
        assert cov_rel_err(cpi_amd.unpack_sym(big[:W]).cpu().numpy(), dense["P"].cpu().numpy()) <= 1e-13
    for k in ("DT", "alpha", "beta", "q"):
        assert torch.equal(out[k], dense[k])


def test_P_sym_through_the_stream_entry(eng):
    stream, upd, lin, q = synth.make_stream(500, 17, seed=44, device=eng.device, phase=0.37)
    for model in (1, 2, 3):
        prm = eng.make_params(model)
        qq = q if model != 3 else None
        dense = eng.preintegrate_stream(stream, upd, lin, qq, prm, N=18)
        packed = eng.preintegrate_stream(stream, upd, lin, qq, prm, N=18, want=("mean", "jac", "cov_sym"))
        torch.cuda.synchronize()
        assert "P" not in packed
        assert torch.equal(packed["P_sym"], cpi_amd.pack_sym(dense["P"]))
        if model != 3:
            assert torch.equal(cpi_amd.unpack_sym(packed["P_sym"]), dense["P"])
        assert torch.equal(packed["alpha"], dense["alpha"])


def _sweep_inputs(eng, F, model, N=50, seed=91):
    kn, lin, q = synth.make_windows(F, N, seed=seed, device=eng.device, edge_cases=False)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac", "cov", "cov_sym"))
    xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
    states = torch.cat([xi, xj[-1:]], dim=0).contiguous()
    return meas, lin, (q if model == 2 else None), states


@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 7, 9, 64, 257, 700, 703])
def test_sqrt_information_packed_is_the_dense_result_bit_for_bit(eng, F):
    meas, _, _, _ = _sweep_inputs(eng, F, 1, N=20, seed=300 + F)
    R = eng.sqrt_information(meas["P"])
    big = torch.full((F + 3, 120), -7.0, dtype=torch.float64, device=eng.device)
    eng.sqrt_information(meas["P_sym"], out=big[:F])
    torch.cuda.synchronize()
    assert torch.all(big[F:] == -7.0)
    assert torch.equal(big[:F], cpi_amd.pack_tri(R))
    assert torch.equal(cpi_amd.unpack_tri(big[:F]), R)       # the dense form's strict lower part is zeros


def test_sqrt_information_packed_not_positive_definite(eng):
    """A matrix that is not positive definite poisons ITS factor with NaNs and no other factor of the wavefront -- as the dense entry."""
    rng = np.random.default_rng(5)
    F = 9
    A = rng.standard_normal((F, 15, 15))
    P = A @ A.transpose(0, 2, 1) + 15.0 * np.eye(15)
    P[4] = np.eye(15)
    P[6, 3, 3] = -1.0
    Pd = torch.tensor(P.reshape(F, 225), device=eng.device)
    Rt = eng.sqrt_information(cpi_amd.pack_sym(Pd))
    R = eng.sqrt_information(Pd)
    torch.cuda.synchronize()
    ok = [f for f in range(F) if f != 6]
    assert torch.equal(Rt[ok], cpi_amd.pack_tri(R)[ok])
    assert torch.isnan(Rt[6]).any()
    assert torch.equal(cpi_amd.unpack_tri(Rt[4:5]).reshape(15, 15), torch.eye(15, dtype=torch.float64, device=eng.device))


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("F", [1, 3, 4, 5, 13, 700, 1003])
def test_whitened_and_hessian_sweeps_with_packed_R_are_bit_identical(eng, model, F):
    meas, lin, qq, states = _sweep_inputs(eng, F, model, N=20, seed=500 + F)
    R = eng.sqrt_information(meas["P"])
    Rt = eng.sqrt_information(meas["P_sym"])
    m = {k: v for k, v in meas.items() if k not in ("P", "P_sym")}
    dense = eng.factor_eval(model, m, lin, qq, states, sqrt_info=R)
    big = {"err": torch.full((F + 2, 15), -7.0, dtype=torch.float64, device=eng.device),
           "H1": torch.full((F + 2, 225), -7.0, dtype=torch.float64, device=eng.device),
           "H2": torch.full((F + 2, 225), -7.0, dtype=torch.float64, device=eng.device)}
    eng.factor_eval(model, m, lin, qq, states, sqrt_info=Rt, out={k: v[:F] for k, v in big.items()})
    hd = eng.factor_hessian(model, m, lin, qq, states, R)
    hbig = torch.full((F + 2, 496), -7.0, dtype=torch.float64, device=eng.device)
    eng.factor_hessian(model, m, lin, qq, states, Rt, out=hbig[:F])
    torch.cuda.synchronize()
    for k in ("err", "H1", "H2"):
        assert torch.all(big[k][F:] == -7.0)
        assert torch.equal(big[k][:F], dense[k]), (model, F, k)
    assert torch.all(hbig[F:] == -7.0)
    assert torch.equal(hbig[:F], hd), (model, F)
    # gathered (non-chained) state indices and err only (H1 = H2 = NULL)
    if F >= 5:
        g = torch.Generator().manual_seed(F)
        ii = torch.randint(0, F + 1, (F,), dtype=torch.int32, generator=g).to(eng.device)
        jj = torch.randint(0, F + 1, (F,), dtype=torch.int32, generator=g).to(eng.device)
        a = eng.factor_eval(model, m, lin, qq, states, idx_i=ii, idx_j=jj, sqrt_info=R)
        b = eng.factor_eval(model, m, lin, qq, states, idx_i=ii, idx_j=jj, sqrt_info=Rt)
        c = eng.factor_eval(model, m, lin, qq, states, idx_i=ii, idx_j=jj, sqrt_info=Rt, want_H=False)
        torch.cuda.synchronize()
        assert all(torch.equal(a[k], b[k]) for k in a) and torch.equal(c["err"], a["err"])


def test_packed_forms_on_a_strided_sample_of_a_1M_launch(eng):
    """The bench rows' size: 1 M factors.  The packed pipeline (P_sym -> R_tri -> whitened / Hessian) on the whole batch against
    the dense pipeline on every 7 919th factor (128 + factors, dense forms rebuilt from the packed ones for the comparison)."""
    F, N = 1000000, 10
    kn, lin, q = synth.make_windows(F, N, seed=77, device=eng.device, edge_cases=False)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(1), want=("mean", "jac", "cov_sym"))
    del kn
    xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, 1, device=eng.device)
    states = torch.cat([xi, xj[-1:]], dim=0).contiguous()
    del xi, xj
    Rt = eng.sqrt_information(meas["P_sym"])
    m = {k: v for k, v in meas.items() if k != "P_sym"}
    hess = eng.factor_hessian(1, m, lin, None, states, Rt)
    white = eng.factor_eval(1, m, lin, None, states, sqrt_info=Rt)
    torch.cuda.synchronize()
    sel = torch.arange(0, F, 7919, device=eng.device)
    ms = {k: v[sel].contiguous() for k, v in m.items()}
    ii = sel.to(torch.int32)
    jj = (sel + 1).to(torch.int32)
    Pd = cpi_amd.unpack_sym(meas["P_sym"][sel].contiguous())
    Rd = eng.sqrt_information(Pd)
    assert torch.equal(cpi_amd.pack_tri(Rd), Rt[sel])
    wd = eng.factor_eval(1, ms, lin[sel].contiguous(), None, states, idx_i=ii, idx_j=jj, sqrt_info=Rd)
    hd = eng.factor_hessian(1, ms, lin[sel].contiguous(), None, states, Rd, idx_i=ii, idx_j=jj)
    torch.cuda.synchronize()
    for k in ("err", "H1", "H2"):
        assert torch.equal(white[k][sel], wd[k]), k
    assert torch.equal(hess[sel], hd)


def test_bad_arguments(eng):
    z = torch.zeros((4, 120), dtype=torch.float64, device=eng.device)
    lib = eng.lib
    assert lib.cpi_sqrt_information_packed_batch(eng.ctx, 4, None, z.data_ptr()) == 1
    assert lib.cpi_sqrt_information_packed_batch(eng.ctx, -1, z.data_ptr(), z.data_ptr()) == 1
    assert lib.cpi_sqrt_information_packed_batch(eng.ctx, 0, None, None) == 0
    meas, lin, qq, states = _sweep_inputs(eng, 4, 1, N=5)
    with pytest.raises(cpi_amd.CpiError):
        m = eng._outputs_struct(meas)
        import ctypes as C
        g = (C.c_double * 3)(0, 0, 9.8)
        e = torch.zeros((4, 15), dtype=torch.float64, device=eng.device)
        eng._check(lib.cpi_factor_eval_whitened_tri_batch(eng.ctx, 1, g, 4, C.byref(m), lin.data_ptr(), None, states.data_ptr(), 5, None, None,
                                                          None, e.data_ptr(), None, None))
    # the tiled (mean-only) entry refuses a covariance request in either layout
    tiles = eng.tile_knots(synth.make_windows(64, 5, seed=1, device=eng.device)[0])
    out = eng.alloc_outputs(64, ("mean", "cov_sym"), 1)
    with pytest.raises(cpi_amd.CpiError):
        eng.preintegrate_tiled(tiles, 64, lin.new_zeros((64, 6)), None, eng.make_params(1), out=out)
