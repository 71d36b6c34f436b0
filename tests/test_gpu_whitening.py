"""GPU: SURVEY.md section 8 row f1 -- noise-model whitening (parity UNPINNED: the arithmetic lives in GTSAM @ c21186c6,
absent from the reference tree).  GTSAM's published algorithm is restated with numpy/LAPACK as the checker:
noiseModel::Gaussian::Covariance(P) -> Information(P^-1) -> R = LLT(P^-1).matrixU(), and
NoiseModelFactor::linearize -> Gaussian::WhitenSystem: e <- R e, H <- R H."""
import numpy as np
import pytest
import torch

from cpi_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import cpi_amd
    return cpi_amd.Engine()


@pytest.mark.parametrize("model", [1, 2])
def test_sqrt_information_and_whitened_sweep(eng, model):
    F = 1003                                              # not a multiple of 4: ragged tail of the grid
    kn, lin, q = synth.make_windows(F, 50, seed=91, device=eng.device, edge_cases=False)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(model))
    R = eng.sqrt_information(meas["P"])
    xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
    states = torch.cat([xi, xj[-1:]], dim=0).contiguous()
    qq = q if model == 2 else None
    plain = eng.factor_eval(model, meas, lin, qq, states)
    white = eng.factor_eval(model, meas, lin, qq, states, sqrt_info=R)
    torch.cuda.synchronize()
    P = meas["P"].cpu().numpy().reshape(F, 15, 15)        # symmetric: storage order irrelevant
    Rg = R.cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)   # column-major -> [row][col]
    # (1) structure: upper triangular with positive diagonal
    assert np.all(np.tril(Rg, -1) == 0.0)
    assert np.all(np.diagonal(Rg, axis1=1, axis2=2) > 0)
    # (2) R^T R = P^-1  <=>  R P R^T = I
    I = np.einsum("fij,fjk,flk->fil", Rg, P, Rg)
    assert np.abs(I - np.eye(15)).max() < 1e-7
    # (3) equals LAPACK's chol_upper(inv(P)) (unique), up to the conditioning of P (~1e8)
    Rn = np.stack([np.linalg.cholesky(np.linalg.inv(P[f])).T for f in range(F)])
    scale = np.abs(Rn).max(axis=(1, 2))[:, None, None]
    assert (np.abs(Rg - Rn) / scale).max() < 1e-6
    # (3b) regression gate ("nothing moved"): against the same factorisation carried out in longdouble, relative to the
    # largest entry of the factor's R -- measured floor 3.5e-16 on MI355X (LAPACK's f64 result itself: 2.3e-16)
    from tests.tol import REG_SQRT_INFO, sqrt_info_longdouble
    Rl = sqrt_info_longdouble(P)
    assert (np.abs(Rg - Rl) / np.abs(Rl).max(axis=(1, 2))[:, None, None]).max() < REG_SQRT_INFO
    # (4) whitened sweep = R @ unwhitened sweep
    e = plain["err"].cpu().numpy(); H1 = plain["H1"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
    H2 = plain["H2"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
    ew = white["err"].cpu().numpy(); H1w = white["H1"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
    H2w = white["H2"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
    for got, ref in ((ew, np.einsum("fij,fj->fi", Rg, e)), (H1w, Rg @ H1), (H2w, Rg @ H2)):
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # (5) the whitened residual's squared norm is the Mahalanobis distance e^T P^-1 e
    maha = np.einsum("fi,fij,fj->f", e, np.linalg.inv(P), e)
    assert np.abs((ew ** 2).sum(1) - maha).max() <= 1e-6 * max(1.0, maha.max())


def test_sqrt_information_well_conditioned_and_not_positive_definite(eng):
    """Well-conditioned random SPD matrices must agree with LAPACK to rounding; a matrix that is not positive
    definite poisons ITS factor with NaNs (include/cpi_amd.h) and no other factor of the wavefront."""
    rng = np.random.default_rng(5)
    F = 9                                                  # 2 wavefronts of 4 factors + a tail of 1
    A = rng.standard_normal((F, 15, 15))
    P = A @ A.transpose(0, 2, 1) + 15.0 * np.eye(15)
    P[4] = np.eye(15)                                      # identity -> identity
    P[6, 3, 3] = -1.0                                      # indefinite
    R = eng.sqrt_information(torch.tensor(P.reshape(F, 225), device=eng.device))
    torch.cuda.synchronize()
    Rg = R.cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
    ok = [f for f in range(F) if f != 6]
    Rn = np.stack([np.linalg.cholesky(np.linalg.inv(P[f])).T for f in ok])
    assert np.abs(Rg[ok] - Rn).max() < 1e-13                # well-conditioned: LAPACK is the reference, to rounding
    assert np.array_equal(Rg[4], np.eye(15))
    assert np.isnan(Rg[6]).any()


@pytest.mark.parametrize("F", [1, 2, 3, 5, 6, 7, 12, 13, 700, 703])
def test_hessian_blocks_ragged_grid_and_unwritten_neighbours(eng, F):
    """Four factors per wavefront (16 lanes = one DPP row each), the last wavefront ragged -- F % 4 = 1, 2, 3 and 0, i.e. every
    partial-wavefront shape of the shadow-lane / DPP path: every factor complete, nothing written past F (the output buffer
    is allocated larger and pre-filled), chained states, both models; equal to the F = whole-batch run bit for bit."""
    kn, lin, q = synth.make_windows(F, 20, seed=300 + F, device=eng.device, edge_cases=False)
    for model in (1, 2):
        meas = eng.preintegrate(kn, lin, q, eng.make_params(model))
        R = eng.sqrt_information(meas["P"])
        xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
        states = torch.cat([xi, xj[-1:]], dim=0).contiguous()
        qq = q if model == 2 else None
        big = torch.full((F + 3, 496), -7.0, dtype=torch.float64, device=eng.device)
        eng.factor_hessian(model, meas, lin, qq, states, R, out=big[:F])
        white = eng.factor_eval(model, meas, lin, qq, states, sqrt_info=R)
        torch.cuda.synchronize()
        assert torch.all(big[F:] == -7.0)
        A1 = white["H1"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
        A2 = white["H2"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
        Ab = np.concatenate([A1, A2, -white["err"].cpu().numpy()[:, :, None]], axis=2)
        ref = np.einsum("fki,fkj->fij", Ab, Ab)
        want = np.stack([ref[:, i, d] for d in range(31) for i in range(d + 1)], axis=1)
        got = big[:F].cpu().numpy()
        assert (np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)).max() < 1e-12, (model, F)
        if F > 7:       # a factor's result does not depend on where it sits in the wavefront (7: another slot of the DPP row group)
            sl = slice(7, F)
            part = eng.factor_hessian(model, {k: v[sl].contiguous() for k, v in meas.items()}, lin[sl].contiguous(),
                                      None if qq is None else qq[sl].contiguous(), states[7:].contiguous(), R[sl].contiguous())
            torch.cuda.synchronize()
            assert torch.equal(part, big[7:F])


@pytest.mark.parametrize("model", [1, 2])
def test_hessian_blocks_of_the_whitened_factor(eng, model):
    """cpi_factor_hessian_batch: packed upper triangle of [A1 A2 b]^T [A1 A2 b], A = R H, b = -R e --
    the augmented information matrix a GTSAM HessianFactor built from the linearised factor holds.  Checked against numpy
    on the outputs of the whitened sweep (itself checked above), with gathered (non-chained) state indices; G must be
    symmetric positive semi-definite and f = the Mahalanobis distance."""
    F = 1001
    kn, lin, q = synth.make_windows(F, 50, seed=92, device=eng.device, edge_cases=False)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(model))
    R = eng.sqrt_information(meas["P"])
    xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
    states = torch.cat([xi, xj], dim=0).contiguous()
    ii = torch.arange(F, dtype=torch.int32, device=eng.device)
    jj = ii + F
    qq = q if model == 2 else None
    white = eng.factor_eval(model, meas, lin, qq, states, ii, jj, sqrt_info=R)
    hess = eng.factor_hessian(model, meas, lin, qq, states, R, ii, jj)
    torch.cuda.synchronize()
    A1 = white["H1"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
    A2 = white["H2"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
    b = -white["err"].cpu().numpy()
    Ab = np.concatenate([A1, A2, b[:, :, None]], axis=2)                 # [F, 15, 31]
    ref = np.einsum("fki,fkj->fij", Ab, Ab)                              # [F, 31, 31]
    iu = [(i, d) for d in range(31) for i in range(d + 1)]               # packed 'U' order: i + d (d + 1) / 2
    want = np.stack([ref[:, i, d] for (i, d) in iu], axis=1)
    got = hess.cpu().numpy()
    assert got.shape == (F, 496)
    scale = np.abs(want).max(axis=1, keepdims=True)
    assert (np.abs(got - want) / scale).max() < 1e-12
    # f = b^T b = e^T P^-1 e
    e = eng.factor_eval(model, meas, lin, qq, states, ii, jj, want_H=False)["err"].cpu().numpy()
    P = meas["P"].cpu().numpy().reshape(F, 15, 15)
    maha = np.einsum("fi,fij,fj->f", e, np.linalg.inv(P), e)
    assert np.abs(got[:, 495] - maha).max() <= 1e-6 * max(1.0, maha.max())
