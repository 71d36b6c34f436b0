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
    assert np.abs(Rg[ok] - Rn).max() < 1e-13
    assert np.array_equal(Rg[4], np.eye(15))
    assert np.isnan(Rg[6]).any()
