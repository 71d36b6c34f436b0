"""GPU: cpi_predict_batch (GraphSolver::getpredictedstate_v1 / _v2, GraphSolver_IMU.cpp:263-307) on ragged
grids and gathered state indices -- the kernel moves a wavefront's 64 records as one coalesced burst through LDS (round 6), so
every partial-wavefront shape, the clamp of out-of-range indices and the "nothing written past F" rule are exercised against
the C restatement of the reference's function."""
import numpy as np
import pytest
import torch

import cpi_amd
from cpi_amd import synth
from tests.tol import TOL_FACTOR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return cpi_amd.Engine()


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("F", [1, 2, 7, 8, 9, 63, 64, 65, 127, 1000, 70001])
def test_predict_ragged_chained_and_gathered(eng, model, F):
    from oracle import oracle_py as op
    kn, lin, q = synth.make_windows(F, 12, seed=40 + F, device=eng.device, edge_cases=False)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac"))
    xi, _ = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
    big = torch.full((F + 3, 16), -7.0, dtype=torch.float64, device=eng.device)
    eng.predict(model, meas, xi, out=big[:F])
    torch.cuda.synchronize()
    assert torch.all(big[F:] == -7.0)
    m = {k: v.cpu().numpy() for k, v in meas.items()}
    n = min(F, 3000)                                   # the restatement is a per-factor C call: a bounded sample of the big case
    sel = np.unique(np.linspace(0, F - 1, n).astype(np.int64))
    rec = op.factor_records({k: v[sel] for k, v in m.items()}, lin.cpu().numpy()[sel], q.cpu().numpy()[sel] if model == 2 else None)
    want = op.oracle().predict(model, rec, xi.cpu().numpy()[sel])
    got = big[:F].cpu().numpy()[sel]
    assert np.abs(got - want).max() <= TOL_FACTOR * max(1.0, np.abs(want).max())
    # gathered states: S != F, repeated and out-of-range indices (clamped into [0, S): include/cpi_amd.h)
    S = max(1, F // 2 + 1)
    g = torch.Generator().manual_seed(F)
    idx = torch.randint(-3, S + 3, (F,), dtype=torch.int32, generator=g).to(eng.device)
    states = xi[:S].contiguous()
    out = eng.predict(model, meas, states, idx_i=idx)
    ref = eng.predict(model, meas, states[idx.clamp(0, S - 1).long()].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # a factor's result does not depend on where it sits in the wavefront
    if F > 9:
        part = eng.predict(model, {k: v[9:].contiguous() for k, v in meas.items()}, xi[9:].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(part, big[9:F])
