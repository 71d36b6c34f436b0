"""GPU: the window lengths the reference itself runs -- imurate / camrate = 10 samples per window at 100 Hz
(cpi_compare/launch/synthetic_test.launch:27-28), 20 / 40 / 80 for the 200 / 400 / 800 Hz sets -- through every launch
geometry the automatic rules pick for them (round 6: the lane split of small batches, the two-knot kernel below 16 intervals per
window, the three-knot BIG kernel above; dense layout, tiled layout, stream entry): a strided 128-window sample of each launch against
the COMPILED REFERENCE (oracle/_ref, CpiV1.h / CpiV2.h) at the regression gates (tests/tol.py), models 1 and 2, mean-only and full."""
import os

import numpy as np
import pytest
import torch

import cpi_amd
from cpi_amd import synth
from tests.tol import check_pre

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return cpi_amd.Engine()


def _cpu(mode, kn, lin, q):
    from oracle import oracle_py as op
    ref = op.reference()
    lib = ref if ref is not None else op.oracle()
    return lib.run(op.make_params(*mode), kn, lin, q, nthreads=min(16, os.cpu_count() or 1)), ref is not None


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("N", [10, 20, 40, 80])
def test_mean_only_every_auto_geometry_vs_reference_sample(eng, model, N):
    mode = (model, 0, 1)
    for W in (5000, 10000, 30000, 100000, 1000000):
        kn, lin, q = synth.make_windows(W, N, seed=600 + N + model, device=eng.device)
        pick = np.unique(np.linspace(0, W - 1, 128).astype(np.int64))
        ref, from_ref = _cpu(mode, kn[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
        ref = {k: ref[k] for k in ("DT", "alpha", "beta", "q")}
        for lanes in ((0, 1, 2, 4) if W <= 30000 else (0,)):
            out = eng.preintegrate(kn, lin, q, eng.make_params(model, lanes_per_window=lanes), want=("mean",))
            torch.cuda.synchronize()
            check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, ref, what=("mean",), regression=from_ref,
                      label="N=%d W=%d model %d dense lanes %d" % (N, W, model, lanes))
        if W in (10000, 1000000):
            tiles = eng.tile_knots(kn)
            out = eng.preintegrate_tiled(tiles, W, lin, q, eng.make_params(model))
            torch.cuda.synchronize()
            check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, ref, what=("mean",), regression=from_ref,
                      label="N=%d W=%d model %d tiled" % (N, W, model))
            del tiles
        del kn, lin, q, out
        torch.cuda.empty_cache()


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("N", [10, 20])
def test_stream_entry_short_windows_vs_reference_sample(eng, model, N):
    """One resident stream cut in place, a partial tail interval in every window (N + 1 intervals): small batch (lane split) and
    1 M windows (the fused-cut streaming kernel: two-knot below 16 intervals per window, BIG above)."""
    from cpi_amd import stream as st
    mode = (model, 0, 1)
    for W in (10000, 1000000):
        stream, upd, lin, q = synth.make_stream(W, N, seed=650 + N + model, device=eng.device, phase=0.4)
        pick = np.unique(np.linspace(0, W - 1, 128).astype(np.int64))
        # the picked windows, assembled on the host exactly as the reference's deque loop cuts them (GraphSolver_IMU.cpp:50-69)
        s_np, u_np = stream.cpu().numpy(), upd.cpu().numpy()
        dense = []
        for u in pick:
            lo_t = u_np[u - 1] if u > 0 else -np.inf
            a = int(np.searchsorted(s_np[:, 0], lo_t, side="right")) - 1 if u > 0 else 0
            b = int(np.searchsorted(s_np[:, 0], u_np[u], side="right"))
            k, f, c = st.assemble_windows(s_np[max(a, 0):b + 1], np.array([u_np[u - 1], u_np[u]]) if u > 0 else np.array([u_np[u]]))
            w = len(c) - 1
            rows = k[f[w] + np.minimum(np.arange(N + 2), c[w])]
            dense.append(rows)
        dense = np.stack(dense)
        ref, from_ref = _cpu(mode, dense, lin[pick].cpu().numpy(), q[pick].cpu().numpy())
        for want in (("mean",), ("mean", "jac", "cov")) if W == 10000 else (("mean",),):
            out = eng.preintegrate_stream(stream, upd, lin, q, eng.make_params(model), want=want, N=N + 1)
            torch.cuda.synchronize()
            check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, ref, what=want, v2=(model == 2), regression=from_ref,
                      label="N=%d W=%d model %d stream %s" % (N, W, model, "+".join(want)))
        del stream, upd, lin, q, out
        torch.cuda.empty_cache()


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("N,W", [(10, 200000), (20, 100000), (10, 3000)])
def test_full_outputs_short_windows_vs_reference_sample(eng, model, N, W):
    """Everything out (means, Jacobians, covariance; model 2 through the state-transition read-out) -- the bench rows v1_full / v2_full
    at 10 and 20 samples -- and the packed covariance beside the dense one."""
    mode = (model, 0, 1)
    kn, lin, q = synth.make_windows(W, N, seed=690 + N + model, device=eng.device)
    pick = np.unique(np.linspace(0, W - 1, 128).astype(np.int64))
    ref, from_ref = _cpu(mode, kn[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
    out = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac", "cov", "cov_sym"))
    torch.cuda.synchronize()
    assert torch.equal(cpi_amd.unpack_sym(out["P_sym"][pick]), out["P"][pick])
    check_pre({k: v[pick].cpu().numpy() for k, v in out.items() if k != "P_sym"}, ref, v2=(model == 2), regression=from_ref,
              label="N=%d W=%d model %d full" % (N, W, model))

