"""GPU: BASELINE.json configs[4] at its FULL size on one device -- 8 M windows x 100 samples (45 GB of knots, offsets far
beyond 2^32 bytes) -- through a size-independent property: the batch is one 1 M-window batch repeated 8 times, so every
eighth of every output must equal the first eighth bit for bit (and the first eighth the 1 M-window call on its own).
Dense mean-only, tiled mean-only and "V1 full" (18.6 GB of outputs).  Skipped when the device has < 150 GB free."""
import pytest
import torch

from cpi_amd import synth

pytestmark = pytest.mark.gpu


def test_configs4_full_size_block_periodicity():
    import cpi_amd
    eng = cpi_amd.Engine()
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs 150 GB of free device memory")
    W1, R, N = 1000000, 8, 100
    kn1, lin1, q1 = synth.make_windows(W1, N, seed=4040, device=eng.device)
    prm = eng.make_params(1, lanes_per_window=1)
    ref = eng.preintegrate(kn1, lin1, q1, prm, want=("mean",))
    big, lin, q = kn1.repeat(R, 1, 1), lin1.repeat(R, 1), q1.repeat(R, 1)
    assert big.numel() * 8 > 40 * (1 << 30)
    del kn1
    W = W1 * R

    def periodic(out, keys, first=None):
        torch.cuda.synchronize()
        for k in keys:
            v = out[k].reshape(R, W1, -1)
            for r in range(1, R):
                assert torch.equal(v[r], v[0]), (k, r)
            if first is not None:
                assert torch.equal(v[0], first[k].reshape(W1, -1)), k

    out = eng.preintegrate(big, lin, q, prm, want=("mean",))
    periodic(out, ("DT", "alpha", "beta", "q"), ref)
    tiles = eng.tile_knots(big)
    tout = eng.preintegrate_tiled(tiles, W, lin, q, prm)
    periodic(tout, ("DT", "alpha", "beta", "q"))
    for k in ("alpha", "beta", "q"):
        assert (tout[k][:W1] - ref[k]).abs().max().item() < 1e-13, k
    del tiles, tout, out
    torch.cuda.empty_cache()
    full = eng.preintegrate(big, lin, q, prm)               # V1 full: covariance kernel + Jacobian kernel on 8 M windows
    periodic(full, ("DT", "alpha", "beta", "q", "P", "J_q", "J_a", "J_b", "H_a", "H_b"))
    P = full["P"].reshape(W, 15, 15)
    assert torch.isfinite(P[::9973]).all() and (P[::9973].diagonal(dim1=1, dim2=2) > 0).all()
    del full, big, P
    torch.cuda.empty_cache()


@pytest.mark.parametrize("model", [1, 2])
def test_configs3_full_size_sweep_block_periodicity(model):
    """configs[3] at its full size: 1 M factors (3.7 GB of err / H1 / H2) = 125 k factors repeated 8 times through the index
    arrays; dense, packed and whitened sweeps must be block-periodic bit for bit, and the first block equal to the
    125 k-factor call."""
    import cpi_amd
    eng = cpi_amd.Engine()
    F1, R = 125000, 8
    kn, lin1, q1 = synth.make_windows(F1, 20, seed=606 + model, device=eng.device)
    meas1 = eng.preintegrate(kn, lin1, q1, eng.make_params(model))
    xi, xj = synth.make_states(meas1["alpha"], meas1["beta"], meas1["q"], meas1["DT"], lin1, model, device=eng.device)
    states = torch.cat([xi, xj], 0).contiguous()
    ii1 = torch.arange(F1, dtype=torch.int32, device=eng.device)
    qq1 = q1 if model == 2 else None
    ref = eng.factor_eval(model, meas1, lin1, qq1, states, ii1, ii1 + F1)
    meas = {k: (v.repeat(R, 1) if v.dim() == 2 else v.repeat(R)) for k, v in meas1.items() if not k.startswith("_")}
    lin, qq = lin1.repeat(R, 1), (q1.repeat(R, 1) if model == 2 else None)
    ii = ii1.repeat(R)
    F = F1 * R
    sq = eng.sqrt_information(meas["P"])
    for name, out in (("dense", eng.factor_eval(model, meas, lin, qq, states, ii, ii + F1)),
                      ("packed", {"packed": eng.factor_eval_packed(model, meas, lin, qq, states, ii, ii + F1)}),
                      ("whitened", eng.factor_eval(model, meas, lin, qq, states, ii, ii + F1, sqrt_info=sq))):
        torch.cuda.synchronize()
        for k, v in out.items():
            b = v.reshape(R, F1, -1)
            for r in range(1, R):
                assert torch.equal(b[r], b[0]), (name, k, r)
            if name == "dense":
                assert torch.equal(b[0], ref[k].reshape(F1, -1)), k
        del out
    torch.cuda.empty_cache()
