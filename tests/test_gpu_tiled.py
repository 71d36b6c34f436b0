"""GPU: the tiled input layout (cpi_preintegrate_tiled_batch / cpi_tile_knots, include/cpi_amd.h): knots of 64 windows
interleaved per step.  Same recursion, same parity bar as the dense layout: golden vectors of the compiled reference,
seeded batches whose size is not a multiple of 64, per-window counts with unwritten tails, both models, imu_avg."""
import os

import numpy as np
import pytest
import torch

from cpi_amd import synth
from tests.tol import check_pre

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import cpi_amd
    return cpi_amd.Engine()


def _host(out):
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items() if not k.startswith("_")}


def test_tile_knots_is_the_documented_permutation(eng):
    W, N = 131, 9
    kn = torch.arange(W * (N + 1) * 7, dtype=torch.float64, device=eng.device).reshape(W, N + 1, 7)
    t = eng.tile_knots(kn)
    torch.cuda.synchronize()
    assert t.shape == (3, N + 1, 7, 64)
    ref = torch.empty_like(t)
    for b in range(3):
        for i in range(64):
            w = min(b * 64 + i, W - 1)                     # windows past W replicate the last one
            ref[b, :, :, i] = kn[w]
    assert torch.equal(t, ref)


@pytest.mark.parametrize("mode", [(1, 0), (1, 1), (2, 0), (2, 1)])
def test_tiled_layout_vs_reference_golden(eng, golden_dir, mode):
    d = dict(np.load(os.path.join(golden_dir, "pre_w48.npz")))
    kn = torch.from_numpy(d["knots"]).to(eng.device)
    lin, q = torch.from_numpy(d["lin"]).to(eng.device), torch.from_numpy(d["q_k_lin"]).to(eng.device)
    tiles = eng.tile_knots(kn)
    out = _host(eng.preintegrate_tiled(tiles, kn.shape[0], lin, q, eng.make_params(mode[0], imu_avg=bool(mode[1]))))
    key = "m%d_avg%d_stj1__" % mode
    ref = {k[len(key):]: v for k, v in d.items() if k.startswith(key)}
    assert set(out) == {"DT", "alpha", "beta", "q"}
    check_pre(out, ref, what=("mean",), regression=True)


@pytest.mark.parametrize("split", [1, 2, 5, 8])
def test_tiled_split_over_wavefronts(eng, split):
    """Small batches split a tile's steps over several wavefronts of one workgroup (segments composed in order by the
    first one); the library picks the split itself -- here it is forced, for both models, with counts and imu_avg."""
    from oracle import oracle_py as op
    W, N = 200, 37
    kn, lin, q = synth.make_windows(W, N, seed=99 + split, device=eng.device)
    g = torch.Generator(device="cpu"); g.manual_seed(split)
    cnt = torch.randint(0, N + 1, (W,), generator=g, dtype=torch.int32)
    cnt[:3] = torch.tensor([0, N, 1], dtype=torch.int32)
    tiles = eng.tile_knots(kn)
    lib = op.reference() or op.oracle()
    knh, linh, qh = kn.cpu().numpy(), lin.cpu().numpy(), q.cpu().numpy()
    for model in (1, 2):
        for avg in (0, 1):
            ref = lib.run(op.make_params(model, avg, 1), knh, linh, qh, nthreads=8)
            out = _host(eng.preintegrate_tiled(tiles, W, lin, q, eng.make_params(model, imu_avg=bool(avg), lanes_per_window=split)))
            check_pre(out, ref, what=("mean",), regression=op.reference() is not None)
        rows = [lib.run(op.make_params(model, 0, 1), knh[w:w + 1, :int(cnt[w]) + 1], linh[w:w + 1], qh[w:w + 1]) for w in range(W)]
        ref = {k: np.concatenate([r[k] for r in rows], axis=0) for k in ("DT", "alpha", "beta", "q")}
        out = _host(eng.preintegrate_tiled(tiles, W, lin, q, eng.make_params(model, lanes_per_window=split), count=cnt.to(eng.device)))
        check_pre(out, ref, what=("mean",), regression=op.reference() is not None)


def test_tiled_layout_seeded_counts_and_refusals(eng):
    import cpi_amd
    from oracle import oracle_py as op
    W, N = 1000, 50                                         # 15 full tiles + one of 40 windows
    kn, lin, q = synth.make_windows(W, N, seed=515, device=eng.device)
    tiles = eng.tile_knots(kn)
    lib = op.reference() or op.oracle()
    for model in (1, 2):
        ref = lib.run(op.make_params(model, 0, 1), kn.cpu().numpy(), lin.cpu().numpy(), q.cpu().numpy(), nthreads=8)
        out = _host(eng.preintegrate_tiled(tiles, W, lin, q, eng.make_params(model)))
        check_pre(out, ref, what=("mean",), regression=op.reference() is not None)
        dense = _host(eng.preintegrate(kn, lin, q, eng.make_params(model, lanes_per_window=1), want=("mean",)))
        for k in out:                                       # same arithmetic per lane as the dense one-lane kernel
            assert np.abs(out[k] - dense[k]).max() < 1e-13, k
    # per-window counts; what lies behind a window's last knot in its tile column is never consumed
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    cnt = torch.randint(0, N + 1, (W,), generator=g, dtype=torch.int32)
    knp = kn.clone()
    for w in range(W):
        knp[w, int(cnt[w]) + 1:] = float("nan")
    tiles2 = eng.tile_knots(knp)
    out = _host(eng.preintegrate_tiled(tiles2, W, lin, q, eng.make_params(1), count=cnt.to(eng.device)))
    knh, linh, qh = kn.cpu().numpy(), lin.cpu().numpy(), q.cpu().numpy()
    rows = [lib.run(op.make_params(1, 0, 1), knh[w:w + 1, :int(cnt[w]) + 1], linh[w:w + 1], qh[w:w + 1]) for w in range(W)]
    ref = {k: np.concatenate([r[k] for r in rows], axis=0) for k in ("DT", "alpha", "beta", "q")}
    assert all(np.all(np.isfinite(v)) for v in out.values())
    check_pre(out, ref, what=("mean",))
    # the tiled entry serves the mean outputs only
    full = eng.alloc_outputs(W, ("mean", "jac", "cov"), 1)
    with pytest.raises(cpi_amd.CpiError):
        eng.preintegrate_tiled(tiles, W, lin, q, eng.make_params(1), out=full)


def test_tiled_edge_shapes(eng):
    """W = 1, W = 0, windows without intervals (N = 0), N smaller than the number of wavefronts per tile."""
    from oracle import oracle_py as op
    lib = op.reference() or op.oracle()
    for W, N in ((1, 50), (65, 1), (3, 0), (130, 3)):
        kn, lin, q = synth.make_windows(W, max(N, 1), seed=900 + W + N, device=eng.device)
        kn = kn[:, :N + 1].contiguous()
        tiles = eng.tile_knots(kn)
        assert tiles.shape == ((W + 63) // 64, N + 1, 7, 64)
        for model in (1, 2):
            out = _host(eng.preintegrate_tiled(tiles, W, lin, q, eng.make_params(model)))
            ref = lib.run(op.make_params(model, 0, 1), kn.cpu().numpy(), lin.cpu().numpy(), q.cpu().numpy())
            check_pre(out, ref, what=("mean",))
    empty = torch.empty((0, 51, 7, 64), dtype=torch.float64, device=eng.device)
    out = eng.preintegrate_tiled(empty, 0, torch.empty((0, 6), dtype=torch.float64, device=eng.device), None, eng.make_params(1))
    assert out["DT"].numel() == 0
