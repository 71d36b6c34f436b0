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


# ---------------------------------------------------------------------------------------------------------------------------------
# cpi_mean_block_kernel (round 6): N <= 11, one lane per window -- the wavefront's knots fetched ONCE as a linear LDS block.  Same
# mean_step sequence as the one-lane instantiation of cpi_mean_kernel: BIT FOR BIT against it (a CSR `first` array is never admitted
# to the block kernel, so the same windows through first / count run the chunked kernel).
@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("N", [1, 2, 7, 10, 11])
def test_block_kernel_dense_layout_bitwise_vs_the_chunked_one_lane_kernel(eng, model, N):
    for W in (1, 63, 64, 65, 1000, 200001):
        kn, lin, q = synth.make_windows(W, N, seed=800 + N + W % 97, device=eng.device)
        first = (torch.arange(W, device=eng.device, dtype=torch.int64) * (N + 1)).contiguous()
        count = torch.full((W,), N, dtype=torch.int32, device=eng.device)
        for avg in (False, True):
            prm = eng.make_params(model, avg, lanes_per_window=1)
            blk = eng.preintegrate(kn, lin, q, prm, want=("mean",))                                   # dense, one lane: the block kernel
            csr = eng.preintegrate(kn.reshape(-1, 7), lin, q, prm, want=("mean",), first=first, count=count, N=N)
            torch.cuda.synchronize()
            for k in ("DT", "alpha", "beta", "q"):
                assert torch.equal(blk[k], csr[k]), (model, N, W, avg, k)
    # N = 12 is past the block kernel's limit: same answer from the chunked kernel either way (the launcher's boundary)
    kn, lin, q = synth.make_windows(3000, 12, seed=5, device=eng.device)
    a = eng.preintegrate(kn, lin, q, eng.make_params(model, lanes_per_window=1), want=("mean",))
    b = eng.preintegrate(kn.reshape(-1, 7), lin, q, eng.make_params(model, lanes_per_window=1), want=("mean",),
                         first=(torch.arange(3000, device=eng.device, dtype=torch.int64) * 13).contiguous(),
                         count=torch.full((3000,), 12, dtype=torch.int32, device=eng.device), N=12)
    torch.cuda.synchronize()
    assert all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("model", [1, 2])
def test_block_kernel_stream_cut_ragged_truncated_and_at_the_end_of_the_buffer(eng, model):
    """The fused stream cut inside the block kernel: jittered update times (ragged windows of 6 ... 11 intervals, mostly with a tail),
    one update exactly on a reading, a gap in the update grid that makes one window 40 intervals long (truncated to the bound, and
    the wavefront's run no longer fits the LDS block: the direct-read path), the last window ending PAST the last stamp with NaNs right
    behind the stream in memory -- against the same windows through the CSR layout (chunked one-lane kernel) bit for bit, the TRUE
    counts exactly, and a strided sample against the compiled reference."""
    from tests.test_stream import _torch_cut
    W, N = 70000, 8
    stream, upd, lin, q = synth.make_stream(W, N, seed=31 + model, device=eng.device, phase=0.37)
    g = torch.Generator(device=eng.device); g.manual_seed(9)
    upd = upd + (torch.rand(upd.shape, generator=g, dtype=torch.float64, device=eng.device) * 3.0 - 2.0) / 200.0   # -2 ... +1 samples
    upd = torch.sort(upd).values.contiguous()
    upd[7] = stream[7 * N + 3, 0]                                    # one update exactly ON a reading: no tail interval
    upd[30000:] += 40 * 0.005                                        # a gap: window 30 000 holds ~48 intervals
    upd = torch.sort(upd).values.contiguous()
    upd[-1] = stream[-1, 0] + 0.003                                  # the last window ends past the last stamp
    K = stream.shape[0]
    big = torch.full((K + 64, 7), float("nan"), dtype=torch.float64, device=eng.device)
    big[:K] = stream
    ds = big[:K]
    knots, first, count = _torch_cut(stream, upd)
    cnt_np = count.cpu().numpy()
    assert cnt_np.max() > 40 and np.median(cnt_np) <= 10 and len(np.unique(cnt_np)) >= 5
    Nb = 11                                                          # the bound: the long window is truncated to 11 intervals
    for avg in (False, True):
        prm = eng.make_params(model, avg, lanes_per_window=1)
        out, cnt = eng.preintegrate_stream(ds, upd, lin, q, prm, want=("mean",), N=Nb, return_counts=True, check_counts=False)
        csr = eng.preintegrate(knots, lin, q, prm, want=("mean",), first=first, count=count, N=Nb)
        torch.cuda.synchronize()
        assert torch.equal(cnt.to(torch.int32), count), (model, avg)
        for k in ("DT", "alpha", "beta", "q"):
            assert bool(torch.isfinite(out[k]).all()), (model, avg, k)
            assert torch.equal(out[k], csr[k]), (model, avg, k)
    # sample against the compiled reference: the picked windows' knots from the CSR arrays (truncated like the kernels truncate)
    pick = np.unique(np.concatenate([np.linspace(0, W - 1, 120).astype(np.int64), [7, 29999, 30000, 30001, W - 1]]))
    f_np = first.cpu().numpy()
    dense = np.stack([knots[int(f_np[u]) + np.minimum(np.arange(Nb + 1), min(int(cnt_np[u]), Nb))].cpu().numpy() for u in pick])
    ref, from_ref = _cpu((model, 1, 1), dense, lin[pick].cpu().numpy(), q[pick].cpu().numpy())
    check_pre({k: out[k][pick].cpu().numpy() for k in ("DT", "alpha", "beta", "q")}, {k: ref[k] for k in ("DT", "alpha", "beta", "q")},
              what=("mean",), regression=from_ref, label="block kernel, stream cut, model %d" % model)
