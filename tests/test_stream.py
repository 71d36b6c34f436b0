"""the IMU text wire format (SimParser.h:130-193) and the cutting of one IMU stream
into windows at update times (GraphSolver_IMU.cpp:50-69), on a 700-line excerpt of the reference's own
dataset cpi_simulation/GAZEBO_FREQ_200/rawdata_00/imu_data_meas.dat (data, committed as a fixture).
CPU tests: parser and assembler (Python and C++ twins) against the oracle's literal restatement of the
reference's deque loop.  GPU test: the assembled ragged windows through the HIP path."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from cpi_amd import stream as st
from oracle import oracle_py as op
from tests.tol import check_pre

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "imu_gazebo200_excerpt.dat")


def _updates(kn):
    # 10 Hz camera, deliberately off the IMU grid, plus one update exactly ON an IMU stamp (no tail interval)
    t0 = kn[0, 0]
    ut = t0 + 0.0523 + 0.1 * np.arange(33)
    ut[5] = kn[120, 0]
    return np.sort(ut)


def test_parse_matches_loadtxt():
    kn = st.parse_imu_text(open(DATA).read())
    raw = np.loadtxt(DATA)
    assert kn.shape == (raw.shape[0], 7) and kn.shape[0] == 700
    assert np.array_equal(kn[:, 1:7], raw[:, 0:6])
    assert np.array_equal(kn[:, 0], 1e-3 * raw[:, 7])
    assert abs(kn[0, 0] - 1275.0) < 1e-12            # stamps are milliseconds
    assert st.parse_imu_text("\n  \n1 2 3\n").shape == (0, 7)


@pytest.mark.parametrize("mode", [(1, 0, 1), (2, 0, 1), (1, 1, 1)])
def test_assembly_equals_reference_deque_loop(mode):
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    knots, first, count = st.assemble_windows(kn, ut)
    U = len(ut)
    rng = np.random.default_rng(1)
    lin = np.concatenate([0.01 * rng.standard_normal((U, 3)), 0.05 * rng.standard_normal((U, 3))], axis=1)
    q = rng.standard_normal((U, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    prm = op.make_params(*mode)
    ref = op.oracle().stream(prm, kn, ut, lin, q)          # literal deque-loop restatement
    for u in range(U):
        w = knots[first[u]:first[u] + count[u] + 1][None]
        o = op.oracle().run(prm, w, lin[u:u + 1], q[u:u + 1])
        for k in ("DT", "alpha", "beta", "q", "J_a", "P"):
            assert np.array_equal(o[k][0], ref[k][u]), (u, k)
    # window u ends exactly at the update time; the on-grid update has no tail knot
    ends = knots[first + count, 0]
    assert np.array_equal(ends, ut)
    assert abs(ref["DT"].sum() - (ut[-1] - kn[0, 0])) < 1e-9


def test_default_interval_bound_of_the_stream_entry_is_the_longest_window():
    """Engine.preintegrate_stream without N: the bound is computed from the stamps (the library picks the mean
    kernel's lane split from N; the old default -- the whole stream -- was as loose as a bound gets).  It must never be below a
    true count (a longer window would be truncated) and is at most one above the longest (the tail is assumed)."""
    import torch
    from cpi_amd.engine import Engine
    kn = st.parse_imu_text(open(DATA).read())
    for ut in (_updates(kn), np.array([kn[0, 0] - 1.0, kn[3, 0], kn[3, 0], kn[400, 0] + 1e-4, kn[-1, 0] + 5.0]), kn[:1, 0] + 100.0):
        _, _, count = st.assemble_windows(kn, ut)
        b = Engine._stream_bound(torch.from_numpy(kn), torch.from_numpy(np.ascontiguousarray(ut)))
        assert int(count.max()) <= b <= max(int(count.max()), 0) + 1, (b, count.max())
    assert Engine._stream_bound(torch.zeros((0, 7), dtype=torch.float64), torch.zeros((3,), dtype=torch.float64)) == 1


def test_cpp_twins_match_python():
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    knots, first, count = st.assemble_windows(kn, ut)
    exe = os.path.join(tempfile.mkdtemp(), "test_stream")
    # cpi_host.hpp needs the C-ABI header only at compile time; link against the library for the (unused) symbols
    libdir = os.path.join(ROOT, "cpi_amd")
    from cpi_amd import _lib
    _lib.load()
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_stream.cpp"), "-o", exe,
                           "-L" + libdir, "-lcpi_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("\n".join("%.17g" % t for t in ut))
        upath = f.name
    out = subprocess.run([exe, DATA, upath], stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
    K, U, mx = (int(x) for x in out[0].split())
    assert K == 700 and U == len(ut) and mx == count.max()
    fc = np.array([[int(x) for x in ln.split()] for ln in out[1:1 + U]])
    assert np.array_equal(fc[:, 0], first) and np.array_equal(fc[:, 1], count)
    kk = np.array([[float(x) for x in ln.split()] for ln in out[1 + U:] if ln.strip()])
    assert np.array_equal(kk, knots)
    # the tiled twin: written straight into tiles[ceil(U/64)][N+1][7][64]; every slot a kernel may read is the CSR knot
    out = subprocess.run([exe, DATA, upath, "tiled"], stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
    Wt, Nt = (int(x) for x in out[0].split())
    assert Wt == U and Nt == count.max()
    assert np.array_equal(np.array([int(x) for x in out[1:1 + U]]), count)
    tiles = np.array([[float(x) for x in ln.split()] for ln in out[1 + U:] if ln.strip()]).reshape((U + 63) // 64, Nt + 1, 7, 64)
    for u in range(U):
        assert np.array_equal(tiles[u // 64, :count[u] + 1, :, u % 64], knots[first[u]:first[u] + count[u] + 1]), u


def test_tiled_assembly_places_every_knot_at_its_tile_slot():
    """cpi_amd.stream: assemble_windows(layout="tiled") / tile_windows against the CSR assembly (which the test above pins to
    the reference's deque loop), on the reference's own IMU excerpt and on the dense layout."""
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    knots, first, count = st.assemble_windows(kn, ut)
    tiles, c2 = st.assemble_windows(kn, ut, layout="tiled")
    U, N = len(ut), int(count.max())
    assert np.array_equal(c2, count) and tiles.shape == ((U + 63) // 64, N + 1, 7, 64)
    for u in range(U):
        assert np.array_equal(tiles[u // 64, :count[u] + 1, :, u % 64], knots[first[u]:first[u] + count[u] + 1])
        assert np.array_equal(tiles[u // 64, count[u]:, :, u % 64], np.repeat(knots[first[u] + count[u]][None], N + 1 - count[u], 0))   # finite padding
    assert np.all(np.isfinite(tiles))
    dense = np.arange(131 * 10 * 7, dtype=np.float64).reshape(131, 10, 7)
    t = st.tile_windows(dense)
    assert t.shape == (3, 10, 7, 64)
    for w in (0, 63, 64, 130):
        assert np.array_equal(t[w // 64, :, :, w % 64], dense[w])
    assert np.array_equal(t[2, :, :, 63], dense[130])                      # columns past W repeat window W - 1


def test_device_assembler_closed_form_equals_the_deque_loop():
    """cpi_assemble_tiles cuts every window independently: for non-decreasing stamps the deque state at the start of a window
    is a function of the previous update time alone -- front(T) = max(#{t <= T} - 1, 0), stamp(T) = max(T, t_0)
    (cpi_mean_kernels.hpp).  That closed form, restated in numpy, against the sequential loop: repeated stamps (dt = 0),
    update times on / off / before / after the IMU grid, repeated update times, streams of one knot."""
    def closed_form(s, T):
        t = s[:, 0]
        rows, cnt = [], []
        for u in range(len(T)):
            fp, start = (max(np.searchsorted(t, T[u - 1], side="right") - 1, 0), max(T[u - 1], t[0])) if u else (0, t[0])
            fu = max(np.searchsorted(t, T[u], side="right") - 1, 0, fp)
            m = fu - fp
            front_t = t[fu] if m > 0 else start
            tail = (T[u] - front_t) > 0
            r = [np.concatenate([[start], s[fp, 1:]])] + [s[fp + i] for i in range(1, m + 1)]
            if tail:
                r.append(np.concatenate([[T[u]], s[fu, 1:]]))
            rows.append(np.array(r)); cnt.append(m + int(tail))
        return rows, np.array(cnt)
    rng = np.random.default_rng(1)
    for trial in range(200):
        K = int(rng.integers(1, 60))
        t = np.cumsum(rng.choice([0.0, 0.005, 0.005, 0.01], K)) + 1.0
        s = np.concatenate([t[:, None], rng.normal(size=(K, 6))], 1)
        T = np.sort(rng.choice(np.concatenate([t, t + 0.0025, [t[0] - 0.1, t[-1] + 0.1]]), int(rng.integers(1, 20))))
        knots, first, count = st.assemble_windows(s, T)
        rows, c = closed_form(s, T)
        assert np.array_equal(c, count), trial
        for u in range(len(T)):
            assert np.array_equal(rows[u], knots[first[u]:first[u] + count[u] + 1]), (trial, u)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(1, 0, 1), (2, 0, 1), (1, 1, 1)])
def test_gpu_stream_through_the_tiled_producers_vs_deque_oracle(mode):
    """The mean-only path a caller shaped like GraphSolver_IMU.cpp:43-75 takes: ONE IMU stream + update times ->
    cpi_assemble_tiles on the device (no dense / CSR copy, no cpi_tile_knots pass) -> cpi_preintegrate_tiled_batch, against
    the oracle's literal deque-loop restatement; the device assembler's tiles and counts equal the host assembler's
    (Python and, through cpi_tile_windows, the CSR layout) slot for slot."""
    import torch
    import cpi_amd
    eng = cpi_amd.Engine()
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    ut = np.sort(np.concatenate([ut, [kn[0, 0] - 1.0, ut[7], kn[-1, 0] + 0.5]]))     # + before the stream, repeated, past its end
    knots, first, count = st.assemble_windows(kn, ut)
    tiles_h, _ = st.assemble_windows(kn, ut, layout="tiled")
    U, N = len(ut), int(count.max())
    rng = np.random.default_rng(1)
    lin = np.concatenate([0.01 * rng.standard_normal((U, 3)), 0.05 * rng.standard_normal((U, 3))], axis=1)
    q = rng.standard_normal((U, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    tiles_d, count_d = eng.assemble_tiles(T(kn), T(ut), N + 2)          # N larger than needed: rows past count stay unwritten
    tiles_c = eng.tile_windows(T(knots), T(first), T(count), N + 2)
    torch.cuda.synchronize()
    assert np.array_equal(count_d.cpu().numpy(), count)
    td, tc = tiles_d.cpu().numpy(), tiles_c.cpu().numpy()
    for u in range(U):
        assert np.array_equal(td[u // 64, :count[u] + 1, :, u % 64], knots[first[u]:first[u] + count[u] + 1]), u
        assert np.array_equal(tc[u // 64, :count[u] + 1, :, u % 64], tiles_h[u // 64, :count[u] + 1, :, u % 64]), u
    prm = eng.make_params(*mode)
    ref = op.oracle().stream(op.make_params(*mode), kn, ut, lin, q)
    for tiles in (tiles_d, tiles_c, T(tiles_h)):
        out = eng.preintegrate_tiled(tiles, U, T(lin), T(q), prm, count=T(count))
        torch.cuda.synchronize()
        check_pre({k: v.cpu().numpy() for k, v in out.items()}, ref, what=("mean",), v2=(mode[0] == 2), label="tiled stream %s" % (mode,))
    with pytest.raises(ValueError):
        eng.preintegrate_stream(T(kn), T(ut), T(lin), T(q), prm, want=("mean",), N=N - 1)   # a window does not fit: said, not truncated
    # the same tiles from HOST memory through the chunked pipeline
    hout = eng.preintegrate_tiled_host(torch.from_numpy(tiles_h), U, torch.from_numpy(lin), torch.from_numpy(q), prm,
                                       count=torch.from_numpy(count))
    check_pre({k: v.numpy() for k, v in hout.items()}, ref, what=("mean",), v2=(mode[0] == 2))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(1, 0, 1), (2, 0, 1), (1, 1, 1), (2, 1, 1), (2, 0, 0)])
def test_gpu_stream_entry_reads_the_stream_in_place(mode):
    """cpi_preintegrate_stream: ONE resident IMU stream + update times, every output, NO knot copied -- the kernels cut
    the windows themselves.  Against the oracle's literal deque-loop restatement on the reference's own IMU excerpt (update
    times before the stream, on and off the IMU grid, repeated, past the end), and BIT FOR BIT against cpi_preintegrate_batch
    on the knots / first / count the host assembler makes from the same stream, for every lane split of the mean kernel."""
    import torch
    import cpi_amd
    eng = cpi_amd.Engine()
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    ut = np.sort(np.concatenate([ut, [kn[0, 0] - 1.0, ut[7], kn[-1, 0] + 0.5]]))
    knots, first, count = st.assemble_windows(kn, ut)
    U, N = len(ut), int(count.max())
    rng = np.random.default_rng(1)
    lin = np.concatenate([0.01 * rng.standard_normal((U, 3)), 0.05 * rng.standard_normal((U, 3))], axis=1)
    q = rng.standard_normal((U, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    ref = op.oracle().stream(op.make_params(*mode), kn, ut, lin, q)
    dk, du, dl, dq = T(kn), T(ut), T(lin), T(q)
    ck, cf, cc = T(knots), T(first), T(count)
    for lanes in (0, 1, 2, 3, 5, 8, 16, 64):
        prm = eng.make_params(mode[0], bool(mode[1]), bool(mode[2]), lanes_per_window=lanes)
        for want in (("mean", "jac", "cov"), ("mean",), ("mean", "jac")):
            out, cnt = eng.preintegrate_stream(dk, du, dl, dq, prm, want=want, N=N, return_counts=True)
            csr = eng.preintegrate(ck, dl, dq, prm, want=want, first=cf, count=cc, N=N)
            torch.cuda.synchronize()
            assert np.array_equal(cnt.cpu().numpy(), count)
            for k in csr:
                assert torch.equal(out[k], csr[k]), (lanes, want, k)
            check_pre({k: v.cpu().numpy() for k, v in out.items()}, ref, what=want, v2=(mode[0] == 2), label="stream entry %s L%d" % (mode, lanes))
    # a looser bound on the window length changes nothing; the default bound is the stream's length
    prm = eng.make_params(mode[0], bool(mode[1]), bool(mode[2]))
    a = eng.preintegrate_stream(dk, du, dl, dq, prm, N=N)
    b = eng.preintegrate_stream(dk, du, dl, dq, prm, N=N + 37)
    c = eng.preintegrate_stream(dk, du, dl, dq, prm)
    torch.cuda.synchronize()
    for k in ("DT", "alpha", "beta", "q", "P"):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    with pytest.raises(ValueError):
        eng.preintegrate_stream(dk, du, dl, dq, prm, N=N - 1)          # a window does not fit: said, not truncated


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["long", "mixed", "tiny", "end"])
def test_gpu_device_assembler_every_pass_geometry(case):
    """cpi_assemble_tiles (cpi_mean_kernels.hpp): a wavefront owns a tile of 64 windows, lane = window; a trip moves 8 rows of every
    window -- 32 LDS-DMA instructions, each the 448 contiguous bytes of TWO windows, into a 33-KB image, from which every lane reads
    its window back for seven 512-byte row stores.  Window lengths chosen to exercise the trip logic -- windows of 900 and 2 000
    intervals next to empty ones (lanes that finished long ago re-fetch their last knot), lengths 803 / 804 / 805 (= 5 mod 8 ...:
    the partial last trip), repeated update times, the last windows running past the stream's end (the pull-back of a trip's first
    knot to K - 8), a stream shorter than one trip (K = 3 < 8: the whole launch takes the per-lane loads) -- against the host
    assembler (the deque loop) slot for slot, and the counts exactly.  The two other fallback predicates of the DMA route have
    their own test below (test_gpu_device_assembler_fallback_predicates)."""
    import torch
    import cpi_amd
    eng = cpi_amd.Engine()
    rng = np.random.default_rng(7)
    if case == "tiny":
        K, lens = 3, [1, 0, 1, 0]
    elif case == "long":
        lens = [900, 3, 0, 410, 410, 50, 2000, 1, 1, 805, 804, 803, 2, 0, 0, 60, 17, 300, 300, 300, 300, 300, 300, 300, 300]
        K = sum(lens) + 40
    elif case == "mixed":
        lens = list(rng.integers(0, 120, 200)) + [500, 500, 10] + list(rng.integers(40, 60, 77))
        K = int(sum(lens)) + 5
    else:
        lens = [50] * 40
        K = 50 * 40 - 7          # the last windows run past the stream's end
    t = 100.0 + np.cumsum(np.full(K, 0.005)) - 0.005
    stream = np.concatenate([t[:, None], rng.standard_normal((K, 6))], axis=1)
    edges = np.minimum(np.cumsum(lens), K - 1 + 30)
    ut = 100.0 + 0.005 * edges + 0.002                     # every window ends 2 ms behind a reading: a tail interval
    ut[::7] -= 0.002                                        # ... except some that end exactly on a reading
    ut = np.maximum.accumulate(ut)
    knots, first, count = st.assemble_windows(stream, ut)
    U, N = len(ut), max(int(count.max()), 1)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    tiles = torch.full(((U + 63) // 64, N + 1, 7, 64), float("nan"), dtype=torch.float64, device=eng.device)
    _, cnt = eng.assemble_tiles(T(stream), T(ut), N, tiles=tiles)
    torch.cuda.synchronize()
    assert np.array_equal(cnt.cpu().numpy(), count), case
    td = tiles.cpu().numpy()
    for u in range(U):
        assert np.array_equal(td[u // 64, :count[u] + 1, :, u % 64], knots[first[u]:first[u] + count[u] + 1]), (case, u, count[u])
    # a bound N smaller than the longest window: rows beyond N are not written, the true counts still are
    if N > 4:
        small = torch.full(((U + 63) // 64, 5, 7, 64), float("nan"), dtype=torch.float64, device=eng.device)
        _, cnt2 = eng.assemble_tiles(T(stream), T(ut), 4, tiles=small)
        torch.cuda.synchronize()
        assert np.array_equal(cnt2.cpu().numpy(), count)
        sd = small.cpu().numpy()
        for u in range(U):
            n = min(int(count[u]), 4)
            assert np.array_equal(sd[u // 64, :n + 1, :, u % 64], knots[first[u]:first[u] + n + 1]), (case, u)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["k_lt_8", "span_2_24", "unsorted"])
def test_gpu_device_assembler_fallback_predicates(case):
    """the DMA route of cpi_assemble_tiles is taken per wavefront when (a) K >= 8, (b) every window of the wavefront
    starts at or behind lane 0's (`fp >= fp of lane 0`) and (c) all of them lie within 2^24 knots above it (the DMA offsets are 32-bit
    byte offsets from a wave-uniform base); any other wavefront takes per-lane loads.  Each predicate failing, next to wavefronts
    where it holds:
      k_lt_8     streams of 1 ... 7 readings;
      span_2_24  17 M readings; one tile whose last window ends 16.9 M readings behind the others (count >> N: rows beyond N are not
                 written, the TRUE count is) between tiles of ordinary windows;
      unsorted   update times that go BACKWARDS inside a tile (outside the contract -- the reference's deque cannot be restated for
                 them -- but the kernel must stay inside the stream and follow its documented closed form: window u starts where the
                 deque would stand after update u - 1 ALONE): expectation = that closed form (_torch_cut).
    Slot for slot against the host assembler (k_lt_8, span_2_24) / the closed form, counts exactly; and the fused cut of
    cpi_preintegrate_stream (same arithmetic in the mean kernel's prologue) gives the same counts."""
    import torch
    import cpi_amd
    eng = cpi_amd.Engine()
    rng = np.random.default_rng(11)
    dev = eng.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if case == "k_lt_8":
        for K in range(1, 8):
            t = 50.0 + 0.005 * np.arange(K)
            stream = np.concatenate([t[:, None], rng.standard_normal((K, 6))], axis=1)
            ut = np.sort(np.concatenate([t[0] + 0.005 * rng.uniform(-1, K + 1, 70), [t[-1], t[0]]]))
            knots, first, count = st.assemble_windows(stream, ut)
            N = max(int(count.max()), 1)
            tiles = torch.full(((len(ut) + 63) // 64, N + 1, 7, 64), float("nan"), dtype=torch.float64, device=dev)
            _, cnt = eng.assemble_tiles(T(stream), T(ut), N, tiles=tiles)
            torch.cuda.synchronize()
            assert np.array_equal(cnt.cpu().numpy(), count), K
            td = tiles.cpu().numpy()
            for u in range(len(ut)):
                assert np.array_equal(td[u // 64, :count[u] + 1, :, u % 64], knots[first[u]:first[u] + count[u] + 1]), (K, u)
        return
    if case == "span_2_24":
        K = (1 << 24) + 300000
        stream = torch.empty((K, 7), dtype=torch.float64, device=dev)
        stream[:, 0] = 100.0 + torch.arange(K, dtype=torch.float64, device=dev) * 0.005
        g = torch.Generator(device=dev); g.manual_seed(3)
        stream[:, 1:] = torch.randn((K, 6), generator=g, dtype=torch.float64, device=dev)
        # 3 tiles: windows of ~40 intervals; window 100 (tile 1) ends 16.9 M readings later, windows 101 ... go on from there
        edges = np.concatenate([40 * np.arange(1, 101), (1 << 24) + 100000 + 40 * np.arange(1, 92)]).astype(np.int64)
        ut = torch.from_numpy(100.0 + 0.005 * edges + 0.002).to(dev)
        N = 64
        knots, first, count = _torch_cut(stream, ut)
        cnt_np = count.cpu().numpy()
        assert int(cnt_np[100]) > (1 << 24) and set(cnt_np[:100].tolist()) <= {40, 41} and int(cnt_np[101:].max()) <= 41
        tiles = torch.full(((len(edges) + 63) // 64, N + 1, 7, 64), float("nan"), dtype=torch.float64, device=dev)
        _, cnt = eng.assemble_tiles(stream, ut, N, tiles=tiles)
        torch.cuda.synchronize()
        assert torch.equal(cnt.to(torch.int32), count)
        f_np = first.cpu().numpy()
        for u in list(range(0, 191, 7)) + [63, 64, 99, 100, 101, 127, 128, 190]:
            n = min(int(cnt_np[u]), N)
            assert torch.equal(tiles[u // 64, :n + 1, :, u % 64], knots[int(f_np[u]):int(f_np[u]) + n + 1]), u
        # the head of the closed form against the deque loop (host), and the fused cut's counts
        hk, hf, hc = st.assemble_windows(stream[:5000].cpu().numpy(), ut[:100].cpu().numpy())
        assert np.array_equal(hc, cnt_np[:100]) and np.array_equal(hk, knots[:hk.shape[0]].cpu().numpy())
        lin = torch.zeros((len(edges), 6), dtype=torch.float64, device=dev)
        _, c2 = eng.preintegrate_stream(stream, ut, lin, None, eng.make_params(1), want=("mean",), N=N, return_counts=True, check_counts=False)
        torch.cuda.synchronize()
        assert torch.equal(c2.to(torch.int32), count)
        return
    K = 6000
    t = 10.0 + 0.005 * np.arange(K)
    stream = np.concatenate([t[:, None], rng.standard_normal((K, 6))], axis=1)
    ut = 10.0 + 0.005 * (30 * np.arange(1, 193)) + 0.0021
    ut[5], ut[6] = ut[6], ut[5]                       # a swap inside tile 0 (lane 6's window starts before lane 0's? no: before lane 5's)
    ut[70:80] = ut[70:80][::-1].copy()                # a reversed run inside tile 1
    ut[128] = t[0] - 1.0                              # the first window of tile 2 ends before the stream starts: lane 1 starts at knot 0, below nothing -- and
    ut[129] = ut[127] - 0.3                           # ... lane 1 of tile 2 ends 60 readings BEFORE where lane 0 started
    ds, du = T(stream), T(ut)
    knots, first, count = _torch_cut(ds, du)
    cnt_np, f_np = count.cpu().numpy(), first.cpu().numpy()
    N = max(int(cnt_np.max()), 1)
    tiles = torch.full(((len(ut) + 63) // 64, N + 1, 7, 64), float("nan"), dtype=torch.float64, device=dev)
    _, cnt = eng.assemble_tiles(ds, du, N, tiles=tiles)
    torch.cuda.synchronize()
    assert torch.equal(cnt.to(torch.int32), count)
    for u in range(len(ut)):
        n = int(cnt_np[u])
        assert torch.equal(tiles[u // 64, :n + 1, :, u % 64], knots[int(f_np[u]):int(f_np[u]) + n + 1]), u
    lin = torch.zeros((len(ut), 6), dtype=torch.float64, device=dev)
    for lanes in (0, 1, 4):
        _, c2 = eng.preintegrate_stream(ds, du, lin, None, eng.make_params(1, lanes_per_window=lanes), want=("mean",), N=N, return_counts=True, check_counts=False)
        torch.cuda.synchronize()
        assert torch.equal(c2.to(torch.int32), count), lanes


@pytest.mark.gpu
@pytest.mark.parametrize("model", [1, 2])
def test_gpu_stream_entry_never_reads_behind_the_last_reading(model):
    """with several lanes per window an EMPTY trailing lane segment of a tail window started on the virtual tail
    knot -- whose address, for a window that ends on the stream's LAST reading (update time past the last stamp), lies 56 bytes
    behind the caller's buffer; with imu_avg the loaded w1 / a1 reached the state through 0 * NaN.  Here the stream is the
    front of a larger allocation whose remainder is NaN: every lane split (n % L != 0 included), imu_avg on and off, mean-only
    and everything-out must be finite and bit-identical to the host-assembled ragged layout (which holds no such knot)."""
    import torch
    import cpi_amd
    from cpi_amd import synth
    eng = cpi_amd.Engine()
    stream, upd, lin, q = synth.make_stream(37, 13, seed=11, phase=0.45)       # every window: 13 whole intervals + a tail
    K = stream.shape[0]
    upd = upd.clone(); upd[-1] = stream[-1, 0] + 0.003                          # the last window ends PAST the last stamp: fu == K - 1
    knots, first, count = st.assemble_windows(stream.numpy(), upd.numpy())
    # the last window: its front reading + whole intervals reach the stream's LAST reading, then a tail -- the virtual tail
    # knot would be knot K, one past the buffer
    front = int(np.searchsorted(stream[:, 0].numpy(), float(upd[-2]), side="right")) - 1
    assert front + int(count[-1]) == K and int(count[-1]) >= 2 and float(upd[-1]) > float(stream[-1, 0])
    N = int(count.max())
    big = torch.full((K + 8, 7), float("nan"), dtype=torch.float64, device=eng.device)
    big[:K] = stream.to(eng.device)
    ds = big[:K]                                                                # contiguous view: NaN knots right behind the stream
    T = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(eng.device)
    du, dl, dq, ck, cf, cc = T(upd), T(lin), T(q), T(knots), T(first), T(count)
    for avg in (True, False):
        for lanes in (0, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64):
            prm = eng.make_params(model, avg, True, lanes_per_window=lanes)
            for want in (("mean",), ("mean", "jac"), ("mean", "jac", "cov")):
                for Nb in (N, N + 9, min(K, 65535)):                            # a loose bound picks larger auto lane splits
                    out = eng.preintegrate_stream(ds, du, dl, dq, prm, want=want, N=Nb, check_counts=False)
                    csr = eng.preintegrate(ck, dl, dq, prm, want=want, first=cf, count=cc, N=Nb)
                    torch.cuda.synchronize()
                    for k in csr:
                        assert bool(torch.isfinite(out[k]).all()), (avg, lanes, want, Nb, k)
                        assert torch.equal(out[k], csr[k]), (avg, lanes, want, Nb, k)
    ref = op.oracle().stream(op.make_params(model, 1, 1), stream.numpy(), upd.numpy(), lin.numpy(), q.numpy())
    out = eng.preintegrate_stream(ds, du, dl, dq, eng.make_params(model, True, True), N=N)
    torch.cuda.synchronize()
    check_pre({k: v.cpu().numpy() for k, v in out.items() if not k.startswith("_")}, ref, v2=(model == 2), label="stream tail at the end of the buffer m%d" % model)


@pytest.mark.gpu
@pytest.mark.parametrize("phase", [0.0, 0.4])
def test_gpu_stream_entry_large_batches_take_the_three_knot_kernel_bitwise(phase):
    """From 100 000 windows the mean-only stream entry of model 1 runs cpi_mean_kernel<..., BIG> (three knots per chunk, 32-bit
    staging offsets from a wave-uniform base; cpi_mean.hip; model 2 and the dense layout from 700 000).  Chunking does not touch
    the arithmetic: 600 000 windows x 12 intervals cut out of one stream must equal, BIT FOR BIT, cpi_preintegrate_batch on the
    same windows laid out densely on the device (first knot under the previous update time, the tail knot = the front reading
    under the update time) -- the two-knot kernel at this size, with and without per-window counts --, with and without tail
    intervals, both imu_avg settings, models 1 and 2; a strided sample is held against the oracle.  (A request with Jacobians
    takes the workspace route and the analytic-Jacobian kernel at the same size: its means are compared too.  The dense layout's
    own three-knot launches: tests/test_gpu_parity.py::test_three_knot_kernel_on_the_dense_layout_...)"""
    import torch
    import cpi_amd
    from cpi_amd import synth
    eng = cpi_amd.Engine()
    W, N = 600000, 12
    stream, upd, lin, q = synth.make_stream(W, N, seed=77, device=eng.device, phase=phase)
    K = stream.shape[0]
    n = N + (1 if phase > 0 else 0)
    # the windows, densely: window u = readings u N .. (u + 1) N (+ the tail knot); knot 0 carries the previous update time
    idx = (torch.arange(W, device=eng.device) * N)[:, None] + torch.arange(N + 1, device=eng.device)[None, :]
    dense = stream[idx]                                                     # [W, N + 1, 7]
    prev = torch.cat([stream[:1, 0], upd[:-1]])
    dense[:, 0, 0] = torch.maximum(prev, stream[0, 0])
    if phase > 0:
        tailk = dense[:, -1:, :].clone()
        tailk[:, 0, 0] = upd
        dense = torch.cat([dense, tailk], dim=1)
    dense = dense.contiguous()
    full_counts = torch.full((W,), n, dtype=torch.int32, device=eng.device)
    for model, avg in ((1, False), (1, True), (2, False), (2, True)):
        prm = eng.make_params(model, avg)
        out_m, cnt = eng.preintegrate_stream(stream, upd, lin, q, prm, want=("mean",), N=n, return_counts=True)
        one = eng.make_params(model, avg, lanes_per_window=1)
        ref = eng.preintegrate(dense, lin, q, one, want=("mean",))
        ref2 = eng.preintegrate(dense, lin, q, one, want=("mean",), count=full_counts)
        torch.cuda.synchronize()
        assert int(cnt.min()) == n and int(cnt.max()) == n
        for k in ("DT", "alpha", "beta", "q"):
            assert torch.equal(out_m[k], ref[k]), (phase, model, avg, k)
            assert torch.equal(ref[k], ref2[k]), ("dense with / without counts", phase, model, avg, k)
        if model == 1:
            full = eng.preintegrate_stream(stream, upd, lin, q, prm, want=("mean", "jac"), N=n)
            torch.cuda.synchronize()
            for k in ("DT", "alpha", "beta", "q"):
                # (the analytic-Jacobian kernel forms R by a 3x3 product instead of rotating its columns: equal to rounding, not bitwise)
                assert (full[k] - ref[k]).abs().max().item() < 1e-12, ("workspace route", phase, avg, k)
            if avg:
                out = out_m
    pick = torch.arange(0, W, 9973, device=eng.device)
    o = op.oracle().run(op.make_params(1, 1, 1), dense[pick].cpu().numpy(), lin[pick].cpu().numpy(), q[pick].cpu().numpy())
    check_pre({k: v[pick].cpu().numpy() for k, v in out.items()}, o, what=("mean",), label="BIG stream kernel, phase %.1f" % phase)


def _torch_cut(stream, upd):
    """The closed form of the deque loop (cpi_cut_windows_kernel / the fused cut), vectorised on the stream's device: ->
    (knots [M, 7], first [U] int64, count [U] int32) in the CSR layout of st.assemble_windows.  Valid for non-decreasing stamps
    and update times; the callers below hold its head against st.assemble_windows bit for bit before using it at full size."""
    import torch
    t = stream[:, 0].contiguous()
    K, U = t.shape[0], upd.shape[0]
    cT = torch.searchsorted(t, upd, right=True)
    fp = torch.zeros_like(cT)
    fp[1:] = (cT[:-1] - 1).clamp_min(0)
    start = torch.cat([t[:1], torch.maximum(upd[:-1], t[0])])
    fu = torch.maximum((cT - 1).clamp_min(0), fp)
    m = fu - fp
    front_t = torch.where(m > 0, t[fu], start)
    tail = (upd - front_t) > 0
    count = m + tail.long()
    first = torch.cumsum(count + 1, 0) - (count + 1)
    M = int((count + 1).sum().item())
    u_of = torch.repeat_interleave(torch.arange(U, device=t.device), count + 1, output_size=M)
    j = torch.arange(M, device=t.device) - first[u_of]
    knots = stream[fp[u_of] + torch.minimum(j, m[u_of])].clone()
    knots[first, 0] = start
    last = first + count
    knots[last[tail], 0] = upd[tail]
    return knots.contiguous(), first.contiguous(), count.to(torch.int32).contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("model,W", [(1, 120000), (2, 720000)])
def test_gpu_stream_entry_jittered_updates_three_knot_kernel_ragged_bitwise(model, W):
    """the shipped cpi_mean_kernel<MODEL, false, AVG, 1, CUT = 2, BIG> instantiations on the grids a camera-rate
    caller produces -- update times that are image stamps, not multiples of the IMU period: every window holds N - 2 ... N + 1
    whole intervals plus (mostly) a tail, so NO wavefront has equally long lane-segments and every one runs BIG's per-element
    staging path (the 32-bit offsets advanced per chunk, elements that stop at their own segment's last chunk).  Model 1 from
    100 000 windows, model 2 from 700 000 (cpi_mean.hip: CPI_MEAN_BIG_W / CPI_MEAN_BIG_W_M2).  Expectation: the SAME windows in
    the ragged CSR layout (knots, first, count) through cpi_preintegrate_batch -- the two-knot kernel; a CSR `first` array is
    never admitted to BIG -- bit for bit, the counts exactly, and a strided sample against the oracle."""
    import torch
    import cpi_amd
    from cpi_amd import synth
    eng = cpi_amd.Engine()
    N = 12
    stream, upd, lin, q = synth.make_stream(W, N, seed=91 + model, device=eng.device, phase=0.37)
    g = torch.Generator(device=eng.device); g.manual_seed(5)
    upd = upd + (torch.rand(upd.shape, generator=g, dtype=torch.float64, device=eng.device) * 3.0 - 2.0) / 200.0   # -2 ... +1 samples
    upd = torch.sort(upd).values.contiguous()
    upd[7] = stream[7 * N + 3, 0]                                    # one update exactly ON a reading: no tail interval
    upd = torch.sort(upd).values.contiguous()
    knots, first, count = _torch_cut(stream, upd)
    # the vectorised cut against the host assembler (the deque loop, pinned to the reference by the CPU tests above)
    H = 3000
    hk, hf, hc = st.assemble_windows(stream[: H * N + 4 * N].cpu().numpy(), upd[:H].cpu().numpy())
    assert np.array_equal(hc, count[:H].cpu().numpy()) and np.array_equal(hf, first[:H].cpu().numpy())
    assert np.array_equal(hk, knots[: hk.shape[0]].cpu().numpy())
    cnt_np = count.cpu().numpy()
    assert cnt_np.min() <= N - 1 and cnt_np.max() >= N + 2 and len(np.unique(cnt_np)) >= 4      # ragged for real
    Nb = int(cnt_np.max())
    for avg in (False, True):
        prm = eng.make_params(model, avg, lanes_per_window=1)
        out, cnt = eng.preintegrate_stream(stream, upd, lin, q, prm, want=("mean",), N=Nb, return_counts=True)
        csr = eng.preintegrate(knots, lin, q, prm, want=("mean",), first=first, count=count, N=Nb)
        torch.cuda.synchronize()
        assert torch.equal(cnt.to(torch.int32), count), (model, avg)
        for k in ("DT", "alpha", "beta", "q"):
            assert torch.equal(out[k], csr[k]), (model, avg, k)
    pick = torch.arange(0, W, 7919, device=eng.device)
    pk = pick.cpu().numpy()
    f_np, k_np = first.cpu().numpy(), knots
    wins = [k_np[int(f_np[u]): int(f_np[u]) + int(cnt_np[u]) + 1].cpu().numpy() for u in pk]
    oo = [op.oracle().run(op.make_params(model, 1, 1), w_[None], lin[u:u + 1].cpu().numpy(), q[u:u + 1].cpu().numpy()) for w_, u in zip(wins, pk)]
    ref = {k: np.concatenate([o[k] for o in oo]) for k in ("DT", "alpha", "beta", "q")}
    check_pre({k: out[k][pick].cpu().numpy() for k in ref}, ref, what=("mean",), v2=(model == 2), label="BIG stream kernel, jittered updates, model %d" % model)


@pytest.mark.gpu
def test_gpu_stream_entry_refuses_invalid_calls_before_enqueueing_anything():
    """the entry used to launch the cut kernel -- which writes the caller's workspace -- before prm / out / lin,
    the model, N and q_k_lin were validated.  An invalid call must return CPI_ERR_INVALID and leave the workspace untouched."""
    import ctypes as C
    import torch
    import cpi_amd
    from cpi_amd import _lib, synth
    eng = cpi_amd.Engine()
    lib = eng.lib
    stream, upd, lin, q = synth.make_stream(64, 10, seed=3, device=eng.device, phase=0.3)
    U, K = upd.shape[0], stream.shape[0]
    out = eng.alloc_outputs(U, ("mean", "jac", "cov"), 2)
    o = eng._outputs_struct(out)
    ws = eng.stream_workspace(U)
    ws.fill_(-7.0)
    torch.cuda.synchronize()
    good = eng.make_params(2)
    bad_model = eng.make_params(2); bad_model.model = 9
    bad_lanes = eng.make_params(1); bad_lanes.lanes_per_window = 7
    P = lambda t: C.c_void_p(t.data_ptr())
    calls = [
        (None, 11, P(lin), P(q), C.byref(o)),                        # prm NULL
        (C.byref(good), 11, P(lin), P(q), None),                     # out NULL
        (C.byref(good), 11, None, P(q), C.byref(o)),                 # lin NULL
        (C.byref(good), 11, P(lin), None, C.byref(o)),               # model 2 without q_k_lin
        (C.byref(bad_model), 11, P(lin), P(q), C.byref(o)),          # unknown model
        (C.byref(good), 70000, P(lin), P(q), C.byref(o)),            # N > 65535
        (C.byref(bad_lanes), 11, P(lin), P(q), C.byref(o)),          # unsupported lane split
    ]
    for prm, N, l, qq, oo in calls:
        rc = lib.cpi_preintegrate_stream(eng.ctx, prm, K, P(stream), U, P(upd), N, l, qq, P(ws), oo)
        assert rc == _lib.CPI_ERR_INVALID, (rc, N)
    torch.cuda.synchronize()
    assert bool((ws == -7.0).all()), "an invalid call wrote the workspace"
    rc = lib.cpi_preintegrate_stream(eng.ctx, C.byref(good), K, P(stream), U, P(upd), 11, P(lin), P(q), P(ws), C.byref(o))
    assert rc == 0
    torch.cuda.synchronize()
    assert not bool((ws == -7.0).all())


@pytest.mark.gpu
def test_gpu_stream_entry_forster_and_many_windows():
    """The same entry for the Forster comparator, and on a synthetic stream cut into thousands of windows (every window ends
    in a partial tail interval; wavefronts of the covariance kernels mix windows of different lengths): bit for bit equal to
    the host-assembled ragged layout."""
    import torch
    import cpi_amd
    from cpi_amd import synth
    eng = cpi_amd.Engine()
    stream, upd, lin, q = synth.make_stream(3001, 17, seed=5, phase=0.37)
    upd = upd.clone(); upd[100:200] += 0.02; upd = torch.sort(upd).values      # a few longer / shorter windows
    knots, first, count = st.assemble_windows(stream.numpy(), upd.numpy())
    N = int(count.max())
    T = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(eng.device)
    ds, du, dl, dq, ck, cf, cc = T(stream), T(upd), T(lin), T(q), T(knots), T(first), T(count)
    for model in (1, 2, 3):
        prm = eng.make_params(model)
        qq = dq if model != 3 else None
        out = eng.preintegrate_stream(ds, du, dl, qq, prm, N=N)
        csr = eng.preintegrate(ck, dl, qq, prm, first=cf, count=cc, N=N)
        torch.cuda.synchronize()
        for k in csr:
            assert torch.equal(out[k], csr[k]), (model, k)
        ref = op.oracle().run(op.make_params(model, 0, 1), knots[first[5]:first[5] + count[5] + 1][None], lin[5:6].numpy(), q[5:6].numpy())
        for k in ("DT", "alpha", "beta", "q"):
            assert np.abs(out[k][5].cpu().numpy() - ref[k][0]).max() < 1e-12, (model, k)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(1, 0, 1), (2, 0, 1)])
def test_gpu_stream_windows_vs_deque_oracle(mode):
    import torch
    import cpi_amd
    eng = cpi_amd.Engine()
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    knots, first, count = st.assemble_windows(kn, ut)
    U = len(ut)
    rng = np.random.default_rng(1)
    lin = np.concatenate([0.01 * rng.standard_normal((U, 3)), 0.05 * rng.standard_normal((U, 3))], axis=1)
    q = rng.standard_normal((U, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    out = eng.preintegrate(T(knots), T(lin), T(q), eng.make_params(*mode), first=T(first), count=T(count), N=int(count.max()))
    torch.cuda.synchronize()
    ref = op.oracle().stream(op.make_params(*mode), kn, ut, lin, q)
    check_pre({k: v.cpu().numpy() for k, v in out.items()}, ref, v2=(mode[0] == 2), label="stream %s" % (mode,))


@pytest.mark.gpu
@pytest.mark.parametrize("model", [1, 2, 3])
def test_gpu_stream_host_entry_equals_the_device_entry(model):
    """cpi_preintegrate_stream_host: the same call for a caller that holds the stream, the update times and the
    linearisation points in HOST memory -- bit for bit the device entry's results, the true counts, and a window that
    does not fit its bound is an error."""
    import torch
    import cpi_amd
    eng = cpi_amd.Engine()
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    ut = np.sort(np.concatenate([ut, [kn[0, 0] - 1.0, ut[7], kn[-1, 0] + 0.5]]))
    _, _, count = st.assemble_windows(kn, ut)
    U = len(ut)
    rng = np.random.default_rng(1)
    lin = np.concatenate([0.01 * rng.standard_normal((U, 3)), 0.05 * rng.standard_normal((U, 3))], axis=1)
    q = rng.standard_normal((U, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    H = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    prm = eng.make_params(model)
    qq = None if model == 3 else H(q)
    host, cnt = eng.preintegrate_stream_host(H(kn), H(ut), H(lin), qq, prm, return_counts=True)
    dev = eng.preintegrate_stream(H(kn).to(eng.device), H(ut).to(eng.device), H(lin).to(eng.device),
                                  None if qq is None else qq.to(eng.device), prm)
    torch.cuda.synchronize()
    assert np.array_equal(cnt.numpy(), count)
    for k, v in host.items():
        assert torch.equal(v, dev[k].cpu()), (model, k)
    only_means = eng.preintegrate_stream_host(H(kn), H(ut), H(lin), qq, prm, want=("mean",), N=int(count.max()), pinned=True)
    # (a mean-only request runs the composing mean kernel, a full one carries the means in the covariance kernel: same to rounding)
    assert sorted(only_means) == ["DT", "alpha", "beta", "q"] and (only_means["alpha"] - host["alpha"]).abs().max().item() < 1e-12
    with pytest.raises(ValueError):
        eng.preintegrate_stream_host(H(kn), H(ut), H(lin), qq, prm, N=int(count.max()) - 1)


@pytest.mark.gpu
@pytest.mark.parametrize("model", [1, 2])
def test_gpu_cpp_imu_stream_facade(model):
    """cpi_host::ImuStream -- the caller's loop of GraphSolver_IMU.cpp:34-134 for every update time in one call, from C++
    through the C-ABI: the reference's IMU text excerpt -> parse_imu_text -> ImuStream::preintegrate -> one CpiResult per
    window, against the oracle's restatement of the reference's deque loop; then window 1 wrapped in an ImuFactorCPI and
    evaluated, against the oracle's evaluateError."""
    from cpi_amd import _lib
    _lib.load()
    kn = st.parse_imu_text(open(DATA).read())
    ut = _updates(kn)
    _, _, count = st.assemble_windows(kn, ut)
    U = len(ut)
    rng = np.random.default_rng(5)
    lin = np.concatenate([0.01 * rng.standard_normal((U, 3)), 0.05 * rng.standard_normal((U, 3))], axis=1)
    q = rng.standard_normal((U, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
    tmp = tempfile.mkdtemp()
    exe, libdir = os.path.join(tmp, "test_imu_stream"), os.path.join(ROOT, "cpi_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_imu_stream.cpp"), "-o", exe,
                           "-L" + libdir, "-lcpi_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    np.savetxt(os.path.join(tmp, "ut.txt"), ut, fmt="%.17g")
    np.savetxt(os.path.join(tmp, "lin.txt"), np.concatenate([lin, q], axis=1), fmt="%.17g")
    p = subprocess.run([exe, DATA, os.path.join(tmp, "ut.txt"), os.path.join(tmp, "lin.txt"), str(model)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().split("\n")
    rows = np.array([[float(x) for x in ln.split()] for ln in lines[:U]])
    names = [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4), ("J_q", 9), ("J_a", 9), ("J_b", 9), ("H_a", 9), ("H_b", 9),
             ("O_a", 9), ("O_b", 9), ("P", 225)]
    out, o = {}, 0
    for name, n in names:
        out[name] = rows[:, o] if n == 1 else rows[:, o:o + n]
        o += n
    oprm = op.make_params(model, 0, 1)
    ref = op.oracle().stream(oprm, kn, ut, lin, q)
    check_pre(out, ref, v2=(model == 2), label="c++ ImuStream m%d" % model)
    assert lines[U].split()[0] == "COUNT" and np.array_equal(np.array(lines[U].split()[1:], dtype=np.int32), count)
    assert lines[U + 1] == "BOUND 1"
    err = np.array([float(x) for x in lines[U + 2].split()[1:]])
    rec = op.factor_records({k: v[1:2] for k, v in ref.items()}, lin[1:2], q[1:2] if model == 2 else None)
    x = np.array([[0, 0, 0, 1, *lin[1, 0:3], 0.1, -0.2, 0.05, *lin[1, 3:6], 1, 2, 3]], dtype=np.float64)
    e_ref, _, _ = op.oracle().factor(model, rec, x, x)
    assert np.abs(err - e_ref[0]).max() <= 1e-9 * max(1.0, np.abs(e_ref).max())
