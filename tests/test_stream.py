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
