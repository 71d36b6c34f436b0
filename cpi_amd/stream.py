"""Caller-side data formats of the hot path:

  parse_imu_text     the reference's simulated-IMU wire format, one reading per line
                     ``wx wy wz ax ay az 0 t_ms`` (cpi_compare/src/sim/SimParser.h:130-193:
                     fields split on single spaces, empty fields skipped, column 8 is the stamp in ms)
  assemble_windows   cutting ONE IMU stream into preintegration windows at successive update (camera)
                     times exactly like GraphSolver::createimufactor_cpi_v1/v2
                     (cpi_compare/src/solvers/GraphSolver_IMU.cpp:50-69): whole intervals while
                     imu_times[1] <= updatetime, then the partial tail interval with the front reading
                     repeated, after which the front stamp is overwritten by the update time.

Both produce the knot layout of include/cpi_amd.h (knots[K][7] + first[W] + count[W]), ready for
cpi_preintegrate_batch -- or, with layout="tiled", the TILED layout of cpi_preintegrate_tiled_batch
(tiles[ceil(W/64)][N+1][7][64] + count[W]): the mean-only recursion is HBM-bound and that is the layout it streams
fastest; tile_windows() converts windows a caller already holds.  The C++ twin lives in
cpi_amd/csrc/cpi_host.hpp; the device-side assembler is cpi_assemble_tiles (Engine.assemble_tiles).
"""
import numpy as np


def tile_windows(knots, first=None, count=None, N=None):
    """knots [W,N+1,7] (dense) or a shared stream [K,7] with first[W] / count[W]  ->  tiles [ceil(W/64), N+1, 7, 64]:
    tiles[b, s, k, i] = field k of knot s of window 64 b + i.  Rows past a window's last knot repeat that knot, columns
    past W repeat window W - 1 (finite padding the kernels never integrate).  Mirrors cpi_tile_windows."""
    knots = np.asarray(knots, dtype=np.float64)
    if first is None:
        W, n1, _ = knots.shape
        N = n1 - 1 if N is None else N
        flat = knots.reshape(-1, 7)
        first = np.arange(W, dtype=np.int64) * n1
        count = np.full(W, n1 - 1, dtype=np.int64) if count is None else np.asarray(count, dtype=np.int64)
    else:
        flat = knots.reshape(-1, 7)
        first = np.asarray(first, dtype=np.int64)
        count = np.asarray(count, dtype=np.int64)
        W = first.shape[0]
        N = int(count.max()) if N is None else N
    nb = (W + 63) // 64
    w = np.minimum(np.arange(nb * 64), W - 1)
    rows = first[w][:, None] + np.minimum(np.arange(N + 1)[None, :], np.clip(count[w], 0, N)[:, None])   # [nb*64, N+1]
    g = flat[rows]                                                                                        # [nb*64, N+1, 7]
    return np.ascontiguousarray(g.reshape(nb, 64, N + 1, 7).transpose(0, 2, 3, 1))


def parse_imu_text(text):
    """-> knots [K,7] = {t[s], w[3], a[3]}.  Malformed / short lines are skipped like blank ones."""
    rows = []
    for line in text.splitlines():
        f = [x for x in line.split(" ") if x]
        if len(f) < 8:
            continue
        wx, wy, wz, ax, ay, az = (float(f[i]) for i in range(6))
        rows.append((1e-3 * float(f[7]), wx, wy, wz, ax, ay, az))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 7)


def assemble_windows(stream, update_times, layout="csr", N=None):
    """stream [K,7], update_times [U] (non-decreasing) -> (knots [M,7], first [U] int64, count [U] int32).
    Window u covers (previous update time, update_times[u]]; its first knot carries the reading that was at
    the front of the reference's deque when the window started.
    layout="tiled": -> (tiles [ceil(U/64), N+1, 7, 64], count [U] int32) with N = the largest count (or the given N)."""
    stream = np.asarray(stream, dtype=np.float64)
    K = stream.shape[0]
    out, first, count = [], [], []
    front = 0
    front_t = stream[0, 0]
    for T in np.asarray(update_times, dtype=np.float64):
        first.append(len(out))
        out.append(np.concatenate([[front_t], stream[front, 1:]]))
        n = 0
        while K - front > 1 and stream[front + 1, 0] <= T:
            front += 1
            front_t = stream[front, 0]
            out.append(stream[front].copy())      # intervals with dt < 0 stay in the list: the kernels skip them
            n += 1
        if T - front_t > 0:
            out.append(np.concatenate([[T], stream[front, 1:]]))
            front_t = T
            n += 1
        count.append(n)
    knots, first, count = (np.asarray(out, dtype=np.float64).reshape(-1, 7), np.asarray(first, dtype=np.int64),
                           np.asarray(count, dtype=np.int32))
    if layout == "tiled":
        return tile_windows(knots, first, count, N), count
    assert layout == "csr", layout
    return knots, first, count
