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
cpi_preintegrate_batch.  The C++ twin lives in cpi_amd/csrc/cpi_host.hpp.
"""
import numpy as np


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


def assemble_windows(stream, update_times):
    """stream [K,7], update_times [U] (non-decreasing) -> (knots [M,7], first [U] int64, count [U] int32).
    Window u covers (previous update time, update_times[u]]; its first knot carries the reading that was at
    the front of the reference's deque when the window started."""
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
    return (np.asarray(out, dtype=np.float64).reshape(-1, 7), np.asarray(first, dtype=np.int64),
            np.asarray(count, dtype=np.int32))
