"""PCIe-inclusive rate of the host-pointer convenience entry point (development tool; never the benchmarked path)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cpi_amd
from cpi_amd import synth
from cpi_amd._lib import CpiOutputs

eng = cpi_amd.Engine(device=0)
lib = eng.lib
for W in (10000, 100000):
    kn, lin, q = synth.make_windows(W, 50, seed=3)
    kn, lin = np.ascontiguousarray(kn.numpy()), np.ascontiguousarray(lin.numpy())
    out = {k: np.zeros((W, n)) for k, n in (("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4))}
    o = CpiOutputs()
    for k, v in out.items():
        setattr(o, k, v.ctypes.data)
    prm = eng.make_params(1)
    dp = lambda a: a.ctypes.data_as(C.c_void_p)
    def call():
        rc = lib.cpi_preintegrate_batch_host(eng.ctx, C.byref(prm), W, 50, dp(kn), None, None, 0, dp(lin), None, C.byref(o))
        assert rc == 0, lib.cpi_last_error(eng.ctx)
    call(); call()
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps): call()
    dt = (time.perf_counter() - t0) / reps
    print("host-pointer path: W=%d  %.3f ms per call  %.1f M windows/s  (%.1f GB/s over PCIe, pageable host memory)" % (
        W, dt * 1e3, W / dt / 1e6, W * 2944 / dt / 1e9))
