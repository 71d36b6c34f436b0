"""CPU baseline thread scaling (development tool): the compiled reference (oracle/_ref) and the C restatement on
W windows x 50 samples with 1..256 threads, output buffer reused.   python tests/tools/cpu_scale.py [W]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cpi_amd import synth  # noqa: E402
from oracle import oracle_py as op  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
kn, lin, q = [t.numpy() for t in synth.make_windows(W, 50, seed=5)]
prm = op.make_params(1, 0, 1)
raw = np.ones((W, op.OUT_DOUBLES))
for name, lib in (("reference", op.reference()), ("port", op.oracle())):
    if lib is None:
        continue
    for nt in (1, 16, 32, 64, 128, 256):
        w = W if nt > 1 else 2000
        lib.run(prm, kn[:w], lin[:w], q[:w], nthreads=nt, raw=raw[:w])
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.5:
            lib.run(prm, kn[:w], lin[:w], q[:w], nthreads=nt, raw=raw[:w]); n += w
        print(name, nt, "%.0f windows/s" % (n / (time.perf_counter() - t0)), flush=True)
