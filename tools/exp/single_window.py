"""Latency of ONE window through the host-pointer entry (BASELINE configs[0]: 1 window x 100 samples, model 1, everything
out) -- what the CpiV1-shaped facade's finalize() costs -- and of small batches; development measurement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cpi_amd
from cpi_amd import synth

eng = cpi_amd.Engine(device=0)
for model in (1, 2):
    for W in (1, 16, 256):
        kn, lin, q = synth.make_windows(W, 100, seed=5)
        kn, lin, q = kn.pin_memory(), lin.pin_memory(), q.pin_memory()
        prm = eng.make_params(model)
        out = eng.preintegrate_host(kn, lin, q, prm)
        for _ in range(20):
            eng.preintegrate_host(kn, lin, q, prm, out=out)
        t0 = time.perf_counter(); reps = 200
        for _ in range(reps):
            eng.preintegrate_host(kn, lin, q, prm, out=out)
        dt = (time.perf_counter() - t0) / reps
        print("model %d  W=%-4d x 100 samples, everything out, host pointers: %7.1f us per call (%.1f us per window)" % (model, W, dt * 1e6, dt * 1e6 / W), flush=True)
