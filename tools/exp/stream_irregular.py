"""Timing of cpi_preintegrate_stream (mean-only, model 1 / 2) on IRREGULAR update grids -- the shape a camera-rate caller produces
(GraphSolver_IMU.cpp:43-75: update times are image stamps, not multiples of the IMU period): every window holds N - 2 ... N + 1
whole intervals plus a partial tail, so no wavefront has equally long lane-segments and every one takes the per-element staging path.
    CPI_AMD_LIB=cpi_amd/libcpi_amd_<tag>.so python tools/exp/stream_irregular.py [windows] [samples] [jitter 0|1]
Prints microseconds per launch (HIP events over 10 launches, best of 3) and a checksum of the outputs (same number under every
library that computes the same thing)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpi_amd  # noqa: E402
from cpi_amd import synth  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    jitter = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    eng = cpi_amd.Engine(device=0)
    tag = os.path.basename(os.environ.get("CPI_AMD_LIB", "default"))
    stream, upd, lin, q = synth.make_stream(W, N, seed=77, device=eng.device, phase=0.37)
    if jitter:
        g = torch.Generator(device=eng.device); g.manual_seed(5)
        upd = upd + (torch.rand(upd.shape, generator=g, dtype=torch.float64, device=eng.device) * 3.0 - 2.0) / 200.0   # -2 ... +1 samples
        upd = torch.sort(upd).values.contiguous()
    for model in (1, 2):
        prm = eng.make_params(model=model)
        out = eng.alloc_outputs(W, ("mean",), model)
        ws = eng.stream_workspace(W)
        run = lambda: eng.preintegrate_stream(stream, upd, lin, q if model == 2 else None, prm, want=("mean",), N=N + 3, out=out,
                                              check_counts=False, workspace=ws)
        run(); torch.cuda.synchronize()
        best = 1e30
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100.0)
        chk = float(out["alpha"].double().abs().sum().item()) + float(out["q"].abs().sum().item())
        print("%-24s stream %s W=%d N=%d model %d: %9.1f us per launch   checksum %.15e" % (
            tag, "jittered" if jitter else "uniform", W, N, model, best, chk), flush=True)


if __name__ == "__main__":
    main()
