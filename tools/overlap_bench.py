"""Throughput of the headline workload when independent batches are issued through SEVERAL engine contexts (one HIP
stream each), so that consecutive launches may overlap (development tool; NOT what bench.py's `value` reports -- with
overlapping launches "the duration of a launch" stops being a meaningful quantity for the roofline line).
   python tools/overlap_bench.py [windows] [contexts ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cpi_amd  # noqa: E402
from cpi_amd import synth  # noqa: E402


def run(W, nctx, steps=4000, N=50):
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nctx)]
    engs = [cpi_amd.Engine(device=0, stream=s) for s in streams]
    nb = max(nctx, -(-(320 << 20) // (W * (N + 1) * 56)))
    batches = [synth.make_windows(W, N, seed=100 + b, device=dev) for b in range(nb)]
    outs = [engs[0].alloc_outputs(W, ("mean",), 1) for _ in range(max(4, 2 * nctx))]
    prm = engs[0].make_params(1)
    torch.cuda.synchronize()

    def go(k):
        for i in range(k):
            kn, lin, q = batches[i % nb]
            engs[i % nctx].preintegrate(kn, lin, q, prm, want=("mean",), out=outs[i % len(outs)])
    go(400)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        go(steps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("W=%d contexts=%d  %.2f us per batch  %.4g windows/s" % (W, nctx, best / steps * 1e6, W * steps / best), flush=True)


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    for n in ([int(x) for x in sys.argv[2:]] or [1, 2, 3, 4]):
        run(W, n)
