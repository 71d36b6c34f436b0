"""V1 full (100 k x 50) and its parts: the covariance kernel alone (want = cov + mean), the analytic-Jacobian mean kernel alone
(want = mean + jac), both (as shipped: the Jacobian kernel on a side stream).  HIP events over 20 launches, best of 3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpi_amd  # noqa: E402
from cpi_amd import synth  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    eng = cpi_amd.Engine(device=0)
    batches = [synth.make_windows(W, 50, seed=5 + b, device=eng.device) for b in range(3)]
    for model in (1, 2):
        prm = eng.make_params(model)
        for want in (("mean", "jac", "cov"), ("mean", "cov"), ("cov",), ("mean", "jac"), ("mean",)):
            outs = [eng.alloc_outputs(W, want, model) for _ in range(3)]
            run = lambda i: eng.preintegrate(*batches[i % 3], prm, want=want, out=outs[i % 3])
            for i in range(3):
                run(i)
            torch.cuda.synchronize()
            best = 1e30
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(20):
                    run(i)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            print("model %d  W=%d  want=%-22s %8.3f ms" % (model, W, "+".join(want), best), flush=True)


if __name__ == "__main__":
    main()
