"""Development check of the LDS-DMA mean kernel (CPI_AMD_MEAN_DMA=KC,S,A) against the oracle: dense batches whose size
is not a multiple of 64 (DMA blocks + cpi_mean_kernel tail), odd window lengths, per-window counts, models 1 / 2, imu_avg.
(Test infrastructure: lives under tests/tools; imports oracle/.)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpi_amd  # noqa: E402
from cpi_amd import synth  # noqa: E402
from cpi_amd.engine import _ptr  # noqa: E402
from oracle import oracle_py as op  # noqa: E402


def run(eng, kn, lin, q, prm, cnt, N):
    W = kn.shape[0]
    out = eng.alloc_outputs(W, ("mean",), prm.model)
    o = eng._outputs_struct(out)
    eng._check(eng.lib.cpi_preintegrate_batch(eng.ctx, C.byref(prm), W, N, _ptr(kn), None, _ptr(cnt), _ptr(lin), _ptr(q), C.byref(o)))
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def main():
    eng = cpi_amd.Engine(device=0)
    worst = 0.0
    cases = [(64 * 20 + 37, 50, False), (64 * 9 + 5, 23, False), (64 * 6 + 1, 50, True), (700, 100, False), (64 * 4, 8, False)]
    if os.environ.get("CPI_AMD_MEAN_LINE"):
        # cpi_mean_line_kernel takes the leading whole groups of 64 P windows (P = phase classes of the window stride): batches of
        # several groups + a tail for P = 16 (N = 50, 100), P = 4 (N = 51: 7 x 52 = 4 x 91), P = 2 (N = 9: 70), P = 1 (N = 15: 112)
        cases = [(1024 * 3 + 37, 50, False), (1024 * 2 + 1, 100, False), (256 * 5 + 70, 51, False), (128 * 9, 9, False),
                 (64 * 21 + 3, 15, False), (1024 + 5, 50, True)]
    for (W, N, use_count) in cases:
        kn, lin, q = synth.make_windows(W, N, seed=77 + W + N, device=eng.device)
        cnt = None
        if use_count:
            g = torch.Generator(device="cpu"); g.manual_seed(5)
            cnt = torch.randint(0, N + 1, (W,), generator=g, dtype=torch.int32).to(eng.device)
        knh, linh, qh = kn.cpu().numpy(), lin.cpu().numpy(), q.cpu().numpy()
        for model in (1, 2):
            for avg in (0, 1):
                prm = eng.make_params(model, imu_avg=bool(avg), lanes_per_window=1)
                out = run(eng, kn, lin, q, prm, cnt, N)
                oprm = op.make_params(model, avg, 1)
                if cnt is None:
                    ref = op.oracle().run(oprm, knh, linh, qh, nthreads=8)
                else:
                    ref = {k: np.zeros_like(v) for k, v in out.items()}
                    ch = cnt.cpu().numpy()
                    for w in range(W):
                        r = op.oracle().run(oprm, knh[w:w + 1, :ch[w] + 1], linh[w:w + 1], qh[w:w + 1])
                        for k in ref:
                            ref[k][w] = r[k][0]
                for k in ("DT", "alpha", "beta", "q"):
                    e = float(np.abs(out[k] - ref[k]).max())
                    worst = max(worst, e)
                    assert e < 1e-11, (W, N, model, avg, k, e)
    cfg = ",".join("%s=%s" % (k[8:], os.environ[k]) for k in ("CPI_AMD_MEAN_DMA", "CPI_AMD_MEAN_BLK", "CPI_AMD_MEAN_LINE") if os.environ.get(k))
    print("dma_check ok cfg=%s worst=%.3e" % (cfg or "default", worst), flush=True)


if __name__ == "__main__":
    main()
