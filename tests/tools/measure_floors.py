"""GPU (development): measured floors behind the regression gates of tests/tol.py for SURVEY 8(f1) / 8(f4) -- the distance of
the HIP results from the best available reference on realistic inputs.  Prints; the gates are set to ~100 x these."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import cpi_amd
    from cpi_amd import synth
    from oracle import oracle_py as op
    from tests.tol import cov_rel_err, sqrt_info_longdouble
    eng = cpi_amd.Engine(device=0)
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "pre_w48.npz")))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    worst = {}
    for kn, lin in ((d["knots"], d["lin"]),) + tuple((lambda t: (t[0].numpy(), t[1].numpy()))(synth.make_windows(2000, n, seed=s)) for n, s in ((50, 1), (100, 2), (7, 3))):
        out = eng.preintegrate(T(kn), T(lin), None, eng.make_params(3))
        torch.cuda.synchronize()
        out = {k: v.cpu().numpy() for k, v in out.items()}
        ref = op.oracle().run(op.make_params(3), kn, lin)
        for k in out:
            e = cov_rel_err(out[k], ref[k]) if k == "P" else float(np.abs(out[k] - ref[k]).max())
            worst[k] = max(worst.get(k, 0.0), e)
    print("forster vs restatement:", {k: "%.2e" % v for k, v in worst.items()})
    for model in (1, 2):
        kn, lin, q = synth.make_windows(2003, 50, seed=91, device=eng.device, edge_cases=False)
        meas = eng.preintegrate(kn, lin, q, eng.make_params(model))
        R = eng.sqrt_information(meas["P"])
        torch.cuda.synchronize()
        P = meas["P"].cpu().numpy().reshape(-1, 15, 15)
        Rg = R.cpu().numpy().reshape(-1, 15, 15).transpose(0, 2, 1)
        Rl = sqrt_info_longdouble(P)
        Rn = np.stack([np.linalg.cholesky(np.linalg.inv(P[f])).T for f in range(P.shape[0])])
        scale = np.abs(Rl).max(axis=(1, 2))[:, None, None]
        print("model %d sqrt_info: HIP vs longdouble %.2e   LAPACK(f64) vs longdouble %.2e   (relative to max |R| per factor)" % (
            model, float((np.abs(Rg - Rl) / scale).max()), float((np.abs(Rn - Rl) / scale).max())))


if __name__ == "__main__":
    main()
