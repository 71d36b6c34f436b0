"""GPU (development): the measured floor behind REG_FACTOR (tests/tol.py) -- max |HIP - restatement| of evaluateError on the golden
factor cases (tests/golden/factor_256.npz), per output and model.  Prints; the gate is ~100 x the floor."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import cpi_amd
    from oracle import oracle_py as orc
    eng = cpi_amd.Engine(device=0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "factor_256.npz")))
    for model in (1, 2):
        rec, xi, xj = d["v%d_rec" % model], d["v%d_xi" % model], d["v%d_xj" % model]
        F = rec.shape[0]
        cols, o = {}, 0
        for name, n in orc.FACTOR_FIELDS:
            cols[name] = rec[:, o:o + n]; o += n
        meas = dict(DT=cols["deltatime"][:, 0], alpha=cols["alpha"], beta=cols["beta"], q=cols["q_KtoK1"], J_q=cols["J_q"], J_b=cols["J_beta"],
                    J_a=cols["J_alpha"], H_b=cols["H_beta"], H_a=cols["H_alpha"], O_b=cols["O_beta"], O_a=cols["O_alpha"])
        meas = {k: T(v) for k, v in meas.items()}
        lin = np.concatenate([cols["bg_lin"], cols["ba_lin"]], axis=1)
        states = np.concatenate([xi, xj], axis=0)
        ii = np.arange(F, dtype=np.int32)
        out = eng.factor_eval(model, meas, T(lin), T(cols["q_K_lin"]), T(states), T(ii), T(ii + F))
        torch.cuda.synchronize()
        print("model %d evaluateError vs restatement (%d golden cases, max |entry| %.1f):" % (model, F, float(np.abs(d["v%d_H1" % model]).max())),
              {k: "%.2e" % float(np.abs(out[k].cpu().numpy() - d["v%d_%s" % (model, k)]).max()) for k in ("err", "H1", "H2")})


if __name__ == "__main__":
    main()
