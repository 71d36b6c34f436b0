"""Timing of the 'next' rows of SURVEY.md section 8(f) (development tool): sqrt-information, whitened
evaluateError and state prediction on a 1 M-factor sweep, HIP events on the engine's stream.
   python tools/aux_bench.py [F]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cpi_amd  # noqa: E402
from cpi_amd import synth  # noqa: E402


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    eng = cpi_amd.Engine(device=0)
    dev = eng.device
    for model in (1, 2):
        kn, lin, q = synth.make_windows(F, 50, seed=7, device=dev)
        W0 = min(F, 100000)   # covariances of a 100 k slice, tiled: the Cholesky cost does not depend on the values
        full = eng.preintegrate(kn[:W0], lin[:W0], q[:W0], eng.make_params(model), want=("mean", "jac", "cov"))
        P = full["P"].repeat((F + W0 - 1) // W0, 1)[:F].contiguous()
        del full
        meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac"))
        torch.cuda.synchronize()
        del kn
        xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=dev)
        states = torch.cat([xi, xj[-1:]], dim=0).contiguous()
        qq = q if model == 2 else None
        R = eng.sqrt_information(P)
        out = {"err": torch.empty((F, 15), dtype=torch.float64, device=dev),
               "H1": torch.empty((F, 225), dtype=torch.float64, device=dev),
               "H2": torch.empty((F, 225), dtype=torch.float64, device=dev)}
        t_sq = timeit(lambda: eng.lib.cpi_sqrt_information_batch(eng.ctx, F, P.data_ptr(), R.data_ptr()))
        t_pl = timeit(lambda: eng.factor_eval(model, meas, lin, qq, states, out=out))
        t_wh = timeit(lambda: eng.factor_eval(model, meas, lin, qq, states, out=out, sqrt_info=R))
        t_pr = timeit(lambda: eng.predict(model, meas, states))
        pk = torch.empty((F, 72), dtype=torch.float64, device=dev)
        t_pk = timeit(lambda: eng.factor_eval_packed(model, meas, lin, qq, states, out=pk))
        hs = torch.empty((F, 496), dtype=torch.float64, device=dev)
        t_hs = timeit(lambda: eng.factor_hessian(model, meas, lin, qq, states, R, out=hs))
        del hs
        gb = lambda b, ms: b * F / (ms * 1e-3) / 1e9
        in_b = 776 if model == 1 else 952
        print("model %d F=%d  sqrt_info %.3f ms (%.0f GB/s)  factor %.3f ms (%.0f GB/s)  whitened %.3f ms (%.0f GB/s)  "
              "predict %.3f ms (%.0f GB/s)  packed factor %.3f ms (%.0f GB/s)  hessian blocks %.3f ms (%.0f GB/s)" % (
                  model, F, t_sq, gb(3600, t_sq), t_pl, gb(in_b + 3720, t_pl), t_wh, gb(in_b + 3720 + 1800, t_wh),
                  t_pr, gb(88 + 128 + 128, t_pr), t_pk, gb(in_b + 576, t_pk), t_hs, gb(in_b + 1800 + 3968, t_hs)), flush=True)
        del P, R, out, meas, states
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
