"""Run-to-run determinism of every sweep (development check, GPU): the kernels order their LDS exchanges by in-order DS
execution within single-wavefront workgroups (compiler-only fences) -- a race there would show up as outputs that differ
between repetitions of the same call.  python tests/tools/determinism.py [repetitions]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpi_amd  # noqa: E402
from cpi_amd import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    eng = cpi_amd.Engine(device=0)
    bad = 0
    for model in (1, 2, 3):
        for W, N in ((100003, 50), (4099, 100), (517, 23)):
            kn, lin, q = synth.make_windows(W, N, seed=1000 + model * 10 + N, device=eng.device)
            prm = eng.make_params(model)
            ref = {k: v.clone() for k, v in eng.preintegrate(kn, lin, q, prm).items() if not k.startswith("_")}
            for r in range(reps):
                out = eng.preintegrate(kn, lin, q, prm)
                torch.cuda.synchronize()
                for k, v in ref.items():
                    if not torch.equal(out[k], v):
                        bad += 1
                        print("DIFF preintegrate model %d W %d N %d field %s repetition %d" % (model, W, N, k, r), flush=True)
            if model < 3:
                tiles = eng.tile_knots(kn)
                tref = {k: v.clone() for k, v in eng.preintegrate_tiled(tiles, W, lin, q, prm).items() if not k.startswith("_")}
                for r in range(reps):
                    out = eng.preintegrate_tiled(tiles, W, lin, q, prm)
                    torch.cuda.synchronize()
                    bad += sum(0 if torch.equal(out[k], v) else 1 for k, v in tref.items())
    # the stream producers / the zero-copy stream entry (round 3)
    for model in (1, 2, 3):
        stream, upd, lin, q = synth.make_stream(50001, 23, seed=9 + model, device=eng.device, phase=0.4)
        prm = eng.make_params(model)
        qq = q if model != 3 else None
        ref = {k: v.clone() for k, v in eng.preintegrate_stream(stream, upd, lin, qq, prm, N=24).items() if not k.startswith("_")}
        for r in range(reps):
            out = eng.preintegrate_stream(stream, upd, lin, qq, prm, N=24)
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(out[k], v) else 1 for k, v in ref.items())
        if model == 1:
            t0, c0 = eng.assemble_tiles(stream, upd, 24)
            t0, c0 = t0.clone(), c0.clone()
            for r in range(reps):
                t1, c1 = eng.assemble_tiles(stream, upd, 24)
                torch.cuda.synchronize()
                # rows past a window's count are never written: compare what the contract defines
                bad += 0 if (torch.equal(c1, c0) and torch.equal(t1[:, :25], t0[:, :25])) else 1
    for model in (1, 2):
        F = 200003
        kn, lin, q = synth.make_windows(F, 20, seed=77 + model, device=eng.device)
        meas = eng.preintegrate(kn, lin, q, eng.make_params(model))
        xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device)
        states = torch.cat([xi, xj[-1:]], 0).contiguous()
        qq = q if model == 2 else None
        R = eng.sqrt_information(meas["P"])
        calls = {"dense": lambda: eng.factor_eval(model, meas, lin, qq, states), "packed": lambda: {"p": eng.factor_eval_packed(model, meas, lin, qq, states)},
                 "whitened": lambda: eng.factor_eval(model, meas, lin, qq, states, sqrt_info=R), "hessian": lambda: {"h": eng.factor_hessian(model, meas, lin, qq, states, R)},
                 "sqrt_info": lambda: {"r": eng.sqrt_information(meas["P"])}}
        for name, fn in calls.items():
            ref = {k: v.clone() for k, v in fn().items()}
            for r in range(reps):
                out = fn()
                torch.cuda.synchronize()
                for k, v in ref.items():
                    if not torch.equal(out[k], v):
                        bad += 1
                        print("DIFF %s model %d field %s repetition %d" % (name, model, k, r), flush=True)
    print("determinism: %d repetitions per sweep, %d differing outputs" % (reps, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
