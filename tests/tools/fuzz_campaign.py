"""Randomised parity campaign on the GPU (test infrastructure -- it uses the oracle, so it lives under tests/; not collected by pytest): random shapes, layouts,
models (1, 2, Forster comparator), flags, lane splits and output subsets against the CPU oracle; then random factor
sweeps (dense, packed, whitened) against the oracle.   python tests/tools/fuzz_campaign.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpi_amd  # noqa: E402
from cpi_amd import synth  # noqa: E402
from oracle import oracle_py as op  # noqa: E402
from tests.tol import check_pre  # noqa: E402


def dev(a, eng):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


def host(out):
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    eng = cpi_amd.Engine(device=0)
    lanes_all = [0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64]
    wants = [("mean",), ("mean", "jac"), ("mean", "jac", "cov"), ("cov",), ("jac",), ("mean", "cov")]
    fails = 0
    for case in range(cases):
        W = int(rng.integers(1, 700))
        N = int(rng.integers(1, 130))
        model = int(rng.integers(1, 4))
        avg = int(rng.integers(0, 2)) if model < 3 else 0
        stj = int(rng.integers(0, 2)) if model == 2 else 1
        ragged = bool(rng.integers(0, 2))
        lanes = int(rng.choice(lanes_all))
        want = wants[int(rng.integers(0, len(wants)))]
        label = "case %d W%d N%d m%d avg%d stj%d %s L%d %s" % (case, W, N, model, avg, stj, "ragged" if ragged else "dense", lanes, want)
        oprm = op.make_params(model, avg, stj)
        prm = eng.make_params(model, avg, stj, lanes_per_window=lanes)
        try:
            if not ragged:
                kn, lin, q = synth.make_windows(W, N, seed=seed * 100000 + case)
                kn, lin, q = kn.numpy(), lin.numpy(), q.numpy()
                ref = op.oracle().run(oprm, kn, lin, q, nthreads=8)
                out = host(eng.preintegrate(dev(kn, eng), dev(lin, eng), dev(q, eng), prm, want=want))
                # the same batch through the host-pointer entry (one pipeline chunk here: same kernels, same bits) ...
                hout = eng.preintegrate_host(torch.from_numpy(kn), torch.from_numpy(lin), torch.from_numpy(q), prm, want=want,
                                             pinned=bool(rng.integers(0, 2)))
                for k in out:
                    assert np.array_equal(hout[k].numpy(), out[k]), "host entry differs in " + k
                # ... and, for models 1 / 2, the mean outputs through the tiled layout with a random split over wavefronts
                if model < 3:
                    split = int(rng.choice([0, 1, 2, 3, 4, 5, 8]))
                    tiles = eng.tile_knots(dev(kn, eng))
                    tprm = eng.make_params(model, imu_avg=bool(prm.imu_avg), state_transition_jacobians=bool(prm.state_transition_jacobians),
                                           lanes_per_window=split)   # tiled entry: wavefronts per tile
                    tout = host(eng.preintegrate_tiled(tiles, W, dev(lin, eng), dev(q, eng), tprm))
                    check_pre(tout, ref, what=("mean",), v2=(model == 2), label=label + " tiled split %d" % split)
            else:
                lens = rng.integers(0, N + 1, W).astype(np.int32)
                lens[rng.integers(0, W)] = N
                K = int(lens.sum()) + 1
                kn1, _, _ = synth.make_windows(1, max(K - 1, 1), seed=seed * 100000 + 50000 + case, edge_cases=False)
                stream = kn1.numpy()[0][:K]
                first = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
                _, lin, q = synth.make_windows(W, 4, seed=seed * 100000 + 70000 + case)
                lin, q = lin.numpy(), q.numpy()
                out = host(eng.preintegrate(dev(stream, eng), dev(lin, eng), dev(q, eng), prm, want=want,
                                            first=dev(first, eng), count=dev(lens, eng), N=N))
                ref = None
                for w in range(W):
                    n = int(lens[w])
                    r = op.oracle().run(oprm, stream[first[w]:first[w] + n + 1][None], lin[w:w + 1], q[w:w + 1])
                    if ref is None:
                        ref = {k: np.zeros((W,) + v.shape[1:]) for k, v in r.items()}
                    for k in ref:
                        ref[k][w] = r[k][0]
            check_pre(out, ref, what=want, v2=(model == 2), label=label)
        except AssertionError as ex:
            fails += 1
            print("FAIL", label, ex, flush=True)
    # factor sweeps
    for case in range(max(4, cases // 10)):
        F = int(rng.integers(1, 3000))
        model = int(rng.integers(1, 3))
        kn, lin, q = synth.make_windows(F, 20, seed=seed * 1000 + case, device=eng.device)
        meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac", "cov"))
        xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device,
                                   seed=seed * 31 + case)
        states = torch.cat([xi, xj], 0).contiguous()
        ii = torch.arange(F, dtype=torch.int32, device=eng.device)
        qq = q if model == 2 else None
        dense = eng.factor_eval(model, meas, lin, qq, states, ii, ii + F)
        packed = eng.factor_eval_packed(model, meas, lin, qq, states, ii, ii + F)
        torch.cuda.synchronize()
        m = {k: v.cpu().numpy() for k, v in meas.items()}
        rec = op.factor_records(m, lin.cpu().numpy(), q.cpu().numpy() if model == 2 else None)
        st = states.cpu().numpy()
        err, H1, H2 = op.oracle().factor(model, rec, st[:F], st[F:])
        ok = all(np.abs(g.cpu().numpy() - w).max() <= 1e-9 * max(1.0, np.abs(w).max())
                 for g, w in ((dense["err"], err), (dense["H1"], H1), (dense["H2"], H2)))
        e2, H12, H22 = cpi_amd.unpack_factor(packed, meas)
        ok = ok and torch.equal(e2, dense["err"]) and (H12 - dense["H1"]).abs().max().item() == 0.0 \
            and (H22 - dense["H2"]).abs().max().item() == 0.0
        if not ok:
            fails += 1
            print("FAIL factor case", case, F, model, flush=True)
    # Hessian blocks of random sweeps against the numpy definition on the whitened sweep (round 3: 3x3 block algebra)
    for case in range(max(4, cases // 20)):
        F = int(rng.integers(1, 2000))
        model = int(rng.integers(1, 3))
        kn, lin, q = synth.make_windows(F, 15, seed=seed * 2000 + case, device=eng.device, edge_cases=False)
        meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac", "cov"))
        R = eng.sqrt_information(meas["P"])
        xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, model, device=eng.device, seed=seed * 17 + case)
        states = torch.cat([xi, xj], 0).contiguous()
        ii = torch.arange(F, dtype=torch.int32, device=eng.device)
        qq = q if model == 2 else None
        white = eng.factor_eval(model, meas, lin, qq, states, ii, ii + F, sqrt_info=R)
        hess = eng.factor_hessian(model, meas, lin, qq, states, R, ii, ii + F)
        torch.cuda.synchronize()
        A1 = white["H1"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1); A2 = white["H2"].cpu().numpy().reshape(F, 15, 15).transpose(0, 2, 1)
        Ab = np.concatenate([A1, A2, -white["err"].cpu().numpy()[:, :, None]], axis=2)
        M = np.einsum("fki,fkj->fij", Ab, Ab)
        want_h = np.stack([M[:, i, d] for d in range(31) for i in range(d + 1)], axis=1)
        if not (np.abs(hess.cpu().numpy() - want_h) / np.abs(want_h).max(axis=1, keepdims=True)).max() < 1e-12:
            fails += 1
            print("FAIL hessian case", case, F, model, flush=True)
    # ABI 3 (round 6): the packed triangles are the dense forms' own entries -- P_sym out of random preintegrations (every model, averaging,
    # ragged counts), R_tri and the whitened / Hessian sweeps on it against the dense pipeline, BIT FOR BIT; and the state prediction
    # (rebuilt in round 6: one coalesced burst per wavefront) against the restatement with random gathered indices
    npk = 0
    for case in range(max(6, cases // 10)):
        F = int(rng.integers(1, 2500))
        N = int(rng.integers(1, 60))
        model = int(rng.integers(1, 4))
        avg = int(rng.integers(0, 2)) if model < 3 else 0
        kn, lin, q = synth.make_windows(F, N, seed=seed * 3000 + case, device=eng.device, edge_cases=bool(rng.integers(0, 2)))
        cnt = torch.from_numpy(rng.integers(0, N + 1, F).astype(np.int32)).to(eng.device) if rng.integers(0, 2) else None
        prm = eng.make_params(model, avg, 1)
        qq = q if model != 3 else None
        both = eng.preintegrate(kn, lin, qq, prm, want=("mean", "jac", "cov", "cov_sym"), count=cnt)
        only = eng.preintegrate(kn, lin, qq, prm, want=("cov_sym",), count=cnt)
        torch.cuda.synchronize()
        ok = torch.equal(both["P_sym"], cpi_amd.pack_sym(both["P"])) and torch.equal(only["P_sym"], both["P_sym"])
        if model != 3:
            ok = ok and torch.equal(cpi_amd.unpack_sym(both["P_sym"]), both["P"])
        if ok and model != 3 and cnt is None and N >= 8:
            R, Rt = eng.sqrt_information(both["P"]), eng.sqrt_information(both["P_sym"])
            xi, xj = synth.make_states(both["alpha"], both["beta"], both["q"], both["DT"], lin, model, device=eng.device, seed=seed * 13 + case)
            S = int(rng.integers(1, F + 2))
            states = torch.cat([xi, xj], 0)[:max(S, 2)].contiguous()
            ii = torch.from_numpy(rng.integers(-2, states.shape[0] + 2, F).astype(np.int32)).to(eng.device)
            jj = torch.from_numpy(rng.integers(-2, states.shape[0] + 2, F).astype(np.int32)).to(eng.device)
            m = {k: v for k, v in both.items() if k not in ("P", "P_sym")}
            q2 = q if model == 2 else None
            wd, wt = eng.factor_eval(model, m, lin, q2, states, ii, jj, sqrt_info=R), eng.factor_eval(model, m, lin, q2, states, ii, jj, sqrt_info=Rt)
            hd, ht = eng.factor_hessian(model, m, lin, q2, states, R, ii, jj), eng.factor_hessian(model, m, lin, q2, states, Rt, ii, jj)
            xp = eng.predict(model, m, states, idx_i=ii)
            torch.cuda.synchronize()
            ok = ok and torch.equal(cpi_amd.unpack_tri(Rt), R) and all(torch.equal(wd[k], wt[k]) for k in wd) and torch.equal(hd, ht)
            n = min(F, 400)
            sel = ii[:n].clamp(0, states.shape[0] - 1).long()
            rec = op.factor_records({k: v[:n].cpu().numpy() for k, v in m.items()}, lin[:n].cpu().numpy(), q[:n].cpu().numpy() if model == 2 else None)
            want_x = op.oracle().predict(model, rec, states[sel].cpu().numpy())
            ok = ok and np.abs(xp[:n].cpu().numpy() - want_x).max() <= 1e-9 * max(1.0, np.abs(want_x).max())
        npk += 1
        if not ok:
            fails += 1
            print("FAIL packed-forms case", case, F, N, model, avg, flush=True)
    print("campaign seed %d: %d packed-triangle / prediction cases" % (seed, npk), flush=True)
    # the zero-copy stream entry: random streams (irregular sampling, repeated stamps) cut at random update times -- against
    # the oracle's deque-loop restatement and BIT FOR BIT against the host-assembled ragged layout, random lane splits / outputs
    from cpi_amd import stream as st
    for case in range(max(6, cases // 10)):
        K = int(rng.integers(2, 4000))
        U = int(rng.integers(1, 300))
        model = int(rng.integers(1, 4))
        avg = int(rng.integers(0, 2)) if model < 3 else 0
        t = 100.0 + np.cumsum(rng.choice([0.0, 0.0025, 0.005, 0.005, 0.005, 0.01, 0.025], K))
        sm = np.concatenate([t[:, None], rng.normal(size=(K, 3)), rng.normal(size=(K, 3)) * 3 + np.array([0, 0, 9.8])], axis=1)
        ut = np.sort(rng.choice(np.concatenate([t, t + 0.0013, [t[0] - 1.0, t[-1] + 0.3]]), U))
        knots, first, count = st.assemble_windows(sm, ut)
        N = int(count.max())
        _, lin, q = synth.make_windows(U, 2, seed=seed * 3000 + case)
        lanes = int(rng.choice(lanes_all))
        want = wants[int(rng.integers(0, len(wants)))]
        prm = eng.make_params(model, avg, 1, lanes_per_window=lanes)
        qq = dev(q.numpy(), eng) if model != 3 else None
        try:
            out = eng.preintegrate_stream(dev(sm, eng), dev(ut, eng), dev(lin.numpy(), eng), qq, prm, want=want, N=max(N, 1))
            csr = eng.preintegrate(dev(knots, eng), dev(lin.numpy(), eng), qq, prm, want=want, first=dev(first, eng), count=dev(count, eng), N=max(N, 1))
            torch.cuda.synchronize()
            for k in csr:
                assert torch.equal(out[k], csr[k]), "stream entry differs from the ragged layout in " + k
            if model < 3:
                ref = op.oracle().stream(op.make_params(model, avg, 1), sm, ut, lin.numpy(), q.numpy())
                check_pre({k: v.cpu().numpy() for k, v in out.items()}, ref, what=want, v2=(model == 2), label="stream case %d" % case)
        except AssertionError as ex:
            fails += 1
            print("FAIL stream case", case, K, U, model, lanes, want, ex, flush=True)
    print("campaign seed %d: %d preintegration cases, %d failures" % (seed, cases, fails), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
