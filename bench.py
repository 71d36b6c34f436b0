#!/usr/bin/env python
"""bench.py -- preintegration windows/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-extra] [--no-cpu]

A "step" is one pass of the hot path (one cpi_preintegrate_batch call = one kernel launch) over one
batch of synthetic windows already resident in HBM.  The default workload is BASELINE.json
configs[1]: 10 000 windows x 50 samples, CPI model 1, mean-only.  Successive steps walk a pool of
distinct batches larger than the 256 MiB Infinity Cache, so the inputs really stream from HBM.
Prints ONE JSON line (rank 0).  For N>1 launch with torch.distributed.run (one rank per GPU, RCCL):
every rank processes its own pool (weak scaling) with no data-path collective; the output slabs of
the last step are all-gathered once at the end, inside the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
FP64_PEAK_TFLOPS = 78.6         # vector FP64 (datasheet); the covariance kernels are VALU-bound
# Algorithmic HBM bytes per unit: read + write, f64, compulsory traffic only
BYTES = {"v1_mean": 2856 + 88, "v2_mean": 2888 + 88, "v1_full": 2856 + 2320, "v2_full": 2888 + 2392,
         "forster_full": 2856 + 2320,   # Forster / GTSAM comparator (CPI_MODEL_FORSTER): same I/O as model 1 full
         "factor_v1": 776 + 3720, "factor_v2": 952 + 3720,
         # packed evaluateError (state-dependent blocks only, include/cpi_amd.h): NOT the dense GTSAM-shaped output
         "factor_v1_packed": 776 + 576, "factor_v2_packed": 952 + 576}
# sparse-minimal FP64 flop per 50-sample window (0.35-0.5 M and 0.65-0.8 M; midpoints) -- an estimate
FLOP_EST = {"v1_full": 0.425e6, "v2_full": 0.725e6}


def bytes_per_unit(workload, samples=50):
    """a window reads samples*56 + 8 + 48 (+32 for q_k_lin) bytes; the table above is that figure at
    50 samples.  Factor workloads do not depend on the window length."""
    if workload.startswith("factor"):
        return BYTES[workload]
    return BYTES[workload] + (samples - 50) * 56


MALL_BYTES = 256 << 20
# HBM traffic per launch measured with separate rocprofv3 --pmc passes of the same workloads
# (profiles/r01_pmc_counters.md) and corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE / WRITE_SIZE are
# KiB and gfx950's FETCH_SIZE reports half of a streaming read -> bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
# bench.py cannot collect PMC counters itself (that needs the rocprofv3 wrapper), so the figure is only reported for
# the exact (workload, size) it was measured on; any other configuration reports null.
PMC_TRAFFIC_KIB = {("v1_mean", 10000, 50): (15620.2, 906.25), ("v1_mean", 1000000, 50): (1803690.0, 85947.2),
                   ("v1_full", 100000, 50): (144556.0 + 147757.0, 184375.0 + 35156.6),   # covariance + Jacobian kernels
                   ("v2_full", 100000, 50): (149170.0, 242188.0),
                   ("forster_full", 100000, 50): (144690.0, 219531.0),
                   ("factor_v1", 1000000, 50): (351659.0, 3632840.0), ("factor_v2", 1000000, 50): (445426.0, 3632830.0)}


def pmc_traffic(workload, W, N):
    t = PMC_TRAFFIC_KIB.get((workload, W, N))
    return None if t is None else (2.0 * t[0] + t[1]) * 1024.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default="v1_mean", choices=sorted(BYTES))
    ap.add_argument("--windows", type=int, default=0, help="windows (factors) per step; 0 = BASELINE config size")
    ap.add_argument("--samples", type=int, default=50)
    ap.add_argument("--lanes", type=int, default=0, help="mean kernel lanes per window (0 = auto)")
    ap.add_argument("--no-extra", action="store_true", help="skip the additional BASELINE configs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    return ap.parse_args()


def default_size(workload):
    return {"v1_mean": 10000, "v2_mean": 10000, "v1_full": 100000, "v2_full": 100000, "forster_full": 100000,
            "factor_v1": 1000000, "factor_v2": 1000000,
            "factor_v1_packed": 1000000, "factor_v2_packed": 1000000}[workload]


class Workload:
    """A pool of resident batches + preallocated outputs + a step() closure."""

    def __init__(self, eng, name, W, N, seed, lanes=0, pool_bytes=MALL_BYTES * 5 // 4):
        from cpi_amd import synth
        self.name, self.W, self.N = name, W, N
        dev = eng.device
        self.eng = eng
        if name.startswith("factor"):
            model = 1 if "v1" in name else 2
            self.packed = name.endswith("_packed")
            self.model = model
            kn, lin, q = synth.make_windows(W, N, seed=seed, device=dev)
            self.meas = eng.preintegrate(kn, lin, q, eng.make_params(model), want=("mean", "jac"))
            torch.cuda.synchronize()
            del kn
            xi, xj = synth.make_states(self.meas["alpha"], self.meas["beta"], self.meas["q"], self.meas["DT"], lin,
                                       model, device=dev)
            self.states = torch.cat([xi, xj[-1:]], dim=0).contiguous()   # chained states: idx_i=f, idx_j=f+1
            self.lin, self.q = lin, (q if model == 2 else None)
            if self.packed:
                self.out = torch.empty((W, 72), dtype=torch.float64, device=dev)
            else:
                self.out = {"err": torch.empty((W, 15), dtype=torch.float64, device=dev),
                            "H1": torch.empty((W, 225), dtype=torch.float64, device=dev),
                            "H2": torch.empty((W, 225), dtype=torch.float64, device=dev)}
            self.nbatch = 1   # 4.5 GB per sweep: far beyond the Infinity Cache by itself
            return
        model = 2 if name.startswith("v2") else (3 if name.startswith("forster") else 1)
        self.model = model
        want = ("mean",) if name.endswith("mean") else ("mean", "jac", "cov")
        self.want = want
        self.prm = eng.make_params(model, lanes_per_window=lanes)
        batch_bytes = W * (N + 1) * 56
        self.nbatch = max(1, min(64, -(-pool_bytes // batch_bytes)))
        self.batches = [synth.make_windows(W, N, seed=seed + 101 * b, device=dev) for b in range(self.nbatch)]
        # one flat buffer per output set: a rank's outputs are one contiguous slab, so the multi-GPU gather is ONE collective
        self.outs = [eng.alloc_outputs(W, want, model, packed=True) for _ in range(min(self.nbatch, 4))]
        self.i = 0

    def step(self):
        if self.name.startswith("factor"):
            if self.packed:
                self.eng.factor_eval_packed(self.model, self.meas, self.lin, self.q, self.states, out=self.out)
                return {"packed": self.out}
            self.eng.factor_eval(self.model, self.meas, self.lin, self.q, self.states, out=self.out)
            return self.out
        kn, lin, q = self.batches[self.i % self.nbatch]
        out = self.outs[self.i % len(self.outs)]
        self.i += 1
        self.eng.preintegrate(kn, lin, q, self.prm, want=self.want, out=out)
        return out


def final_gather(out, W_local):
    """All ranks' outputs of the last step on every rank: one all_gather of the packed per-rank slabs when the outputs
    are views of one flat buffer (preintegration workloads), else one all_gather per field."""
    import torch.distributed as dist
    from cpi_amd.dist import gather_outputs, gather_packed
    if "_flat" in out:
        return gather_packed(out["_flat"], out["_fields"], W_local)
    return gather_outputs({k: v for k, v in out.items() if not k.startswith("_")}, W_local * dist.get_world_size())


def time_steps(wl, steps, warmup, dist_on=False):
    import torch.distributed as dist
    out = None
    for _ in range(warmup):
        out = wl.step()
    torch.cuda.synchronize()
    if dist_on:
        if out is not None:
            final_gather(out, wl.W)              # untimed: first use of the collective (RCCL channel set-up, buffers)
            torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    out = None
    for _ in range(steps):
        out = wl.step()
    e1.record()                                  # HIP events on the launch stream: kernel time only
    gathered = None
    if dist_on:                                  # the one exchange step: final gather of the output slabs
        gathered = final_gather(out, wl.W)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1)
    del gathered
    return wall, kern_ms


def usable_cpus():
    """Threads worth starting: the affinity mask capped by the cgroup CPU quota.  (The GPU boxes of this pool show 256
    logical CPUs but run the container under cpu.max = 16 CPUs; with 256 threads the same code is 2x SLOWER than with
    16 -- measured with tests/tools/cpu_scale.py: 42 k windows/s at 16 threads, 20 k at 256.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                quota, period = int(f.read()), int(g.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n


def cpu_baseline(wl, min_seconds=8.0):
    """The reference's own CpiV1/CpiV2 (oracle/_ref, kind 'reference') or the C restatement (kind
    'port') timed on this box's host cores on a bounded sample of the SAME windows."""
    from oracle import oracle_py as op
    ref = op.reference()
    lib, kind = (ref, "reference") if ref is not None else (op.oracle(), "port")
    if wl.model == 3:   # the Forster comparator lives in GTSAM (absent): only the restatement exists
        lib, kind = op.oracle(), "port"
    cores = usable_cpus()
    kn, lin, q = [t.cpu().numpy() for t in wl.batches[0]]
    Wc = min(wl.W, 10000)
    kn, lin, q = kn[:Wc], lin[:Wc], q[:Wc]
    prm = op.make_params(wl.model, 0, 1)
    import numpy as np
    from oracle.oracle_py import OUT_DOUBLES
    raw = np.ones((Wc, OUT_DOUBLES))                               # reused, already touched output buffer
    lib.run(prm, kn[:256], lin[:256], q[:256], nthreads=cores)     # warm
    t0, done = time.perf_counter(), 0
    while True:
        lib.run(prm, kn, lin, q, nthreads=cores, raw=raw)
        done += Wc
        el = time.perf_counter() - t0
        if el >= min_seconds:
            break
    # single-thread figure on a smaller slice
    ws = min(Wc, 1500 if wl.model == 1 else 600)
    t1 = time.perf_counter(); lib.run(prm, kn[:ws], lin[:ws], q[:ws], nthreads=1); t1 = time.perf_counter() - t1
    return {"value": done / el, "unit": "windows/s", "cores": cores, "kind": kind,
            "single_core_value": ws / t1,
            "sample": "%d passes over %d of the workload's %d-sample windows, %d threads (= usable CPUs: affinity mask "
                      "capped by the cgroup quota; %d logical CPUs visible); the reference feed_IMU always integrates "
                      "means + bias Jacobians + covariance (it has no mean-only mode)"
                      % (done // Wc, Wc, wl.N, cores, os.cpu_count() or 1)}


def overlapped_rate(W, N, nctx, steps):
    """Whole-job rate when independent batches are issued round-robin through nctx engine contexts (one HIP stream
    each, cpi_amd.EnginePool), so that consecutive launches overlap.  Reported as an `extra` row only: with overlapping
    launches the duration of one launch is no longer the inverse of the throughput, which is what the roofline line
    is defined on."""
    import cpi_amd
    from cpi_amd import synth
    dev = torch.device("cuda", torch.cuda.current_device())
    pool = cpi_amd.EnginePool(nctx, device=dev)
    nb = max(nctx, -(-(MALL_BYTES * 5 // 4) // (W * (N + 1) * 56)))
    batches = [synth.make_windows(W, N, seed=977 + b, device=dev) for b in range(nb)]
    outs = [pool.engines[0].alloc_outputs(W, ("mean",), 1) for _ in range(2 * nctx)]
    prm = pool.engines[0].make_params(1)

    def go(k):
        for i in range(k):
            kn, lin, q = batches[i % nb]
            pool.engines[i % nctx].preintegrate(kn, lin, q, prm, want=("mean",), out=outs[i % len(outs)])
    torch.cuda.synchronize()
    go(max(50, steps // 10))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    pool.close()
    return wall / steps


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or bool(os.environ.get("CPI_BENCH_FORCE_DIST"))  # env: exercise the RCCL path with one rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import cpi_amd
    eng = cpi_amd.Engine(device=local_rank)
    W = a.windows or default_size(a.workload)
    wl = Workload(eng, a.workload, W, a.samples, seed=20190101 + 7919 * rank, lanes=a.lanes)
    # Untimed clock pre-ramp, separate from the W warm-up steps: an idle MI355X needs tens of milliseconds of load to reach
    # its steady shader clock, so a short (K, W) would otherwise time the ramp (K = 20: 14.9 us per launch instead of
    # 12.0).  Reported in config.clock_preramp_ms; the W warm-up steps and the K timed steps follow unchanged.
    PRERAMP_MS = 60.0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < PRERAMP_MS:
        for _ in range(50):
            wl.step()
        torch.cuda.synchronize()
    wall, kern_ms = time_steps(wl, a.steps, a.warmup, dist_on)
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([wall, kern_ms], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = t[0].item(), t[1].item()
    units = W * world * a.steps
    value = units / wall
    launch_s = kern_ms * 1e-3 / a.steps
    bpu = bytes_per_unit(a.workload, a.samples)
    achieved = bpu * W / launch_s / 1e9
    is_factor = a.workload.startswith("factor")
    res = {
        "metric": "evaluateError factors/sec" if is_factor else "preintegration windows/sec (50-sample windows)",
        "value": value, "unit": "factors/s" if is_factor else "windows/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall * 1e3 / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d %s x %d samples per GPU per step%s" % (
            a.workload, W, "factors" if is_factor else "windows", a.samples,
            ", CPI model 1, mean-only (BASELINE.json configs[1])" if a.workload == "v1_mean" and W == 10000 else ""),
            "pool_batches": wl.nbatch, "clock_preramp_ms": PRERAMP_MS, "parallelism": "windows sharded over %d GPU(s), final all_gather" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(a.workload, W, a.samples),
                     "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r01_pmc_counters.md)",
                     "algorithmic_bytes_per_launch": bpu * W,
                     "kernel": {"v1_mean": "cpi_mean_kernel", "v2_mean": "cpi_mean_kernel", "v1_full": "cpi_cov_kernel<1>", "v2_full": "cpi_cov_kernel<2>",
                                "forster_full": "cpi_forster_kernel", "factor_v1": "cpi_factor_kernel<1,false,8>", "factor_v2": "cpi_factor_kernel<2,false,8>",
                                "factor_v1_packed": "cpi_factor_packed_kernel<1>",
                                "factor_v2_packed": "cpi_factor_packed_kernel<2>"}[a.workload],
                     "launch_us": launch_s * 1e6, "algorithmic_bytes_per_unit": bpu},
    }
    if rank == 0 and world == 1 and not a.no_cpu and not is_factor:
        res["cpu_baseline"] = cpu_baseline(wl)
    if rank == 0 and world == 1 and not a.no_extra:
        extra = []
        del wl
        torch.cuda.empty_cache()
        for name, Wx, steps in (("v1_mean", 30000, 1000), ("v1_mean", 100000, 300), ("v1_mean", 1000000, 40),
                                ("v1_full", 100000, 30), ("v2_full", 100000, 30), ("forster_full", 100000, 30),
                                ("factor_v1", 1000000, 40), ("factor_v2", 1000000, 40),
                                ("factor_v1_packed", 1000000, 40), ("factor_v2_packed", 1000000, 40)):
            try:
                w2 = Workload(eng, name, Wx, a.samples, seed=4242, pool_bytes=MALL_BYTES * 5 // 4)
                wall2, k2 = time_steps(w2, steps, max(10, steps // 10))
                ls = k2 * 1e-3 / steps
                ach = bytes_per_unit(name, a.samples) * Wx / ls / 1e9
                row = {"workload": name, "units_per_step": Wx, "value": Wx * steps / wall2,
                       "unit": "factors/s" if name.startswith("factor") else "windows/s",
                       "launch_ms": ls * 1e3, "hbm_GBs": ach, "hbm_frac": ach / HBM_PEAK_GBS}
                if name in FLOP_EST:   # the covariance workloads are FP64-bound: estimate from SURVEY.md 8(d)'s
                    row["fp64_TFLOPs_est"] = FLOP_EST[name] * Wx / ls / 1e12   # sparse-minimal flop counts (midpoints)
                    row["fp64_frac_est"] = row["fp64_TFLOPs_est"] / FP64_PEAK_TFLOPS
                extra.append(row)
                del w2
                torch.cuda.empty_cache()
            except Exception as ex:  # an extra config must never take the headline down
                extra.append({"workload": name, "error": repr(ex)})
        try:   # the headline workload again, issued through 3 contexts so that consecutive launches overlap
            per = overlapped_rate(10000, a.samples, 3, 3000)
            ach = bytes_per_unit("v1_mean", a.samples) * 10000 / per / 1e9
            extra.append({"workload": "v1_mean", "units_per_step": 10000, "value": 10000 / per, "unit": "windows/s",
                          "contexts": 3, "us_per_batch": per * 1e6, "hbm_GBs": ach, "hbm_frac": ach / HBM_PEAK_GBS,
                          "note": "independent batches round-robin over 3 engine contexts (3 HIP streams): launches "
                                  "overlap, so this is an aggregate rate, not a per-launch duration"})
        except Exception as ex:
            extra.append({"workload": "v1_mean (3 contexts)", "error": repr(ex)})
        res["extra"] = extra
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its banner through C stdio: flush that first so the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
