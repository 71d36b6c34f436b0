#!/usr/bin/env python
"""bench.py -- preintegration windows/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--scaling weak|strong] [--no-extra] [--no-cpu]

A "step" is one pass of the hot path (one cpi_preintegrate_batch / cpi_factor_eval_batch call) over one batch of
synthetic windows already resident in HBM.  The default workload is BASELINE.json configs[1]: 10 000 windows x 50
samples, CPI model 1, mean-only.  Successive steps walk a pool of distinct batches larger than the 256 MiB Infinity
Cache, so the inputs really stream from HBM.  Prints ONE JSON line (rank 0).

N > 1: one rank per GPU over RCCL.  Launched either by the driver (`python -m torch.distributed.run ... bench.py --gpus N`,
WORLD_SIZE set) or by itself: with WORLD_SIZE unset `python bench.py --gpus N` re-executes under torch.distributed.run.
Windows shard with no data-path collective ("weak": every rank runs the per-GPU workload; "strong": the workload's
windows are split N ways); the one exchange step is the final gather of the last step's output slabs TO RANK 0
(cpi_amd.dist.gather_to_root: each peer sends straight to the root over its own xGMI link), inside the timed region.
The rate without the gather is reported beside it.  `--workload cfg5_mean | cfg5_full` is BASELINE configs[4]:
1 M windows x 100 samples per GPU, generated on the device.

Every measured row (the headline and each `extra` row) carries its own `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
FP64_PEAK_TFLOPS = 78.6         # vector FP64 (datasheet); the covariance kernels are VALU / LDS-bound
MALL_BYTES = 256 << 20
PMC_FILE = os.path.join(ROOT, "profiles", "r02_pmc.json")

# name -> model, outputs, default units per step, samples, algorithmic HBM bytes per unit at 50 samples (SURVEY.md 8(d):
# read + write, f64, compulsory traffic only), dominant kernel
WORKLOADS = {
    "v1_mean": dict(model=1, want=("mean",), W=10000, N=50, bytes=2856 + 88, kernel="cpi_mean_kernel<1,false,false,L>"),
    "v2_mean": dict(model=2, want=("mean",), W=10000, N=50, bytes=2888 + 88, kernel="cpi_mean_kernel<2,false,false,L>"),
    "v1_full": dict(model=1, want=("mean", "jac", "cov"), W=100000, N=50, bytes=2856 + 2320, kernel="cpi_cov_kernel<1,false>"),
    "v2_full": dict(model=2, want=("mean", "jac", "cov"), W=100000, N=50, bytes=2888 + 2392, kernel="cpi_cov_kernel<2,false>"),
    # Forster / GTSAM comparator (CPI_MODEL_FORSTER): same I/O as model 1 full
    "forster_full": dict(model=3, want=("mean", "jac", "cov"), W=100000, N=50, bytes=2856 + 2320, kernel="cpi_forster_kernel"),
    "factor_v1": dict(model=1, factor=True, W=1000000, N=50, bytes=776 + 3720, kernel="cpi_factor_kernel<1,false,8>"),
    "factor_v2": dict(model=2, factor=True, W=1000000, N=50, bytes=952 + 3720, kernel="cpi_factor_kernel<2,false,8>"),
    # packed evaluateError (state-dependent blocks only, include/cpi_amd.h): NOT the dense GTSAM-shaped output
    "factor_v1_packed": dict(model=1, factor=True, packed=True, W=1000000, N=50, bytes=776 + 576, kernel="cpi_factor_packed_kernel<1,L>"),
    "factor_v2_packed": dict(model=2, factor=True, packed=True, W=1000000, N=50, bytes=952 + 576, kernel="cpi_factor_packed_kernel<2,L>"),
    # the same mean-only recursion on the TILED input layout (knots of 64 windows interleaved per step; include/cpi_amd.h)
    "v1_mean_tiled": dict(model=1, want=("mean",), W=1000000, N=50, bytes=2856 + 88, tiled=True, kernel="cpi_mean_tiled_kernel<1,false,false,SPLIT>"),
    "v2_mean_tiled": dict(model=2, want=("mean",), W=1000000, N=50, bytes=2888 + 88, tiled=True, kernel="cpi_mean_tiled_kernel<2,false,false,SPLIT>"),
    # BASELINE configs[4]: one GPU's share of 8 M windows x 100 samples (EuRoC-rate synthetic IMU), generated on the device
    "cfg5_mean": dict(model=1, want=("mean",), W=1000000, N=100, bytes=2856 + 88, kernel="cpi_mean_kernel<1,false,false,1>"),
    "cfg5_full": dict(model=1, want=("mean", "jac", "cov"), W=1000000, N=100, bytes=2856 + 2320, kernel="cpi_cov_kernel<1,false>"),
}
# sparse-minimal FP64 flop per 50-sample window (0.35-0.5 M and 0.65-0.8 M; midpoints) -- an ESTIMATE, used
# only when no counter-derived figure is available for the loaded library
FLOP_EST = {"v1_full": 0.425e6, "v2_full": 0.725e6}


def bytes_per_unit(workload, samples=50):
    """a window reads samples*56 + 8 + 48 (+32 for q_k_lin) bytes; WORKLOADS holds that figure at 50
    samples.  Factor workloads do not depend on the window length."""
    w = WORKLOADS[workload]
    return w["bytes"] if w.get("factor") else w["bytes"] + (samples - 50) * 56


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default="v1_mean", choices=sorted(WORKLOADS))
    ap.add_argument("--windows", type=int, default=0, help="windows (factors) per step and GPU; 0 = the workload's size")
    ap.add_argument("--samples", type=int, default=0, help="samples per window; 0 = the workload's (50; cfg5: 100)")
    ap.add_argument("--lanes", type=int, default=0, help="mean kernel lanes per window (0 = auto)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1: weak = every rank runs the per-GPU workload; strong = the workload's windows are split N ways")
    ap.add_argument("--gather", default="root", choices=("root", "all", "none"), help="N > 1: the final exchange step")
    ap.add_argument("--no-extra", action="store_true", help="skip the additional BASELINE configs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ workloads
class Workload:
    """A pool of resident batches + preallocated outputs + a step() closure."""

    def __init__(self, eng, name, W, N, seed, lanes=0, pool_bytes=MALL_BYTES * 5 // 4):
        from cpi_amd import synth
        spec = WORKLOADS[name]
        self.name, self.W, self.N, self.spec = name, W, N, spec
        self.model = spec["model"]
        self.is_factor = bool(spec.get("factor"))
        dev = eng.device
        self.eng = eng
        if self.is_factor:
            model = self.model
            self.packed = bool(spec.get("packed"))
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
        self.want = spec["want"]
        self.prm = eng.make_params(self.model, lanes_per_window=lanes)
        batch_bytes = W * (N + 1) * 56
        self.nbatch = max(1, min(64, -(-pool_bytes // batch_bytes)))
        self.batches = [synth.make_windows(W, N, seed=seed + 101 * b, device=dev) for b in range(self.nbatch)]
        # one flat buffer per output set: a rank's outputs are one contiguous slab, so the multi-GPU gather is ONE collective
        out_bytes = W * sum(n for _, n in eng.alloc_outputs(1, self.want, self.model, packed=True)["_fields"]) * 8
        self.outs = [eng.alloc_outputs(W, self.want, self.model, packed=True)
                     for _ in range(1 if out_bytes > (1 << 30) else min(self.nbatch, 4))]
        self.i = 0
        self.tiled = bool(spec.get("tiled"))
        import math
        if self.tiled:   # the batches converted once, untimed: the timed step reads tiles only
            self.tiles = [eng.tile_knots(b[0]) for b in self.batches]
            torch.cuda.synchronize()
            for bi in range(self.nbatch):
                self.batches[bi] = (self.batches[bi][0][:64].clone(), self.batches[bi][1], self.batches[bi][2])   # the dense copy is dropped (lin / q stay)
        # every (batch, output set) pair of the walk pre-bound: a step is one foreign call (Engine.bind_preintegrate)
        period = self.nbatch * len(self.outs) // math.gcd(self.nbatch, len(self.outs))
        self.calls = []
        for i in range(period):
            kn, lin, q = self.batches[i % self.nbatch]
            if self.tiled:
                call, _ = eng.preintegrate_tiled(self.tiles[i % self.nbatch], W, lin, q, self.prm, out=self.outs[i % len(self.outs)], bind=True)
            else:
                call, _ = eng.bind_preintegrate(kn, lin, q if self.model != 3 else None, self.prm, want=self.want,
                                                out=self.outs[i % len(self.outs)])
            self.calls.append(call)

    def step(self):
        if self.is_factor:
            if self.packed:
                self.eng.factor_eval_packed(self.model, self.meas, self.lin, self.q, self.states, out=self.out)
                return {"packed": self.out}
            self.eng.factor_eval(self.model, self.meas, self.lin, self.q, self.states, out=self.out)
            return self.out
        out = self.outs[self.i % len(self.outs)]
        self.calls[self.i % len(self.calls)]()
        self.i += 1
        return out


# ------------------------------------------------------------------------------------------------ multi-GPU exchange
def final_gather(out, W_local, mode="root", dst=0, recv=None):
    """The path's one exchange step: the per-rank output slabs of the last step.  mode "root": to rank `dst` only
    (every peer sends straight to the root); "all": all-gather.  Returns the gathered blocks
    (name -> [world, W_local, n]) on the ranks that hold them, else None.  Works on any backend (gloo in the CPU tests)."""
    from cpi_amd.dist import gather_packed, gather_to_root
    if mode == "none":
        return None
    assert "_flat" in out, "the final gather moves ONE packed slab per rank (Engine.alloc_outputs(packed=True))"
    if mode == "all":
        return gather_packed(out["_flat"], out["_fields"], W_local)
    return gather_to_root(out["_flat"], out["_fields"], W_local, dst=dst, out=recv)


def time_steps(wl, steps, warmup, dist_on=False, gather="root"):
    """W untimed warm-up steps, then EXACTLY `steps` timed steps bracketed by barrier + synchronize on both sides.
    Returns (wall seconds, kernel milliseconds by HIP events on the launch stream)."""
    import torch.distributed as dist
    out = None
    for _ in range(warmup):
        out = wl.step()
    torch.cuda.synchronize()
    recv = None
    do_gather = dist_on and gather != "none" and not wl.is_factor
    if dist_on:
        if do_gather and out is None:
            out = wl.step()
        if do_gather:
            # untimed: first use of the collective (RCCL channel set-up), and the root's receive buffer
            if gather == "root" and dist.get_rank() == 0:
                recv = torch.empty((dist.get_world_size(), out["_flat"].numel()), dtype=torch.float64, device=out["_flat"].device)
            final_gather(out, wl.W, gather, recv=recv)
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
    gathered = final_gather(out, wl.W, gather, recv=recv) if do_gather else None
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1)
    del gathered
    return wall, kern_ms


# ------------------------------------------------------------------------------------------------ CPU legs
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
    """The CPU path timed beside a row, on this box's host cores, on a bounded sample of the SAME inputs.
    Preintegration rows: the reference's own CpiV1 / CpiV2 (oracle/_ref, kind "reference") -- which always integrates means
    + bias Jacobians + covariance, so it is like-for-like for the *_full rows and does MORE than the GPU for the
    mean-only rows (the reference has no mean-only mode; said in `sample`).  Forster comparator and the evaluateError
    sweep: the C restatement (kind "port"; GTSAM / the factor TUs cannot be built here)."""
    import numpy as np
    from oracle import oracle_py as op
    from oracle.oracle_py import OUT_DOUBLES
    cores = usable_cpus()
    if wl.is_factor:
        F = min(wl.W, 20000)
        meas = {k: v[:F].cpu().numpy() for k, v in wl.meas.items()}
        rec = op.factor_records(meas, wl.lin[:F].cpu().numpy(), wl.q[:F].cpu().numpy() if wl.q is not None else None)
        st = wl.states[:F + 1].cpu().numpy()
        xi, xj = np.ascontiguousarray(st[:-1]), np.ascontiguousarray(st[1:])
        buf = (np.ones((F, 15)), np.ones((F, 225)), np.ones((F, 225)))
        orc = op.oracle()
        orc.factor_batch(wl.model, rec[:256], xi[:256], xj[:256], nthreads=cores)
        t0, done = time.perf_counter(), 0
        while True:
            orc.factor_batch(wl.model, rec, xi, xj, nthreads=cores, out=buf)
            done += F
            el = time.perf_counter() - t0
            if el >= min_seconds:
                break
        fs = min(F, 5000)
        t1 = time.perf_counter(); orc.factor_batch(wl.model, rec[:fs], xi[:fs], xj[:fs], nthreads=1); t1 = time.perf_counter() - t1
        return {"value": done / el, "unit": "factors/s", "cores": cores, "kind": "port", "single_core_value": fs / t1,
                "sample": "%d passes over %d of the sweep's factors (residual + dense H1 / H2, the C restatement of "
                          "ImuFactorCPIv%d::evaluateError -- the reference's factor TUs need GTSAM), %d threads" % (done // F, F, wl.model, cores)}
    ref = op.reference()
    lib, kind = (ref, "reference") if ref is not None else (op.oracle(), "port")
    if wl.model == 3:   # the Forster comparator lives in GTSAM (absent): only the restatement exists
        lib, kind = op.oracle(), "port"
    kn, lin, q = [t[:10000].cpu().numpy() for t in wl.batches[0]]
    Wc = kn.shape[0]
    prm = op.make_params(wl.model, 0, 1)
    raw = np.ones((Wc, OUT_DOUBLES))                               # reused, already touched output buffer
    lib.run(prm, kn[:256], lin[:256], q[:256], nthreads=cores)     # warm
    t0, done = time.perf_counter(), 0
    while True:
        lib.run(prm, kn, lin, q, nthreads=cores, raw=raw)
        done += Wc
        el = time.perf_counter() - t0
        if el >= min_seconds:
            break
    port = None
    if wl.model in (1, 2):
        port = sparse_port_rate(wl, kn, lin, q, cores, min(2.0, min_seconds / 3))
    ws = min(Wc, (1500 if wl.model == 1 else 600) * 50 // max(50, wl.N))
    t1 = time.perf_counter(); lib.run(prm, kn[:ws], lin[:ws], q[:ws], nthreads=1); t1 = time.perf_counter() - t1
    what = {1: "CpiV1::feed_IMU", 2: "CpiV2::feed_IMU (state_transition_jacobians = true)", 3: "the Forster comparator restatement"}[wl.model]
    note = "" if "cov" in wl.want else "; the reference has no mean-only mode: this CPU figure includes bias Jacobians and covariance, the GPU row does not"
    res = {"value": done / el, "unit": "windows/s", "cores": cores, "kind": kind, "single_core_value": ws / t1,
           "sample": "%d passes over %d of the row's %d-sample windows through %s, %d threads (= usable CPUs: affinity mask "
                     "capped by the cgroup quota; %d logical CPUs visible)%s"
                     % (done // Wc, Wc, wl.N, what, cores, os.cpu_count() or 1, note)}
    if port:
        res["sparse_port"] = port
    return res


def sparse_port_rate(wl, kn, lin, q, cores, seconds):
    """Second CPU figure beside the dense reference (SURVEY.md 8(d) asks for both): the kernels' OWN sparse arithmetic
    (cpi_amd/csrc/cpi_math.hpp compiled for the host by tests/hostsim -- column-lane covariance recursion, or the mean-only
    recursion for mean-only rows, which is the LIKE-FOR-LIKE CPU figure the reference cannot give: it has no mean-only
    mode), on the same windows, all usable cores and one core.  g++ -O2, no hand vectorisation: a port, not a tuned CPU code."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import hostsim_py as hs
    mean_only = "cov" not in wl.want
    Wc = kn.shape[0]

    def run(lo, hi):
        if mean_only:
            hs.mean(wl.model, 0, 0, 1, kn[lo:hi], lin[lo:hi], q[lo:hi])
        else:
            hs.cov(wl.model, 0, kn[lo:hi], lin[lo:hi], q[lo:hi])
            if wl.model == 1:
                hs.mean(1, 1, 0, 1, kn[lo:hi], lin[lo:hi], q[lo:hi])      # model 1: the analytic Jacobians are a second pass
    hs.lib()
    n1 = min(Wc, 2000 if mean_only else 200)
    t = time.perf_counter(); run(0, n1); t = time.perf_counter() - t
    single = n1 / t
    Wp = int(min(Wc, max(cores * 16, single * cores * seconds)))
    cuts = [Wp * i // cores for i in range(cores + 1)]
    passes = 0
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        while True:
            list(ex.map(lambda i: run(cuts[i], cuts[i + 1]), range(cores)))
            passes += 1
            t = time.perf_counter() - t0
            if t >= seconds:
                break
    Wp *= passes
    return {"value": Wp / t, "unit": "windows/s", "cores": cores, "kind": "port", "single_core_value": single,
            "what": ("mean-only recursion (like for like with the GPU row)" if mean_only else
                     "sparse column-lane covariance recursion" + (" + analytic Jacobians" if wl.model == 1 else " with the state-transition Jacobians")),
            "sample": "%d window evaluations (%d passes) over %d threads; %d windows on one thread" % (Wp, passes, cores, n1)}



# ------------------------------------------------------------------------------------------------ counters
def load_pmc(build_id):
    """profiles/r02_pmc.json: rocprofv3 --pmc passes of tools/pmc_collect.sh, stamped with the build id of the library
    they were collected on.  Used only when that stamp equals the LOADED library's cpi_build_id(); otherwise the
    counter-derived fields are null (the file is stale for this library)."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
    except Exception:
        return {}, "no %s" % os.path.relpath(PMC_FILE, ROOT)
    if d.get("build_id") != build_id:
        return {}, "%s was collected on build %s, the loaded library is %s" % (os.path.relpath(PMC_FILE, ROOT), d.get("build_id"), build_id)
    return d.get("rows", {}), "%s (build %s)" % (os.path.relpath(PMC_FILE, ROOT), build_id)


def roofline_of(name, W, N, launch_s, pmc_rows, pmc_note):
    """The contract's roofline object for one row: achieved = ALGORITHMIC bytes per launch / launch duration."""
    bpu = bytes_per_unit(name, N)
    achieved = bpu * W / launch_s / 1e9
    row = pmc_rows.get("%s:%d:%d" % (name, W, N))
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": row.get("traffic_bytes") if row else None,
         "traffic_unit": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 --pmc passes; " + pmc_note,
         "algorithmic_bytes_per_launch": bpu * W, "algorithmic_bytes_per_unit": bpu,
         "kernel": WORKLOADS[name]["kernel"], "launch_us": launch_s * 1e6}
    if row and row.get("fp64_flop"):
        # FP64 work counted by the SQ instruction counters of the same library: (2 FMA + MUL + ADD + TRANS) x 64 lanes
        tf = row["fp64_flop"] / launch_s / 1e12
        r["fp64"] = {"TFLOPs": tf, "peak": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS, "source": "counters",
                     "flop_per_launch": row["fp64_flop"], "valu_insts_per_launch": row.get("valu_insts"),
                     "fp64_insts_per_launch": row.get("fp64_insts")}
    elif name in FLOP_EST:
        tf = FLOP_EST[name] * W / launch_s / 1e12
        r["fp64"] = {"TFLOPs": tf, "peak": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS,
                     "source": "estimate (SURVEY.md 8(d) sparse-minimal flop midpoints; no counters for this build)"}
    return r


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


# ------------------------------------------------------------------------------------------------ launch
def respawn(gpus):
    """`python bench.py --gpus N` with no launcher: become the launcher (one rank per GPU, RCCL)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execve(sys.executable, cmd, env)


def preramp(wl, ms):
    """Untimed clock pre-ramp, separate from the W warm-up steps: an idle MI355X needs tens of milliseconds of load to reach
    its steady shader clock, so a short (K, W) would otherwise time the ramp (K = 20: 14.9 us per launch instead of 12.0)."""
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        for _ in range(50 if wl.W <= 100000 else 2):
            wl.step()
        torch.cuda.synchronize()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        respawn(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or bool(os.environ.get("CPI_BENCH_FORCE_DIST"))  # env: exercise the RCCL path with one rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    if rank == 0 and world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d\n" % (a.gpus, world, world))
    # CPI_BENCH_SINGLE_DEVICE=1: rehearsal of the N-rank flow on a box with ONE GPU (tests/test_gpu_group.py) -- every rank
    # computes on cuda:0 and the exchange goes through gloo (RCCL refuses two ranks on one device).  Not a measurement.
    rehearsal = bool(os.environ.get("CPI_BENCH_SINGLE_DEVICE"))
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d needs device %d but this node shows %d GPU(s) -- one rank per GPU (--gpus N <= devices); "
                 "CPI_BENCH_SINGLE_DEVICE=1 rehearses the N-rank flow on one device" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import cpi_amd
    from cpi_amd.dist import shard_bounds
    eng = cpi_amd.Engine(device=local_rank)
    build_id = (eng.lib.cpi_build_id() or b"").decode()
    pmc_rows, pmc_note = load_pmc(build_id)
    spec = WORKLOADS[a.workload]
    N = a.samples or spec["N"]
    W_job = a.windows or spec["W"]                   # per GPU (weak) or in total (strong)
    if a.scaling == "strong" and world > 1:
        lo, hi, W = shard_bounds(W_job, rank, world)  # equal padded block per rank; W = block size
        total_units = W_job
    else:
        W = W_job
        total_units = W_job * world
    wl = Workload(eng, a.workload, W, N, seed=20190101 + 7919 * rank, lanes=a.lanes)
    PRERAMP_MS = 60.0
    preramp(wl, PRERAMP_MS)
    wall, kern_ms = time_steps(wl, a.steps, a.warmup, dist_on, a.gather)
    wall_ng = None
    if dist_on and a.gather != "none" and not wl.is_factor:     # the same K steps without the exchange step, beside it
        wall_ng, _ = time_steps(wl, a.steps, min(a.warmup, 5), dist_on, "none")
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([wall, kern_ms, wall_ng or 0.0], dtype=torch.float64, device="cpu" if rehearsal else eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms, wall_ng = t[0].item(), t[1].item(), (t[2].item() if wall_ng is not None else None)
    value = total_units * a.steps / wall
    launch_s = kern_ms * 1e-3 / a.steps
    is_factor = wl.is_factor
    unit = "factors" if is_factor else "windows"
    cfg_note = ""
    if a.workload == "v1_mean" and W_job == 10000 and N == 50:
        cfg_note = ", CPI model 1, mean-only (BASELINE.json configs[1])"
    elif a.workload.startswith("cfg5"):
        cfg_note = ", BASELINE.json configs[4]: 8 M windows x 100 samples over 8 GPUs = this per-GPU share, generated on the device"
    res = {
        "metric": "evaluateError factors/sec" if is_factor else "preintegration windows/sec (%d-sample windows)" % N,
        "value": value, "unit": unit + "/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall * 1e3 / a.steps,
        "higher_is_better": True, "scaling": a.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo exchange -- not a measurement)",
        "config": {"workload": "%s: %d %s x %d samples per GPU per step%s" % (a.workload, W, unit, N, cfg_note),
                   "pool_batches": wl.nbatch, "clock_preramp_ms": PRERAMP_MS, "library_build": build_id,
                   "parallelism": ("1 GPU" if world == 1 else
                                   "%s scaling: %d windows per step on each of %d GPUs, no data-path collective; final gather of the "
                                   "last step's outputs %s, inside the timed region" % (
                                       a.scaling, W, world, {"root": "to rank 0 (each peer sends straight to the root)",
                                                             "all": "to every rank (all_gather)", "none": "skipped"}[a.gather]))},
        "roofline": roofline_of(a.workload, W, N, launch_s, pmc_rows, pmc_note),
    }
    if wall_ng is not None:
        res["config"]["value_without_gather"] = total_units * a.steps / wall_ng
        res["config"]["ms_final_gather"] = max(0.0, (wall - wall_ng) * 1e3)
    if rank == 0 and world == 1 and not a.no_cpu:
        res["cpu_baseline"] = cpu_baseline(wl, 8.0)
    if rank == 0 and world == 1 and not a.no_extra:
        extra = []
        del wl
        torch.cuda.empty_cache()
        for name, Wx, steps in (("v1_mean", 30000, 1000), ("v1_mean", 100000, 300), ("v1_mean", 1000000, 40),
                                ("v1_full", 100000, 30), ("v2_full", 100000, 30), ("forster_full", 100000, 30),
                                ("factor_v1", 1000000, 40), ("factor_v2", 1000000, 40),
                                ("factor_v1_packed", 1000000, 40), ("factor_v2_packed", 1000000, 40),
                                ("cfg5_mean", 1000000, 10), ("cfg5_full", 1000000, 3), ("v1_mean_tiled", 1000000, 40),
                                ("v1_mean_tiled", 10000, 1000)):
            try:
                Nx = WORKLOADS[name]["N"]
                w2 = Workload(eng, name, Wx, Nx, seed=4242)
                wall2, k2 = time_steps(w2, steps, max(2, steps // 10))
                ls = k2 * 1e-3 / steps
                row = {"workload": name, "units_per_step": Wx, "samples": Nx, "value": Wx * steps / wall2,
                       "unit": ("factors" if w2.is_factor else "windows") + "/s", "launch_ms": ls * 1e3,
                       "roofline": roofline_of(name, Wx, Nx, ls, pmc_rows, pmc_note)}
                if not a.no_cpu and not name.endswith("_packed") and not name.endswith("_tiled") and not (name == "v1_mean" and Wx != 1000000):
                    row["cpu_baseline"] = cpu_baseline(w2, 2.5)     # bounded: ~2.5 s of CPU work per row
                extra.append(row)
                del w2
                torch.cuda.empty_cache()
            except Exception as ex:  # an extra config must never take the headline down
                extra.append({"workload": name, "error": repr(ex)})
        try:   # the headline workload again, issued through 3 contexts so that consecutive launches overlap
            per = overlapped_rate(10000, 50, 3, 3000)
            ach = bytes_per_unit("v1_mean", 50) * 10000 / per / 1e9
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
