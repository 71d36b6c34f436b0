"""Host-side mirror of the reference's preintegrator / factor interface over the C-ABI.

    Engine                 one (device, stream) context; batched device-resident entry points
    CpiV1 / CpiV2          same constructor, setLinearizationPoints(), feed_IMU() and public result
                           fields as the reference classes (cpi_compare/src/cpi/CpiBase.h:40-145,
                           CpiV1.h:62, CpiV2.h:84); the recursion runs on the GPU when a result is read
    ImuFactorCPIv1 / v2    constructor argument order of ImuFactorCPIv1.h:78-81 / ImuFactorCPIv2.h:82-85
                           and evaluateError(state_i, state_j) -> (error, H1, H2)

PyTorch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import OUT_FIELDS, CpiError, CpiOutputs, CpiParams

DEFAULT_SIGMAS = (0.005, 4e-6, 0.01, 2e-4)   # ADIS16448, cpi_compare/launch/synthetic_test.launch:13-17
DEFAULT_GRAV = (0.0, 0.0, 9.8)
MEAN_FIELDS = ("DT", "alpha", "beta", "q")
JAC_FIELDS = ("J_q", "J_a", "J_b", "H_a", "H_b", "O_a", "O_b")


def _group_of(name):
    """want-group of an output field: "mean", "jac", "cov" (P, dense 15 x 15) or "cov_sym" (P_sym: its packed upper triangle, 120
    doubles -- include/cpi_amd.h CPI_TRI_INDEX)."""
    if name in MEAN_FIELDS:
        return "mean"
    return "cov" if name == "P" else ("cov_sym" if name == "P_sym" else "jac")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _tri_index(device=None):
    """(rows, cols) of the packed upper triangle in storage order: entry k of a packed matrix is (rows[k], cols[k])."""
    cols = torch.repeat_interleave(torch.arange(15, device=device), torch.arange(1, 16, device=device))
    start = cols * (cols + 1) // 2
    rows = torch.arange(120, device=device) - start
    return rows, cols


def pack_sym(M):
    """Dense [F, 225] (column-major 15 x 15; symmetric or upper triangular) -> packed upper triangle [F, 120]
    (include/cpi_amd.h: entry (i, j), i <= j, at i + j (j + 1) / 2).  pack_tri is the same gather."""
    rows, cols = _tri_index(M.device)
    return M.reshape(-1, 225)[:, cols * 15 + rows].contiguous()


pack_tri = pack_sym


def unpack_sym(Ps):
    """Packed upper triangle [F, 120] of a SYMMETRIC matrix (cpi_outputs.P_sym) -> dense [F, 225], both halves filled."""
    rows, cols = _tri_index(Ps.device)
    M = torch.zeros((Ps.shape[0], 225), dtype=Ps.dtype, device=Ps.device)
    M[:, rows * 15 + cols] = Ps      # lower half (element (j, i) of the column-major matrix sits at i * 15 + j)
    M[:, cols * 15 + rows] = Ps
    return M


def unpack_tri(Rt):
    """Packed [F, 120] of an UPPER-TRIANGULAR matrix (R_tri of cpi_sqrt_information_packed_batch) -> dense [F, 225]
    column-major with zeros below the diagonal (what cpi_sqrt_information_batch writes)."""
    rows, cols = _tri_index(Rt.device)
    M = torch.zeros((Rt.shape[0], 225), dtype=Rt.dtype, device=Rt.device)
    M[:, cols * 15 + rows] = Rt
    return M


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class Engine:
    """One C-ABI context.  stream=None: the engine FOLLOWS torch's current stream of its device -- every call is issued
    on torch.cuda.current_stream() as it is at that call (so inputs produced and outputs allocated under
    `with torch.cuda.stream(s):` are used on the stream that owns them).  An explicit stream pins the engine to it; the
    caller then orders that stream against the producers / consumers of the tensors (EnginePool does that bookkeeping)."""

    def __init__(self, device=None, stream=None):
        self.lib = _lib.load()
        self._follow = stream is None
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        ctx = C.c_void_p()
        if stream is None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(self.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else None
        rc = self.lib.cpi_ctx_create(self.device.index or 0, sptr, C.byref(ctx))
        if rc != 0:
            raise CpiError(rc, (self.lib.cpi_last_error(None) or b"").decode())
        self.ctx = ctx
        self.stream = stream

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.cpi_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CpiError(rc, (self.lib.cpi_last_error(self.ctx) or b"").decode())

    def _sync_stream(self):
        """Follow torch's current stream (see the class docstring); one pointer comparison per call."""
        if self._follow:
            cur = torch.cuda.current_stream(self.device)
            if cur.cuda_stream != self.stream.cuda_stream:
                self._check(self.lib.cpi_ctx_set_stream(self.ctx, C.c_void_p(cur.cuda_stream)))
                self.stream = cur

    def synchronize(self):
        self._check(self.lib.cpi_ctx_synchronize(self.ctx))

    # ------------------------------------------------------------------ preintegration
    @staticmethod
    def make_params(model=1, imu_avg=False, state_transition_jacobians=True, sigmas=DEFAULT_SIGMAS,
                    grav=DEFAULT_GRAV, lanes_per_window=0):
        p = CpiParams()
        p.sigma_w, p.sigma_wb, p.sigma_a, p.sigma_ab = sigmas
        p.grav[:] = grav
        p.model, p.imu_avg = int(model), int(bool(imu_avg))
        p.state_transition_jacobians = int(bool(state_transition_jacobians))
        p.lanes_per_window = int(lanes_per_window)
        return p

    def alloc_outputs(self, W, want=("mean", "jac", "cov"), model=1, packed=False):
        """packed=True: all fields are views of ONE flat buffer (field-major, cpi_amd.dist.pack_layout), returned under
        the extra key "_flat" together with "_fields" -- a rank's whole output is then one contiguous slab and the
        multi-GPU gather one collective (cpi_amd.dist.gather_packed)."""
        names = []
        for name, n in OUT_FIELDS:
            grp = _group_of(name)
            if grp not in want:
                continue
            if model != 2 and name in ("O_a", "O_b"):
                continue
            names.append((name, n))
        out = {}
        if packed:
            from .dist import alloc_packed
            flat, out = alloc_packed(names, W, self.device)
            out["_flat"], out["_fields"] = flat, names
            return out
        for name, n in names:
            shape = (W,) if n == 1 else (W, n)
            out[name] = torch.empty(shape, dtype=torch.float64, device=self.device)
        return out

    @staticmethod
    def _outputs_struct(out):
        o = CpiOutputs()
        for name, _ in OUT_FIELDS:
            t = out.get(name)
            setattr(o, name, t.data_ptr() if t is not None else None)
        return o

    def preintegrate(self, knots, lin, q_k_lin=None, params=None, want=("mean", "jac", "cov"), first=None,
                     count=None, N=None, out=None):
        """knots [W,N+1,7] (dense) or [K,7] with first[W] (int64) / count[W] (int32); lin [W,6];
        q_k_lin [W,4].  All CUDA float64 tensors.  Returns a dict of device tensors (matrices flat,
        column-major).  Asynchronous on the engine's stream."""
        params = params or self.make_params()
        if first is None:
            W, n1, seven = knots.shape
            N = n1 - 1
        else:
            W = first.shape[0]
            assert N is not None, "ragged layout needs N = max intervals per window"
        for t in (knots, lin, q_k_lin, first, count):
            assert t is None or (t.is_cuda and t.is_contiguous()), "inputs must be contiguous CUDA tensors"
        assert knots.dtype == torch.float64 and lin.dtype == torch.float64
        if out is None:
            out = self.alloc_outputs(W, want, params.model)
        o = self._outputs_struct(out)
        self._sync_stream()
        self._check(self.lib.cpi_preintegrate_batch(self.ctx, C.byref(params), W, N, _ptr(knots), _ptr(first), _ptr(count),
                                                    _ptr(lin), _ptr(q_k_lin), C.byref(o)))
        return out

    def preintegrate_host(self, knots, lin, q_k_lin=None, params=None, want=("mean", "jac", "cov"), count=None, pinned=True, out=None):
        """Dense batch held in HOST memory (CPU float64 tensors; pinned ones overlap upload / kernels / download):
        cpi_preintegrate_batch_host.  Returns a dict of CPU tensors (page-locked when pinned=True; out= re-uses the
        dict of an earlier call); synchronous."""
        params = params or self.make_params()
        W, n1, _ = knots.shape
        for t in (knots, lin, q_k_lin, count):
            assert t is None or (not t.is_cuda and t.is_contiguous()), "inputs must be contiguous CPU tensors"
        assert knots.dtype == torch.float64 and lin.dtype == torch.float64 and (count is None or count.dtype == torch.int32)
        if out is None:
            out = {}
            for name, n in OUT_FIELDS:
                grp = _group_of(name)
                if grp in want and (params.model == 2 or name not in ("O_a", "O_b")):
                    out[name] = torch.empty((W,) if n == 1 else (W, n), dtype=torch.float64, pin_memory=pinned)
        o = self._outputs_struct(out)
        self._sync_stream()
        self._check(self.lib.cpi_preintegrate_batch_host(self.ctx, C.byref(params), W, n1 - 1, _ptr(knots), None, _ptr(count), 0,
                                                         _ptr(lin), _ptr(q_k_lin), C.byref(o)))
        return out

    def tile_knots(self, knots):
        """dense knots [W, N+1, 7] -> tiles [ceil(W/64), N+1, 7, 64] on the device (cpi_tile_knots)."""
        W, n1, _ = knots.shape
        tiles = torch.empty(((W + 63) // 64, n1, 7, 64), dtype=torch.float64, device=self.device)
        self._sync_stream()
        self._check(self.lib.cpi_tile_knots(self.ctx, W, n1 - 1, _ptr(knots), _ptr(tiles)))
        return tiles

    def tile_windows(self, knots, first, count, N):
        """a shared knot stream [K, 7] indexed by first[W] (int64) / count[W] (int32) -> tiles [ceil(W/64), N+1, 7, 64] on the
        device (cpi_tile_windows).  A full extra pass: one-off use and tests."""
        W = first.shape[0]
        tiles = torch.empty(((W + 63) // 64, N + 1, 7, 64), dtype=torch.float64, device=self.device)
        self._sync_stream()
        self._check(self.lib.cpi_tile_windows(self.ctx, W, N, _ptr(knots), _ptr(first), _ptr(count), _ptr(tiles)))
        return tiles

    def assemble_tiles(self, stream, update_times, N, tiles=None, count=None):
        """Window assembly ON THE DEVICE, straight into the tiled layout (cpi_assemble_tiles; GraphSolver_IMU.cpp:50-69 for
        every window at once): stream [K, 7] with non-decreasing stamps, update_times [U] non-decreasing, both CUDA float64.
        Returns (tiles [ceil(U/64), N+1, 7, 64], count [U] int32 = the TRUE interval counts: check count.max() <= N)."""
        K, U = stream.shape[0], update_times.shape[0]
        assert stream.is_cuda and stream.is_contiguous() and update_times.is_cuda and update_times.is_contiguous()
        if tiles is None:
            tiles = torch.empty(((U + 63) // 64, N + 1, 7, 64), dtype=torch.float64, device=self.device)
        if count is None:
            count = torch.empty((U,), dtype=torch.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.cpi_assemble_tiles(self.ctx, K, _ptr(stream), U, _ptr(update_times), N, _ptr(tiles), _ptr(count)))
        return tiles, count

    def preintegrate_stream(self, stream, update_times, lin, q_k_lin=None, params=None, want=("mean", "jac", "cov"), N=None, out=None,
                            return_counts=False, check_counts=True, workspace=None):
        """One IMU stream cut at update times and preintegrated IN PLACE (cpi_preintegrate_stream: the caller-side loop of
        GraphSolver_IMU.cpp:43-75 for all windows at once, zero copies of the IMU data, every model and output).
        stream [K, 7] with non-decreasing stamps, update_times [U] non-decreasing, lin [U, 6], q_k_lin [U, 4]: CUDA float64.
        N = upper bound of the intervals per window; with check_counts (one synchronisation) raises when a window holds more.
        N=None (default): the exact bound, computed from the stamps by stream_bound() -- one searchsorted over the K stamps and ONE
        HOST SYNCHRONISATION on first use of a (stream, update_times) pair (cached afterwards by storage and version), issued on
        the engine's stream: that first call is NOT asynchronous and NOT graph-capturable.  Callers that need either pass an integer
        N (e.g. stream_bound() taken once, outside the capture) or N="loose" = min(K, 65535): no pass over the stamps, no
        synchronisation -- the library picks the mean kernel's lane split from N, so a loose bound costs speed on small batches
        (and check_counts=False keeps the whole call free of synchronisations)."""
        params = params or self.make_params()
        K, U = stream.shape[0], update_times.shape[0]
        for t in (stream, update_times, lin, q_k_lin):
            assert t is None or (t.is_cuda and t.is_contiguous() and t.dtype == torch.float64), "inputs must be contiguous CUDA float64 tensors"
        if N is None:
            N = self.stream_bound(stream, update_times)
        elif isinstance(N, str):
            assert N == "loose", 'N: an integer, None (exact bound, synchronises once per pair) or "loose"'
            N = max(1, min(K, 65535))
        if out is None:
            out = self.alloc_outputs(U, want, params.model)
        ws = workspace if workspace is not None else self.stream_workspace(U)
        o = self._outputs_struct(out)
        self._sync_stream()
        self._check(self.lib.cpi_preintegrate_stream(self.ctx, C.byref(params), K, _ptr(stream), U, _ptr(update_times), int(N), _ptr(lin),
                                                     _ptr(q_k_lin), _ptr(ws), C.byref(o)))
        # The workspace (28 bytes per window) must outlive the kernels that read it.  Allocated here it came from torch's
        # caching allocator on the stream the kernels were just issued on (follow mode), so dropping the reference is safe: the
        # block can only be handed to later, stream-ordered allocations of that stream.  An engine pinned to an explicit stream
        # says so to the allocator.  Nothing but output fields is stored in the returned dict.
        if workspace is None and not self._follow and self.stream is not None:
            ws.record_stream(self.stream)
        counts = torch.empty((0,), dtype=torch.int32, device=self.device)
        if U and (return_counts or check_counts):
            off = (self.lib.cpi_stream_counts(_ptr(ws), U) - ws.data_ptr()) // 4
            with torch.cuda.stream(self.stream) if (not self._follow and self.stream is not None) else _nullctx():
                counts = ws.view(torch.int32)[off:off + U].clone()   # a copy: a re-used workspace is overwritten by the next call
            if check_counts and int(counts.max().item()) > N:
                raise ValueError("preintegrate_stream: a window has %d intervals, more than N = %d" % (int(counts.max().item()), N))
        return (out, counts) if return_counts else out

    def stream_bound(self, stream, update_times):
        """Longest window (whole intervals + a tail) that cutting `stream` at `update_times` can produce: the closed form of
        the deque loop (GraphSolver_IMU.cpp:50-69; cpi_cut_windows_kernel) on the stamps, tail assumed.  Device tensors: issued
        on the ENGINE's stream (not torch's current one) and followed by one host synchronisation (.item()); the result is cached
        per (storage, shape, version) of the two tensors, so a loop over the same buffers pays it once."""
        key = tuple((t.data_ptr(), tuple(t.shape), t._version) for t in (stream, update_times))
        hit = getattr(self, "_bound_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        pinned = stream.is_cuda and not self._follow and self.stream is not None
        with torch.cuda.stream(self.stream) if pinned else _nullctx():
            n = self._stream_bound(stream, update_times)
        self._bound_cache = (key, n)
        return n

    @staticmethod
    def _stream_bound(stream, update_times):
        K, U = stream.shape[0], update_times.shape[0]
        if K == 0 or U == 0:
            return 1
        c = torch.searchsorted(stream[:, 0].contiguous(), update_times, right=True)
        fp = torch.zeros_like(c)
        fp[1:] = (c[:-1] - 1).clamp_min(0)
        m = torch.maximum((c - 1).clamp_min(0), fp) - fp
        return max(1, min(int(m.max().item()) + 1, 65535))

    def stream_workspace(self, U):
        """Device workspace of preintegrate_stream for U windows (28 bytes per window; re-usable across calls)."""
        return torch.empty((self.lib.cpi_stream_workspace_bytes(U) // 8 + 1,), dtype=torch.float64, device=self.device)

    def preintegrate_stream_host(self, stream, update_times, lin, q_k_lin=None, params=None, want=("mean", "jac", "cov"), N=None,
                                 pinned=False, return_counts=False):
        """preintegrate_stream for a caller that holds everything in HOST memory (cpi_preintegrate_stream_host): the IMU
        stream [K,7], the update times [U], one linearisation point per window -- CPU float64 tensors in, CPU tensors out;
        synchronous.  A window longer than N intervals raises (N defaults to the longest window the stamps allow: nothing can be)."""
        params = params or self.make_params()
        K, U = stream.shape[0], update_times.shape[0]
        N = int(N) if N is not None else self._stream_bound(stream, update_times)     # the same default as the device entry: same lane split, same bits
        for t in (stream, update_times, lin, q_k_lin):
            assert t is None or (not t.is_cuda and t.is_contiguous() and t.dtype == torch.float64), "inputs must be contiguous CPU float64 tensors"
        out = {}
        for name, n in OUT_FIELDS:
            grp = _group_of(name)
            if grp in want and (params.model == 2 or name not in ("O_a", "O_b")):
                out[name] = torch.empty((U,) if n == 1 else (U, n), dtype=torch.float64, pin_memory=pinned)
        cnt = torch.empty((U,), dtype=torch.int32)
        o = self._outputs_struct(out)
        self._sync_stream()
        self._check(self.lib.cpi_preintegrate_stream_host(self.ctx, C.byref(params), K, _ptr(stream), U, _ptr(update_times), N,
                                                          _ptr(lin), _ptr(q_k_lin), C.byref(o), _ptr(cnt)))
        if U and int(cnt.max()) > N:
            raise ValueError("preintegrate_stream_host: a window has %d intervals, N = %d" % (int(cnt.max()), N))
        return (out, cnt) if return_counts else out

    def preintegrate_tiled_host(self, tiles, W, lin, q_k_lin=None, params=None, count=None, pinned=True, out=None):
        """Mean outputs from tiles held in HOST memory (cpi_preintegrate_tiled_batch_host: chunked upload / kernel /
        download pipeline).  CPU float64 tensors; returns CPU tensors; synchronous."""
        params = params or self.make_params()
        N = tiles.shape[1] - 1
        for t in (tiles, lin, q_k_lin, count):
            assert t is None or (not t.is_cuda and t.is_contiguous()), "inputs must be contiguous CPU tensors"
        assert tiles.shape[0] == (W + 63) // 64 and tiles.shape[2:] == (7, 64)
        if out is None:
            out = {name: torch.empty((W,) if n == 1 else (W, n), dtype=torch.float64, pin_memory=pinned)
                   for name, n in OUT_FIELDS if name in MEAN_FIELDS}
        o = self._outputs_struct(out)
        self._sync_stream()
        self._check(self.lib.cpi_preintegrate_tiled_batch_host(self.ctx, C.byref(params), W, N, _ptr(tiles), _ptr(count), _ptr(lin),
                                                               _ptr(q_k_lin), C.byref(o)))
        return out

    def preintegrate_tiled(self, tiles, W, lin, q_k_lin=None, params=None, count=None, out=None, bind=False):
        """Mean outputs from the tiled layout (include/cpi_amd.h: cpi_preintegrate_tiled_batch).  bind=True: returns
        (call, out) with the foreign call pre-bound, like bind_preintegrate."""
        params = params or self.make_params()
        N = tiles.shape[1] - 1
        assert tiles.is_cuda and tiles.is_contiguous() and tiles.shape[0] == (W + 63) // 64 and tiles.shape[2:] == (7, 64)
        if out is None:
            out = self.alloc_outputs(W, ("mean",), params.model)
        o = self._outputs_struct(out)
        args = (self.ctx, C.byref(params), W, N, _ptr(tiles), _ptr(count), _ptr(lin), _ptr(q_k_lin), C.byref(o))
        fn, check, sync = self.lib.cpi_preintegrate_tiled_batch, self._check, self._sync_stream

        def call():
            sync()
            rc = fn(*args)
            if rc:
                check(rc)
        call._keep = (o, params, tiles, lin, q_k_lin, count, out)
        if bind:
            return call, out
        call()
        return out

    def bind_preintegrate(self, knots, lin, q_k_lin=None, params=None, want=("mean", "jac", "cov"), first=None, count=None,
                          N=None, out=None):
        """A zero-argument callable that issues exactly this preintegrate() call again and again: the ctypes argument
        objects are built once, so a step costs one foreign call (~2 us of host time instead of ~10) -- for callers that
        re-run fixed buffers in a loop (bench.py; a C or C++ host has no such overhead to begin with).  Returns
        (call, out)."""
        params = params or self.make_params()
        if first is None:
            W, n1, _ = knots.shape
            N = n1 - 1
        else:
            W = first.shape[0]
            assert N is not None
        for t in (knots, lin, q_k_lin, first, count):
            assert t is None or (t.is_cuda and t.is_contiguous()), "inputs must be contiguous CUDA tensors"
        if out is None:
            out = self.alloc_outputs(W, want, params.model)
        o = self._outputs_struct(out)
        args = (self.ctx, C.byref(params), W, N, _ptr(knots), _ptr(first), _ptr(count), _ptr(lin), _ptr(q_k_lin), C.byref(o))
        fn, check, sync = self.lib.cpi_preintegrate_batch, self._check, self._sync_stream

        def call():
            sync()
            rc = fn(*args)
            if rc:
                check(rc)
        call._keep = (o, params, knots, lin, q_k_lin, first, count, out)   # the buffers live as long as the callable
        return call, out

    # ------------------------------------------------------------------ factors
    def sqrt_information(self, P, out=None):
        """R = chol_upper(P^-1) per factor (GTSAM noiseModel::Gaussian::Covariance).  P [F,225] (dense) -> R [F,225] with zeros
        below the diagonal; P [F,120] (the packed upper triangle: outputs' P_sym) -> R_tri [F,120], the same values
        (cpi_sqrt_information_packed_batch)."""
        F, n = P.shape
        assert n in (225, 120) and P.is_contiguous(), "P must be [F,225] (dense) or [F,120] (packed upper triangle)"
        R = out if out is not None else torch.empty((F, n), dtype=torch.float64, device=self.device)
        assert R.shape == (F, n) and R.is_contiguous()
        self._sync_stream()
        fn = self.lib.cpi_sqrt_information_batch if n == 225 else self.lib.cpi_sqrt_information_packed_batch
        self._check(fn(self.ctx, F, _ptr(P), _ptr(R)))
        return R

    def factor_eval(self, model, meas, lin, q_k_lin, states, idx_i=None, idx_j=None, want_H=True, grav=DEFAULT_GRAV,
                    out=None, sqrt_info=None):
        """sqrt_info [F,225] or its packed triangle [F,120] (from sqrt_information): return the WHITENED residual / Jacobians
        (R e, R H1, R H2); the packed form reads 840 bytes less per factor and gives the same bits."""
        F = lin.shape[0]
        if out is None:
            out = {"err": torch.empty((F, 15), dtype=torch.float64, device=self.device)}
            if want_H:
                out["H1"] = torch.empty((F, 225), dtype=torch.float64, device=self.device)
                out["H2"] = torch.empty((F, 225), dtype=torch.float64, device=self.device)
        m = self._outputs_struct(meas)
        g = (C.c_double * 3)(*grav)
        self._sync_stream()
        if sqrt_info is None:
            self._check(self.lib.cpi_factor_eval_batch(self.ctx, int(model), g, F, C.byref(m), _ptr(lin), _ptr(q_k_lin),
                                                       _ptr(states), states.shape[0], _ptr(idx_i), _ptr(idx_j),
                                                       _ptr(out["err"]), _ptr(out.get("H1")), _ptr(out.get("H2"))))
        else:
            assert sqrt_info.shape == (F, 225) or sqrt_info.shape == (F, 120)
            fn = self.lib.cpi_factor_eval_whitened_batch if sqrt_info.shape[1] == 225 else self.lib.cpi_factor_eval_whitened_tri_batch
            self._check(fn(self.ctx, int(model), g, F, C.byref(m), _ptr(lin), _ptr(q_k_lin), _ptr(states), states.shape[0], _ptr(idx_i),
                           _ptr(idx_j), _ptr(sqrt_info), _ptr(out["err"]), _ptr(out.get("H1")), _ptr(out.get("H2"))))
        return out

    def factor_eval_packed(self, model, meas, lin, q_k_lin, states, idx_i=None, idx_j=None, grav=DEFAULT_GRAV, out=None):
        """State-dependent part of evaluateError only: [F,72] = err[15] + the 3x3 blocks H1(0,0), H1(6,0), H1(12,0),
        H1(0,3), R(q_GtoK), H2(0,0) (column-major) + 3 zeros; see include/cpi_amd.h.  unpack_factor() rebuilds the
        dense pair."""
        F = lin.shape[0]
        if out is None:
            out = torch.empty((F, 72), dtype=torch.float64, device=self.device)
        m = self._outputs_struct(meas)
        g = (C.c_double * 3)(*grav)
        self._sync_stream()
        self._check(self.lib.cpi_factor_eval_packed_batch(self.ctx, int(model), g, F, C.byref(m), _ptr(lin), _ptr(q_k_lin),
                                                          _ptr(states), states.shape[0], _ptr(idx_i), _ptr(idx_j), _ptr(out)))
        return out

    def factor_hessian(self, model, meas, lin, q_k_lin, states, sqrt_info, idx_i=None, idx_j=None, grav=DEFAULT_GRAV, out=None):
        """[F, 496]: packed upper triangle of the augmented information matrix [A1 A2 b]^T [A1 A2 b] with A = R H, b = -R e
        (what a GTSAM HessianFactor built from the linearised factor holds); see include/cpi_amd.h."""
        F = lin.shape[0]
        if out is None:
            out = torch.empty((F, 496), dtype=torch.float64, device=self.device)
        m = self._outputs_struct(meas)
        g = (C.c_double * 3)(*grav)
        self._sync_stream()
        assert sqrt_info.shape == (F, 225) or sqrt_info.shape == (F, 120), "sqrt_info: [F,225] dense or [F,120] packed triangle"
        fn = self.lib.cpi_factor_hessian_batch if sqrt_info.shape[1] == 225 else self.lib.cpi_factor_hessian_tri_batch
        self._check(fn(self.ctx, int(model), g, F, C.byref(m), _ptr(lin), _ptr(q_k_lin), _ptr(states), states.shape[0], _ptr(idx_i),
                       _ptr(idx_j), _ptr(sqrt_info), _ptr(out)))
        return out

    def predict(self, model, meas, states_i, idx_i=None, grav=DEFAULT_GRAV, out=None):
        F = meas["DT"].shape[0]
        xj = out if out is not None else torch.empty((F, 16), dtype=torch.float64, device=self.device)
        m = self._outputs_struct(meas)
        g = (C.c_double * 3)(*grav)
        self._sync_stream()
        self._check(self.lib.cpi_predict_batch(self.ctx, int(model), g, F, C.byref(m), _ptr(states_i), states_i.shape[0],
                                               _ptr(idx_i), _ptr(xj)))
        return xj


def unpack_factor(packed, meas):
    """Dense (err [F,15], H1 [F,225], H2 [F,225], column-major) from the packed evaluation and the measurement it was
    computed from -- the block table of include/cpi_amd.h (ImuFactorCPIv1.cpp:109-143,169-185)."""
    F = packed.shape[0]
    dev, f64 = packed.device, packed.dtype
    blk = lambda o: packed[:, o:o + 9].reshape(F, 3, 3).transpose(1, 2)      # column-major -> [row][col]
    cm = lambda t: t.reshape(F, 3, 3).transpose(1, 2)
    H1 = torch.zeros((F, 15, 15), dtype=f64, device=dev)
    H2 = torch.zeros((F, 15, 15), dtype=f64, device=dev)
    eye = torch.eye(3, dtype=f64, device=dev).expand(F, 3, 3)
    Rk = blk(51)
    H1[:, 0:3, 0:3], H1[:, 6:9, 0:3], H1[:, 12:15, 0:3], H1[:, 0:3, 3:6] = blk(15), blk(24), blk(33), blk(42)
    H1[:, 3:6, 3:6] = -eye
    H1[:, 9:12, 9:12] = -eye
    H1[:, 6:9, 3:6], H1[:, 6:9, 6:9], H1[:, 6:9, 9:12] = -cm(meas["J_b"]), -Rk, -cm(meas["H_b"])
    H1[:, 12:15, 3:6], H1[:, 12:15, 6:9] = -cm(meas["J_a"]), -meas["DT"][:, None, None] * Rk
    H1[:, 12:15, 9:12], H1[:, 12:15, 12:15] = -cm(meas["H_a"]), -Rk
    H2[:, 0:3, 0:3] = blk(60)
    H2[:, 3:6, 3:6] = eye
    H2[:, 6:9, 6:9] = Rk
    H2[:, 9:12, 9:12] = eye
    H2[:, 12:15, 12:15] = Rk
    return (packed[:, 0:15].contiguous(), H1.transpose(1, 2).reshape(F, 225).contiguous(),
            H2.transpose(1, 2).reshape(F, 225).contiguous())


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine


# ---------------------------------------------------------------------- reference-shaped classes
class _CpiBase:
    """Mirror of CpiBase (CpiBase.h:40-145).  feed_IMU() records the interval; reading any result
    field runs the whole window through cpi_preintegrate_batch on the GPU (one window = one batch).
    Unlike the reference's feed_IMU (which integrates a negative dt; only its caller skips it, GraphSolver_IMU.cpp:52),
    an interval with t_1 - t_0 <= 0 is skipped."""
    _model = 0

    def __init__(self, sigma_w, sigma_wb, sigma_a, sigma_ab, imu_avg_=False, engine=None):
        self._sig = (sigma_w, sigma_wb, sigma_a, sigma_ab)
        self.imu_avg = bool(imu_avg_)
        self.state_transition_jacobians = True
        self.b_w_lin = np.zeros(3); self.b_a_lin = np.zeros(3)
        self.q_k_lin = np.zeros(4); self.grav = np.zeros(3)
        self._iv = []      # (t0, t1, w0, a0, w1, a1)
        self._res = None
        self._engine = engine

    def setLinearizationPoints(self, b_w_lin_, b_a_lin_, q_k_lin_=None, grav_=None):
        self.b_w_lin = np.asarray(b_w_lin_, dtype=np.float64).reshape(3)
        self.b_a_lin = np.asarray(b_a_lin_, dtype=np.float64).reshape(3)
        self.q_k_lin = np.zeros(4) if q_k_lin_ is None else np.asarray(q_k_lin_, dtype=np.float64).reshape(4)
        self.grav = np.zeros(3) if grav_ is None else np.asarray(grav_, dtype=np.float64).reshape(3)
        self._res = None

    def feed_IMU(self, t_0, t_1, w_m_0, a_m_0, w_m_1=None, a_m_1=None):
        z = np.zeros(3)
        self._iv.append((float(t_0), float(t_1), np.asarray(w_m_0, float).reshape(3), np.asarray(a_m_0, float).reshape(3),
                         z if w_m_1 is None else np.asarray(w_m_1, float).reshape(3),
                         z if a_m_1 is None else np.asarray(a_m_1, float).reshape(3)))
        self._res = None

    def _knots(self):
        """Intervals -> knot records.  Consecutive intervals that chain (t1 == next t0 and the next
        reading equals this interval's w1/a1) share a knot.  The reference's feed_IMU only ever uses
        t1 - t0, so intervals need not chain: a knot whose time is NaN acts as a separator (both
        intervals touching it have a NaN dt and are skipped by the kernels)."""
        rows = []
        for (t0, t1, w0, a0, w1, a1) in self._iv:
            if rows and not self.imu_avg and rows[-1][0] == t0:
                # imu_avg == False: closing readings take no part in the arithmetic (CpiV1.h:77-86); when the TIMES chain the
                # previous closing knot simply takes this interval's opening reading (one knot per interval, as in cpi_host.hpp)
                rows[-1] = np.concatenate([[t0], w0, a0])
            chained = bool(rows) and rows[-1][0] == t0 and np.array_equal(rows[-1][1:4], w0) and np.array_equal(rows[-1][4:7], a0)
            if not chained:
                if rows:
                    rows.append(np.concatenate([[np.nan], np.zeros(6)]))
                rows.append(np.concatenate([[t0], w0, a0]))
            rows.append(np.concatenate([[t1], w1, a1]))
        return np.stack(rows) if rows else np.zeros((1, 7))

    def _run(self):
        if self._res is not None:
            return self._res
        eng = self._engine or default_engine()
        kn = self._knots()
        dev = eng.device
        knots = torch.from_numpy(kn[None]).to(dev)
        lin = torch.from_numpy(np.concatenate([self.b_w_lin, self.b_a_lin])[None]).to(dev)
        q = torch.from_numpy(self.q_k_lin[None]).to(dev)
        prm = eng.make_params(self._model, self.imu_avg, self.state_transition_jacobians, self._sig, tuple(self.grav))
        out = eng.preintegrate(knots, lin, q, prm)
        eng.synchronize()
        self._res = {k: v.cpu().numpy()[0] for k, v in out.items()}
        return self._res

    def _m3(self, name):
        return self._run()[name].reshape(3, 3).T  # column-major -> [row][col]

    DT = property(lambda s: float(s._run()["DT"]))
    alpha_tau = property(lambda s: s._run()["alpha"])
    beta_tau = property(lambda s: s._run()["beta"])
    q_k2tau = property(lambda s: s._run()["q"])
    J_q = property(lambda s: s._m3("J_q"))
    J_a = property(lambda s: s._m3("J_a"))
    J_b = property(lambda s: s._m3("J_b"))
    H_a = property(lambda s: s._m3("H_a"))
    H_b = property(lambda s: s._m3("H_b"))
    P_meas = property(lambda s: s._run()["P"].reshape(15, 15).T)


class CpiV1(_CpiBase):
    _model = 1


class CpiV2(_CpiBase):
    _model = 2
    O_a = property(lambda s: s._m3("O_a"))
    O_b = property(lambda s: s._m3("O_b"))


class ForsterDiscrete(_CpiBase):
    """The "Forster discrete" comparator as GraphSolver::createimufactor_discrete (GraphSolver_IMU.cpp:141-232) uses
    it: GTSAM's PreintegratedCombinedMeasurements driven by integrateMeasurement(acc, omega, dt), read back through the
    call site's conversions (:204-225) into the CpiV1-shaped result fields.  GTSAM is absent from the reference tree:
    parity of this model is unpinned (oracle/forster_oracle.c)."""
    _model = 3

    def __init__(self, sigma_g, sigma_wg, sigma_a, sigma_wa, engine=None):
        super().__init__(sigma_g, sigma_wg, sigma_a, sigma_wa, False, engine)
        self._t = 0.0

    def integrateMeasurement(self, measuredAcc, measuredOmega, dt):
        if not dt > 0:
            return
        w, a = np.asarray(measuredOmega, float).reshape(3), np.asarray(measuredAcc, float).reshape(3)
        if self._iv:   # the previous interval's closing knot opens this one: it carries this reading
            t0, t1, w0, a0, _, _ = self._iv[-1]
            self._iv[-1] = (t0, t1, w0, a0, w, a)
        self._iv.append((self._t, self._t + float(dt), w, a, w, a))
        self._t += float(dt)
        self._res = None

    deltaTij = property(lambda s: s.DT)


class _ImuFactorBase:
    _model = 0

    def _setup(self, covariance, deltatime, grav, alpha, beta, q_KtoK1, ba_lin, bg_lin, J_q, J_beta, J_alpha, H_beta,
               H_alpha, q_K_lin=None, O_beta=None, O_alpha=None, engine=None):
        self.covariance = np.asarray(covariance, float)
        self._grav = tuple(np.asarray(grav, float).reshape(3))
        cm = lambda M: np.asarray(M, float).reshape(3, 3).T.reshape(9)  # -> column-major flat
        f = lambda v, n: np.asarray(v, float).reshape(n)
        self._meas = dict(DT=np.array([float(deltatime)]), alpha=f(alpha, 3)[None], beta=f(beta, 3)[None],
                          q=f(q_KtoK1, 4)[None], J_q=cm(J_q)[None], J_b=cm(J_beta)[None], J_a=cm(J_alpha)[None],
                          H_b=cm(H_beta)[None], H_a=cm(H_alpha)[None])
        if self._model == 2:
            self._meas["O_b"] = cm(O_beta)[None]; self._meas["O_a"] = cm(O_alpha)[None]
        self._lin = np.concatenate([f(bg_lin, 3), f(ba_lin, 3)])[None]
        self._qk = None if q_K_lin is None else f(q_K_lin, 4)[None]
        self._engine = engine

    def evaluateError(self, state_i, state_j, want_H=True):
        """state = 16-vector [q(4) bg(3) v(3) ba(3) p(3)].  Returns error[15] (and H1, H2 as 15x15)."""
        eng = self._engine or default_engine()
        dev = eng.device
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        meas = {k: T(v) for k, v in self._meas.items()}
        states = T(np.stack([np.asarray(state_i, float).reshape(16), np.asarray(state_j, float).reshape(16)]))
        out = eng.factor_eval(self._model, meas, T(self._lin), None if self._qk is None else T(self._qk), states,
                              want_H=want_H, grav=self._grav)
        eng.synchronize()
        e = out["err"].cpu().numpy()[0]
        if not want_H:
            return e
        return e, out["H1"].cpu().numpy()[0].reshape(15, 15).T, out["H2"].cpu().numpy()[0].reshape(15, 15).T


class ImuFactorCPIv1(_ImuFactorBase):
    """Argument order of ImuFactorCPIv1.h:78-81 (keys omitted)."""
    _model = 1

    def __init__(self, covariance, deltatime, grav, alpha, beta, q_KtoK1, ba_lin, bg_lin, J_q, J_beta, J_alpha, H_beta,
                 H_alpha, engine=None):
        self._setup(covariance, deltatime, grav, alpha, beta, q_KtoK1, ba_lin, bg_lin, J_q, J_beta, J_alpha, H_beta,
                    H_alpha, engine=engine)


class ImuFactorCPIv2(_ImuFactorBase):
    """Argument order of ImuFactorCPIv2.h:82-85 (keys omitted)."""
    _model = 2

    def __init__(self, covariance, deltatime, grav, alpha, beta, q_KtoK1, q_K_lin, ba_lin, bg_lin, J_q, J_beta, J_alpha,
                 H_beta, H_alpha, O_beta, O_alpha, engine=None):
        self._setup(covariance, deltatime, grav, alpha, beta, q_KtoK1, ba_lin, bg_lin, J_q, J_beta, J_alpha, H_beta,
                    H_alpha, q_K_lin=q_K_lin, O_beta=O_beta, O_alpha=O_alpha, engine=engine)
