"""Several engine contexts for INDEPENDENT batches.

Calls on one cpi_ctx execute in order on its stream.  Small batches leave the GPU partly idle while one launch ramps up
and the previous one drains (10 k-window mean-only batches: 12.4 us per batch on one context, 7.8 us round-robin over
three -- tools/overlap_bench.py), so a caller that has independent batches -- different inputs, different outputs --
can issue them through several contexts, one HIP stream each.  EnginePool does the stream bookkeeping:

    pool = cpi_amd.EnginePool(3)
    jobs = [pool.preintegrate(kn, lin, q, prm, want=("mean",)) for kn, lin, q in batches]   # returns at once
    for job in jobs:
        out = job.result()        # makes the CURRENT torch stream wait for that batch; out = dict of device tensors

Ordering rules: a submitted batch starts after everything already queued on the caller's current stream at submit time
(its inputs are ready); nothing on the caller's stream waits for it until result() is called.  Batches submitted to the
same context run in order; batches on different contexts may overlap, so they must not share output buffers.
"""
import torch

from .engine import Engine


class Job:
    def __init__(self, out, event):
        self._out, self._event = out, event

    def result(self):
        """Orders the caller's current stream after this batch and returns its outputs."""
        torch.cuda.current_stream().wait_event(self._event)
        return self._out

    def synchronize(self):
        self._event.synchronize()
        return self._out


class EnginePool:
    def __init__(self, n=3, device=None):
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        self.engines = [Engine(device=self.device.index, stream=s) for s in self.streams]
        self._next = 0

    def __len__(self):
        return len(self.engines)

    def _pick(self):
        i = self._next
        self._next = (i + 1) % len(self.engines)
        self.streams[i].wait_stream(torch.cuda.current_stream(self.device))   # inputs queued so far are ready first
        return self.engines[i], self.streams[i]

    @staticmethod
    def _tensors(obj):
        if isinstance(obj, torch.Tensor):
            yield obj
        elif isinstance(obj, dict):
            for v in obj.values():
                yield from EnginePool._tensors(v)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                yield from EnginePool._tensors(v)

    def _submit(self, method, *args, **kw):
        eng, stream = self._pick()
        out = getattr(eng, method)(*args, **kw)
        # inputs and outputs were allocated on the caller's stream but are used on the pool's: tell the caching
        # allocator, so that freeing one early cannot hand its memory out while the batch is still running
        for t in self._tensors((args, kw, out)):
            if t.is_cuda:
                t.record_stream(stream)
        ev = torch.cuda.Event()
        ev.record(stream)
        return Job(out, ev)

    def preintegrate(self, *args, **kw):
        return self._submit("preintegrate", *args, **kw)

    def factor_eval(self, *args, **kw):
        return self._submit("factor_eval", *args, **kw)

    def factor_eval_packed(self, *args, **kw):
        return self._submit("factor_eval_packed", *args, **kw)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        for e in self.engines:
            e.close()
