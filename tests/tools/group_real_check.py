"""GPU, >= 2 devices: the device-set entries of the C-ABI on REAL RCCL (run by tests/test_gpu_multi.py in a process of its
own; on a 1-GPU box the test is skipped and tests/tools/group_check.py covers the same code through the stand-in).

    python tests/tools/group_real_check.py [n_devices]

cpi_group_create(n) -> ncclCommInitAll on n distinct devices (librccl.so.1, bound lazily by the PRODUCT library); rank r
computes its cpi_shard_bounds block on ITS OWN device through its own context / stream; cpi_group_gather sends every peer's
slab straight to the root over xGMI; the gathered arrays must equal the UNSHARDED call (made on the root's device) bit for
bit -- roots first and last, windows that do not divide (a short trailing rank), both models, slab-packed outputs (one
ncclSend per peer) and separately allocated fields (one per field)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import cpi_amd
    from cpi_amd import _lib, synth
    from cpi_amd._lib import CpiOutputs
    ndev = torch.cuda.device_count()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else ndev
    assert 2 <= n <= ndev, "needs >= 2 devices (n = %d, visible %d)" % (n, ndev)
    assert "CPI_AMD_RCCL_LIB" not in os.environ and "CPI_AMD_LIB" not in os.environ, "this check is for the product library on real RCCL"
    lib = _lib.load()
    N = 37
    ncase = 0
    g = C.c_void_p()
    rc = lib.cpi_group_create(n, None, C.byref(g))
    assert rc == 0, (rc, lib.cpi_group_last_error(None))
    try:
        assert lib.cpi_group_size(g) == n
        for W, root, model, want, packed in [
                (n * 4096, 0, 1, ("mean",), True), (n * 1000 + 3, n - 1, 2, ("mean", "jac", "cov"), True),
                (n * 500 - 1, 0, 1, ("mean", "cov"), False), (n * 2000 + 1, n - 1, 1, ("mean", "jac", "cov"), True),
                (n + 1, 0, 2, ("mean", "jac"), False),
                # ABI 3: the slab carries the covariance as its packed upper triangle (120 doubles per window instead of 225)
                (n * 1500 + 7, 0, 1, ("mean", "jac", "cov_sym"), True), (n * 700 + 1, n - 1, 2, ("mean", "jac", "cov_sym"), True)]:
            rdev = torch.device("cuda", root)
            eng = cpi_amd.Engine(device=root)
            kn, lin, q = synth.make_windows(W, N, seed=300 + n + W)           # host copies: every rank uploads its block
            prm = eng.make_params(model, lanes_per_window=1)                    # pinned: the auto choice depends on the batch size
            with torch.cuda.device(rdev):
                ref = eng.preintegrate(kn.to(rdev), lin.to(rdev), q.to(rdev), prm, want=want)
                torch.cuda.synchronize(rdev)
            root_out = {k: torch.full_like(v, float("nan")) for k, v in ref.items()}
            ro = eng._outputs_struct(root_out)
            locs = (CpiOutputs * n)()
            keep = []
            per = (W + n - 1) // n
            for r in range(n):
                lo, hi = C.c_int64(), C.c_int64()
                lib.cpi_shard_bounds(W, r, n, C.byref(lo), C.byref(hi))
                lo, hi = lo.value, hi.value
                w = hi - lo
                if w == 0:
                    continue
                dev = torch.device("cuda", r)
                dk, dl, dq = kn[lo:hi].to(dev), lin[lo:hi].to(dev), q[lo:hi].to(dev)
                if packed:
                    slab = torch.empty((lib.cpi_outputs_slab_doubles(C.byref(ro), per),), dtype=torch.float64, device=dev)
                    o = CpiOutputs()
                    assert lib.cpi_outputs_bind_slab(C.byref(ro), per, slab.data_ptr(), C.byref(o)) == 0
                    keep.append((slab, dk, dl, dq))
                else:
                    loc = {k: torch.empty((w,) + tuple(v.shape[1:]), dtype=torch.float64, device=dev) for k, v in ref.items()}
                    o = eng._outputs_struct(loc)
                    keep.append((loc, dk, dl, dq))
                locs[r] = o
                torch.cuda.synchronize(dev)                                     # uploads done before the group's own stream reads them
                ctx = lib.cpi_group_ctx(g, r)
                rc = lib.cpi_preintegrate_batch(ctx, C.byref(prm), w, N, dk.data_ptr(), None, None, dl.data_ptr(), dq.data_ptr(), C.byref(o))
                assert rc == 0, lib.cpi_last_error(ctx)
            assert lib.cpi_group_gather(g, root, W, locs, C.byref(ro)) == 0, lib.cpi_group_last_error(g)
            assert lib.cpi_group_synchronize(g) == 0
            assert lib.cpi_group_last_gather_messages(g) == (1 if packed else len(ref))
            for k in ref:
                assert torch.equal(root_out[k], ref[k]), (n, W, root, model, packed, k)
            for v in root_out.values():                                         # again: the staging area is re-used
                v.fill_(float("nan"))
            assert lib.cpi_group_gather(g, root, W, locs, C.byref(ro)) == 0, lib.cpi_group_last_error(g)
            assert lib.cpi_group_synchronize(g) == 0
            for k in ref:
                assert torch.equal(root_out[k], ref[k]), ("second gather", n, W, root, k)
            ncase += 1
            del keep
            # ---- the exchange INSIDE one batch (cpi_group_gather_chunk): k sub-blocks per rank, each with a slab of its own, sub-block
            # c on the wire (the group's exchange streams) while c + 1 computes; the last call joins
            if packed:
                for k in (1, 3, 8):
                    for v in root_out.values():
                        v.fill_(float("nan"))
                    keepc = []
                    cper = (per + k - 1) // k
                    for c in range(k):
                        locs = (CpiOutputs * n)()
                        for r in range(n):
                            lo, hi = C.c_int64(), C.c_int64()
                            lib.cpi_shard_chunk_bounds(W, r, n, c, k, C.byref(lo), C.byref(hi))
                            lo, hi = lo.value, hi.value
                            if hi <= lo:
                                continue
                            dev = torch.device("cuda", r)
                            dk, dl, dq = kn[lo:hi].to(dev), lin[lo:hi].to(dev), q[lo:hi].to(dev)
                            slab = torch.empty((lib.cpi_outputs_slab_doubles(C.byref(ro), cper),), dtype=torch.float64, device=dev)
                            o = CpiOutputs()
                            assert lib.cpi_outputs_bind_slab(C.byref(ro), cper, slab.data_ptr(), C.byref(o)) == 0
                            keepc.append((slab, dk, dl, dq))
                            locs[r] = o
                            torch.cuda.synchronize(dev)
                            ctx = lib.cpi_group_ctx(g, r)
                            assert lib.cpi_preintegrate_batch(ctx, C.byref(prm), hi - lo, N, dk.data_ptr(), None, None, dl.data_ptr(), dq.data_ptr(),
                                                              C.byref(o)) == 0, lib.cpi_last_error(ctx)
                        assert lib.cpi_group_gather_chunk(g, root, W, c, k, locs, C.byref(ro)) == 0, lib.cpi_group_last_error(g)
                        assert lib.cpi_group_last_gather_messages(g) in (0, 1)
                    assert lib.cpi_group_synchronize(g) == 0
                    for kk in ref:
                        assert torch.equal(root_out[kk], ref[kk]), ("chunked", n, W, root, model, k, kk)
                    ncase += 1
                    del keepc
    finally:
        lib.cpi_group_destroy(g)
    print("group_real_check ok: %d cases on %d devices over real RCCL" % (ncase, n))


if __name__ == "__main__":
    main()
