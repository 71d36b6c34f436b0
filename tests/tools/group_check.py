"""GPU: the n > 1 code paths of the device-set entries on ONE device (run by tests/test_gpu_group.py in a process of its
own, because the RCCL binding is made once per process):

    CPI_AMD_RCCL_LIB=tests/fake_rccl/libfake_rccl.so python tests/tools/group_check.py

cpi_test_group_create_shared(n) -> ncclCommInitAll through the stand-in; every rank runs cpi_preintegrate_batch on its block
through its own context / stream; cpi_group_gather moves the blocks to the root; the result must equal the UNSHARDED call
bit for bit -- for n = 2..8, roots first / last / middle, windows that do not divide (short and empty trailing ranks), both
models, slab-packed outputs (one message per peer) and separately allocated fields (one per field)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def chunked_cases(lib, fake, eng):
    """cpi_group_gather_chunk: every rank's block in k sub-blocks, sub-block c computed into a slab of its own and sent on the
    exchange streams while c + 1 computes.  n = 2 ... 16 ranks, k = 1 ... 8, windows that divide neither by n nor by k (short and empty
    sub-blocks), every root position; the result must equal the UNSHARDED call bit for bit and the stand-in must have seen exactly
    one message per (peer, non-empty sub-block)."""
    from cpi_amd import synth
    from cpi_amd._lib import CpiOutputs
    dev, N = eng.device, 23
    stats = (C.c_longlong * 5)()
    ncase = 0
    for n, W, root, k, model, want, packed in [
            (2, 1000, 0, 1, 1, ("mean", "jac", "cov"), True), (2, 1000, 1, 2, 1, ("mean", "jac", "cov_sym"), True),
            (3, 1001, 2, 3, 2, ("mean", "jac", "cov_sym"), True), (4, 999, 0, 4, 1, ("mean",), True),
            (5, 1003, 3, 5, 1, ("mean", "cov_sym"), True), (8, 4099, 0, 8, 1, ("mean", "jac", "cov_sym"), True),
            (8, 4099, 7, 6, 2, ("mean", "jac", "cov"), True), (8, 13, 0, 8, 1, ("mean", "cov_sym"), True),
            (16, 333, 9, 7, 1, ("mean", "cov_sym"), True), (4, 1001, 1, 3, 1, ("mean", "cov"), False), (2, 7, 0, 8, 2, ("mean", "jac"), True)]:
        kn, lin, q = synth.make_windows(W, N, seed=900 + n + W + k, device=dev)
        prm = eng.make_params(model, lanes_per_window=1)
        ref = eng.preintegrate(kn, lin, q, prm, want=want)
        torch.cuda.synchronize()
        g = C.c_void_p()
        assert lib.cpi_test_group_create_shared(n, 0, C.byref(g)) == 0, lib.cpi_group_last_error(None)
        try:
            root_out = {kk: torch.full_like(v, float("nan")) for kk, v in ref.items()}
            ro = eng._outputs_struct(root_out)
            per = (W + n - 1) // n
            cper = (per + k - 1) // k
            keep, expect_msgs = [], 0
            fake.fake_rccl_stats(stats)
            pairs0, groups0 = stats[1], stats[0]
            for c in range(k):
                locs = (CpiOutputs * n)()
                for r in range(n):
                    lo, hi = C.c_int64(), C.c_int64()
                    lib.cpi_shard_chunk_bounds(W, r, n, c, k, C.byref(lo), C.byref(hi))
                    lo, hi = lo.value, hi.value
                    blo, bhi = C.c_int64(), C.c_int64()
                    lib.cpi_shard_bounds(W, r, n, C.byref(blo), C.byref(bhi))
                    assert blo.value <= lo <= hi <= bhi.value and lo == min(bhi.value, blo.value + c * cper) and hi == min(bhi.value, lo + cper)
                    w = hi - lo
                    if w == 0:
                        continue
                    if r != root:
                        expect_msgs += 1 if packed else len(ref)
                    if packed:
                        slab = torch.empty((lib.cpi_outputs_slab_doubles(C.byref(ro), cper),), dtype=torch.float64, device=dev)
                        o = CpiOutputs()
                        assert lib.cpi_outputs_bind_slab(C.byref(ro), cper, slab.data_ptr(), C.byref(o)) == 0
                        keep.append(slab)
                    else:
                        loc = eng.alloc_outputs(w, want, model)
                        o = eng._outputs_struct(loc)
                        keep.append(loc)
                    locs[r] = o
                    ctx = lib.cpi_group_ctx(g, r)
                    rc = lib.cpi_preintegrate_batch(ctx, C.byref(prm), w, N, kn[lo:hi].data_ptr(), None, None, lin[lo:hi].data_ptr(),
                                                    q[lo:hi].data_ptr(), C.byref(o))
                    assert rc == 0, lib.cpi_last_error(ctx)
                assert lib.cpi_group_gather_chunk(g, root, W, c, k, locs, C.byref(ro)) == 0, lib.cpi_group_last_error(g)
            # the last chunk joined: synchronising the CONTEXTS alone must cover the whole exchange
            for r in range(n):
                assert lib.cpi_ctx_synchronize(lib.cpi_group_ctx(g, r)) == 0
            for kk in ref:
                assert torch.equal(root_out[kk], ref[kk]), ("chunked", n, W, root, k, model, packed, kk)
            assert lib.cpi_group_synchronize(g) == 0
            fake.fake_rccl_stats(stats)
            assert stats[0] == groups0 + k and stats[1] - pairs0 == expect_msgs, (stats[1] - pairs0, expect_msgs, n, W, k)
            # out-of-range chunk arguments are refused
            assert lib.cpi_group_gather_chunk(g, root, W, k, k, locs, C.byref(ro)) == 1
            assert lib.cpi_group_gather_chunk(g, root, W, 0, 0, locs, C.byref(ro)) == 1
        finally:
            lib.cpi_group_destroy(g)
        ncase += 1
    return ncase


def main():
    import cpi_amd
    from cpi_amd import _lib, synth
    from cpi_amd._lib import CpiOutputs
    assert os.environ.get("CPI_AMD_RCCL_LIB"), "run with CPI_AMD_RCCL_LIB pointing at the stand-in"
    fake = C.CDLL(os.environ["CPI_AMD_RCCL_LIB"])
    lib = _lib.load()
    eng = cpi_amd.Engine(device=0)
    dev = eng.device
    N = 37
    stats = (C.c_longlong * 5)()
    ncase = 0
    for n, W, root, model, want, packed in [
            (2, 1000, 0, 1, ("mean", "jac", "cov"), True), (2, 1000, 1, 2, ("mean", "jac", "cov"), True),
            (3, 1000, 2, 1, ("mean",), True), (3, 1000, 1, 1, ("mean", "cov"), False),
            (5, 1003, 0, 2, ("mean", "jac", "cov"), True), (8, 1001, 7, 1, ("mean", "jac", "cov"), True),
            (8, 5, 0, 1, ("mean", "cov"), True), (8, 5, 3, 2, ("mean", "jac"), False), (4, 3, 3, 1, ("mean",), True),
            (8, 4096, 0, 1, ("mean",), True), (16, 333, 9, 1, ("mean", "cov"), True),
            # ABI 3: the slab carries the covariance as its packed upper triangle (P_sym: 120 instead of 225 doubles per window)
            (4, 1001, 2, 1, ("mean", "jac", "cov_sym"), True), (8, 777, 0, 2, ("mean", "jac", "cov_sym"), True),
            (3, 50, 1, 1, ("mean", "cov", "cov_sym"), False)]:
        kn, lin, q = synth.make_windows(W, N, seed=100 + n + W, device=dev)
        prm = eng.make_params(model, lanes_per_window=1)     # pinned: the auto choice depends on the batch size
        ref = eng.preintegrate(kn, lin, q, prm, want=want)
        torch.cuda.synchronize()
        g = C.c_void_p()
        rc = lib.cpi_test_group_create_shared(n, 0, C.byref(g))
        assert rc == 0, (rc, lib.cpi_group_last_error(None))
        try:
            assert lib.cpi_group_size(g) == n
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
                    continue    # an empty trailing rank: its (zeroed) cpi_outputs is never read
                if packed:      # one slab per rank, laid out by the library (padded to the common block size)
                    slab = torch.empty((lib.cpi_outputs_slab_doubles(C.byref(ro), per),), dtype=torch.float64, device=dev)
                    o = CpiOutputs()
                    assert lib.cpi_outputs_bind_slab(C.byref(ro), per, slab.data_ptr(), C.byref(o)) == 0
                    keep.append(slab)
                else:
                    loc = eng.alloc_outputs(w, want, model)
                    o = eng._outputs_struct(loc)
                    keep.append(loc)
                locs[r] = o
                ctx = lib.cpi_group_ctx(g, r)
                rc = lib.cpi_preintegrate_batch(ctx, C.byref(prm), w, N, kn[lo:hi].data_ptr(), None, None, lin[lo:hi].data_ptr(),
                                                q[lo:hi].data_ptr(), C.byref(o))
                assert rc == 0, lib.cpi_last_error(ctx)
            fake.fake_rccl_stats(stats)
            pairs0, groups0 = stats[1], stats[0]
            assert lib.cpi_group_gather(g, root, W, locs, C.byref(ro)) == 0, lib.cpi_group_last_error(g)
            assert lib.cpi_group_synchronize(g) == 0
            fake.fake_rccl_stats(stats)
            peers = sum(1 for r in range(n) if r != root and min(W, (r + 1) * per) > min(W, r * per))
            nfields = len(ref)
            assert stats[0] == groups0 + 1
            assert stats[1] - pairs0 == (peers if packed else peers * nfields), (stats[1] - pairs0, peers, nfields, packed)
            assert peers == 0 or stats[4] == (1 if packed else nfields)
            assert peers == 0 or lib.cpi_group_last_gather_messages(g) == (1 if packed else nfields)
            for k in ref:
                assert torch.equal(root_out[k], ref[k]), (n, W, root, model, packed, k)
            # a second gather re-uses the staging area; a different root re-targets it
            root2 = (root + 1) % n
            for v in root_out.values():
                v.fill_(float("nan"))
            assert lib.cpi_group_gather(g, root2, W, locs, C.byref(ro)) == 0, lib.cpi_group_last_error(g)
            assert lib.cpi_group_synchronize(g) == 0
            for k in ref:
                assert torch.equal(root_out[k], ref[k]), ("second gather", n, W, root2, k)
        finally:
            lib.cpi_group_destroy(g)
        ncase += 1
    ncase += chunked_cases(lib, fake, eng)
    # a send / recv count mismatch must surface as CPI_ERR_RCCL (the stand-in checks what real RCCL would hang on)
    fake.fake_rccl_stats(stats)
    assert stats[3] >= 2 and stats[2] > 0
    print("group_check ok: %d cases, %d communicators, %d messages, %.1f MB through the stand-in" % (ncase, stats[3], stats[1], stats[2] / 1e6))


if __name__ == "__main__":
    main()
