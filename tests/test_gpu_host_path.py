"""GPU: cpi_preintegrate_batch_host on dense batches -- the chunked upload / kernels / download pipeline must return what the
device-pointer entry returns on the whole batch (same kernels per window once the lanes per window are pinned), from pinned
and from pageable host memory, with counts, for several chunks and for a single window."""
import numpy as np
import pytest
import torch

from cpi_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import cpi_amd
    return cpi_amd.Engine()


@pytest.mark.parametrize("pinned", [True, False])
def test_host_pipeline_equals_device_path_full_v1(eng, pinned):
    W, N = 150001, 12                                    # 3 chunks of 50048 / 50048 / 49905 windows
    kn, lin, q = synth.make_windows(W, N, seed=31, device=eng.device)
    prm = eng.make_params(1, lanes_per_window=1)
    dev = eng.preintegrate(kn, lin, q, prm)
    torch.cuda.synchronize()
    knh, linh = kn.cpu(), lin.cpu()
    if pinned:
        knh, linh = knh.pin_memory(), linh.pin_memory()
    for _ in range(2):                                   # second call: the context's staging buffers are reused
        out = eng.preintegrate_host(knh, linh, None, prm, pinned=pinned)
        assert set(out) == {k for k in dev if not k.startswith("_")}
        for k, v in out.items():
            d = dev[k].cpu()
            if k in ("DT", "alpha", "beta", "q", "P"):
                assert torch.equal(v, d), k
            else:
                assert (v - d).abs().max().item() < 1e-11, k


def test_host_pipeline_model2_counts_and_single_window(eng):
    W, N = 70000, 9                                      # 2 chunks
    kn, lin, q = synth.make_windows(W, N, seed=32, device=eng.device)
    g = torch.Generator(device="cpu"); g.manual_seed(8)
    cnt = torch.randint(0, N + 1, (W,), generator=g, dtype=torch.int32)
    prm = eng.make_params(2, lanes_per_window=1)
    dev = eng.preintegrate(kn, lin, q, prm, want=("mean",), count=cnt.to(eng.device))
    torch.cuda.synchronize()
    out = eng.preintegrate_host(kn.cpu(), lin.cpu(), q.cpu(), prm, want=("mean",), count=cnt, pinned=False)
    for k, v in out.items():
        assert torch.equal(v, dev[k].cpu()), k
    one = eng.preintegrate_host(kn[:1].cpu(), lin[:1].cpu(), q[:1].cpu(), eng.make_params(2), want=("mean", "jac", "cov"))
    ref = eng.preintegrate(kn[:1].contiguous(), lin[:1].contiguous(), q[:1].contiguous(), eng.make_params(2))
    torch.cuda.synchronize()
    for k, v in one.items():
        assert torch.equal(v, ref[k].cpu()), k


def test_pinned_allocator_round_trip(eng):
    import ctypes as C
    p = eng.lib.cpi_host_alloc(1 << 20)
    assert p
    buf = (C.c_double * 16).from_address(p)
    buf[3] = 2.5
    assert buf[3] == 2.5
    eng.lib.cpi_host_free(p)
    eng.lib.cpi_host_free(None)
