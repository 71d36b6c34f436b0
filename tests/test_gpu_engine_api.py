"""GPU: host-side behaviour of the Python mirror that ADVICE round 5 asked about -- Engine.preintegrate_stream's default bound
(exact, one synchronisation per (stream, update_times) pair, cached; N="loose" = no pass over the stamps, no synchronisation)."""
import numpy as np
import pytest
import torch

import cpi_amd
from cpi_amd import synth
from tests.tol import REG_MEAN

pytestmark = pytest.mark.gpu


def test_stream_bound_is_cached_per_pair_and_the_loose_bound_needs_no_pass():
    eng = cpi_amd.Engine()
    stream, upd, lin, q = synth.make_stream(3000, 17, seed=12, device=eng.device, phase=0.37)
    n = eng.stream_bound(stream, upd)
    assert n == 18                                         # 17 whole intervals + the partial tail
    assert eng._bound_cache[1] == n
    key = eng._bound_cache[0]
    assert eng.stream_bound(stream, upd) == n and eng._bound_cache[0] == key      # served from the cache: same storage, same version
    upd2 = upd.clone()
    assert eng.stream_bound(stream, upd2) == n and eng._bound_cache[0] != key     # another tensor: recomputed
    upd2[1::2] += 0.01                                     # in-place edit bumps the version: the cached bound must not be trusted
    k2 = eng._bound_cache[0]
    n2 = eng.stream_bound(stream, upd2)
    assert eng._bound_cache[0] != k2 and n2 >= n
    exact = eng.preintegrate_stream(stream, upd, lin, q, eng.make_params(1), want=("mean",))                 # N=None: the exact bound
    pinned = eng.preintegrate_stream(stream, upd, lin, q, eng.make_params(1), want=("mean",), N=n)
    loose, counts = eng.preintegrate_stream(stream, upd, lin, q, eng.make_params(1), want=("mean",), N="loose", check_counts=False,
                                            return_counts=True)
    torch.cuda.synchronize()
    for k in exact:
        assert torch.equal(exact[k], pinned[k])
        # a looser bound may pick another lane split (the library chooses it from N): same values to the regression gate
        assert (exact[k] - loose[k]).abs().max().item() <= REG_MEAN * max(1.0, exact[k].abs().max().item())
    assert int(counts.max()) == n
    with pytest.raises(AssertionError):
        eng.preintegrate_stream(stream, upd, lin, q, eng.make_params(1), want=("mean",), N="tight")
