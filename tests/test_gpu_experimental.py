"""GPU: the measurement-only mean kernels (profiles/r02_mean_lds_dma.md) must stay correct, or the A/B numbers
they produce mean nothing.  They are NOT in libcpi_amd.so: `python -m cpi_amd.build --experiments` compiles them (and the
CPI_AMD_MEAN_DMA / CPI_AMD_MEAN_BLK / CPI_AMD_MEAN_LINE switches) into cpi_amd/libcpi_amd_exp.so, which CPI_AMD_LIB selects.  The switches are read
once per process, so each configuration runs tests/tools/dma_check.py (oracle comparison over ragged sizes, counts,
both models, imu_avg) in its own process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"CPI_AMD_MEAN_DMA": "4,2,0"}, {"CPI_AMD_MEAN_DMA": "4,2,1"}, {"CPI_AMD_MEAN_DMA": "2,3,0"},
                                 {"CPI_AMD_MEAN_BLK": "8"}, {"CPI_AMD_MEAN_BLK": "16"}, {"CPI_AMD_MEAN_LINE": "1"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_experimental_mean_kernels_match_the_oracle(env):
    from cpi_amd import build
    build.build(experiments=True)
    env = dict(env, CPI_AMD_LIB=build.LIB_EXP)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "dma_check.py")], env=dict(os.environ, **env),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "dma_check ok" in p.stdout, p.stdout[-2000:]
