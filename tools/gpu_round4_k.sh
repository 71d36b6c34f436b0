#!/bin/bash
# Round-4 session K: the GPU suite on the final tree, the randomised campaign (streams, layouts, lane splits, sweeps against the
# oracle) and the run-to-run determinism check on the final library.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r04_pytest_final.txt
for seed in 11 12 13; do timeout 900 python tests/tools/fuzz_campaign.py 300 $seed 2>&1 | tail -4; done | tee gpurun_out/r04_fuzz.txt
timeout 900 python tests/tools/determinism.py 100 2>&1 | tail -6 | tee gpurun_out/r04_determinism.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
