#!/bin/bash
# Round-2 measurement artefacts (run through gpurun): PMC counters of bench.py's rows (build-stamped JSON), then the
# kernel-trace summary of the default bench command.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash tools/pmc_collect.sh gpurun_out/r02_pmc.json > gpurun_out/r02_pmc_collect.log 2>&1; tail -40 gpurun_out/r02_pmc_collect.log
bash tools/prof_run.sh gpurun_out/r02_kernel_stats.md --steps 20 --warmup 5 > /dev/null 2>&1; cat gpurun_out/r02_kernel_stats.md | head -40
