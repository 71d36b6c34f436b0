#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_stream.py -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r04_pytest_l.txt
