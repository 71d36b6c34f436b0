#!/bin/bash
# Same-box A/B of --custom variant libraries (python -m cpi_amd.build --custom <tag> -D...) on microbench rows, two alternating rounds:
#   bash tools/exp/nt_ab.sh "<tag> <tag> ..." "<row> <row> ..."      ("" = the default library)
# e.g. the round-4 non-temporal hints (profiles/r04_nt_hints.md):
#   bash tools/exp/nt_ab.sh "'' ntl8" "factor_v1_packed:1000000:0 factor_v2_packed:1000000:0"
cd ${GRAFT_REPO_ROOT:-.}
TAGS=${1:-"''"}
ROWS=${2:-"v1_mean:1000000:0"}
mb() { local lib=cpi_amd/libcpi_amd${1:+_$1}.so; echo "== $lib"; CPI_AMD_LIB=$PWD/$lib python tools/microbench.py $ROWS 2>&1 | grep "launch_us"; }
for round in 1 2; do for t in $TAGS; do [ "$t" = "''" ] && t=""; mb "$t"; done; done
