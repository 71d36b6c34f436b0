cd ${GRAFT_REPO_ROOT:-.}
mb() { local lib=cpi_amd/libcpi_amd${1:+_$1}.so; CPI_AMD_LIB=$PWD/$lib python tools/microbench.py "${@:2}" 2>&1 | grep "launch_us"; }
for round in 1 2; do for t in "" c3; do mb "$t" v1_mean:10000:0:400 v1_mean:10000:5:400 v1_mean:10000:8:400 v1_mean:5000:0:400 v1_mean:20000:0:300 v1_mean:30000:0:300 v1_mean:60000:0:100 v2_mean:10000:0:400; done; done | tee gpurun_out/r04_c3_small.txt
