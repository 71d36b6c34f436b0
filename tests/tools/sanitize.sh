#!/bin/bash
# host sanitizer pass over everything that runs on the CPU side of the boundary.
#   tests/tools/sanitize.sh cpu    (here, no GPU)   the C restatement (oracle/*.c), the host emulation of the kernel arithmetic
#                                                   (tests/hostsim) and the C++ facade's CPU-only parts (parse / assemble twins)
#                                                   under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the pytest
#                                                   modules that exercise them
#   tests/tools/sanitize.sh gpu    (GPU box)        the C-ABI's host translation unit (cpi_abi.hip: contexts, device sets, the
#                                                   slab gather, the three-stream host pipeline) rebuilt with ASan + UBSan and
#                                                   linked with the ordinary kernel objects into libcpi_amd_asan.so; the C++
#                                                   hosts (tests/cpp/test_facade / test_imu_stream / test_group incl. the 5-rank shared mode /
#                                                   test_threads) compiled with the same runtime against it; ThreadSanitizer on
#                                                   test_threads
# Writes gpurun_out/sanitize_<part>.txt; exit code 0 = no sanitizer report.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
R=$PWD
PART=${1:-cpu}
mkdir -p gpurun_out build/san
OUT=gpurun_out/sanitize_$PART.txt
: > $OUT
FAIL=0
say() { echo "$@" | tee -a $OUT; }
run() {   # run <label> <cmd...>: a sanitizer report (or a non-zero exit) fails the pass
  local label=$1; shift
  local log=build/san/$(echo "$label" | tr ' /' '__').log
  "$@" > $log 2>&1; local rc=$?
  # Known artefact of ROCm's ASan runtime, NOT a report about this code: at process exit libamdhip64's finaliser tears libhsa-runtime64
  # down, an `operator delete` there makes ASan recycle a quarantined chunk of its DEVICE allocator after that allocator has flagged the
  # runtime as unloaded, and the sanitizer aborts on its own CHECK (sanitizer_allocator_device.h: "!dev_runtime_unloaded_").  Seen since
  # round 6 on the facade host that lets worker threads create thread-local contexts.  Accepted only when the report is that CHECK,
  # there is no AddressSanitizer ERROR / UBSan report, and NO frame of the stack lies in libcpi_amd, the facade or the test program.
  if grep -q "CHECK failed: sanitizer_allocator_device.h" $log && ! grep -qE "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer|ERROR: LeakSanitizer" $log \
     && ! grep -E "^ +#[0-9]+ " $log | grep -qE "libcpi_amd|cpi_host|test_[a-z_]+_asan|cpi_abi"; then
    say "clean $label   [ASan-runtime CHECK in the HSA teardown at exit (all frames inside the sanitizer / libhsa-runtime64 / libamdhip64 / libc): ignored]"
    return
  fi
  if [ $rc -ne 0 ] || grep -qE "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer|ERROR: LeakSanitizer" $log; then
    say "FAIL  $label (rc=$rc)"; grep -E "Sanitizer|runtime error" $log | head -5 | tee -a $OUT; tail -3 $log | tee -a $OUT; FAIL=1
  else
    say "clean $label   [$(tail -1 $log | cut -c1-100)]"
  fi
}
if [ "$PART" = cpu ]; then
  SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
  gcc -std=c99 -O1 $SAN -fPIC -shared -ffp-contract=off -o build/san/liboracle_asan.so oracle/cpi_oracle.c oracle/forster_oracle.c -lm -lpthread || exit 2
  g++ -std=c++17 -O1 $SAN -fPIC -shared -Wno-unknown-pragmas -ffp-contract=off -o build/san/libhostsim_asan.so tests/hostsim/hostsim.cpp || exit 2
  g++ -std=c++17 -O1 $SAN tests/cpp/test_stream.cpp -o build/san/test_stream_asan -Lcpi_amd -lcpi_amd -Wl,-rpath,$R/cpi_amd -Wl,-rpath,/opt/rocm/lib || exit 2
  PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
  run "oracle/*.c under pytest (golden vectors, factor, Forster, stream, quat_ops)" env LD_PRELOAD=$PRE CPI_ORACLE_LIB=$R/build/san/liboracle_asan.so \
      python -m pytest -x -q -p no:cacheprovider tests/test_oracle_golden.py tests/test_factor_oracle.py tests/test_forster_oracle.py tests/test_stream.py tests/test_quat_ops.py -m "not gpu"
  run "tests/hostsim (cpi_math.hpp on the host) under pytest" env LD_PRELOAD=$PRE CPI_HOSTSIM_LIB=$R/build/san/libhostsim_asan.so \
      python -m pytest -x -q -p no:cacheprovider tests/test_hostsim.py -m "not gpu"
  printf '%s\n' 1275.06 1275.4 1276.0 1277.5 > build/san/ut.txt
  run "cpi_host.hpp parse_imu_text + assemble_windows" build/san/test_stream_asan tests/golden/imu_gazebo200_excerpt.dat build/san/ut.txt
  run "cpi_host.hpp assemble_windows_tiled" build/san/test_stream_asan tests/golden/imu_gazebo200_excerpt.dat build/san/ut.txt tiled
  g++ -std=c++17 -O1 $SAN tests/cpp/test_facade_lazy_cpu.cpp -o build/san/test_facade_lazy_asan -Lcpi_amd -lcpi_amd -Wl,-rpath,$R/cpi_amd -Wl,-rpath,/opt/rocm/lib || exit 2
  run "cpi_host.hpp lazy result members (fresh / stored / copied / a read that needs the device)" build/san/test_facade_lazy_asan
else
  CXX=/opt/rocm/lib/llvm/bin/clang++
  SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -shared-libsan"
  RT=$(dirname $($CXX -print-file-name=libclang_rt.asan-x86_64.so))
  python -m cpi_amd.build --test-hooks > /dev/null || exit 2   # the hosts below use the shared-device hook: objects with -DCPI_TEST_HOOKS
  BID=$(python -c "from cpi_amd import build; print(build.source_id())")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC $SAN -fno-gpu-sanitize -DCPI_TEST_HOOKS -DCPI_BUILD_ID=\"$BID\" -c -o build/san/cpi_abi_asan.o cpi_amd/csrc/cpi_abi.hip || exit 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -o build/san/libcpi_amd.so build/san/cpi_abi_asan.o cpi_amd/csrc/_obj/cpi_mean.o cpi_amd/csrc/_obj/cpi_cov.o cpi_amd/csrc/_obj/cpi_factor_test.o -ldl || exit 2
  LNK="-Lbuild/san -lcpi_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/build/san -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT"
  for t in test_facade test_group test_threads test_imu_stream; do
    $CXX -std=c++17 -O1 $SAN -pthread -D__HIP_PLATFORM_AMD__ -DCPI_TEST_HOOKS -I/opt/rocm/include tests/cpp/$t.cpp -o build/san/${t}_asan $LNK || exit 2
  done
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
  python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from cpi_amd import synth
d = dict(np.load('tests/golden/pre_w48.npz'))
with open('build/san/w48.bin', 'wb') as f:
    np.array([d['knots'].shape[0], d['knots'].shape[1] - 1], dtype=np.float64).tofile(f); d['knots'].tofile(f); d['lin'].tofile(f); d['q_k_lin'].tofile(f)
kn, lin, q = synth.make_windows(70001, 10, seed=77)
with open('build/san/thr.bin', 'wb') as f:
    np.array([70001, 10], dtype=np.float64).tofile(f); kn.numpy().tofile(f); lin.numpy().tofile(f); q.numpy().tofile(f)
PY
  FAKE=$(python -c "from tests import fake_rccl_py as f; print(f.lib_path())")
  for m in 1 2 3; do run "test_facade model $m (ASan+UBSan: facade, C-ABI host code, host pipeline)" build/san/test_facade_asan build/san/w48.bin $m; done
  run "test_group 1 device" build/san/test_group_asan build/san/w48.bin 1 1
  run "test_group 5 ranks on one device (slab gather, RCCL stand-in)" env CPI_AMD_RCCL_LIB=$FAKE build/san/test_group_asan build/san/w48.bin 2 5 shared
  run "test_group 1 device, exchange in 3 sub-blocks (cpi_group_gather_chunk, packed covariance in the slab)" build/san/test_group_asan build/san/w48.bin 1 1 native 3
  run "test_group 5 ranks on one device, exchange in 3 sub-blocks (RCCL stand-in)" env CPI_AMD_RCCL_LIB=$FAKE build/san/test_group_asan build/san/w48.bin 2 5 shared 3
  run "test_threads 4 host threads (ASan+UBSan)" build/san/test_threads_asan build/san/thr.bin 4
  python - <<'PY'
import numpy as np
kn = np.loadtxt('tests/golden/imu_gazebo200_excerpt.dat')
t0 = 1e-3 * kn[0, 7]
ut = t0 + 0.0523 + 0.1 * np.arange(20)
np.savetxt('build/san/ut20.txt', ut, fmt='%.17g')
r = np.random.default_rng(2)
q = r.standard_normal((20, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True); q[q[:, 3] < 0] *= -1
np.savetxt('build/san/lin20.txt', np.concatenate([0.01 * r.standard_normal((20, 6)), q], axis=1), fmt='%.17g')
PY
  for m in 1 2 3; do run "test_imu_stream model $m (ASan+UBSan: ImuStream facade, cpi_preintegrate_stream_host)" build/san/test_imu_stream_asan tests/golden/imu_gazebo200_excerpt.dat build/san/ut20.txt build/san/lin20.txt $m; done
  # ThreadSanitizer: the facade + test are instrumented, the HIP runtime is not (its internal threads are invisible to TSan)
  $CXX -std=c++17 -O1 -fsanitize=thread -g -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/test_threads.cpp -o build/san/test_threads_tsan \
      -Lcpi_amd -lcpi_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/cpi_amd -Wl,-rpath,/opt/rocm/lib 2>> $OUT || say "TSan build failed"
  # libhsa-runtime64 / libamdhip64 are not instrumented: TSan cannot see the synchronisation inside them and reports their
  # internal queue / signal handling as races.  Those (every frame inside the two libraries) are suppressed; a report that
  # involves the facade, the test or libcpi_amd.so still fails the pass.  The raw log is kept (gpurun_out/sanitize_tsan.log).
  printf 'race:libhsa-runtime64.so\nrace:libamdhip64.so\ncalled_from_lib:libhsa-runtime64.so\ncalled_from_lib:libamdhip64.so\nrace:librocprofiler\n' > build/san/tsan.supp
  [ -x build/san/test_threads_tsan ] && run "test_threads 4 host threads (ThreadSanitizer; HIP / HSA runtime internals suppressed)" env TSAN_OPTIONS="report_signal_unsafe=0 suppressions=$R/build/san/tsan.supp" build/san/test_threads_tsan build/san/thr.bin 4
  cp "build/san/test_threads_4_host_threads_(ThreadSanitizer;_HIP___HSA_runtime_internals_suppressed).log" gpurun_out/sanitize_tsan.log 2>/dev/null
fi
say "sanitize $PART: $([ $FAIL = 0 ] && echo PASS || echo FAIL)"
exit $FAIL
