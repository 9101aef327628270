#!/bin/bash
# The CPU test suites against the host-sanitized build of the C-ABI (make -C pygpso_amd/csrc asan): AddressSanitizer + UBSan on
# the library's host code.  CPU ONLY -- never on the GPU box (the pool refuses sanitizer runs on the device).
# usage: tools/run_asan_cpu.sh [LOG]      (default profiles/r06_asan_cpu.log)     exit status = pytest's (86: a sanitizer report)
set -u -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$R/profiles/r06_asan_cpu.log}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
make -C $R/pygpso_amd/csrc asan -j8 > /dev/null 2>&1 || { echo "asan build failed"; exit 1; }
trap "rm -rf $R/pygpso_amd/libgpso_hip_asan.so $R/pygpso_amd/csrc/build_asan" EXIT  # (71 MB that would travel to the GPU box with every gpurun call)
cd $R
{
  echo "# host-side ASan + UBSan run of the CPU suites (tools/run_asan_cpu.sh); library: pygpso_amd/libgpso_hip_asan.so"
  echo "# runtime: $RT"
  # python itself is not instrumented: leak checking off (the interpreter's arenas), everything else fatal
  GPSO_HIP_LIB=$R/pygpso_amd/libgpso_hip_asan.so LD_PRELOAD=$RT \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=86 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_cabi_cpu.py tests/test_distributed_cpu.py tests/test_host_cpu.py -q -x -m "not gpu" -p no:cacheprovider 2>&1
  rc=$?
  echo "# exit code: $rc"
  exit $rc   # (of this brace group's subshell: the pipeline's status under pipefail)
} | tee $LOG | tail -5
exit $?
