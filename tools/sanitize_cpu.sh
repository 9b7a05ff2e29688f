#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer pass over the CPU-side native code: the host library (grids, background,
# thermodynamics, non-cold species, the C++ shim classes), the C++ shim demo's host side and the CPU oracle, driven by the CPU test suite.
# (GPU sanitizers are not available on the pool; the HIP library itself is loaded uninstrumented.)
#   bash tools/sanitize_cpu.sh            -> profiles/r03_sanitizers_cpu.txt
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/profiles/r03_sanitizers_cpu.txt}
B=/tmp/cpt_san
mkdir -p $B
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g -O1"
HOST=$ROOT/classpp_public_amd/host
g++ $SAN -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $B/libcpt_host_san.so $HOST/cpt_grids.cpp $HOST/cpt_cosmo.cpp $HOST/cpt_ncdm.cpp \
    $HOST/cpt_modules.cpp -L$ROOT/classpp_public_amd/csrc -lcpt -Wl,-rpath,$ROOT/classpp_public_amd/csrc
g++ $SAN -std=c++17 -fPIC -shared -o $B/libcpt_oracle_san.so $ROOT/oracle/restate/*.cpp $ROOT/oracle/restate/host/*.cpp -lm -lpthread
# the shim demo (tests/host_shim_demo.cpp) against the sanitized host library: constructing the modules without a GPU must throw cleanly
g++ $SAN -std=c++17 -I$ROOT/include -o $B/host_shim_demo_san $ROOT/tests/host_shim_demo.cpp -L$B -l:libcpt_host_san.so -L$ROOT/classpp_public_amd/csrc -lcpt \
    -Wl,-rpath,$B -Wl,-rpath,$ROOT/classpp_public_amd/csrc
{
  echo "# ASan + UBSan, CPU build: $(g++ --version | head -1); $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "# flags: $SAN"
  echo "# libraries: libcpt_host (cpt_grids, cpt_cosmo, cpt_ncdm, cpt_modules), libcpt_oracle (oracle/restate), host_shim_demo"
  export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  export CPT_HOST_LIB=$B/libcpt_host_san.so CPT_ORACLE_LIB=$B/libcpt_oracle_san.so
  cd $ROOT
  echo "## pytest -m 'not gpu' (host library + oracle instrumented)"
  python -m pytest tests -q -x -m "not gpu" -p no:cacheprovider 2>&1 | tail -5
  echo "## host_shim_demo without a GPU: the module constructors must fail cleanly (no device), not crash"
  set +e
  python - <<PY
import sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
from classpp_public_amd.inputs import Inputs
from test_gpu_host_shim import write_inputs
write_inputs(Inputs("small"), "$B/in.bin")
PY
  ASAN_OPTIONS=detect_leaks=1:halt_on_error=1 $B/host_shim_demo_san $B/in.bin $B/out.bin 0 2>&1 | tail -6
  echo "exit code $? (the demo maps std::runtime_error to 11: the constructor threw as it must without a device and released what it had built - leak detection on; a sanitizer report would show above)"
} 2>&1 | tee $OUT
