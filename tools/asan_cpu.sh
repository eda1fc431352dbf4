#!/bin/bash
# CPU-side sanitizer pass (GPU AddressSanitizer is not available on this pool): the C oracle and the CPython list <-> buffer extension
# rebuilt with -fsanitize=address,undefined and driven by the CPU test-suite modules that use them; the product builds are restored
# afterwards.  Run from the repository root:  bash tools/asan_cpu.sh
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
restore() { make -s -C oracle -B; make -s -C bindings/fastseq -B; }
trap restore EXIT
make -s -C oracle -B CFLAGS="$SAN -fPIC -std=c11 -fopenmp -mavx2 -mfma -ffp-contract=fast -Wall -Wno-unknown-pragmas -Wno-maybe-uninitialized" || exit 1
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))"); INC=$(python3 -c "import sysconfig; print('-I' + sysconfig.get_paths()['include'])")
gcc $SAN -std=c11 -shared -fPIC -fvisibility=hidden -Wall $INC -Iinclude bindings/fastseq/gato_fastseq.c -o gato_python_amd/_gato_fastseq$EXT || exit 1
echo "instrumented: oracle $(nm -D oracle/libgato_oracle.so | grep -c __asan_) asan symbols, fastseq $(nm -D gato_python_amd/_gato_fastseq$EXT | grep -c __asan_)"
export LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_oracle.py tests/test_host_logic_cpu.py tests/test_kkt.py tests/test_capi_cpu.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
rc=${PIPESTATUS[0]}
unset LD_PRELOAD
echo "sanitizer pass exit code $rc"
exit $rc
