#!/bin/bash
# The host library (libunfazed_io.so: BGZF / BAM / BAI / VCF / TBI readers, the staging passes -- everything that parses file bytes)
# built with AddressSanitizer + UBSan and run under the CPU tests that drive it.  CPU only (the GPU pool has no sanitizer runs).
# usage: scripts/sanitize_io.sh [pytest args]      (default: the io tests, -m "not gpu")
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${UZ_SAN_DIR:-/tmp/uz_san}
mkdir -p $OUT
SRCS=$(python3 -c "
import sys; sys.path.insert(0, '$ROOT')
from unfazed_amd import build
import os
print(' '.join(os.path.join('$ROOT', 'unfazed_amd', 'csrc', s) for s in build.IO_SRC))")
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer \
    -Wall -Wno-unused-parameter -I $ROOT/include -I $ROOT/unfazed_amd/csrc $SRCS -lz -ldl -o $OUT/libunfazed_io_san.so
cd $ROOT
TESTS=${@:-tests/test_io_native.py tests/test_io_csi.py tests/test_io_stage.py tests/test_pack_select.py tests/test_index_refdata.py tests/test_synth_files.py tests/test_refdata_plumbing.py}
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 \
UBSAN_OPTIONS=print_stacktrace=1 UZ_IO_LIB=$OUT/libunfazed_io_san.so python3 -m pytest $TESTS -x -q -s -m "not gpu" -p no:cacheprovider 2>&1 | tee $OUT/log.txt | grep -v "^    #" | tail -40
echo "sanitizer findings:"; grep -c "runtime error\|ERROR: AddressSanitizer" $OUT/log.txt || true
# the oracle (the checker every parity test trusts) the same way, on its golden vectors
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I $ROOT/include -shared -o $OUT/liboracle_san.so $ROOT/oracle/uz_oracle.c -lm
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 \
UBSAN_OPTIONS=print_stacktrace=1 UZ_ORACLE_LIB=$OUT/liboracle_san.so python3 -m pytest tests/test_oracle_golden.py -x -q -s -p no:cacheprovider 2>&1 | tee $OUT/log_oracle.txt | grep -v "^    #" | tail -5
echo "sanitizer findings (oracle):"; grep -c "runtime error\|ERROR: AddressSanitizer" $OUT/log_oracle.txt || true
