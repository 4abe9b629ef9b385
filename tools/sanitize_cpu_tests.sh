#!/bin/bash
# The host C code (library + CPython extension) under AddressSanitizer + UBSan, on the CPU test suite.
# GPU sanitizers are not available on the pool; the kernels are covered by the parity tests instead.
# Builds in a scratch copy of the tree so the product .so files stay as they are.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=${1:-/tmp/canvas_sanitize}
rm -rf "$WORK" && mkdir -p "$WORK"
tar -C "$ROOT" --exclude=.git --exclude=gpurun_out --exclude='*.so' --exclude=build --exclude=__pycache__ -cf - . | tar -C "$WORK" -xf -
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
LIBDIR=$(dirname "$(gcc -print-file-name=libasan.so)")
make -s -C "$WORK/canvas_amd/csrc" -j8 CFLAGS_EXTRA="$SAN" LDFLAGS_EXTRA="-L$LIBDIR -lasan -lubsan"
make -s -C "$WORK/canvas_amd/pyext" CFLAGS_EXTRA="$SAN" LDFLAGS_EXTRA="-L$LIBDIR -lasan -lubsan"
make -s -C "$WORK/oracle"
cd "$WORK"
# CPython leaks by design at exit; interceptors must be loaded before the interpreter
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
