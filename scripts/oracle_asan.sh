#!/bin/bash
# The oracle's C restatement under AddressSanitizer + UBSan: the known-answer suite and the NumPy<->C agreement tests.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
ORC_LIB_SUFFIX=_asan LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest tests/test_oracle_known_answers.py tests/test_oracle_independent.py -x -q -p no:cacheprovider 2>&1 | tail -15
