#!/bin/bash
# CPU sanitizer run: builds the checker with AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle san) and runs the
# CPU test suite of the checker on it.  (GPU sanitizers are not available on the pool; the product's device code is covered by
# the parity tests instead.)   usage: bash tools/run_san.sh [pytest args]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C "$R/oracle" san
ASAN=$(gcc -print-file-name=libasan.so)
export PRS_ORACLE_LIB="$R/oracle/libproslam_oracle_san.so"
export LD_PRELOAD="$ASAN"
# python itself leaks by design at exit; the checker's allocations are what is of interest
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
cd "$R"
exec python -m pytest tests/test_oracle_stereo.py tests/test_oracle_clipper.py tests/test_oracle_finder_aligner.py tests/test_oracle_aligner_ext.py tests/test_oracle_mapping.py \
  tests/test_oracle_features.py tests/test_ref_pins.py tests/test_ref_mapping.py tests/test_ref_tracker.py -q -m "not gpu" "$@"
