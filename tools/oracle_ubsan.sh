#!/bin/bash
# tools/oracle_ubsan.sh -- the CPU restatement (oracle/lives_oracle.c) under gcc -fsanitize=undefined (-fno-sanitize-recover: the first report aborts
# the test run) over the CPU suites that exercise it: the golden fixtures and, where oracle/_ref is built, the live reference builds.  TEST TOOL.
set -e
cd "$(dirname "$0")/.."
export LGPU_ORACLE_UBSAN=1
export LD_PRELOAD="$(gcc -print-file-name=libubsan.so)${LD_PRELOAD:+:$LD_PRELOAD}"
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_oracle_golden.py tests/test_oracle_cpu.py -x -q "$@"
