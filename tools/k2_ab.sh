#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do python tools/bench_one.py k2 2>/dev/null | awk '{printf "%s %s | ", $1, $2}'; python tools/bench_one.py --cold k2 2>/dev/null | awk '{printf "cold %s | ", $2}'; python tools/bench_one.py --two-streams k2 2>/dev/null | awk '{printf "two streams %s\n", $2}'; done
python tools/bench_ops.py 2>/dev/null | grep -i "yuv420\|k2" | head -5
