#!/bin/bash
# tools/ab.sh [-r REPS] [-o] "ENV=.. ENV=.." ["ENV=.. ..." ...] -- ARGS
# One interleaved A / B / ... comparison of launch-shape switches (LGPU_* variables, lgpu_common.h's Tune table): every setting in turn, REPS rounds (default 2), so that
# clock and box drift hit all of them alike.  Default: `python bench.py --no-cpu ARGS` and its roofline.launch_us; -o: `python tools/bench_one.py ARGS` (single
# entry points by graph replay) and its lines.  "-" stands for "no variable set".  Replaces the per-experiment *_ab.sh scripts of rounds 3 and 4, e.g.
#   tools/ab.sh - LGPU_PBH_ORDER=1 LGPU_PBH_ORDER=2 -- --tracks 16          tools/ab.sh -o - LGPU_GCK_TH=12 -- --cold c4rgba fx8:c4rgba
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
reps=2; one=0
while [ "$1" = "-r" ] || [ "$1" = "-o" ]; do if [ "$1" = "-r" ]; then reps=$2; shift 2; else one=1; shift; fi; done
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
for rep in $(seq 1 $reps); do
  for s in "${sets[@]}"; do
    e=""; [ "$s" != "-" ] && e="$s"
    if [ $one = 1 ]; then env $e python tools/bench_one.py "$@" 2>/dev/null | sed "s|^|[$s] |"
    else env $e timeout 300 python bench.py --no-cpu --steps 300 --warmup 50 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('[$s]', ' '.join(sys.argv[1:]), j['roofline']['launch_us'], 'us', j['roofline']['frac'])" "$@"
    fi
  done
done
