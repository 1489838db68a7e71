#!/bin/bash
# config 4 / config 5 kernels after an arithmetic change: graph-replay timings + the blur legs of the bench
cd $GRAFT_REPO_ROOT
for rep in 1 2; do python tools/bench_one.py c4rgba c4rgb24 c3 chain1 2>/dev/null | awk '{printf "%s %s | ", $1, $2}'; echo; done
python bench.py --no-cpu --blur 1 --steps 200 --warmup 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('blur 16 tracks', j['roofline']['launch_us'], j['value'])"
python bench.py --no-cpu --blur 1 --tracks 1 --sets 32 --steps 400 --warmup 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('blur 1 track', j['roofline']['launch_us'], j['value'])"
