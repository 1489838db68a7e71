#!/bin/bash
# non-temporal inner rows on COLD buffers (the legs rotate through 32 / 64 frames of every kind: 1.6 / 3.2 GB): tracks per launch x LGPU_PBH_NT_IN, interleaved
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 300 --warmup 60 "${@:2}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3; do
  for cfg in "2 16" "4 8" "6 5" "8 4" "12 3" "16 2"; do
    set -- $cfg
    echo "rep $rep tracks $1 sets $2: plain $(LGPU_PBH_NT_IN=0 one x --tracks $1 --sets $2) nt $(LGPU_PBH_NT_IN=1 one x --tracks $1 --sets $2) plain $(LGPU_PBH_NT_IN=0 one x --tracks $1 --sets $2) nt $(LGPU_PBH_NT_IN=1 one x --tracks $1 --sets $2)"
  done
done
