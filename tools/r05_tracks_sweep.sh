#!/bin/bash
# round 5: launch duration against the number of tracks per launch (slope = steady-state cost of a track, intercept = what a launch pays once), chain with and without the gaussian
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
run() { env $1 timeout 200 python bench.py --no-cpu $2 --steps 200 --warmup 40 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', '$2', j['roofline']['launch_us'])"
}
for rep in 1 2; do
  for n in 1 2 4 8 12 16 24 32; do
    run "X=0" "--blur 1 --tracks $n"
    run "LGPU_PBH_TH=100096" "--blur 1 --tracks $n"
    run "LGPU_PBH_TH=100024" "--blur 1 --tracks $n"
    run "X=0" "--tracks $n"
  done
done > $O/tracks_sweep.txt
cat $O/tracks_sweep.txt
