#!/bin/bash
# how much of the 8-track launch's better rate per byte is the memory-side cache seeing the same buffers again: tracks x rotated buffer sets
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 300 --warmup 60 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2; do
  for tr in 4 8 16; do
    line="rep $rep tracks $tr:"
    for sets in 1 2 4 8; do
      if [ $((tr * sets)) -le 64 ]; then line="$line sets $sets $(one --tracks $tr --sets $sets) us |"; fi
    done
    echo "$line"
  done
done
