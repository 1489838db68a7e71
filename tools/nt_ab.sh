#!/bin/bash
# k_pb_half: non-temporal loads for a band's inner source rows (LGPU_PBH_NT_IN=1) against plain loads, interleaved, by tracks per launch
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 600 --warmup 100 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2; do
  line="rep $rep:"
  for t in 2 4 6 8 10 12 16; do line="$line | $t tracks: plain $(one --tracks $t) nt $(LGPU_PBH_NT_IN=1 one --tracks $t) plain $(one --tracks $t) nt $(LGPU_PBH_NT_IN=1 one --tracks $t)"; done
  echo "$line"
done
