#!/bin/bash
# k_pb_half band heights: balanced (whole generations of workgroups, the default) against uniform forced heights, interleaved
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 400 --warmup 100 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3; do
  echo "rep $rep: 16 tracks: auto $(one) th6 $(LGPU_PBH_TH=6 one) th5 $(LGPU_PBH_TH=5 one) th7 $(LGPU_PBH_TH=7 one) auto $(one) | 8 tracks: auto $(one --tracks 8) th6 $(LGPU_PBH_TH=6 one --tracks 8) th5 $(LGPU_PBH_TH=5 one --tracks 8) | 4 tracks: auto $(one --tracks 4) th6 $(LGPU_PBH_TH=6 one --tracks 4) | 1 track: auto $(one --tracks 1 --steps 2000)"
done
