#!/bin/bash
# one 4K frame per launch: k_pb_half with every load of a band up front (DEEP, the default for launches that leave the device part empty) against the two-rows-per-trip loop
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 3000 --warmup 300 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3; do
  echo "rep $rep: 1 track: deep $(one --tracks 1) loop $(LGPU_PBH_DEEP=0 one --tracks 1) deep th5 $(LGPU_PBH_TH=5 one --tracks 1) deep th4 $(LGPU_PBH_TH=4 one --tracks 1) deep th3 $(LGPU_PBH_TH=3 one --tracks 1) deep 62-lane $(LGPU_PBH_ALIGNED=0 one --tracks 1) | blur 1 track $(one --tracks 1 --blur 1) | 2 tracks: auto $(one --tracks 2) forced deep $(LGPU_PBH_DEEP=1 one --tracks 2)"
done
