#!/bin/bash
# k_pb_half strips: 62 storing lanes + 2 feeder lanes against 64 storing lanes on 128-byte lines, interleaved, at 16 / 8 / 1 tracks per launch
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 400 --warmup 100 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3; do
  echo "rep $rep: 16 tracks: 62-lane $(LGPU_PBH_ALIGNED=0 one) 64-lane $(LGPU_PBH_ALIGNED=1 one) 62 $(LGPU_PBH_ALIGNED=0 one) 64 $(LGPU_PBH_ALIGNED=1 one) | 64-lane th=4 $(LGPU_PBH_ALIGNED=1 LGPU_PBH_TH=4 one) th=5 $(LGPU_PBH_ALIGNED=1 LGPU_PBH_TH=5 one) th=8 $(LGPU_PBH_ALIGNED=1 LGPU_PBH_TH=8 one) | 8 tracks: 62 $(LGPU_PBH_ALIGNED=0 one --tracks 8) 64 $(LGPU_PBH_ALIGNED=1 one --tracks 8) | 1 track: 62 $(LGPU_PBH_ALIGNED=0 one --tracks 1 --steps 2000) 64 $(LGPU_PBH_ALIGNED=1 one --tracks 1 --steps 2000) 64 th=4 $(LGPU_PBH_ALIGNED=1 LGPU_PBH_TH=4 one --tracks 1 --steps 2000) 62 th=4 $(LGPU_PBH_ALIGNED=0 LGPU_PBH_TH=4 one --tracks 1 --steps 2000)"
done
