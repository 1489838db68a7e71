#!/bin/bash
# k_pb_half strips: 62 storing lanes + 2 feeder lanes (LGPU_PBH_ALIGNED=0) against 64 storing lanes on 128-byte lines with one extra 4-byte load per row (=1), interleaved
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 300 --warmup 100 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3 4; do
  echo "rep $rep: 16 tracks  62-lane $(LGPU_PBH_ALIGNED=0 one)  64-lane $(LGPU_PBH_ALIGNED=1 one)  62-lane $(LGPU_PBH_ALIGNED=0 one)  64-lane $(LGPU_PBH_ALIGNED=1 one) | 1 track  62-lane $(LGPU_PBH_ALIGNED=0 one --tracks 1)  64-lane $(LGPU_PBH_ALIGNED=1 one --tracks 1)"
done
