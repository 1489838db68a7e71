#!/bin/bash
# k_pb_half work order within a track: bands fastest (PBH_ORDER=0, rounds 3 / 4) against column groups fastest (default), interleaved, cold buffers
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 300 --warmup 60 "${@:2}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3; do
  echo "rep $rep: 16 tracks bands-fastest $(LGPU_PBH_ORDER=0 one x) row-major $(one x) bands-fastest $(LGPU_PBH_ORDER=0 one x) row-major $(one x) | 8 tracks (4 groups) $(LGPU_PBH_ORDER=0 one x --tracks 8 --sets 4) $(one x --tracks 8 --sets 4) | 4 tracks (8 sets) $(LGPU_PBH_ORDER=0 one x --tracks 4 --sets 8) $(one x --tracks 4 --sets 8) | 1 track (32 sets) $(LGPU_PBH_ORDER=0 one x --tracks 1 --sets 32) $(one x --tracks 1 --sets 32) | blur 16 tracks $(LGPU_PBH_ORDER=0 one x --blur 1) $(one x --blur 1) | blur 1 track $(LGPU_PBH_ORDER=0 one x --blur 1 --tracks 1 --sets 32) $(one x --blur 1 --tracks 1 --sets 32)"
done
