#!/bin/bash
# k_pb_half work order: PBH_ORDER 0 (bands fastest, XCD runs: rounds 3 / 4), 1 (column groups fastest, XCD runs), 2 (column groups fastest, bands round robin over the XCDs), interleaved, cold buffers
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 300 --warmup 60 "${@:2}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
o() { LGPU_PBH_ORDER=$1 one x "${@:2}"; }
for rep in 1 2 3; do
  echo "rep $rep: 16 tracks order 0 / 1 / 2 / 0 / 1 / 2: $(o 0) $(o 1) $(o 2) $(o 0) $(o 1) $(o 2) | 8 tracks (4 groups): $(o 0 --tracks 8 --sets 4) $(o 1 --tracks 8 --sets 4) $(o 2 --tracks 8 --sets 4) | 4 tracks (8 sets): $(o 0 --tracks 4 --sets 8) $(o 1 --tracks 4 --sets 8) $(o 2 --tracks 4 --sets 8) | 1 track (32 sets): $(o 0 --tracks 1 --sets 32) $(o 1 --tracks 1 --sets 32) $(o 2 --tracks 1 --sets 32) | blur 16 tracks: $(o 0 --blur 1) $(o 1 --blur 1) $(o 2 --blur 1) | blur 1 track: $(o 0 --blur 1 --tracks 1 --sets 32) $(o 1 --blur 1 --tracks 1 --sets 32) $(o 2 --blur 1 --tracks 1 --sets 32)"
done
