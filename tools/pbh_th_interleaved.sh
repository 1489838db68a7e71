#!/bin/bash
# k_pb_half band height on the full-device launch (16 tracks), interleaved and repeated inside one call (a box drifts by 3-6 % over a minute: sequential sweeps mislead)
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 300 --warmup 100 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3 4; do
  line="rep $rep:"
  for th in 6 8 10 12 16 24 6; do line="$line th=$th $(LGPU_PBH_TH=$th one)"; done
  echo "$line"
done
