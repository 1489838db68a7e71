#!/bin/bash
# k_pb_half (register form, band heights, 64-lane strips) against k_pb_half_ld (loader wave + LDS ring; ring depth, band height, nt), interleaved
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --steps 400 --warmup 100 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('%.2f' % j['roofline']['launch_us'])"; }
for rep in 1 2 3; do
  line="rep $rep: reg $(one)"
  line="$line | reg th=9 $(LGPU_PBH_TH=9 one) | reg th=12 $(LGPU_PBH_TH=12 one) | reg64 $(LGPU_PBH_ALIGNED=1 one) | reg64 th=12 $(LGPU_PBH_ALIGNED=1 LGPU_PBH_TH=12 one)"
  for cfg in "4 12" "20 12" "20 17" "4 17" "20 8"; do set -- $cfg; line="$line | ld=$1 th=$2 $(LGPU_PBH_LOADER=$1 LGPU_PBH_TH=$2 one)"; done
  echo "$line | reg $(one)"
done
