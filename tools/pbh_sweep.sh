#!/bin/bash
# timing sweep of k_pb_half: profiling variants (tools/pbh_variants.sh) x band heights, standalone single frame and the fused chain at 1 / 16 tracks
cd "$(dirname "$0")/.."
for v in ""; do
  so=""; [ -n "$v" ] && so="$PWD/lives_amd/liblivesgpu_pbh$v.so"
  for th in 3 4 5 6 8; do
    echo "variant=${v:-0} th=$th"
    LGPU_SO=$so LGPU_PBH_TH=$th timeout 120 python tools/bench_resize.py --pixbuf 2>/dev/null | head -1
    LGPU_SO=$so LGPU_PBH_TH=$th timeout 120 python bench.py --resize-backend pixbuf --no-cpu --tracks 1 --steps 300 --warmup 50 2>/dev/null | python -c "import sys,json; [print('  chain t1', json.loads(l)['roofline']['launch_us']) for l in sys.stdin if l.startswith('{')]"
    LGPU_SO=$so LGPU_PBH_TH=$th timeout 120 python bench.py --resize-backend pixbuf --no-cpu --steps 200 --warmup 50 2>/dev/null | python -c "import sys,json; [print('  chain t16', json.loads(l)['roofline']['launch_us']) for l in sys.stdin if l.startswith('{')]"
  done
done
