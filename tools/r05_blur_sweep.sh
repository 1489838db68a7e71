#!/bin/bash
# round 5: the blur instantiation of k_pb_half under band counts / work orders / workgroups per CU (LGPU_PBH_TH = 100000 + bands, LGPU_PBH_ORDER, LGPU_PBH_OCC)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
run() { # env-string, bench args
  env $1 timeout 200 python bench.py --no-cpu $2 --steps 300 --warmup 50 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', '$2', j['roofline']['launch_us'])"
}
for rep in 1 2; do
  for b in 48 40 32 24 16; do
    for o in 1 2; do
      run "LGPU_PBH_TH=$((100000+b)) LGPU_PBH_ORDER=$o" "--blur 1"
      run "LGPU_PBH_TH=$((100000+b)) LGPU_PBH_ORDER=$o" "--blur 1 --tracks 8"
    done
  done
  for occ in 4 5 6; do
    run "LGPU_PBH_TH=100024 LGPU_PBH_OCC=$occ" "--blur 1"
    run "LGPU_PBH_TH=100048 LGPU_PBH_OCC=$occ" "--blur 1"
  done
done > $O/blur_band_order_sweep.txt
cat $O/blur_band_order_sweep.txt
