#!/bin/bash
# gdk-pixbuf integer ratios other than 2:1 / 1:2 on 4-byte pixels: k_pb_gather (default there) against k_pb_pairs (LGPU_PB_NO_GATHER=1), alternated in one call
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== k_pb_pairs (LGPU_PB_NO_GATHER=1)"; LGPU_PB_NO_GATHER=1 python tools/bench_resize.py --pixbuf --only="-> 1280x720" --only=1706 --only="-> 1920x1080 HYPER rgba" --only="1280x720 -> 1920" --only="-> 2560" 2>&1 | grep '^{'
  echo "== k_pb_gather, enlargements too (LGPU_PB_GATHER_ALL=1)"; LGPU_PB_GATHER_ALL=1 python tools/bench_resize.py --pixbuf --only="1280x720 -> 1920" --only="-> 2560" 2>&1 | grep '^{'
  echo "== k_pb_gather"; python tools/bench_resize.py --pixbuf --only="-> 1280x720" --only=1706 --only="-> 1920x1080 HYPER rgba" --only="1280x720 -> 1920" --only="-> 2560" 2>&1 | grep '^{'
done
