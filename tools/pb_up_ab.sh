#!/bin/bash
# gdk-pixbuf enlargements on 4-byte pixels: k_pb_up (register window, band heights) against k_pb_pairs (LGPU_PB_NO_UP=1), alternated in one call
cd $GRAFT_REPO_ROOT
SEL='--only=1280x720_->_1920 --only=->_2560 --only=1280x720_->_3840'
for rep in 1 2; do
  echo "== k_pb_pairs (LGPU_PB_NO_UP=1)"; LGPU_PB_NO_UP=1 python tools/bench_resize.py --pixbuf --only="1280x720 -> 1920" --only="-> 2560" --only="1280x720 -> 3840" 2>&1 | grep '^{'
  for rb in 8 16 32 64; do
    echo "== k_pb_up, $rb rows per band"; LGPU_PB_UP_RB=$rb python tools/bench_resize.py --pixbuf --only="1280x720 -> 1920" --only="-> 2560" --only="1280x720 -> 3840" 2>&1 | grep '^{'
  done
done
