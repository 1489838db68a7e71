#!/bin/bash
# the gdk-pixbuf ratios off 2:1 by graph replay: tools/pb_ratios.sh [env assignments...]
cd $GRAFT_REPO_ROOT
R="pb:3840x2160:1706x960:3 pb:1920x1080:1280x720:3 pb:2560x1440:1920x1080:3 pb:3840x2160:1706x960:2 pb:3840x2160:2560x1440:3 pb:1280x720:1920x1080:3 pb:1920x1080:2560x1440:3 pb:1280x720:3840x2160:3 pb:3840x2160:1280x720:3"
env "$@" python tools/bench_one.py $R 2>/dev/null | awk '{printf "%s %s | ", $1, $2}'; echo
