#!/bin/bash
# counters of the single-frame kernels (softlight, edge, YUV411 -> RGBA, YUVA premultiply, composite, K2, C3, C4) and of the gdk-pixbuf ratios off 2:1: tools/pmc_ops.sh <case> ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/pmc
run() { name=$1; pat=$2; shift 2; bash tools/pmc_case.sh gpurun_out/r04/pmc/$name "$pat" python tools/prof_one.py "$@" > gpurun_out/r04/pmc_$name.md 2>&1; rm -rf gpurun_out/r04/pmc/$name; }
for c in "$@"; do
  case $c in
    softlight) run softlight k_softlight softlight;;
    edge) run edge k_edge edge;;
    yuv411) run yuv411 k_yuv411_to_rgb yuv411;;
    premult_yuva) run premult_yuva k_premult_yuva premult_yuva;;
    composite) run composite k_composite composite;;
    k2) run k2 k_yuv420p_to_rgb k2;;
    c3) run c3 k_pb_half c3;;
    c4rgb24) run c4rgb24 k_gauss5_colorkey c4rgb24;;
    pb1) run pb_1080p_to_720p k_pb_ pb:1920x1080:1280x720:3;;
    pb2) run pb_720p_to_1080p k_pb_ pb:1280x720:1920x1080:3;;
    pb3) run pb_4k_to_1706x960 k_pb_ pb:3840x2160:1706x960:3;;
    pb4) run pb_1080p_to_1440p k_pb_ pb:1920x1080:2560x1440:3;;
    pb5) run pb_720p_to_4k k_pb_ pb:1280x720:3840x2160:3;;
  esac
done
