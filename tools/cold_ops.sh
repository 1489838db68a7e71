#!/bin/bash
# single-frame op timings by graph replay: buffers inside the memory-side cache (six sets), cold (1.6 GB of sets), and both again with the launches alternating between two streams
cd $GRAFT_REPO_ROOT
OPS="copy1 copy4 chain1 chain8 chain16 c3 c4rgba c4rgb24 k2 premult_yuva yuv411 composite softlight pb:3840x2160:1920x1080:3 pb:1920x1080:1280x720:3 pb:1280x720:1920x1080:3 pb:3840x2160:1706x960:3 pb:1920x1080:2560x1440:3 pb:1280x720:3840x2160:3"
# the batch entry points (lgpu_fx_batch, lgpu_pixbuf_scale_batch): N frames of one geometry per launch
BATCH="fx8:c4rgba fx8:c4rgb24 fx8:softlight fx16:softlight fx8:yuv411 fx16:yuv411 fx1:transition fx8:transition fx16:transition fx1:chroma fx8:chroma fx16:chroma fx8:luma fx8:multi pb8:3840x2160:1920x1080:3 pb8:3840x2160:1706x960:3 pb16:3840x2160:1706x960:3 pb8:1920x1080:1280x720:3 pb16:1920x1080:1280x720:3 pb8:1280x720:1920x1080:3 pb16:1280x720:1920x1080:3 pb8:1920x1080:2560x1440:3 pb8:1280x720:3840x2160:3"
echo "== six buffer sets"; python tools/bench_one.py $OPS 2>/dev/null
echo "== cold"; python tools/bench_one.py --cold $OPS 2>/dev/null
echo "== cold, batch entry points (x8 / x16 frames per launch)"; python tools/bench_one.py --cold $BATCH 2>/dev/null
echo "== six buffer sets, two streams"; python tools/bench_one.py --two-streams $OPS 2>/dev/null
echo "== cold, two streams"; python tools/bench_one.py --cold --two-streams $OPS 2>/dev/null
