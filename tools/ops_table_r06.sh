#!/bin/bash
# the per-op table of the build that ships (profiles/r06/final/ops_roofline.md): device time per launch by HIP-graph replay over COLD buffers (as many sets as make
# 1.6 GB: nothing a launch reads is still in the 256 MiB memory-side cache), algorithmic GB/s, fraction of 8 TB/s.  x1 / x8 / x16 = frames per launch.
cd $GRAFT_REPO_ROOT
echo "# per-op roofline table, round 6 ($(cat tools/_commit 2>/dev/null || echo unknown)), cold buffers, 1920 x 1080 frames unless the row says otherwise"
echo
echo '```'
echo "== batch forms of the CONVERT-step kernels and single-plane effects (lgpu_*_batch), x1 / x8 / x16"
for k in swz swz34 gamma premult mirror letterbox colorkey r2y420 r2uyvy r2y888 y8882rgb uyvy2rgb; do python tools/bench_one.py --cold b1:$k b8:$k b16:$k 2>/dev/null; done
echo "== lgpu_fx_batch (two-input effects, softlight, YUV411, config 4), x1 / x8 / x16"
python tools/bench_one.py --cold fx1:chroma fx8:chroma fx16:chroma fx1:luma fx8:luma fx16:luma fx1:multi fx8:multi fx16:multi fx1:transition fx8:transition fx16:transition 2>/dev/null
echo "   (in place: out channel = in channel 0, as weed_apply_instance sets the channels up for CAN_DO_INPLACE classes; out of place the 4-byte blends read the destination as well -- its alpha byte is never written)"
python tools/bench_one.py --cold fx1:chroma_ip fx8:chroma_ip fx16:chroma_ip fx8:luma_ip fx16:luma_ip 2>/dev/null
python tools/bench_one.py --cold softlight fx8:softlight fx16:softlight yuv411 fx8:yuv411 fx16:yuv411 c4rgba fx8:c4rgba c4rgb24 fx8:c4rgb24 2>/dev/null
echo "== K2 (lgpu_yuv420p_to_rgb[_batch]) and the chain (lgpu_chain, 4K tracks)"
python tools/bench_one.py --cold k2 k2b8 k2b16 chain1 chain8 chain16 c3 2>/dev/null
echo "   (with config 5's gaussian; chainbluropN: sources whose alpha is 255 everywhere + LGPU_INTERP_OPAQUE)"
python tools/bench_one.py --cold chainblur1 chainblur4 chainblur8 chainblur16 chainblurop1 chainblurop4 chainblurop8 chainblurop16 2>/dev/null
echo "== the chain off the headline shape (lgpu_chain / lgpu_chain_canvas / lgpu_chain_amounts: one launch each): other ratios, a letterbox canvas, nb = no layer 2 (LGPU_INTERP_NOBLEND)"
python tools/bench_one.py --cold chg1:1920x1080:1280x720 chg16:1920x1080:1280x720 chg16:1280x720:1920x1080 chg8:3840x2160:1706x960 chg16:3840x2160:1280x720 chg16:1920x1080:1280x540:1280x720 \
       chg1:3840x2160:1920x1080:nb chg16:3840x2160:1920x1080:nb chg1:1920x1080:1280x720:nb chg16:1920x1080:1280x720:nb chg1:1920x1080:1280x720:1280x800:nb 2>/dev/null
echo "== gdk-pixbuf scaler (lgpu_pixbuf_scale[_batch])"
for r in 3840x2160:1920x1080 1920x1080:1280x720 1280x720:1920x1080 3840x2160:1706x960 1920x1080:2560x1440 1280x720:3840x2160; do python tools/bench_one.py --cold pb:$r:3 pb8:$r:3 pb16:$r:3 2>/dev/null; done
echo "   (BILINEAR, and sources whose alpha is 255 everywhere with LGPU_INTERP_OPAQUE: 3o)"
for r in 1920x1080:1280x720 1280x720:1920x1080 3840x2160:1706x960 1280x720:3840x2160; do python tools/bench_one.py --cold pb16:$r:2 pb:$r:3o pb16:$r:3o 2>/dev/null; done
echo "== single-frame entry points (1080p; a plain copy of one 1080p frame is copy4: what one launch of this size can reach)"
python tools/bench_one.py --cold copy1 copy4 2>/dev/null
for c in s:lb s:gauss5 s:resize s:deint s:tsplit s:dissolve s:slide s:transition s:chroma s:luma s:multi s:colorkey premult_yuva s:clamp s:r2y411 s:r2y444p s:y444p2rgb softlight yuv411 composite \
         s:repack:512:564 s:repack:522:564 s:repack:564:512 s:repack:512:588 s:repack:512:522 s:repack:564:565 s:repack:564:588 s:repack:588:564 s:repack:588:512 s:repack:588:522 s:repack:564:544 s:repack:544:588 s:repack:588:544 s:repack:595:512 s:repack:595:522; do
  python tools/bench_one.py --cold $c 2>&1 | tail -1
done
echo '```'
