#!/usr/bin/env python3
"""tests/golden/rgb_to_yuv411.npz: convert_rgb_to_yuv411_frame / _bgr_ / _argb_ (src/colourspace.c:6499-6615) run from the reference
slice on seeded frames.  TEST INFRASTRUCTURE ONLY; fixtures are data."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
# (width in pixels, height, in_order 0 RGB 1 BGR 2 ARGB, in_alpha, unclamped, extra bytes of row padding)
CASES = [(4, 2, 0, 0, 0, 0), (9, 3, 0, 1, 1, 4), (16, 4, 1, 0, 0, 8), (23, 2, 1, 1, 1, 0), (12, 5, 2, 1, 0, 0), (64, 3, 2, 1, 1, 16), (7, 1, 0, 0, 1, 3)]


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    R = po.csref()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    R.csref_rgb_to_yuv411.argtypes = [vp, ci, ci, ci, ci, ci, vp, ci]
    rec = {"cases": np.array(CASES, np.int32)}
    for n, (w, h, order, ia, uncl, pad) in enumerate(CASES):
        rng = np.random.default_rng(1140 + n)
        ips = 4 if (order == 2 or ia) else 3
        src = rng.integers(0, 256, (h, w * ips + pad), dtype=np.uint8)
        out = np.full((h, (w >> 2) * 6), 0xA5, np.uint8)
        R.csref_rgb_to_yuv411(src.ctypes.data, src.strides[0], w, h, order, ia, out.ctypes.data, 1 if uncl else 0)
        rec["src%d" % n], rec["out%d" % n] = src, out
    np.savez_compressed(os.path.join(OUT, "rgb_to_yuv411.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["rgb_to_yuv411.npz"] = "src/colourspace.c:6499-6615 through csref_rgb_to_yuv411: cases = (width, height, in_order, in_alpha, unclamped, row padding); src / out (compact macropixel rows)"
    json.dump(man, open(mpath, "w"), indent=1)
    print("rgb_to_yuv411.npz: %d cases" % len(CASES))


if __name__ == "__main__":
    main()
