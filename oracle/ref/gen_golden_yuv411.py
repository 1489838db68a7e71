#!/usr/bin/env python3
"""tests/golden/yuv411.npz: convert_yuv411_to_rgb_frame / _bgr_frame / _argb_frame (src/colourspace.c:8305-8620) run from the
reference slice on seeded frames.  TEST INFRASTRUCTURE ONLY; fixtures are data (inputs, the destination's initial bytes, outputs)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
# (width in macropixels, height, out_order 0 RGB 1 BGR 2 ARGB, out_alpha, unclamped, extra bytes of row padding)
CASES = [(1, 3, 0, 0, 0, 0), (2, 2, 0, 1, 0, 8), (5, 4, 1, 0, 1, 0), (7, 3, 1, 1, 0, 4), (9, 5, 2, 1, 1, 0), (16, 6, 0, 1, 1, 12), (3, 2, 2, 1, 0, 8),
         (33, 4, 0, 0, 1, 3)]


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    R = po.csref()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    R.csref_yuv411_to_rgb.argtypes = [vp, ci, ci, vp, ci, ci, ci, ci]
    rec = {"cases": np.array(CASES, np.int32)}
    for n, (wm, h, order, oa, uncl, pad) in enumerate(CASES):
        rng = np.random.default_rng(4110 + n)
        ps = 4 if (order == 2 or oa) else 3
        src = rng.integers(0, 256, (h, wm * 6), dtype=np.uint8)
        init = rng.integers(0, 256, (h, wm * 4 * ps + pad), dtype=np.uint8)
        out = init.copy()
        R.csref_yuv411_to_rgb(src.ctypes.data, wm, h, out.ctypes.data, out.strides[0], order, oa, 1 if uncl else 0)
        rec["src%d" % n], rec["init%d" % n], rec["out%d" % n] = src, init, out
    np.savez_compressed(os.path.join(OUT, "yuv411.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["yuv411.npz"] = "src/colourspace.c:8305-8620 through csref_yuv411_to_rgb: cases = (width_mp, height, out_order, out_alpha, unclamped, row padding); src / init (destination before) / out"
    json.dump(man, open(mpath, "w"), indent=1)
    print("yuv411.npz: %d cases" % len(CASES))


if __name__ == "__main__":
    main()
