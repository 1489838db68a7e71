#!/usr/bin/env python3
"""tests/golden/k4_lut16.npz: RGB24 / RGBA32 / BGR24 / BGRA32 / ARGB32 -> UYVY / YUYV with the 16-bit gamma LUT inline (rgb2uyvy_with_gamma /
rgb2yuyv_with_gamma, src/colourspace.c:2146-2159, :2194-2207) through the reference's own frame functions (slice built by build_cs_slice.py,
csref_k4_lut16).  The LUTs are the reference-made ones of tests/golden/lut16.npz.  TEST INFRASTRUCTURE ONLY; fixtures are data.  Own seed stream.
Also checks the C restatement against every record."""
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests import golden_util as gu  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
P = po.P


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    R, O = po.csref(), po.oracle()
    R.csref_set_prefs(2, 1, 1.4)
    L = gu.load("lut16.npz")
    rng = np.random.default_rng(0x4B16)
    rec, names = {}, []
    for lname in ("-1_1", "1_-1", "1_2"):
        lut = np.ascontiguousarray(gu.lut16(L, lname))
        for (order, alpha) in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 1)):
            for fmt in (2, 3):
                for unc in (0, 1):
                    for (w, h, pad) in ((12, 5, 0), (10, 4, 8)):
                        ips = 4 if alpha else 3
                        src = rng.integers(0, 256, (h, po.align(w * ips) + pad), dtype=np.uint8)
                        ref = np.full((h, w * 2), 0x5A, np.uint8)        # compact destination: the reference's UYVY row step is only right there
                        got = ref.copy()
                        assert R.csref_k4_lut16(order, alpha, P(src), w, h, src.strides[0], fmt, P(ref), ref.strides[0], unc, P(lut)) == 0
                        assert O.orc_rgb_to_yuv_lut16(P(src), src.strides[0], w, h, order, alpha, P(got), got.strides[0], fmt, unc, P(lut)) == 0
                        key = "kl|%s|%d|%d|%d|%d|%d|%d|%d" % (lname, order, alpha, fmt, unc, w, h, pad)
                        assert np.array_equal(ref, got), "oracle differs from the reference: %s\n%s\n%s" % (key, ref, got)
                        rec[key + "|i"] = src
                        rec[key + "|o"] = ref
                        names.append(key)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "k4_lut16.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["k4_lut16.npz"] = ("src/colourspace.c:5129-5698 with a gamma LUT (rgb2uyvy_with_gamma :2146-2159, rgb2yuyv_with_gamma :2194-2207) through csref_k4_lut16; record "
                                     "kl|LUT of lut16.npz (from_to)|in order (0 RGB 1 BGR 2 ARGB)|in alpha|out fmt (2 UYVY 3 YUYV)|clamping (0 clamped 1 unclamped)|w|h|row padding; "
                                     "i = source frame (padded rows where the last field says so), o = destination (compact rows)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("k4_lut16.npz: %d records, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "k4_lut16.npz")) // 1024))


if __name__ == "__main__":
    main()
