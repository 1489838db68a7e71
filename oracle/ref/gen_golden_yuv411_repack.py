#!/usr/bin/env python3
"""tests/golden/yuv411_repack.npz from the reference's own YUV411 <-> YUV conversion functions (line-range slices of src/colourspace.c :7755-7798,
:7973-8033, :8272-8303, :8622-9196 compiled by build_cs_slice.py; csref_yuv_repack calls each with the arguments the dispatcher passes).
TEST INFRASTRUCTURE ONLY; fixtures are data.  Own seed stream.  Also checks the C restatement against every record."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    R, O = po.csref(), po.oracle()
    R.csref_set_prefs(2, 1, 1.4)
    rng = np.random.default_rng(0x411411)
    rec, names = {}, []
    for (ip, op, padok) in po.YUV411_REPACK_PAIRS:
        for clamp_unclamped in (0, 1):
            for pad in ((0, 8) if padok else (0,)):
                for (w, h) in ((16, 6), (8, 5), (4, 2)):
                    if (ip in (512, 513) or op in (512, 513)) and ip != 595 and (h & 1):
                        continue                        # a 4:2:0 source has an even height
                    src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                    ref = po.yuv_planes(op, w, h + (h & 1 if op in (512, 513) else 0), fill=0x5A, pad=0)
                    got = [a.copy() for a in ref]
                    sp, ss = po.planes_args(src)
                    rp, rs = po.planes_args(ref)
                    gp, gs = po.planes_args(got)
                    assert R.csref_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(rp), ctypes.addressof(rs),
                                              w, h, clamp_unclamped, 0) == 0, (ip, op)
                    assert O.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(gp), ctypes.addressof(gs),
                                            w, h, clamp_unclamped, 0) == 0, (ip, op)
                    key = "rp|%d|%d|%d|%d|%d|%d" % (ip, op, clamp_unclamped, pad, w, h)
                    for i, (a, b) in enumerate(zip(ref, got)):
                        assert np.array_equal(a, b), "oracle differs from the reference: %s plane %d\n%s\n%s" % (key, i, a, b)
                    for i, a in enumerate(src):
                        rec[key + "|i%d" % i] = a
                    for i, a in enumerate(ref):
                        rec[key + "|o%d" % i] = a
                    names.append(key)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "yuv411_repack.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["yuv411_repack.npz"] = ("slices of src/colourspace.c (:7755-7798, :7973-8033, :8272-8303, :8622-9196) through csref_yuv_repack; record "
                                          "rp|in palette|out palette|clamping (0 clamped, 1 unclamped)|source row padding|width in pixels|height; planes i<k> in, o<k> out "
                                          "(destination compact and pre-filled with 0x5A: bytes the reference leaves alone keep it -- most of the chroma planes of "
                                          "YUV411 -> 4:2:0, the rows YUV888 -> YUV411 never reaches)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("yuv411_repack.npz: %d records, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "yuv411_repack.npz")) // 1024))


if __name__ == "__main__":
    main()
