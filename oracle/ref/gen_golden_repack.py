#!/usr/bin/env python3
"""tests/golden/yuv_repack.npz from the reference's own YUV -> YUV conversion functions (line-range slices of
src/colourspace.c compiled by build_cs_slice.py: csref_yuv_repack calls each with the arguments the dispatcher passes).
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
    rng = np.random.default_rng(0x4E9AC4)
    rec, names = {}, []
    for (ip, op, padok) in po.YUV_REPACK_PAIRS:
        for clamp_unclamped in (0, 1):
            for pad in ((0, 8) if padok else (0,)):
                w, h = 12, 6
                src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                ref = po.yuv_planes(op, w, h, fill=0x5A, pad=pad)
                got = [a.copy() for a in ref]
                sp, ss = po.planes_args(src)
                rp, rs = po.planes_args(ref)
                gp, gs = po.planes_args(got)
                # WEED_YUV_CLAMPING_CLAMPED = 0, UNCLAMPED = 1
                assert R.csref_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(rp), ctypes.addressof(rs),
                                          w, h, clamp_unclamped, 0) == 0, (ip, op)
                assert O.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(gp), ctypes.addressof(gs),
                                        w, h, clamp_unclamped, 0) == 0, (ip, op)
                key = "rp|%d|%d|%d|%d|%d|%d" % (ip, op, clamp_unclamped, pad, w, h)
                for i, (a, b) in enumerate(zip(ref, got)):
                    assert np.array_equal(a, b), "oracle differs from the reference: %s plane %d" % (key, i)
                for i, a in enumerate(src):
                    rec[key + "|i%d" % i] = a
                for i, a in enumerate(ref):
                    rec[key + "|o%d" % i] = a
                names.append(key)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "yuv_repack.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["yuv_repack.npz"] = ("slices of src/colourspace.c (:7104-7198, :7500-7753, :7800-7971, :9198-9257, :10517-10639 + K1 addpost / delpost) "
                                       "through csref_yuv_repack; record rp|in palette|out palette|clamping (0 clamped, 1 unclamped)|row padding|w|h; "
                                       "planes i<k> in, o<k> out (destination pre-filled with 0x5A: bytes the reference leaves alone keep it)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("yuv_repack.npz: %d records, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "yuv_repack.npz")) // 1024))


if __name__ == "__main__":
    main()
