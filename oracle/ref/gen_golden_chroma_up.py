#!/usr/bin/env python3
"""tests/golden/chroma_up.npz from the reference's convert_quad_chroma_packed / convert_double_chroma_packed (src/colourspace.c:10715-10873, sliced by
build_cs_slice.py): YUV420P / YUV422P -> YUV888 / YUVA8888 with the arguments the dispatcher passes (:13624-13635, :13731-13742).
TEST INFRASTRUCTURE ONLY; fixtures are data.  Own seed stream.  Also checks the C restatement against every record.
mask = 0 marks the bytes that come from the reference's read one sample past a COMPACT chroma plane (undefined there)."""
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
    rng = np.random.default_rng(0xC40A)
    rec, names = {}, []
    for (ip, op) in po.CHROMA_UP_PAIRS:
        for unc in (0, 1):
            for sampling in (0, 1):
                for pad in (0, 8):
                    for (w, h) in ((12, 6), (16, 4), (4, 2)):
                        src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                        ref = po.yuv_planes(op, w, h, fill=0x5A, pad=pad)
                        got = [a.copy() for a in ref]
                        sp, ss = po.planes_args(src)
                        rp, rs = po.planes_args(ref)
                        gp, gs = po.planes_args(got)
                        assert R.csref_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(rp), ctypes.addressof(rs), w, h, unc, sampling) == 0
                        assert O.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(gp), ctypes.addressof(gs), w, h, unc, sampling) == 0
                        mask = po.chroma_up_mask(ip, op, w, h, pad, ref[0].shape)
                        key = "cu|%d|%d|%d|%d|%d|%d|%d" % (ip, op, unc, sampling, pad, w, h)
                        assert np.array_equal(ref[0] * mask, got[0] * mask), "oracle differs from the reference: %s\n%s\n%s" % (key, ref[0], got[0])
                        for i, a in enumerate(src):
                            rec[key + "|i%d" % i] = a
                        rec[key + "|o0"] = ref[0] * mask
                        rec[key + "|m"] = mask
                        names.append(key)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "chroma_up.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["chroma_up.npz"] = ("slices of src/colourspace.c (:10715-10873) through csref_yuv_repack; record cu|in palette|out palette|clamping (0 clamped, 1 unclamped)|"
                                      "YUV_sampling of the source (0 JPEG / default, 1 MPEG)|row padding|w|h; planes i<k> in, o0 out (destination pre-filled with 0x5A: bytes the "
                                      "reference leaves alone keep it -- the last odd row's chroma, the alpha bytes it skips), m = 0 where the reference reads one sample past a "
                                      "compact chroma plane (those bytes are zeroed in o0)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("chroma_up.npz: %d records, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "chroma_up.npz")) // 1024))


if __name__ == "__main__":
    main()
