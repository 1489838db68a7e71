#!/usr/bin/env python3
"""tests/golden/k34_palette.npz from the reference's own RGB <-> YUV conversion functions (line-range slices of
src/colourspace.c compiled by build_cs_slice.py: csref_k4 / csref_k3).  TEST INFRASTRUCTURE ONLY; fixtures are data.
Own seed stream.  prefs: pb_quality = PB_QUALITY_MED, nfx_threads = 1.  Destination strides are compact: the reference's
4:2:0 / UYVY row arithmetic only works there (see oracle/lives_oracle.h)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
P = po.P


def k4_cases():
    for in_order in (0, 1, 2):
        for in_alpha in ((1,) if in_order == 2 else (0, 1)):
            for out_fmt in range(6):
                if out_fmt >= 4 and in_order == 2:
                    continue            # reference reads the wrong bytes (src/colourspace.c:6353): no fixture
                for out_alpha in ((0, 1) if out_fmt <= 1 else (0,)):
                    for which in ((0, 1, 2, 3) if out_fmt >= 4 else (0, 1)):
                        yield in_order, in_alpha, out_fmt, out_alpha, which


def k3_cases():
    for in_fmt in range(4):
        for in_alpha in ((0, 1) if in_fmt <= 1 else (0,)):
            for out_order in (0, 1, 2):
                for out_alpha in ((1,) if out_order == 2 else (0, 1)):
                    if in_fmt == 1 and (out_order == 2 or (out_order == 1 and not out_alpha)):
                        continue        # reference row arithmetic broken (:7475-7476, :7313): no fixture
                    for which in ((0, 1, 2, 3) if in_fmt == 0 else (0, 1)):
                        yield in_fmt, in_alpha, out_order, out_alpha, which


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    R = po.csref()
    R.csref_set_prefs(2, 1, 1.4)
    rng = np.random.default_rng(0x9A1E77E)
    rec, names = {}, []
    for (in_order, in_alpha, out_fmt, out_alpha, which) in k4_cases():
        w, h = (21, 6) if (out_fmt <= 1 and (which & 1)) else (20, 6)
        ips = 4 if (in_order == 2 or in_alpha) else 3
        src = po.make_frame(rng, w, h, ips)
        src[0, :2 * ips] = [0] * ips + [255] * ips             # extremes: black and white pixel
        out, _ = po.k4_out_planes(0x5A, w, h, out_fmt, out_alpha)
        op, os_ = po.planes_args(out)
        assert R.csref_k4(in_order, in_alpha, out_fmt, out_alpha, P(src), src.strides[0], w, h, ctypes.addressof(op), ctypes.addressof(os_),
                          which & 1, which >> 1) == 0
        key = "k4|%d|%d|%d|%d|%d|%d|%d" % (in_order, in_alpha, out_fmt, out_alpha, which, w, h)
        rec[key + "|in"] = src
        for i, a in enumerate(out):
            rec[key + "|o%d" % i] = a
        names.append(key)
    for (in_fmt, in_alpha, out_order, out_alpha, which) in k3_cases():
        w, h = (21, 5) if (in_fmt <= 1 and (which & 1)) else (20, 5)
        if in_fmt == 0:
            planes = [po.make_frame(rng, w, h, 4 if in_alpha else 3)]
        elif in_fmt == 1:
            planes = [po.make_frame(rng, w, h, 1, stride_align=16) for _ in range(4 if in_alpha else 3)]
        else:
            planes = [po.make_frame(rng, w, h, 2, stride_align=4)]
        ops = 4 if (out_order == 2 or out_alpha) else 3
        out = np.full((h, po.align(w * ops)), 0x5A, np.uint8)
        sp, ss = po.planes_args(planes)
        assert R.csref_k3(in_fmt, in_alpha, out_order, out_alpha, ctypes.addressof(sp), ctypes.addressof(ss), w, h, P(out), out.strides[0],
                          which & 1, which >> 1) == 0
        key = "k3|%d|%d|%d|%d|%d|%d|%d" % (in_fmt, in_alpha, out_order, out_alpha, which, w, h)
        for i, a in enumerate(planes):
            rec[key + "|i%d" % i] = a
        rec[key + "|out"] = out
        names.append(key)
    t = [np.zeros(256, np.uint8) for _ in range(4)]                     # K5 tables (init_YUV_to_YUV_tables, :1108-1139)
    R.csref_yuv_yuv_tables(*[P(x) for x in t])
    np.savez_compressed(os.path.join(OUT, "yuvyuv.npz"), yc2u=t[0], uvc2u=t[1], yu2c=t[2], uvu2c=t[3])
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "k34_palette.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["k34_palette.npz"] = ("src/colourspace.c:5129-6440 (RGB/BGR/ARGB -> YUV888, YUVA8888, YUV(A)444(4)P, UYVY, YUYV, YUV420P, YUV422P; record "
                                        "k4|in_order|in_alpha|out_fmt|out_alpha|which|w|h) and :2750-3258, :6616-7102, :7200-7498 (the reverse; record "
                                        "k3|in_fmt|in_alpha|out_order|out_alpha|which|w|h); which: bit0 unclamped, bit1 BT.709; nfx_threads = 1; compact destination strides")
    man["groups"]["yuvyuv.npz"] = "src/colourspace.c:1108-1139 init_YUV_to_YUV_tables: Yclamped_to_Yunclamped, UVclamped_to_UVunclamped, Yunclamped_to_Yclamped, UVunclamped_to_UVclamped"
    json.dump(man, open(mpath, "w"), indent=1)
    print("k34_palette.npz: %d records, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "k34_palette.npz")) // 1024))


if __name__ == "__main__":
    main()
