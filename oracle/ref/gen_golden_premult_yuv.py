#!/usr/bin/env python3
"""tests/golden/premult_yuv.npz: the four clamped-YUV premultiply tables of init_unal (src/colourspace.c:1141-1160) from the reference slice
(csref_unal_yuv), as the bytes alpha_premult stores.  TEST INFRASTRUCTURE ONLY; fixtures are data.  Also checks the C restatement."""
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
    ref = [np.zeros(65536, np.int32) for _ in range(4)]
    R.csref_unal_yuv.argtypes = [ctypes.c_void_p] * 4
    R.csref_unal_yuv(*[a.ctypes.data for a in ref])          # unalcy, alcy, unalcuv, alcuv
    got = [np.zeros(65536, np.uint8) for _ in range(4)]
    O.orc_premult_yuv_tables(*[a.ctypes.data for a in got])
    names = ("unalcy", "alcy", "unalcuv", "alcuv")
    rec = {}
    for n, a, b in zip(names, ref, got):
        assert a.min() >= 0 and a.max() <= 255 and np.array_equal(a.astype(np.uint8), b), "oracle differs from the reference: " + n
        rec[n] = a.astype(np.uint8).reshape(256, 256)
    np.savez_compressed(os.path.join(OUT, "premult_yuv.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["premult_yuv.npz"] = "src/colourspace.c:1141-1160 (init_unal) through csref_unal_yuv: unalcy / alcy / unalcuv / alcuv, [alpha][value], as bytes"
    json.dump(man, open(mpath, "w"), indent=1)
    print("premult_yuv.npz: %d KB" % (os.path.getsize(os.path.join(OUT, "premult_yuv.npz")) // 1024))


if __name__ == "__main__":
    main()
