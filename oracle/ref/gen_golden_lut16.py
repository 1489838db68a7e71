#!/usr/bin/env python3
"""tests/golden/lut16.npz: create_gamma_lut (src/colourspace.c:738-808), each LUT built in a FRESH process (the builder
caches under the wrong key), and convert_yuv420p_to_rgb_frame (:3260-3904) with such a LUT16 fused
(xyuv2rgb_with_gamma, :2386-2390).  TEST INFRASTRUCTURE ONLY; fixtures are data."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
P = po.P


def ref_lut16(fileg, f, t):
    code = ("import ctypes,sys,numpy as np;R=ctypes.CDLL(%r);R.csref_gamma_lut16.argtypes=[ctypes.c_double,ctypes.c_int,ctypes.c_int,ctypes.c_void_p];"
            "R.csref_set_prefs.argtypes=[ctypes.c_int,ctypes.c_int,ctypes.c_double];R.csref_set_prefs(2,1,1.4);a=np.zeros(65536,np.uint16);"
            "r=R.csref_gamma_lut16(%r,%d,%d,a.ctypes.data);sys.stdout.buffer.write(bytes([r])+a.tobytes())" % (os.path.join(po.REFDIR, "libcsref.so"), fileg, f, t))
    out = subprocess.check_output([sys.executable, "-c", code])
    return out[0], np.frombuffer(out[1:], np.uint16).copy()


def main():
    assert po.have_ref()
    R = po.csref()
    R.csref_set_prefs(2, 1, 1.4)
    rng = np.random.default_rng(0x1607)
    rec = {}
    luts = []
    full = {}
    for (f, t) in ((-1, 1), (1, -1), (1, 2), (2, 1), (1, 1024), (1024, 1)):
        ok, lut = ref_lut16(1.0, f, t)
        assert ok == 1
        rec["lut16d_%d_%d" % (f, t)] = np.diff(lut.astype(np.int32), prepend=0).astype(np.int32)   # delta coded (smooth, monotone-ish): compresses 10x
        full["%d_%d" % (f, t)] = lut
        luts.append("%d_%d" % (f, t))
    rec["luts"] = np.array(luts)
    recs = []
    lut = full["-1_1"]
    for which in range(4):
        for opsize in (3, 4):
            for is422 in (0, 1):
                w, h, ys, cs = 32, 12, 32, 16
                chh = h if is422 else h // 2
                Y = rng.integers(0, 256, (h, ys), dtype=np.uint8)
                U = rng.integers(0, 256, (chh * cs + 1,), dtype=np.uint8)
                V = rng.integers(0, 256, (chh * cs + 1,), dtype=np.uint8)
                U[-1] = U[-2]
                V[-1] = V[-2]
                orow = po.align(w * opsize)
                buf = np.full((h + 2) * orow, 0xAB, np.uint8)
                strides = (ctypes.c_int * 3)(ys, cs, cs)
                R.csref_yuv420p_to_rgb(P(Y), P(U), P(V), w, h, strides, orow, ctypes.c_void_p(buf.ctypes.data + orow), int(opsize == 4),
                                       is422, which & 1, 2 if which & 2 else 1, P(lut))
                key = "w%d_h%d_t%d_o%d_q2_s%d" % (w, h, which, opsize, is422)
                rec[key + "_y"], rec[key + "_u"], rec[key + "_v"] = Y, U, V
                rec[key + "_out"] = buf.reshape(h + 2, orow)
                rec[key + "_geom"] = np.array([w, h, ys, cs, which, opsize, 2, is422, orow])
                recs.append(key)
    rec["records"] = np.array(recs)
    np.savez_compressed(os.path.join(OUT, "lut16.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["lut16.npz"] = ("src/colourspace.c:738-808 create_gamma_lut(1.0, from, to), key lut16d_<from>_<to> = first differences (cumsum restores the table), one fresh process per LUT; "
                                  ":3260-3904 convert_yuv420p_to_rgb_frame with lut16_-1_1 fused (records as in k2_yuv420p.npz, same masks)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("lut16.npz: %d luts, %d K2 records, %d KB" % (len(luts), len(recs), os.path.getsize(os.path.join(OUT, "lut16.npz")) // 1024))


if __name__ == "__main__":
    main()
