#!/usr/bin/env python3
"""tests/golden/comp.npz: compositor paint loop of the reference (paint_pixel, compositor.c:120-125, sliced into
oracle/_ref/libcompref.so by build_cs_slice.py) over pre-scaled layers.  TEST INFRASTRUCTURE ONLY; fixtures are data."""
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


def main():
    C = ctypes.CDLL(os.path.join(po.REFDIR, "libcompref.so"))
    C.compref_paint_layer.argtypes = [po.vp, po.ci, po.ci, po.ci, po.ci, po.vp, po.ci, po.ci, po.ci, po.ci, po.ci, po.cd]
    rng = np.random.default_rng(0xC0A9051)
    rec, names = {}, []
    for ps in (3, 4):
        for is_bgr in (0, 1):
            for revz in (0, 1):
                ow, oh, n = 36, 14, 4
                bg = [int(v) for v in rng.integers(0, 256, 3)]
                geo, arrs = [], []
                for z in range(n):
                    w, h = int(rng.integers(6, 40)), int(rng.integers(4, 18))
                    arrs.append(po.make_frame(rng, w, h, ps))
                    geo.append([w, h, int(rng.integers(0, 30)), int(rng.integers(0, 10))])
                alphas = [1.0, 0.5, 0.7312, 0.25]
                out = np.zeros((oh, po.align(ow * ps)), np.uint8)
                r, b = (2, 0) if is_bgr else (0, 2)
                for y in range(oh):                                   # background as compositor.c:171-178 writes it
                    for x in range(ow):
                        out[y, x * ps:x * ps + 3] = [bg[r], bg[1], bg[b]]
                        if ps == 4:
                            out[y, x * ps + 3] = 255
                for z in (range(n) if revz else range(n - 1, -1, -1)):   # z order of :181-189
                    w, h, ox, oy = geo[z]
                    C.compref_paint_layer(P(out), out.strides[0], ow, oh, ps, P(arrs[z]), arrs[z].strides[0], w, h, ox, oy, alphas[z])
                key = "cp|%d|%d|%d|%d|%d" % (ps, is_bgr, revz, ow, oh)
                for z in range(n):
                    rec[key + "|l%d" % z] = arrs[z]
                rec[key + "|geo"] = np.array(geo, np.int32)
                rec[key + "|alpha"] = np.array(alphas)
                rec[key + "|bg"] = np.array(bg, np.int32)
                rec[key + "|o"] = out
                names.append(key)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "comp.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["comp.npz"] = ("lives-plugins/weed-plugins/gdk/compositor.c:120-125 paint_pixel driven over the paint loop of :288-293 for 4 pre-scaled layers; "
                                 "background and z order (:171-189) laid out by the generator; record cp|psize|is_bgr|revz|owidth|oheight")
    json.dump(man, open(mpath, "w"), indent=1)
    print("comp.npz: %d records, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "comp.npz")) // 1024))


if __name__ == "__main__":
    main()
