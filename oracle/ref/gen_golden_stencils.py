#!/usr/bin/env python3
"""tests/golden/stencils.npz from the reference's own softlight.c / edge.c (oracle/_ref/{softlight,edge}.so, built
unmodified by build_ref.sh).  TEST INFRASTRUCTURE ONLY; fixtures are data (inputs + the outputs the reference
produced).  Own seed stream so that the other fixture files stay byte-identical when this one is regenerated."""
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def structured(rng, w, h, ps):
    """noise over a coarse checker so that gradients / histograms are not flat"""
    s = po.make_frame(rng, w, h, ps)
    yy, xx = np.mgrid[0:h, 0:w]
    for c in range(ps):
        s[:, c:w * ps:ps] = ((s[:, c:w * ps:ps] >> 3) + (96 * ((xx // 5 + yy // 4 + c) % 2)).astype(np.uint8) + 40).astype(np.uint8)
    return s


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    rng = np.random.default_rng(0x57E9C11)
    H = po.RefHost()
    rec, names = {}, []
    for pal in (544, 545, 522, 512, 513):
        for (w, h) in ((20, 10), (23, 9)):
            if pal in (512, 513, 522) and w & 1:
                w += 1
            for cl in (0, 1):
                cw = w >> 1 if pal in (512, 513, 522) else w
                ch = h >> 1 if pal in (512, 513) else h
                dims = [(w, h), (cw, ch), (cw, ch)] + ([(w, h)] if pal == 545 else [])
                src = [structured(rng, a, b, 1) for (a, b) in dims]
                dst = [np.full_like(a, 0x5A) for a in src]
                H.run_planar(po.refplugin("softlight"), "softlight", pal, w, h, src, dst, cl)
                key = "sl|%d|%d|%d|%d" % (pal, w, h, cl)
                for i, a in enumerate(src):
                    rec[key + "|i%d" % i] = a
                    rec[key + "|o%d" % i] = dst[i]
                names.append(key)
    for pal in (1, 2, 3, 4, 5):
        ps = 3 if pal <= 2 else 4
        for mode in (0, 1, 2):
            for inplace in (0, 1):
                w, h = (26, 14) if inplace else (31, 12)
                s = structured(rng, w, h, ps)
                d0 = s.copy() if inplace else rng.integers(0, 256, s.shape, dtype=np.uint8)
                d = d0.copy()
                H.run(po.refplugin("edge"), "edge detect", pal, w, h, [d if inplace else s], d, [po.p_int(mode)])
                key = "ed|%d|%d|%d|%d|%d" % (pal, mode, inplace, w, h)
                rec[key + "|a"] = s
                rec[key + "|d"] = d0
                rec[key + "|o"] = d
                names.append(key)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "stencils.npz"), **rec)
    # geometric transitions (multi_transitions.c): own file so that stencils.npz keeps its bytes
    tr, tnames = {}, []
    for t, fn in enumerate(("iris rectangle", "iris circle", "4 way split")):
        for pal, ps in ((1, 3), (4, 4)):
            for amt in (0.0, 0.25, 0.6, 1.0):
                w, h = 26, 11
                s1, s2 = po.make_frame(rng, w, h, ps), po.make_frame(rng, w, h, ps)
                d = np.full_like(s1, 0x5A)
                H.run(po.refplugin("multi_transitions"), fn, pal, w, h, [s1, s2], d, [po.p_double(amt)])
                key = "tr|%d|%d|%s|%d|%d" % (t, pal, amt, w, h)
                tr[key + "|a"], tr[key + "|b"], tr[key + "|o"] = s1, s2, d
                tnames.append(key)
    tr["records"] = np.array(tnames)
    np.savez_compressed(os.path.join(OUT, "transitions.npz"), **tr)
    # slide over (slide_over.c): direction through the radio parameters 1..5 as sover_init reads them (:40-51)
    so, snames = {}, []
    for dirn in (1, 2, 3, 4):
        for pal, ps in ((1, 3), (4, 4)):
            for tv in (0, 77, 200, 255):
                for mvl, mvu in ((1, 0), (0, 1), (1, 1), (0, 0)):
                    w, h = 23, 13
                    s1, s2 = po.make_frame(rng, w, h, ps), po.make_frame(rng, w, h, ps)
                    d = np.full_like(s1, 0x5A)
                    radios = [po.p_bool(False)] + [po.p_bool(dirn == k) for k in (1, 2, 3)] + [po.p_bool(False)]
                    H.run(po.refplugin("slide_over"), "slide over", pal, w, h, [s1, s2], d,
                          [po.p_int(tv)] + radios + [po.p_bool(mvl), po.p_bool(mvu)])
                    key = "so|%d|%d|%d|%d|%d|%d|%d" % (dirn, pal, tv, mvl, mvu, w, h)
                    so[key + "|a"], so[key + "|b"], so[key + "|o"] = s1, s2, d
                    snames.append(key)
    so["records"] = np.array(snames)
    np.savez_compressed(os.path.join(OUT, "slide_over.npz"), **so)
    # deinterlace (deinterlace.c): packed palettes, in place and out of place (ARGB32 in place only, see lives_oracle.c)
    de, dnames = {}, []
    for pal, ps in ((1, 3), (2, 3), (588, 3), (3, 4), (4, 4), (589, 4), (5, 4), (564, 4), (565, 4)):
        for (w, h) in ((12, 9), (13, 8), (6, 4)):
            for inplace in ((1,) if pal == 5 else (0, 1)):
                s1 = structured(rng, w, h, ps)
                if (w + 2) // 3 * 3 * ps > s1.strides[0]:
                    continue
                d = s1.copy() if inplace else np.full_like(s1, 0x5A)
                if inplace:
                    H.run(po.refplugin("deinterlace"), "deinterlace", pal, w, h, [d], d, [])
                else:
                    H.run(po.refplugin("deinterlace"), "deinterlace", pal, w, h, [s1], d, [])
                key = "de|%d|%d|%d|%d" % (pal, inplace, w, h)
                de[key + "|a"], de[key + "|o"] = s1, d
                dnames.append(key)
    de["records"] = np.array(dnames)
    np.savez_compressed(os.path.join(OUT, "deinterlace.npz"), **de)
    # triple split (layout_blends.c)
    ts, tsn = {}, []
    for pal in (1, 2):
        for (start, sym, end, vert, bw) in ((0.666667, 1, 0.333333, 0, 0.), (0.3, 0, 0.8, 0, 0.05), (0.5, 1, 0.2, 1, 0.1), (0.9, 0, 0.1, 1, 0.), (0.0, 1, 0.0, 0, 0.2),
                                            (1.0, 0, 1.0, 0, 0.03)):
            for inplace in (0, 1):
                w, h = 21, 12
                s1, s2 = po.make_frame(rng, w, h, 3), po.make_frame(rng, w, h, 3)
                d = s1.copy() if inplace else np.full_like(s1, 0x5A)
                prm = [po.p_double(start), po.p_bool(sym), po.p_bool(not sym), po.p_double(end), po.p_bool(vert), po.p_double(bw), po.p_rgb(200, 100, 50)]
                H.run(po.refplugin("layout_blends"), "triple split", pal, w, h, [d if inplace else s1, s2], d, prm)
                key = "ts|%d|%r|%d|%r|%d|%r|%d" % (pal, start, sym, end, vert, bw, inplace)
                ts[key + "|a"], ts[key + "|b"], ts[key + "|o"] = s1, s2, d
                tsn.append(key)
    ts["records"] = np.array(tsn)
    np.savez_compressed(os.path.join(OUT, "triple_split.npz"), **ts)
    # dissolve (multi_transitions.c): the instance's "random_seed" leaf is part of the record
    ds, dsn = {}, []
    for pal, ps in ((1, 3), (4, 4)):
        for k, amt in enumerate((0.0, 0.2, 0.5, 0.93, 1.0)):
            for inplace in (0, 1):
                w, h = 17, 9
                seed = 0x1234567ABCDEF + 977 * k + pal
                s1, s2 = po.make_frame(rng, w, h, ps), po.make_frame(rng, w, h, ps)
                d = s1.copy() if inplace else np.full_like(s1, 0x5A)
                H.H.refhost_set_random_seed(seed)
                H.run(po.refplugin("multi_transitions"), "dissolve", pal, w, h, [d if inplace else s1, s2], d, [po.p_double(amt)])
                H.H.refhost_set_random_seed(0)
                key = "ds|%d|%r|%d|%d" % (pal, amt, seed, inplace)
                ds[key + "|a"], ds[key + "|b"], ds[key + "|o"] = s1, s2, d
                dsn.append(key)
    ds["records"] = np.array(dsn)
    np.savez_compressed(os.path.join(OUT, "dissolve.npz"), **ds)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["dissolve.npz"] = ("reference plugin built unmodified: lives-plugins/weed-plugins/multi_transitions.c, filter dissolve; record "
                                     "ds|palette|amount|random_seed leaf of the instance|in place; 17x9; a / b sources, o result")
    man["groups"]["triple_split.npz"] = ("reference plugin built unmodified: lives-plugins/weed-plugins/layout_blends.c; record ts|palette|start|symmetrical|end|"
                                         "split horizontally|border width|in place; border colour 200,100,50; 21x12; a / b sources, o result")
    man["groups"]["deinterlace.npz"] = ("reference plugin built unmodified: lives-plugins/weed-plugins/deinterlace.c; record de|palette|in place|w|h "
                                        "(w in macropixels for 564 / 565); a source, o result (out of place: destination pre-filled with 0x5A)")
    man["groups"]["slide_over.npz"] = ("reference plugin built unmodified: lives-plugins/weed-plugins/slide_over.c; record "
                                       "so|direction(1..4 as sover_init stores it)|palette|amount|slide lower|slide upper|w|h; a / b sources, o result")
    man["groups"]["stencils.npz"] = ("reference plugins built unmodified: lives-plugins/weed-plugins/softlight.c (sl|palette|w|h|clamping(0 clamped,1 unclamped), "
                                     "planes i<k>/o<k>), edge.c (ed|palette|mode|inplace|w|h; a = source, d = destination before, o = after); one process_func call")
    man["groups"]["transitions.npz"] = ("reference plugin built unmodified: lives-plugins/weed-plugins/multi_transitions.c, filters iris rectangle (0), "
                                        "iris circle (1), 4 way split (2); record tr|type|palette|amount|w|h; a / b sources, o result (out of place)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("stencils.npz: %d records, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "stencils.npz")) // 1024))


if __name__ == "__main__":
    main()
