#!/usr/bin/env python3
"""tools/fuzz_seam.py [sequences] [seed] -- randomized differential run over the weed_layer_t seam: a random frame in a random palette goes through a random
sequence of seam calls (palette conversions, gamma, premultiply, resize, letterbox, unletterbox, compact_rowstrides, clear) three times -- as an ordinary layer
(every call uploads and downloads), as a pinned layer on one thread (resident planes, one download at the end), and as a pinned layer with every call made
from a thread of its own (hand-over events) -- and the three must agree in every return value, every leaf and every byte.  Needs oracle/_ref/libweedall.so."""
import ctypes
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RGB24, BGR24, RGBA32, BGRA32, ARGB32, YUV420P, YVU420P, YUV422P, YUV444P, YUVA4444P, UYVY, YUYV, YUV888, YUVA8888, YUV411 = 1, 2, 3, 4, 5, 512, 513, 522, 544, 545, 564, 565, 588, 589, 595
PALS = [RGB24, BGR24, RGBA32, BGRA32, ARGB32, YUV420P, YVU420P, YUV422P, YUV444P, YUVA4444P, UYVY, YUYV, YUV888, YUVA8888, YUV411]
KEYS = ("current_palette", "width", "height", "YUV_clamping", "YUV_subspace", "YUV_sampling", "gamma_type", "host_flags")


def align(n, a=32):
    return (n + a - 1) // a * a


def planes_for(rng, pal, w, h):
    def plane(pw, ph, lo=0, hi=256):
        a = np.zeros((ph, align(pw)), np.uint8)
        a[:, :pw] = rng.integers(lo, hi, (ph, pw), dtype=np.uint8)
        return a
    if pal in (RGB24, BGR24, YUV888):
        return [plane(w * 3, h)], w
    if pal in (RGBA32, BGRA32, ARGB32, YUVA8888):
        return [plane(w * 4, h)], w
    if pal in (UYVY, YUYV):
        return [plane(w * 2, h, 16, 236)], w // 2
    if pal == YUV411:
        return [plane((w // 4) * 6, h, 16, 236)], w // 4
    if pal in (YUV420P, YVU420P):
        return [plane(w, h, 16, 236), plane(w // 2, h // 2, 16, 241), plane(w // 2, h // 2, 16, 241)], w
    if pal == YUV422P:
        return [plane(w, h, 16, 236), plane(w // 2, h, 16, 241), plane(w // 2, h, 16, 241)], w
    if pal == YUV444P:
        return [plane(w, h, 16, 236), plane(w, h, 16, 241), plane(w, h, 16, 241)], w
    return [plane(w, h, 16, 236), plane(w, h, 16, 241), plane(w, h, 16, 241), plane(w, h)], w


def main():
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from lives_amd import lib
    from tests import weedhost as wh
    L = lib.load()
    wh.bind(L)
    rng = np.random.default_rng(seed)
    ncalls = served = 0
    for it in range(nseq):
        pal = PALS[rng.integers(len(PALS))]
        w, h = int(rng.integers(4, 40)) * 4, int(rng.integers(4, 40)) * 2
        planes, wl = planes_for(rng, pal, w, h)
        kw = dict(gamma=int(rng.integers(1, 3))) if pal <= ARGB32 else dict(clamping=int(rng.integers(0, 2)), subspace=1)
        steps = []
        for _ in range(int(rng.integers(2, 6))):
            k = int(rng.integers(0, 9))
            if k <= 2:
                o, c = PALS[rng.integers(len(PALS))], int(rng.integers(0, 2))
                steps.append(("convert %d/%d" % (o, c), lambda lay, o=o, c=c: L.lives_gpu_convert_layer_palette(lay, o, c)))
            elif k == 3:
                g = int(rng.integers(1, 4))
                steps.append(("gamma %d" % g, lambda lay, g=g: L.lives_gpu_gamma_convert_layer(g, lay)))
            elif k == 4:
                d = int(rng.integers(0, 2))
                steps.append(("premult %d" % d, lambda lay, d=d: (L.lives_gpu_alpha_premult(lay, d), 1)[1]))
            elif k == 5:
                nw, nh, ip = int(rng.integers(2, 50)) * 4, int(rng.integers(2, 50)) * 2, int(rng.integers(0, 4))
                steps.append(("resize %dx%d/%d" % (nw, nh, ip), lambda lay, nw=nw, nh=nh, ip=ip: L.lives_gpu_resize_layer(lay, nw, nh, ip, 0, 0)))
            elif k == 6:
                iw, ih = int(rng.integers(2, 30)) * 4, int(rng.integers(2, 30)) * 2
                ow, oh = iw + int(rng.integers(0, 10)) * 4, ih + int(rng.integers(0, 10)) * 2
                steps.append(("letterbox %dx%d in %dx%d" % (iw, ih, ow, oh), lambda lay, a=(ow, oh, iw, ih): L.lives_gpu_letterbox_layer(lay, a[0], a[1], a[2], a[3], 3, 0, 0)))
            elif k == 7:
                steps.append(("compact", lambda lay: L.lives_gpu_compact_rowstrides(lay)))
            else:
                steps.append(("clear", lambda lay: L.lives_gpu_weed_layer_clear_pixel_data(lay)))
        results = []
        for mode in range(3):
            lay = wh.new_layer(pal, wl, h, planes, **kw)
            if mode:
                assert L.lives_gpu_layer_pin(lay) == 0
            rcs = []
            for name, fn in steps:
                if mode == 2:
                    box = []
                    t = threading.Thread(target=lambda: box.append(fn(lay)))
                    t.start(); t.join()
                    rcs.append(box[0])
                else:
                    rcs.append(fn(lay))
            if mode:
                assert L.lives_gpu_layer_unpin(lay) == 0
            pl, _, rs = wh.planes_of(lay)
            results.append((rcs, [wh.geti(lay, k) for k in KEYS], rs, pl))
        ncalls += len(steps)
        served += sum(1 for r in results[0][0] if r)
        for mode in (1, 2):
            a, b = results[0], results[mode]
            same = a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and len(a[3]) == len(b[3]) and all((x == y).all() for x, y in zip(a[3], b[3]))
            if not same:
                print("MISMATCH (ordinary layer against %s) seed %d sequence %d: palette %d %dx%d %s: %s" % ("pinned" if mode == 1 else "pinned, a thread per call", seed, it, pal, w, h, kw,
                                                                                                              [s[0] for s in steps]))
                print("  returns", a[0], b[0], "leaves", a[1], b[1], "rowstrides", a[2], b[2])
                sys.exit(1)
    print("fuzz_seam: %d sequences, %d seam calls (%d served, %d declined), ordinary == pinned == pinned with a thread per call" % (nseq, ncalls, served, ncalls - served))


if __name__ == "__main__":
    main()
