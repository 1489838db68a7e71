#!/usr/bin/env python3
"""tests/golden/blurzoom.npz: the reference's blurzoom.c (oracle/_ref/blurzoom.so, built unmodified) driven over frame
SEQUENCES on one filter instance (refhost_run_seq): it is stateful.  TEST INFRASTRUCTURE ONLY; fixtures are data."""
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    assert po.have_ref()
    H = po.RefHost()
    rng = np.random.default_rng(0xB1A2)
    rec, names = {}, []
    for pal in (3, 4):
        for mode in (0, 1, 2, 3):
            pattern = (mode + pal) % 4
            w, h, n = (70, 10, 6) if mode in (0, 3) else (64, 10, 6)
            stride = po.align(w * 4) if mode in (0, 3) else w * 4          # strobe modes: compact rows (blurzoom.c:391-396)
            srcs = []
            for f in range(n):
                a = rng.integers(0, 256, (h, stride), dtype=np.uint8)
                a[:, :w * 4] = (a[:, :w * 4] >> 4) + 40
                a[1 + f % 3:5 + f % 3, (6 + 6 * f) * 4:(20 + 6 * f) * 4] = 250
                srcs.append(a)
            dsts = [np.full_like(a, 0x5A) for a in srcs]
            H.run_seq(po.refplugin("blurzoom"), "blurzoom", pal, w, h, srcs, dsts, [po.p_int(mode), po.p_int(pattern)])
            key = "bz|%d|%d|%d|%d|%d|%d" % (pal, mode, pattern, w, h, n)
            rec[key + "|in"] = np.stack(srcs)
            rec[key + "|out"] = np.stack(dsts)
            names.append(key)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "blurzoom.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["blurzoom.npz"] = ("reference plugin built unmodified: lives-plugins/weed-plugins/blurzoom.c; record bz|palette|mode|pattern|w|h|nframes, "
                                     "in / out = [nframes][h][rowstride]; ONE instance per record, frames in order (stateful filter)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("blurzoom.npz: %d sequences, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "blurzoom.npz")) // 1024))


if __name__ == "__main__":
    main()
