#!/usr/bin/env python3
"""tests/golden/cavg.npz: the reference's chroma-average tables cavgc / cavgu as init_average() fills them (src/colourspace.c:190-216, read from the slice built by
build_cs_slice.py through csref_cavg).  TEST INFRASTRUCTURE ONLY; fixtures are data.  Also checks orc_cavg against every entry."""
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
    c, u = np.zeros(65536, np.uint8), np.zeros(65536, np.uint8)
    R.csref_cavg(po.P(c), po.P(u))
    for x in range(256):
        for y in range(256):
            assert O.orc_cavg(1, x, y) == c[x * 256 + y] and O.orc_cavg(0, x, y) == u[x * 256 + y], (x, y)
    np.savez_compressed(os.path.join(OUT, "cavg.npz"), cavgc=c.reshape(256, 256), cavgu=u.reshape(256, 256))
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["cavg.npz"] = "src/colourspace.c:190-216 init_average: cavgc (clamped) and cavgu (unclamped), [x][y]"
    json.dump(man, open(mpath, "w"), indent=1)
    print("cavg.npz: %d KB" % (os.path.getsize(os.path.join(OUT, "cavg.npz")) // 1024))


if __name__ == "__main__":
    main()
