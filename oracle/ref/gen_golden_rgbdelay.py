#!/usr/bin/env python3
"""tests/golden/rgbdelay.npz from the reference's own RGBdelay.c (oracle/_ref/RGBdelay.so, built unmodified by build_ref.sh):
one filter instance over a sequence of frames.  TEST INFRASTRUCTURE ONLY; fixtures are data.  Own seed stream.  Also checks the C
restatement against every record."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name -> (filter, palette, yuv clamping leaf (0 clamped, 1 unclamped, -1 none), max cache, {frame: (r, g, b, strength)}, in place)
CASES = {
    "defaults": ("RGBdelay", 1, -1, 20, {0: (1, 0, 0, 1.0), 4: (0, 1, 0, 1.0), 8: (0, 0, 1, 1.0)}, 0),
    "defaults_inplace": ("RGBdelay", 1, -1, 20, {0: (1, 0, 0, 1.0), 4: (0, 1, 0, 1.0), 8: (0, 0, 1, 1.0)}, 1),
    "bgr_mix": ("RGBdelay", 2, -1, 10, {0: (1, 1, 1, 0.7), 2: (1, 0, 1, 0.5), 3: (0, 1, 0, 1.0)}, 0),
    "strong": ("RGBdelay", 1, -1, 6, {0: (1, 1, 1, 1.0), 1: (1, 1, 1, 1.0), 5: (1, 1, 0, 0.9)}, 0),
    "nocache": ("RGBdelay", 1, -1, 20, {0: (1, 0, 1, 0.6)}, 0),
    "nocache_inplace": ("RGBdelay", 2, -1, 20, {0: (0, 1, 1, 0.8)}, 1),
    "cache_cut": ("RGBdelay", 1, -1, 3, {0: (1, 0, 0, 1.0), 2: (0, 1, 0, 1.0), 4: (0, 0, 1, 1.0)}, 0),
    "yuv_clamped": ("YUVdelay", 588, 0, 8, {0: (1, 0, 0, 1.0), 2: (0, 1, 1, 0.8)}, 0),
    "yuv_unclamped": ("YUVdelay", 588, 1, 8, {0: (1, 1, 0, 1.0), 3: (0, 0, 1, 1.0)}, 0),
    "yuv_clamped_nocache": ("YUVdelay", 588, 0, 8, {0: (1, 0, 1, 0.9)}, 1),
}


def param_arrays(groups):
    on = np.zeros(51 * 3, np.int32)
    st = np.ones(51, np.float64)
    for j, (r, g, b, s) in groups.items():
        on[3 * j:3 * j + 3] = (r, g, b)
        st[j] = s
    return on, st


def weed_params(maxcache, on, st):
    out = [po.p_int(maxcache)]
    for j in range(51):
        out += [po.p_bool(on[3 * j]), po.p_bool(on[3 * j + 1]), po.p_bool(on[3 * j + 2]), po.p_double(st[j])]
    return out


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    H, O = po.RefHost(), po.oracle()
    rng = np.random.default_rng(0xD31A7)
    rec, names = {}, []
    w, h, n = 10, 6, 12
    for name, (fn, pal, clamp, maxcache, groups, inplace) in CASES.items():
        on, st = param_arrays(groups)
        frames = [po.make_frame(rng, w, h, 3) for _ in range(n)]
        H.H.refhost_set_yuv_clamping(clamp)
        if inplace:
            ref = [f.copy() for f in frames]
            H.run_seq(po.refplugin("RGBdelay"), fn, pal, w, h, ref, ref, weed_params(maxcache, on, st))
        else:
            ref = [np.full_like(f, 0x5A) for f in frames]
            H.run_seq(po.refplugin("RGBdelay"), fn, pal, w, h, frames, ref, weed_params(maxcache, on, st))
        H.H.refhost_set_yuv_clamping(-1)
        s = O.orc_rgbdelay_new()
        for i in range(n):
            got = frames[i].copy() if inplace else np.full_like(frames[i], 0x5A)
            src = got if inplace else frames[i]
            assert O.orc_rgbdelay_process(s, po.P(src), src.strides[0], po.P(got), got.strides[0], w, h, pal, 1 if clamp == 0 else 0, maxcache,
                                          on.ctypes.data, st.ctypes.data) == 0
            assert np.array_equal(got, ref[i]), "oracle differs from the reference: %s frame %d" % (name, i)
        O.orc_rgbdelay_free(s)
        rec[name + "|in"] = np.stack(frames)
        rec[name + "|out"] = np.stack(ref)
        names.append(name)
    rec["records"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "rgbdelay.npz"), **rec)
    mpath = os.path.join(OUT, "manifest.json")
    man = json.load(open(mpath))
    man["groups"]["rgbdelay.npz"] = ("reference SOURCE unmodified, but compiled with clang -ftrivial-auto-var-init=zero (oracle/ref/build_ref.sh): that flag changes what the reference's "
                                     "uninitialised-int read in weed_param_get_value_boolean does (quirk R1 in DESIGN.md: a plain gcc build reads every switch as off), so these records "
                                     "pin the plugin's evident intent, not a stock build: lives-plugins/weed-plugins/RGBdelay.c (filters RGBdelay, YUVdelay); one instance over 12 frames "
                                     "of 10x6; parameter sets = CASES in oracle/ref/gen_golden_rgbdelay.py (name -> filter, palette, YUV_clamping leaf, cache size, "
                                     "{frame: r, g, b switches + strength}, in place); <name>|in / <name>|out = stacked frames (out of place: destination pre-filled 0x5A)")
    json.dump(man, open(mpath, "w"), indent=1)
    print("rgbdelay.npz: %d sequences, %d KB" % (len(names), os.path.getsize(os.path.join(OUT, "rgbdelay.npz")) // 1024))


if __name__ == "__main__":
    main()
