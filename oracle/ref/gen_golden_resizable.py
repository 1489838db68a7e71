#!/usr/bin/env python3
"""tests/golden/resizable.npz: the answers of the REFERENCE's own get_resizable / get_tgt_gamma / can_inline_gamma / pconv_can_inplace
(src/colourspace.c:14500-14669, :14736-14740, :12128-12157; compiled by build_resizable_slice.py from where they lie, no swscale) for every pair of
the 15 integer palettes (+ the hints NONE / ANY), both scale directions, both clamping hints.  Run in the build container only.

  pals[15], hints[17]
  resizable[15, 17, 2, 2, 6]   [palette, hint, upscale, oclamp_hint] -> result (1 / 0 / -1 = LIVES_FATAL), resolved, xpalette, oclamp_hint, opal_hint, xopal_hint
  tgt_gamma[15, 15], inline_gamma[15, 15], inplace[15, 15]
  is_resizable[15, 2]          weed_palette_is_resizable(pal, ., LIVES_INPUT / LIVES_OUTPUT)"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call([sys.executable, os.path.join(HERE, "build_resizable_slice.py")])
L = ctypes.CDLL(os.path.join(HERE, "..", "_ref", "libresizableref.so"))
PALS = [1, 2, 3, 4, 5, 512, 513, 522, 544, 545, 564, 565, 588, 589, 595]
HINTS = PALS + [0, -1]
res = np.zeros((len(PALS), len(HINTS), 2, 2, 6), np.int32)
for i, p in enumerate(PALS):
    for j, h in enumerate(HINTS):
        for up in (0, 1):
            for cl in (0, 1):
                io = (ctypes.c_int * 5)(p, h, cl, up, 0)
                r = L.rsref_get_resizable(io)
                res[i, j, up, cl] = [r] + (list(io) if r == 1 else [0] * 5)
tg = np.array([[L.rsref_get_tgt_gamma(a, b) for b in PALS] for a in PALS], np.int32)
ig = np.array([[L.rsref_can_inline_gamma(a, b) for b in PALS] for a in PALS], np.int32)
ip = np.array([[L.rsref_pconv_can_inplace(a, b) for b in PALS] for a in PALS], np.int32)
isr = np.array([[L.rsref_is_resizable(a, d) for d in (1, 2)] for a in PALS], np.int32)
out = os.path.normpath(os.path.join(HERE, "..", "..", "tests", "golden", "resizable.npz"))
np.savez_compressed(out, pals=np.array(PALS, np.int32), hints=np.array(HINTS, np.int32), resizable=res, tgt_gamma=tg, inline_gamma=ig, inplace=ip, is_resizable=isr)
print(out, "fatal:", int((res[..., 0] == -1).sum()), "fail:", int((res[..., 0] == 0).sum()), "ok:", int((res[..., 0] == 1).sum()))
