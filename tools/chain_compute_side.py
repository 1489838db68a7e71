#!/usr/bin/env python3
"""tools/chain_compute_side.py -- the default chain launch with its memory side made cheap: all 16 tracks read the SAME source frame and layer 2 (41 MB working set, resident in the
256 MiB Infinity Cache) and write 16 destinations.  What is left is the arithmetic + the write stream; compare with bench.py's launch_us on the same box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lives_amd import ops
from lives_amd.lib import load

ops.init(0)
SW, SH, DW, DH, T = 3840, 2160, 1920, 1080, 16
g = torch.Generator(device="cuda"); g.manual_seed(5)
lut = np.zeros(256, np.uint8)
load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data)
prm = ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=1, interp=3 | 0x100, do_blur=int(os.environ.get("BLUR", "0")), bf=128, lut=lut)
srcs = [torch.randint(0, 256, (SH, SW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
l2s = [torch.randint(0, 256, (DH, DW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
dsts = [torch.zeros((DH, DW * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
for name, s_, l_ in (("16 distinct sources (the bench's launch)", srcs, l2s), ("16 tracks on ONE source + ONE layer 2", [srcs[0]] * T, [l2s[0]] * T)):
    trk = ops.chain_tracks(s_, l_, dsts)
    for _ in range(300):
        ops.chain(prm, trk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.chain(prm, trk)
    e1.record(); torch.cuda.synchronize()
    print("%-45s %.2f us per launch" % (name, e0.elapsed_time(e1) * 1e3 / 200))
