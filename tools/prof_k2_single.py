#!/usr/bin/env python3
"""tools/prof_k2_single.py -- 300 back-to-back launches of the C2 conversion (one 1080p YUV420P frame -> RGBA32 + gamma LUT) and of the in-place gamma pass, for
`rocprofv3 --kernel-trace --stats`: kernel durations against the back-to-back launch interval tools/bench_ops.py reports"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lives_amd import ops
from lives_amd.lib import load
ops.init(0)
w, h = 1920, 1080
g = torch.Generator(device="cuda"); g.manual_seed(1)
Y = [torch.randint(16, 236, (h, w), dtype=torch.uint8, device="cuda", generator=g) for _ in range(4)]
U = [torch.randint(16, 241, (h // 2, w // 2), dtype=torch.uint8, device="cuda", generator=g) for _ in range(4)]
V = [torch.randint(16, 241, (h // 2, w // 2), dtype=torch.uint8, device="cuda", generator=g) for _ in range(4)]
D = [torch.zeros((h, w * 4), dtype=torch.uint8, device="cuda") for _ in range(4)]
lut = np.zeros(256, np.uint8)
load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300):
        ops.yuv420p_to_rgb(Y[i & 3], U[i & 3], V[i & 3], D[i & 3], w, h, lut=lut)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(300):
        ops.gamma_apply(D[i & 3], w, h, 4, lut)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print("back-to-back: K2 %.2f us, gamma %.2f us per launch" % ((t1 - t0) / 300 * 1e6, (t2 - t1) / 300 * 1e6))
