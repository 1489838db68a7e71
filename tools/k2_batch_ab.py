#!/usr/bin/env python3
"""K2 on 16 frames per launch, frames walked in turn (default) against all frames at once (LGPU_K2_WGS=99, the old shape): us per launch on ONE set of 16 frames (182 MB:
inside the memory-side cache) and on nine rotating sets (1.6 GB)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lives_amd import ops
ops.init(0)
w, h, NT = 1920, 1080, 16
g = torch.Generator(device="cuda"); g.manual_seed(5)
def fr(r, c): return torch.randint(0, 256, (r, c), dtype=torch.uint8, device="cuda", generator=g)
lut = np.arange(256, dtype=np.uint8)[::-1].copy()
for nsets in (1, 9):
    sets = [[(fr(h, w), fr(h // 2, w // 2), fr(h // 2, w // 2), torch.zeros((h, w * 4), dtype=torch.uint8, device="cuda")) for _ in range(NT)] for _ in range(nsets)]
    for i in range(20): ops.yuv420p_to_rgb_batch(sets[i % nsets], w, h, lut=lut)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(90): ops.yuv420p_to_rgb_batch(sets[i % nsets], w, h, lut=lut)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 90)
    nbytes = NT * (w * h * 3 // 2 + w * h * 4)
    print("%d set(s): %.2f us  %.3f of 8 TB/s" % (nsets, best, nbytes / best / 1e3 / 8000))
    del sets
