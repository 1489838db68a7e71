#!/usr/bin/env python3
"""tools/bench_chain_geom.py <sw> <sh> [tracks [dw dh]] -- the fused chain (2:1 unless dw dh are given) at another geometry: us per launch and fraction of the 8 TB/s roofline; used to look at
how the persistent kernel's tile-list stride interacts with the number of tiles per row"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from lives_amd import ops   # noqa: E402
from lives_amd.lib import load   # noqa: E402


def main():
    sw, sh = int(sys.argv[1]), int(sys.argv[2])
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    dw, dh = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (sw // 2, sh // 2)
    ops.init(0)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    sets = []
    for _ in range(2):
        srcs = [torch.randint(0, 256, (sh, sw * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
        l2s = [torch.randint(0, 256, (dh, dw * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
        dsts = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
        sets.append((srcs, l2s, dsts, ops.chain_tracks(srcs, l2s, dsts)))
    lut = np.zeros(256, np.uint8)
    load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data)
    prm = ops.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3, do_blur=0, bf=128, lut=lut)
    t_end = time.perf_counter() + 0.08
    i = 0
    while time.perf_counter() < t_end:
        for _ in range(20):
            ops.chain(prm, sets[i & 1][3]); i += 1
        torch.cuda.synchronize()
    reps = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ops.chain(prm, sets[i & 1][3])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ab = T * (sw * sh * 4 + 2 * dw * dh * 4)
    print(json.dumps({"geometry": "%dx%d -> %dx%d x %d tracks" % (sw, sh, dw, dh, T), "tiles_per_row": (dw + 63) // 64, "spare_wgs": os.environ.get("LGPU_CHAIN_SPARE_WGS", "0"),
                      "us_per_launch": round(us, 2), "frac_of_8TBs": round(ab / us / 1e3 / 8000, 4)}))


if __name__ == "__main__":
    main()
