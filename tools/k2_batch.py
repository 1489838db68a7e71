#!/usr/bin/env python3
"""tools/k2_batch.py [reps] -- the K2 batch workload of profiles/*/ops_roofline.md on its own (16 x 1080p YUV420P -> RGBA32 + gamma LUT in one
launch), for rocprofv3 passes (tools/pmc_k2.sh) and quick timing: prints us per launch (HIP events) and the fraction of the 8 TB/s roofline."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from lives_amd import ops   # noqa: E402
from lives_amd.lib import load   # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    with_lut = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0        # 0: plain palette conversion, no gamma on the way
    ops.init(0)
    w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x11FE5)

    def planes(ww, hh, n):
        return [torch.randint(0, 256, (hh, (ww + 31) // 32 * 32), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)]
    lut = np.zeros(256, np.uint8)
    load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data)
    frames = list(zip(planes(w, h, nt), planes(w // 2, h // 2, nt), planes(w // 2, h // 2, nt), planes(w * 4, h, nt)))
    import time
    t_end = time.perf_counter() + 0.08              # ~80 ms of the same launch first: clock / power state of a running pipeline
    while time.perf_counter() < t_end:
        for _ in range(20):
            ops.yuv420p_to_rgb_batch(frames, w, h, lut=lut if with_lut else None)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.yuv420p_to_rgb_batch(frames, w, h, lut=lut if with_lut else None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ab = nt * (w * h * 3 // 2 + w * h * 4)
    print(json.dumps({"op": "yuv420p -> RGBA32%s, %d x %dx%d per launch" % (" + gamma LUT" if with_lut else "", nt, w, h), "us_per_launch": round(us, 2), "algorithmic_bytes": ab,
                      "GBs": round(ab / us / 1e3, 1), "frac_of_8TBs": round(ab / us / 1e3 / 8000, 4)}))


if __name__ == "__main__":
    main()
