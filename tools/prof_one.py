#!/usr/bin/env python3
"""tools/prof_one.py <op> -- one entry point in a loop for rocprofv3 --kernel-trace --stats (which kernels a multi-launch op spends its time in)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from lives_amd import ops   # noqa: E402


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else "edge"
    ops.init(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    w, h = 1920, 1080
    if op == "edge":
        src = torch.randint(0, 256, (h, w * 4), dtype=torch.uint8, device="cuda", generator=g)
        dst = torch.zeros_like(src)
        for _ in range(200):
            ops.edge(src, dst, w, h, 3, 0)
    elif op == "softlight":
        pl = [torch.randint(0, 256, (h, w), dtype=torch.uint8, device="cuda", generator=g)] + [torch.randint(0, 256, (h // 2, w // 2), dtype=torch.uint8, device="cuda", generator=g) for _ in range(2)]
        dl = [torch.zeros_like(t) for t in pl]
        for _ in range(200):
            ops.softlight(pl, dl, w, h, 512, 0)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
