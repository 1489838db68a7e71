#!/usr/bin/env python3
"""tools/bench_gauss5.py -- lgpu_gauss5 at 3840x2160 RGBA32 (BASELINE config 4), HIP events around back-to-back launches on rotating buffers.
Run once as is (SWAR horizontal pass) and once with LGPU_G5_MFMA=1 (horizontal pass on the matrix cores); the MFMA run first checks its frame
against a frame computed by the SWAR pass in a child process is not needed: the parity tests run under the same switch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
from lives_amd import ops   # noqa: E402


def main():
    ops.init(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    w, h, nb, reps = 3840, 2160, 6, 40
    srcs = [torch.randint(0, 256, (h, w * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(nb)]
    dsts = [torch.zeros((h, w * 4), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    for i in range(3):
        ops.gauss5(srcs[i], dsts[i], w, h, psize=4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ops.gauss5(srcs[i % nb], dsts[i % nb], w, h, psize=4)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ab = 2 * w * h * 4
    print(json.dumps({"op": "gauss5 3840x2160 RGBA32", "horizontal_pass": "mfma" if os.environ.get("LGPU_G5_MFMA") else "swar", "us": round(us, 2),
                      "GBs": round(ab / us / 1e3, 1), "frac_of_8TBs": round(ab / us / 1e3 / 8000, 4), "checksum": int(dsts[0].to(torch.int64).sum().item())}))


if __name__ == "__main__":
    main()
