#!/usr/bin/env python3
"""tools/bench_c4.py -- BASELINE config 4 in one launch (lgpu_gauss5_colorkey) on 3840x2160 RGBA32 and RGB24, HIP-graph replay; LGPU_GCK_TH sets the band height"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lives_amd import ops


def main():
    ops.init(0)
    w, h, nb = 3840, 2160, 4
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    for ps in (4, 3):
        a = [torch.randint(0, 256, (h, w * ps), dtype=torch.uint8, device="cuda", generator=g) for _ in range(nb)]
        b = [torch.randint(0, 256, (h, w * ps), dtype=torch.uint8, device="cuda", generator=g) for _ in range(nb)]
        o = [torch.zeros((h, w * ps), dtype=torch.uint8, device="cuda") for _ in range(nb)]

        def run(i):
            ops.gauss5_colorkey(a[i % nb], b[i % nb], o[i % nb], w, h, ps, 0, float(os.environ.get("BENCH_C4_DELTA", "0.3")), 0.8, (128, 128, 128))
        for i in range(20):
            run(i)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for i in range(4 * nb):
                    run(i)
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * 4 * nb)
        ab = 3 * w * h * ps
        print(json.dumps({"op": "C4 gauss5 -> colour key %s 3840x2160, th=%s" % ("RGBA32" if ps == 4 else "RGB24", os.environ.get("LGPU_GCK_TH", "default")), "us": round(us, 2),
                          "algorithmic_bytes": ab, "frac_of_8TBs": round(ab / us / 1e3 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
