#!/usr/bin/env python3
"""A / B of the k_half8s variants (library built with LGPU_EXTRA_FLAGS=-DLGPU_H8S_AB): the bench workload (16 x 4K tracks,
half of layer 2 translucent), variants interleaved in one process, HIP events on the launch stream (lgpu_chain_timed);
every variant is first checked bit for bit against variant 0's output.  Prints median / min us per launch per variant."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from lives_amd import lib, ops   # noqa: E402

SW, SH, DW, DH, T = 3840, 2160, 1920, 1080, 16


def main():
    variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,2,3,16,32".split(","))]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    transl = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    ops.init(0)
    L = lib.load()
    g = torch.Generator(device="cuda")
    g.manual_seed(0x11FE5)
    srcs = [torch.randint(0, 256, (SH, SW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
    l2s = [torch.randint(0, 256, (DH, DW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
    for t in l2s:
        a = t[:, 3::4]
        a[torch.rand(a.shape, device="cuda", generator=g) >= transl] = 255
    dsts = [torch.zeros((DH, DW * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    lut = np.zeros(256, np.uint8)
    assert L.lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data) == 1
    pblock = torch.tensor([107, 0, 0, 0], dtype=torch.int32, device="cuda")
    prm = ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=1, interp=3, do_blur=0, bf=128, lut=lut, param_block=pblock)
    trk = ops.chain_tracks(srcs, l2s, dsts)
    ref = None
    for v in variants:
        assert L.lgpu_h8s_set_opt(v) == 0
        for d in dsts:
            d.zero_()
        ops.chain(prm, trk)
        torch.cuda.synchronize()
        out = [d.clone() for d in dsts]
        if v & 48:
            continue                      # ablation variants (no compute / no window DMA) do not produce the frame
        if ref is None:
            ref = out
        else:
            for a, b in zip(ref, out):
                assert torch.equal(a, b), "variant %d differs from variant %d" % (v, variants[0])
    for _ in range(300):
        ops.chain(prm, trk)
    torch.cuda.synchronize()
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            L.lgpu_h8s_set_opt(v)
            res[v].append(ops.chain_timed(prm, trk, 100) * 1e3 / 100)
    L.lgpu_h8s_set_opt(0)
    algo = (SW * SH * 4 + 2 * DW * DH * 4) * T
    for v in variants:
        x = sorted(res[v])
        med = x[len(x) // 2]
        print(json.dumps({"opt": v, "median_us": round(med, 2), "min_us": round(x[0], 2), "max_us": round(x[-1], 2),
                          "frac_of_8TBs_at_median": round(algo / (med * 1e-6) / 8e12, 4)}))


if __name__ == "__main__":
    main()
