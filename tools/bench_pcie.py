#!/usr/bin/env python3
"""tools/bench_pcie.py [--batches N] [--tracks T] -- the headline chain when the host hands over HOST buffers: frames and layer 2 come from pinned
host memory, results go back to pinned host memory.  Three HIP streams (copy in, compute, copy out) over two device buffer sets, so the upload of
batch n + 1 and the download of batch n - 1 overlap the kernel of batch n.  Reports frames/s including PCIe (DESIGN.md quotes it next to, never as,
the HBM-resident headline)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--tracks", type=int, default=16)
    ap.add_argument("--yuv420p", type=int, default=0, help="1: the tracks arrive as decoder output (YUV420P, 12.4 MB per 4K frame) and are converted on the device (batched K2)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from lives_amd import ops
    from lives_amd.lib import load
    ops.init(0)
    SW, SH, DW, DH, T = 3840, 2160, 1920, 1080, args.tracks
    lut = np.zeros(256, np.uint8)
    assert load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data) == 1
    g = torch.Generator()
    g.manual_seed(7)
    YUV = bool(args.yuv420p)
    if YUV:
        h_src = [(torch.randint(0, 256, (SH, SW), dtype=torch.uint8, generator=g).pin_memory(),
                  torch.randint(0, 256, (SH // 2, SW // 2), dtype=torch.uint8, generator=g).pin_memory(),
                  torch.randint(0, 256, (SH // 2, SW // 2), dtype=torch.uint8, generator=g).pin_memory()) for _ in range(T)]
    else:
        h_src = [torch.randint(0, 256, (SH, SW * 4), dtype=torch.uint8, generator=g).pin_memory() for _ in range(T)]
    h_l2 = [torch.randint(0, 256, (DH, DW * 4), dtype=torch.uint8, generator=g).pin_memory() for _ in range(T)]
    h_out = [[torch.zeros((DH, DW * 4), dtype=torch.uint8).pin_memory() for _ in range(T)] for _ in range(2)]
    sets = []
    for _ in range(2):
        d_src = [torch.empty((SH, SW * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
        d_l2 = [torch.empty((DH, DW * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
        d_dst = [torch.empty((DH, DW * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
        d_yuv = [(torch.empty((SH, SW), dtype=torch.uint8, device="cuda"), torch.empty((SH // 2, SW // 2), dtype=torch.uint8, device="cuda"),
                  torch.empty((SH // 2, SW // 2), dtype=torch.uint8, device="cuda")) for _ in range(T)] if YUV else None
        sets.append((d_src, d_l2, d_dst, ops.chain_tracks(d_src, d_l2, d_dst), d_yuv))
    prm = ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=0 if YUV else 1, interp=3, do_blur=0, bf=128, lut=lut)
    s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_k = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]

    def run(nb):
        for b in range(nb):
            i = b & 1
            d_src, d_l2, d_dst, trk, d_yuv = sets[i]
            with torch.cuda.stream(s_in):
                if b >= 2:
                    s_in.wait_event(ev_k[i])          # the kernel that read this buffer set two batches ago is done
                for t in range(T):
                    if YUV:
                        for p in range(3):
                            d_yuv[t][p].copy_(h_src[t][p], non_blocking=True)
                    else:
                        d_src[t].copy_(h_src[t], non_blocking=True)
                    d_l2[t].copy_(h_l2[t], non_blocking=True)
                ev_in[i].record(s_in)
            with torch.cuda.stream(s_k):
                s_k.wait_event(ev_in[i])
                if b >= 2:
                    s_k.wait_event(ev_out[i])         # its previous result has left the device
                if YUV:                               # decoder planes -> RGBA32 for all tracks in one launch, then the fused chain
                    ops.yuv420p_to_rgb_batch([(d_yuv[t][0], d_yuv[t][1], d_yuv[t][2], d_src[t]) for t in range(T)], SW, SH)
                ops.chain(prm, trk)
                ev_k[i].record(s_k)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_k[i])
                for t in range(T):
                    h_out[i][t].copy_(d_dst[t], non_blocking=True)
                ev_out[i].record(s_out)
        torch.cuda.synchronize()

    run(4)
    t0 = time.perf_counter()
    run(args.batches)
    dt = time.perf_counter() - t0
    frames = args.batches * T
    up = frames * ((SW * SH * 3 // 2 if YUV else SW * SH * 4) + DW * DH * 4)
    down = frames * DW * DH * 4
    print(json.dumps({"metric": "effect-chain frames/sec at 3840x2160 RGBA32, host buffers (PCIe included)", "value": round(frames / dt, 1), "unit": "frames/s",
                      "ms_per_batch": round(dt / args.batches * 1e3, 3), "tracks": T, "h2d_GBps": round(up / dt / 1e9, 1), "d2h_GBps": round(down / dt / 1e9, 1),
                      "source": "YUV420P planes, converted on the device" if YUV else "BGRA32", "streams": "copy-in / compute / copy-out, 2 buffer sets"}))


if __name__ == "__main__":
    main()
