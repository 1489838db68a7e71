#!/usr/bin/env python3
"""tools/corun_probe.py -- can a small kernel (RCCL's broadcast, here a stand-in with a given LDS footprint) start while the persistent chain kernel runs?
Launches the 16-track chain on the main stream and, right behind it on a side stream, a 4-workgroup probe with N KB of LDS; prints when the probe finished relative to
the chain (HIP events).  With `spare` > 0 the chain kernel leaves that many workgroup slots free (LGPU_CHAIN_SPARE_WGS)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402


def main():
    so = os.path.join(ROOT, "tools", "corun_probe.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "corun_probe.hip")])
    from lives_amd import ops
    from lives_amd.lib import load
    ops.init(0)
    P = ctypes.CDLL(so)
    P.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    SW, SH, DW, DH, T = 3840, 2160, 1920, 1080, 16
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    srcs = [torch.randint(0, 256, (SH, SW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
    l2s = [torch.randint(0, 256, (DH, DW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
    dsts = [torch.zeros((DH, DW * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    lut = np.zeros(256, np.uint8)
    load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data)
    prm = ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=1, interp=3, do_blur=0, bf=128, lut=lut)
    trk = ops.chain_tracks(srcs, l2s, dsts)
    out = torch.zeros(64, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    for _ in range(200):
        ops.chain(prm, trk)
    torch.cuda.synchronize()
    for lds_kb in (0, 8, 16, 32, 64):
        res = []
        for rep in range(5):
            e0, e1, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.chain(prm, trk)
            e1.record()
            with torch.cuda.stream(side):
                P.probe_launch(side.cuda_stream, out.data_ptr(), 4, lds_kb * 1024)
                p1.record(side)
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) * 1e3, e0.elapsed_time(p1) * 1e3))
        print("probe with %2d KB LDS: chain %.0f us, probe done %.0f us after the chain's launch (median of 5)" % (lds_kb, sorted(r[0] for r in res)[2], sorted(r[1] for r in res)[2]), flush=True)


if __name__ == "__main__":
    main()
