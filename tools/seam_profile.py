#!/usr/bin/env python3
"""where the host time of bench.py's seam_chain leg goes: tools/seam_host.c's per-call clocks, threads on and off (profiles/r06/seam_host_profile.txt)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lives_amd import lib, ops  # noqa: E402

ops.init(0)
L = lib.load()
Hs = ctypes.CDLL(os.path.join(ROOT, "tools", "libseam_host.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
Hs.seam_host_run.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(vp), ctypes.POINTER(ci)]
Hs.seam_host_profile.argtypes = [ctypes.POINTER(ctypes.c_double), ci]
assert Hs.seam_host_init(os.path.join(ROOT, "lives_amd", "livesgpu_fx.so").encode()) == 0
T, SW, SH, DW, DH = int(os.environ.get("TRACKS", "16")), 3840, 2160, 1920, 1080
g = torch.Generator(device="cuda")
g.manual_seed(1)
srcs = [torch.randint(0, 256, (SH, SW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
l2s = [torch.randint(0, 256, (DH, DW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
torch.cuda.synchronize()
sp, lp = (vp * T)(*[t.data_ptr() for t in srcs]), (vp * T)(*[t.data_ptr() for t in l2s])
names = ["new layer + pin_device", "convert_layer_palette", "resize_layer", "channels + process_func", "gamma_convert_layer", "-", "collector: wait for tracks", "collector: layers_flush"]
for threads in (1, 0, 1):
    ms = ctypes.c_double()
    prof = (ctypes.c_double * 8)()
    Hs.seam_host_profile(prof, 1)
    ticks, warm = 200, 30
    rc = Hs.seam_host_run(T, SW, SH, DW, DH, sp, lp, 128, 2, ticks, warm, threads, ctypes.byref(ms), None, None)
    Hs.seam_host_profile(prof, 1)
    Hs.seam_host_release()
    n = ticks          # the clocks restart after the warm-up ticks (streams, scaler tables and pool buffers exist by then)
    print("threads=%d rc=%d  %.1f us per tick  %.0f frames/s" % (threads, rc, ms.value * 1e3 / ticks, T * ticks / (ms.value * 1e-3)))
    for i, nm in enumerate(names):
        if nm != "-":
            per = prof[i] / n / (T if i < 6 else 1)
            print("    %-28s %8.2f us per %s" % (nm, per, "call" if i < 6 else "tick"))
