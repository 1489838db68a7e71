#!/usr/bin/env python3
"""tools/bench_ops.py -- every entry point of liblivesgpu.so at its BASELINE / SURVEY 8d size against the HBM roofline,
with the CPU oracle (one thread, gcc -O3 -march=native) timed beside it on the same host.

For each op: algorithmic bytes (compulsory reads + writes of one call), mean device time per launch (events on the launch
stream around back-to-back launches AND around replays of a HIP graph of the same launches; the smaller of the two, so that
kernels shorter than a Python call are not charged the host's time), achieved GB/s, fraction of the 8 TB/s HBM3E peak, and the oracle's
time for the same call.  Inputs are resident in HBM; working sets are rotated over `nbuf` buffers so that one pass is
larger than the 256 MiB Infinity Cache where the frame size allows.  Prints a markdown table (and JSON with --json).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    from lives_amd import ops
    from oracle import pyoracle as po
    ops.init(0)
    orc = None if args.no_cpu else ctypes.CDLL(po.build_oracle(native=True))
    P = po.P
    rng = np.random.default_rng(0x11FE5)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x11FE5)

    def dframe(w, h, ps, n=1):
        rs = po.align(w * ps)
        return [torch.randint(0, 256, (h, rs), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)]

    def hframe(w, h, ps):
        return rng.integers(0, 256, (h, po.align(w * ps)), dtype=np.uint8)

    rows = []

    def timeit(fn, nbuf):
        t_end = time.perf_counter() + 0.06          # ~60 ms of the same launch first (clock / power state of a running pipeline, as bench.py does)
        i = 0
        while time.perf_counter() < t_end:
            for _ in range(20):
                fn(i % nbuf)
                i += 1
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.reps):
            fn(i % nbuf)
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) * 1e-3 / args.reps
        # A Python / ctypes call costs ~8 us of host time: back-to-back launches of a kernel shorter than that measure the HOST.  So the same launches are also
        # replayed from a HIP graph (captured on a side stream; no host work between the kernels) and the smaller figure is the one reported.
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            per = 4 * nbuf
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for i in range(per):
                        fn(i % nbuf)
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            n = max(1, args.reps // per)
            e0.record()
            for _ in range(n):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            g = e0.elapsed_time(e1) * 1e-3 / (n * per)
            return min(eager, g)
        except Exception as ex:          # noqa: BLE001 -- an op that allocates or synchronises inside cannot be captured: the eager figure stands
            sys.stderr.write("    (no graph replay for this op: %s)\n" % str(ex).splitlines()[0][:100])
            torch.cuda.synchronize()
            return eager

    def cpu(fn):
        if orc is None:
            return None
        fn()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 0.5:
            fn()
            n += 1
        return (time.perf_counter() - t0) / n

    def add(name, ref, size, abytes, gpu_s, cpu_s):
        gbs = abytes / gpu_s / 1e9
        rows.append(dict(op=name, reference=ref, size=size, algorithmic_bytes=abytes, gpu_us=round(gpu_s * 1e6, 2), gbs=round(gbs, 1),
                         frac=round(gbs / PEAK, 4), cpu_ms=None if cpu_s is None else round(cpu_s * 1e3, 3),
                         speedup=None if cpu_s is None else round(cpu_s / gpu_s, 1)))
        print("  %-34s %9.2f us %8.1f GB/s  frac %.3f" % (name, gpu_s * 1e6, gbs, gbs / PEAK), file=sys.stderr, flush=True)

    NB = 8
    # ---- the floor: a launch that moves nothing (what a dependent 1080p op cannot go below on this stream) ------------------------------------
    tiny = torch.zeros(256, dtype=torch.uint8, device="cuda")
    from lives_amd.lib import call as _call
    t = timeit(lambda i: _call("lgpu_fill", tiny.data_ptr(), 0, 64, None), 1)
    add("a 64-byte lgpu_fill, replayed from a graph (what is left of a launch without the host)", "-", "-", 64, t, None)
    # ---- K1 swizzle: C1 (640x480 RGB24 -> BGRA32) and the same op at 4K -------------------------------------------------
    for (w, h) in ((640, 480), (3840, 2160)):
        src, dst = dframe(w, h, 3, NB), dframe(w, h, 4, NB)
        op = po.OPS.index("swap3addpost")
        t = timeit(lambda i: ops.swizzle(op, src[i], dst[i], w, h), NB)
        hs, hd = hframe(w, h, 3), hframe(w, h, 4)
        c = cpu(lambda: orc.orc_swizzle(op, 0, P(hs), hs.strides[0], P(hd), hd.strides[0], w, h, None))
        add("swizzle RGB24->BGRA32", "colourspace.c:9445-9511", "%dx%d" % (w, h), w * h * 7, t, c)
    # ---- K2 + fused gamma: C2 ----------------------------------------------------------------------------------------
    w, h = 1920, 1080
    lut = np.zeros(256, np.uint8)
    from lives_amd.lib import load
    load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data)
    Y, U, V = dframe(w, h, 1, NB), dframe(w // 2, h // 2, 1, NB), dframe(w // 2, h // 2, 1, NB)
    dst = dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.yuv420p_to_rgb(Y[i], U[i], V[i], dst[i], w, h, lut=lut), NB)
    hy, hu, hv, hd = hframe(w, h, 1), hframe(w // 2, h // 2, 1), hframe(w // 2, h // 2, 1), hframe(w, h, 4)
    st = (ctypes.c_int * 3)(hy.strides[0], hu.strides[0], hv.strides[0])
    if orc:
        orc.orc_yuv420p_to_rgb.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_long, ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_int]
    c = cpu(lambda: orc.orc_yuv420p_to_rgb(P(hy), P(hu), P(hv), st, hu.size, hv.size, P(hd), hd.strides[0], w, h, 4, 0, 0, 0, 2, P(lut), 0))
    add("yuv420p->RGBA32 + gamma LUT (C2)", "colourspace.c:3260-3904, :14034", "1920x1080", w * h * 3 // 2 + w * h * 4, t, c)
    NT = 16
    bY, bU, bV, bD = dframe(w, h, 1, NT), dframe(w // 2, h // 2, 1, NT), dframe(w // 2, h // 2, 1, NT), dframe(w, h, 4, NT)
    bframes = list(zip(bY, bU, bV, bD))
    t = timeit(lambda i: ops.yuv420p_to_rgb_batch(bframes, w, h, lut=lut), 1)
    add("yuv420p->RGBA32 + gamma LUT, 16 frames / launch", "colourspace.c:3260-3904, :14034", "16 x 1920x1080", NT * (w * h * 3 // 2 + w * h * 4), t, None)
    U2, V2 = dframe(w // 2, h, 1, NB), dframe(w // 2, h, 1, NB)
    t = timeit(lambda i: ops.yuv420p_to_rgb(Y[i], U2[i], V2[i], dst[i], w, h, is_422=1, lut=lut), NB)
    add("yuv422p->RGBA32 + gamma LUT", "colourspace.c:3593-3640, :14034", "1920x1080", w * h * 2 + w * h * 4, t, None)
    # ---- K6 gamma apply, K9 premult (in place) ---------------------------------------------------------------------------------
    pix = dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.gamma_apply(pix[i], w, h, 4, lut), NB)
    hp = hframe(w, h, 4)
    c = cpu(lambda: orc.orc_gamma_apply(P(hp), hp.strides[0], w, h, 4, 0, P(lut)))
    add("gamma_apply RGBA32 (in place)", "colourspace.c:14034-14060", "1920x1080", w * h * 8, t, c)
    t = timeit(lambda i: ops.alpha_premult(pix[i], w, h), NB)
    c = cpu(lambda: orc.orc_alpha_premult(P(hp), hp.strides[0], w, h, 0, 0))
    add("alpha_premult RGBA32 (in place)", "colourspace.c:11968-12105", "1920x1080", w * h * 8, t, c)
    # ---- K7 resize alone, K8 letterbox, F1 blend: C3 ----------------------------------------------------------------------------
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    src, dst = dframe(sw, sh, 4, NB), dframe(dw, dh, 4, NB)
    t = timeit(lambda i: ops.resize(src[i], dst[i], sw, sh, dw, dh), NB)
    hs, hd = hframe(sw, sh, 4), hframe(dw, dh, 4)
    if orc:
        orc.orc_resize.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5
    c = cpu(lambda: orc.orc_resize(P(hs), hs.strides[0], sw, sh, P(hd), hd.strides[0], dw, dh, 4, 3))
    add("resize bicubic 0.5x RGBA32 (C3)", "colourspace.c:14759-15328 (own spec)", "3840x2160->1920x1080", sw * sh * 4 + dw * dh * 4, t, c)
    canvas = dframe(1920, 1200, 4, NB)
    black = [0, 0, 0, 255]
    t = timeit(lambda i: ops.letterbox(dst[i], canvas[i], dw, dh, 1920, 1200, 4, black), NB)
    hc = hframe(1920, 1200, 4)
    blk = (ctypes.c_uint8 * 4)(*black)
    c = cpu(lambda: orc.orc_letterbox(P(hd), hd.strides[0], dw, dh, P(hc), hc.strides[0], 1920, 1200, 4, blk))
    add("letterbox into 1920x1200 (C3)", "colourspace.c:15343-15567", "1920x1080", dw * dh * 4 + 1920 * 1200 * 4, t, c)
    # what letterbox_layer does with a frame of another size: the scaler writes the inner frame straight into the canvas, the bars by their own kernel
    from lives_amd import lib as _lib
    blk4 = (ctypes.c_uint8 * 4)(*black)

    def fused(i):
        cv = canvas[i]
        _lib.call("lgpu_letterbox_bars", ops.dptr(cv), cv.stride(0), 1920, 1200, 4, blk4, 0, 60, dw, dh, ops.stream_ptr())
        _lib.call("lgpu_resize", ops.dptr(src[i]), src[i].stride(0), sw, sh, ops.dptr(cv, 60 * cv.stride(0)), cv.stride(0), dw, dh, 4, 3, None, ops.stream_ptr())
    t = timeit(fused, NB)
    add("resize 0.5x straight into the 1920x1200 canvas + bars (C3, letterbox_layer)", "colourspace.c:15343-15567", "3840x2160->1920x1200", sw * sh * 4 + 1920 * 1200 * 4, t, None)
    l1, l2, lo_ = dframe(1920, 1200, 4, NB), dframe(1920, 1200, 4, NB), dframe(1920, 1200, 4, NB)
    t = timeit(lambda i: ops.blend_chroma(l1[i], l2[i], lo_[i], 1920, 1200, 4, 128), NB)
    h1, h2, ho = hframe(1920, 1200, 4), hframe(1920, 1200, 4), hframe(1920, 1200, 4)
    c = cpu(lambda: orc.orc_blend_chroma(P(h1), h1.strides[0], P(h2), h2.strides[0], P(ho), ho.strides[0], 1920, 1200, 4, 0, 128))
    add("chroma blend RGBA32 (C3)", "simple_blend.c:58-150", "1920x1200", 1920 * 1200 * 12, t, c)
    # BASELINE config 3 as ONE launch on the pinned (gdk-pixbuf) arithmetic: resize 0.5x -> letterbox into 1920x1200 -> chroma blend bf = 128 (lgpu_chain_canvas)
    c3d = dframe(1920, 1200, 4, NB)
    for t_ in l2:
        a_ = t_[:, 3::4]
        a_[torch.rand(a_.shape, device="cuda", generator=g) < 0.5] = 255
    prm3 = ops.chain_params(sw, sh, src[0].stride(0), dw, dh, l2[0].stride(0), c3d[0].stride(0), swap_rb=0, interp=3 | 0x100, do_blur=0, bf=128, lut=None)
    trk3 = [ops.chain_tracks([src[i]], [l2[i]], [c3d[i]]) for i in range(NB)]
    t = timeit(lambda i: ops.chain_canvas(prm3, trk3[i], 1920, 1200, 0, 60), NB)
    add("C3 in one launch: resize 0.5x (gdk-pixbuf HYPER) -> letterbox 1920x1200 -> chroma blend (lgpu_chain_canvas)", "colourspace.c:15262-15322, :15343-15567, simple_blend.c:58-150",
        "3840x2160->1920x1200", sw * sh * 4 + 2 * 1920 * 1200 * 4, t, None)
    pbd = dframe(dw, dh, 4, NB)
    t = timeit(lambda i: ops.pixbuf_scale(src[i], pbd[i], sw, sh, dw, dh, channels=4, interp=3), NB)
    if orc:
        orc.orc_pixbuf_scale.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5
    c = cpu(lambda: orc.orc_pixbuf_scale(P(hs), hs.strides[0], sw, sh, P(hd), hd.strides[0], dw, dh, 4, 3))
    add("resize 0.5x RGBA32, gdk-pixbuf HYPER (lgpu_pixbuf_scale, pinned)", "colourspace.c:15262-15322", "3840x2160->1920x1080", sw * sh * 4 + dw * dh * 4, t, c)
    # ---- B1 gaussian, F4 colour key: C4 --------------------------------------------------------------------------------------------
    w, h = 3840, 2160
    src, dst = dframe(w, h, 4, NB), dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.gauss5(src[i], dst[i], w, h), NB)
    hs, hd = hframe(w, h, 4), hframe(w, h, 4)
    c = cpu(lambda: orc.orc_gauss5(P(hs), hs.strides[0], P(hd), hd.strides[0], w, h, 4))
    add("gauss5 RGBA32 (C4)", "(own spec)", "3840x2160", w * h * 8, t, c)
    # BASELINE config 4 as ONE launch: gaussian -> colour key, RGBA32 (the extension SURVEY 8d names: 2 x 33.2 MB read + 33.2 MB written) and RGB24 (the reference's palette)
    k2 = dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.gauss5_colorkey(src[i], k2[i], dst[i], w, h, 4, 0, 0.3, 0.8, (128, 128, 128)), NB)
    add("C4 in one launch: gauss5 -> colour key RGBA32 (lgpu_gauss5_colorkey)", "colorkey.script + own spec", "3840x2160", w * h * 12, t, None)
    a3, b3, o3 = dframe(w, h, 3, NB), dframe(w, h, 3, NB), dframe(w, h, 3, NB)
    t = timeit(lambda i: ops.gauss5_colorkey(a3[i], b3[i], o3[i], w, h, 3, 0, 0.3, 0.8, (128, 128, 128)), NB)
    add("C4 in one launch: gauss5 -> colour key RGB24 (lgpu_gauss5_colorkey)", "colorkey.script + own spec", "3840x2160", w * h * 9, t, None)
    t = timeit(lambda i: ops.colorkey(a3[i], b3[i], o3[i], w, h, 0, 0.2, 1.0, (0, 0, 255)), NB)
    ha, hb, ho = hframe(w, h, 3), hframe(w, h, 3), hframe(w, h, 3)
    if orc:
        orc.orc_colorkey.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    c = cpu(lambda: orc.orc_colorkey(P(ha), ha.strides[0], P(hb), hb.strides[0], P(ho), ho.strides[0], w, h, 0, 0.2, 1.0, 0, 0, 255, 0))
    add("colour key RGB24 (C4)", "scripts/colorkey.script", "3840x2160", w * h * 9, t, c)
    t = timeit(lambda i: ops.mirror(2, src[i], dst[i], w, h, 4), NB)
    c = cpu(lambda: orc.orc_mirror(2, P(hs), hs.strides[0], P(hd), hd.strides[0], w, h, 4))
    add("mirrorxy RGBA32", "mirrors.c:26-122", "3840x2160", w * h * 8, t, c)
    # ---- F6 stencils ------------------------------------------------------------------------------------------------------------
    w, h = 1920, 1080
    yp = [dframe(w, h, 1, NB), dframe(w // 2, h // 2, 1, NB), dframe(w // 2, h // 2, 1, NB)]
    yo = [dframe(w, h, 1, NB), dframe(w // 2, h // 2, 1, NB), dframe(w // 2, h // 2, 1, NB)]
    t = timeit(lambda i: ops.softlight([p[i] for p in yp], [p[i] for p in yo], w, h, 512, 1), NB)
    hs, hd = hframe(w, h, 1), hframe(w, h, 1)
    c = cpu(lambda: orc.orc_softlight_y(P(hs), hs.strides[0], P(hd), hd.strides[0], w, h, 1))
    add("softlight YUV420P", "softlight.c:62-151", "1920x1080", w * h * 3, t, c)
    src, dst = dframe(w, h, 4, NB), dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.edge(src[i], dst[i], w, h, 3, 0), NB)
    hs, hd = hframe(w, h, 4), hframe(w, h, 4)
    m16 = np.zeros(w * h, np.int16)
    c = cpu(lambda: orc.orc_edge(P(hs), hs.strides[0], P(hd), hd.strides[0], w, h, 3, 0, P(m16), 0))
    add("edge detect RGBA32 (mode 0)", "edge.c:129-248", "1920x1080", w * h * (4 + 2 + 2 + 4 + 4), t, c)
    bz = ops.Blurzoom(w, h, 3)
    t = timeit(lambda i: bz.process(src[i], dst[i], 0, 0), NB)
    if orc:
        orc.orc_blurzoom_new.restype = ctypes.c_void_p
        orc.orc_blurzoom_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        zo = orc.orc_blurzoom_new(w, h, 3)
    c = cpu(lambda: orc.orc_blurzoom_process(zo, P(hs), hs.strides[0], P(hd), hd.strides[0], 0, 0))
    add("blurzoom RGBA32 (mode 0)", "blurzoom.c:345-421", "1920x1080", w * h * (4 + 4 + 4 + 4 + 1) + w * h * 6, t, c)
    # ---- K4 / K3 / K5 -----------------------------------------------------------------------------------------------------------
    rgba = dframe(w, h, 4, NB)
    planes = [dframe(w, h, 1, NB), dframe(w // 2, h // 2, 1, NB), dframe(w // 2, h // 2, 1, NB)]
    t = timeit(lambda i: ops.rgb_to_yuv(rgba[i], [p[i] for p in planes], w, h, 0, 1, 4, 0, 0), NB)
    hs = hframe(w, h, 4)
    hpl = [hframe(w, h, 1), hframe(w // 2, h // 2, 1), hframe(w // 2, h // 2, 1)]
    pp, ss = po.planes_args(hpl)
    if orc:
        orc.orc_rgb_to_yuv.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int]
    c = cpu(lambda: orc.orc_rgb_to_yuv(P(hs), hs.strides[0], w, h, 0, 1, ctypes.addressof(pp), ctypes.addressof(ss), 4, 0, 0))
    add("RGBA32 -> YUV420P", "colourspace.c:6250-6322", "1920x1080", w * h * 4 * 3 // 2 + w * h * 3 // 2, t, c)
    uy = dframe(w // 2, h, 4, NB)
    t = timeit(lambda i: ops.rgb_to_yuv(rgba[i], [uy[i]], w, h, 0, 1, 2, 0, 0), NB)
    add("RGBA32 -> UYVY", "colourspace.c:5761-5830, :2162-2176", "1920x1080", w * h * 4 + w * h * 2, t, None)
    uy = dframe(w, h, 2, NB)
    t = timeit(lambda i: ops.yuv_to_rgb([uy[i]], rgba[i], w, h, 2, 0, 0, 1, 0), NB)
    hu2 = hframe(w, h, 2)
    sp, s2 = po.planes_args([hu2])
    if orc:
        orc.orc_yuv_to_rgb.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 4
    c = cpu(lambda: orc.orc_yuv_to_rgb(ctypes.addressof(sp), ctypes.addressof(s2), w, h, 2, 0, P(hs), hs.strides[0], 0, 1, 0))
    add("UYVY -> RGBA32", "colourspace.c:6616-6690", "1920x1080", w * h * 6, t, c)
    t = timeit(lambda i: ops.yuv_switch_clamping([p[i] for p in planes], 512, h, 1), NB)
    add("clamping switch YUV420P (in place)", "colourspace.c:10929-11090", "1920x1080", w * h * 3, t, None)
    # ---- compositor: 8 tracks 960x540 onto 1920x1080 ---------------------------------------------------------------------------------
    lay = [dframe(960, 540, 4, 1)[0] for _ in range(8)]
    out = dframe(w, h, 4, NB)
    layers = [(lay[z], 960, 540, 120 * z, 60 * z, 0.75) for z in range(8)]
    t = timeit(lambda i: ops.composite(out[i], w, h, 4, layers), NB)
    covered = sum(max(0, min(w, 120 * z + 960) - 120 * z) * max(0, min(h, 60 * z + 540) - 60 * z) for z in range(8))
    add("composite 8 x 960x540 layers", "compositor.c:120-293", "1920x1080", covered * 4 + w * h * 4, t, None)

    # ---- K5b: the decoder hand-off repacks ------------------------------------------------------------------------------------------
    w, h = 1920, 1080
    Y, U, V = dframe(w, h, 1, NB), [torch.randint(0, 256, (h // 2, w // 2), dtype=torch.uint8, device="cuda", generator=g) for _ in range(NB)], \
        [torch.randint(0, 256, (h // 2, w // 2), dtype=torch.uint8, device="cuda", generator=g) for _ in range(NB)]
    pk = dframe(w, h, 2, NB)
    t = timeit(lambda i: ops.yuv_repack(512, 564, [Y[i], U[i], V[i]], [pk[i]], w, h, 0), NB)
    add("YUV420P -> UYVY", "colourspace.c:7104-7150", "1920x1080", w * h * 3 // 2 + w * h * 2, t, None)
    p444 = [dframe(w, h, 1, NB) for _ in range(3)]
    p888 = dframe(w, h, 3, NB)
    t = timeit(lambda i: ops.yuv_repack(544, 588, [p444[0][i], p444[1][i], p444[2][i]], [p888[i]], w, h, 0), NB)
    add("YUV444P -> YUV888", "colourspace.c:7593-7641", "1920x1080", w * h * 6, t, None)
    # ---- more plugins: slide over, deinterlace, RGBdelay ------------------------------------------------------------------------------
    s1, s2, d = dframe(w, h, 4, NB), dframe(w, h, 4, NB), dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.slide_over(s1[i], s2[i], d[i], w, h, 4, 100, 1), NB)
    add("slide over RGBA32", "slide_over.c:54-146", "1920x1080", w * h * 8, t, None)
    t = timeit(lambda i: ops.deinterlace(s1[i], d[i], w, h, 3), NB)
    hs = hframe(w, h, 4)
    hd2 = hs.copy()
    if orc:
        orc.orc_deinterlace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    c = cpu(lambda: orc.orc_deinterlace(P(hs), hs.strides[0], P(hd2), hd2.strides[0], w, h, 3))
    add("deinterlace RGBA32", "deinterlace.c:45-308", "1920x1080", w * h * 8, t, c)
    r24, o24 = dframe(w, h, 3, NB), dframe(w, h, 3, NB)
    rd = ops.RgbDelay()
    on = np.zeros(153, np.int32)
    on[[0, 3 * 4 + 1, 3 * 8 + 2]] = 1
    st = np.ones(51)
    for i in range(12):
        rd.process(r24[i % NB], o24[i % NB], w, h, 1, 20, on, st)
    t = timeit(lambda i: rd.process(r24[i], o24[i], w, h, 1, 20, on, st), NB)
    add("RGBdelay RGB24 (default: 3 taps of a 9-frame ring)", "RGBdelay.c:135-416", "1920x1080", w * h * 3 * (1 + 1 + 3 + 1), t, None)
    rd.close()
    # ---- later additions: YUVA premultiply, YUV411 <-> RGB ------------------------------------------------------------------------------
    ya = dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.alpha_premult_yuva([ya[i]], w, h, 589, 1, un=0), NB)
    add("alpha_premult YUVA8888 clamped (in place, 64 KB tables)", "colourspace.c:12087-12096", "1920x1080", w * h * 8, t, None)
    m411 = [torch.randint(0, 256, (h, (w >> 2) * 6), dtype=torch.uint8, device="cuda", generator=g) for _ in range(NB)]
    o32 = dframe(w, h, 4, NB)
    t = timeit(lambda i: ops.yuv411_to_rgb(m411[i], o32[i], w >> 2, h, out_order=0, out_alpha=1), NB)
    add("YUV411 -> RGBA32", "colourspace.c:8305-8411", "1920x1080", w * h * 6 // 4 + w * h * 4, t, None)
    t = timeit(lambda i: ops.rgb_to_yuv411(o32[i], m411[i], w, h, in_order=0, in_alpha=1), NB)
    add("RGBA32 -> YUV411", "colourspace.c:6499-6540", "1920x1080", w * h * 6 // 4 + w * h * 4, t, None)

    print("GPU us = per launch, the smaller of (a) back-to-back launches from Python and (b) the same launches replayed from a HIP graph: a ctypes call costs ~8 us of host time, so (a) alone\nmeasures the host for every kernel shorter than that (the tables up to mid round 2 did).\n")
    print("| op | reference | size | algorithmic bytes | GPU us | GB/s | of 8 TB/s | oracle 1-thread ms | ratio |\n|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | `%s` | %s | %d | %.2f | %.1f | %.3f | %s | %s |" % (r["op"], r["reference"], r["size"], r["algorithmic_bytes"], r["gpu_us"], r["gbs"], r["frac"],
                                                                      "-" if r["cpu_ms"] is None else "%.3f" % r["cpu_ms"], "-" if r["speedup"] is None else "%.0fx" % r["speedup"]))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
