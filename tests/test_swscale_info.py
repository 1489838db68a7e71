"""Informational only, never gating: if the GPU box happens to carry FFmpeg's libswscale (what resize_layer calls, src/colourspace.c:14711;
flags :14991-14997), scale the same frame with SWS_BICUBIC and print PSNR / max |diff| against lgpu-polyphase-v1.  The resize spec of this
repository is its own (DESIGN.md section 5 lists how it differs from swscale's C path); parity with libswscale is UNPINNED because the library
is neither vendored nor version-pinned by the reference and is absent from the build image."""
import ctypes
import ctypes.util

import numpy as np
import pytest

from tests.util import dev, frame, host

pytestmark = pytest.mark.gpu
AV_PIX_FMT_RGBA, SWS_BICUBIC = 26, 4        # libavutil/pixfmt.h, libswscale/swscale.h


def find_swscale():
    names = [ctypes.util.find_library("swscale")] + ["libswscale.so.%d" % v for v in range(9, 3, -1)] + ["libswscale.so"]
    for n in names:
        if not n:
            continue
        try:
            return ctypes.CDLL(n)
        except OSError:
            continue
    return None


def test_informational_comparison_with_libswscale(gpu):
    sws = find_swscale()
    if sws is None:
        pytest.skip("libswscale is not on this box: parity of the resize spec stays unpinned (nothing to compare with)")
    rng = np.random.default_rng(9001)
    sw, sh, dw, dh = 640, 360, 320, 180
    # a smooth test card plus noise: what a decoded frame looks like more than white noise does
    yy, xx = np.mgrid[0:sh, 0:sw]
    base = np.stack([(xx * 255 // sw), (yy * 255 // sh), ((xx + yy) % 256), np.full_like(xx, 255)], axis=2).astype(np.int32)
    src = np.clip(base + rng.integers(-12, 13, base.shape), 0, 255).astype(np.uint8).reshape(sh, sw * 4)
    sws.sws_getContext.restype = ctypes.c_void_p
    sws.sws_getContext.argtypes = [ctypes.c_int] * 3 + [ctypes.c_int] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    ctx = sws.sws_getContext(sw, sh, AV_PIX_FMT_RGBA, dw, dh, AV_PIX_FMT_RGBA, SWS_BICUBIC, None, None, None)
    if not ctx:
        pytest.skip("sws_getContext refused RGBA -> RGBA")
    out = np.zeros((dh, dw * 4), np.uint8)
    srcp = (ctypes.c_void_p * 4)(src.ctypes.data, None, None, None)
    dstp = (ctypes.c_void_p * 4)(out.ctypes.data, None, None, None)
    sst = (ctypes.c_int * 4)(sw * 4, 0, 0, 0)
    dst = (ctypes.c_int * 4)(dw * 4, 0, 0, 0)
    sws.sws_scale.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert sws.sws_scale(ctx, srcp, sst, 0, sh, dstp, dst) == dh
    sws.sws_freeContext.argtypes = [ctypes.c_void_p]
    sws.sws_freeContext(ctx)
    d = dev(np.zeros((dh, dw * 4), np.uint8))
    gpu.resize(dev(src), d, sw, sh, dw, dh, psize=4, interp=3)
    got = host(d).astype(np.int32)
    diff = np.abs(got - out.astype(np.int32)).reshape(dh, dw, 4)[:, :, :3]
    mse = float((diff.astype(np.float64) ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    print("\n[informational] lgpu-polyphase-v1 vs libswscale SWS_BICUBIC %dx%d -> %dx%d RGBA: PSNR %.2f dB, max |diff| %d, mean |diff| %.3f"
          % (sw, sh, dw, dh, psnr, int(diff.max()), float(diff.mean())))
