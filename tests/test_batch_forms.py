"""The batch forms of the CONVERT-step kernels and single-plane effects (include/lives_gpu.h: lgpu_swizzle_batch, lgpu_gamma_apply_batch, lgpu_alpha_premult_batch,
lgpu_mirror_batch, lgpu_letterbox_batch, lgpu_colorkey_batch, lgpu_rgb_to_yuv_batch, lgpu_yuv_to_rgb_batch): n frames of one geometry in one launch.
Every frame equals the single-frame entry point bit for bit (slot order shuffled, guard row behind every output, misaligned frames included), and two slots
of every batch are compared with the oracle directly."""
import ctypes

import numpy as np
import pytest

from lives_amd import lib
from oracle import pyoracle as po
from tests.util import align, dev, frame, host

pytestmark = pytest.mark.gpu
P = po.P
vp = ctypes.c_void_p


def ptrs(ts, off=0):
    return (vp * len(ts))(*[t.data_ptr() + off for t in ts])


def guarded(arr):
    g = np.full((arr.shape[0] + 1, arr.shape[1]), 0xA5, np.uint8)
    g[:-1] = arr
    return dev(g)


NS = (1, 3, 16)


@pytest.mark.parametrize("op,ib,ob", [("swap3postalpha", 4, 4), ("swap3addpost", 3, 4), ("delpost", 4, 3), ("swap3", 3, 3)])
def test_swizzle_batch(gpu, orc, op, ib, ob):
    rng = np.random.default_rng(0xBA00 + ib * 8 + ob)
    w, h, opi = 150, 37, po.OPS.index(op)
    lut = rng.permutation(256).astype(np.uint8)
    for n in NS:
        for off in (0, 4):                       # 4: frames that are 4- but not 16-byte aligned take the byte form -- for the whole batch
            srcs = [frame(rng, w + 2, h, ib) for _ in range(n)]
            d_s = [dev(a) for a in srcs]
            irow, orow = srcs[0].strides[0], align(w * ob + 16, 16)
            outs = [guarded(np.zeros((h, orow), np.uint8)) for _ in range(n)]
            order = list(rng.permutation(n))
            lib.call("lgpu_swizzle_batch", opi, 0, ptrs([d_s[i] for i in order], off), irow, ptrs([outs[i] for i in order], off), orow, w, h, lut.ctypes.data, n, None)
            for f in range(n):
                one = guarded(np.zeros((h, orow), np.uint8))
                lib.call("lgpu_swizzle", opi, 0, d_s[f].data_ptr() + off, irow, one.data_ptr() + off, orow, w, h, lut.ctypes.data, None)
                assert (host(one) == host(outs[f])).all(), (n, f, off)
            for f in (0, n - 1):
                want = np.zeros((h, orow), np.uint8)
                flat = srcs[f].reshape(-1)[off:]
                orc.orc_swizzle(opi, 0, flat.ctypes.data, irow, P(want), orow, w, h, P(lut))
                got = host(outs[f]).reshape(-1)[off:off + (h - 1) * orow + w * ob].copy()
                assert (got == want.reshape(-1)[:got.size]).all(), (n, f, off)


@pytest.mark.parametrize("psize,af", [(4, 0), (4, 1), (3, 0)])
def test_gamma_and_premult_batch(gpu, orc, psize, af):
    rng = np.random.default_rng(0xBA10 + psize + af)
    w, h = 130, 29
    lut = rng.permutation(256).astype(np.uint8)
    for n in NS:
        base = [frame(rng, w, h, psize, alpha_mix=(psize == 4)) for _ in range(n)]
        a, b = [guarded(x) for x in base], [guarded(x) for x in base]
        rs = base[0].strides[0]
        lib.call("lgpu_gamma_apply_batch", ptrs(a), rs, 3, 2, w - 7, h - 5, psize, af, lut.ctypes.data, n, None)
        for f in range(n):
            lib.call("lgpu_gamma_apply", b[f].data_ptr(), rs, 3, 2, w - 7, h - 5, psize, af, lut.ctypes.data, None)
            assert (host(a[f]) == host(b[f])).all(), (n, f)
        for f in (0, n - 1):
            want = base[f].copy()
            sub = want[2:, 3 * psize:]
            orc.orc_gamma_apply(sub.ctypes.data, rs, w - 7, h - 5, psize, af, P(lut))
            assert (host(a[f])[:h] == want).all()
        if psize == 4:
            for un in (0, 1):
                a, b = [guarded(x) for x in base], [guarded(x) for x in base]
                lib.call("lgpu_alpha_premult_batch", ptrs(a), rs, w, h, af, un, n, None)
                for f in range(n):
                    lib.call("lgpu_alpha_premult", b[f].data_ptr(), rs, w, h, af, un, None)
                    assert (host(a[f]) == host(b[f])).all(), (n, f, un)
                for f in (0, n - 1):
                    want = base[f].copy()
                    orc.orc_alpha_premult(P(want), rs, w, h, af, un)
                    assert (host(a[f])[:h] == want).all()


@pytest.mark.parametrize("psize", [3, 4])
def test_mirror_letterbox_colorkey_batch(gpu, orc, psize):
    rng = np.random.default_rng(0xBA20 + psize)
    w, h, nw, nh = 101, 33, 140, 48
    black = (ctypes.c_uint8 * 4)(0, 0, 0, 255)
    for n in NS:
        srcs = [frame(rng, w, h, psize) for _ in range(n)]
        d_s = [dev(x) for x in srcs]
        rs = srcs[0].strides[0]
        for mode in (0, 1, 2):
            outs = [guarded(np.zeros((h, rs), np.uint8)) for _ in range(n)]
            order = list(rng.permutation(n))
            lib.call("lgpu_mirror_batch", mode, ptrs([d_s[i] for i in order]), rs, ptrs([outs[i] for i in order]), rs, w, h, psize, n, None)
            inpl = [guarded(x) for x in srcs]                                    # and in place (mirrors.c works in place)
            lib.call("lgpu_mirror_batch", mode, ptrs(inpl), rs, ptrs(inpl), rs, w, h, psize, n, None)
            for f in range(n):
                one = guarded(np.zeros((h, rs), np.uint8))
                lib.call("lgpu_mirror", mode, d_s[f].data_ptr(), rs, one.data_ptr(), rs, w, h, psize, None)
                assert (host(one) == host(outs[f])).all(), (mode, n, f)
                assert (host(inpl[f])[:h, :w * psize] == host(one)[:h, :w * psize]).all()
            for f in (0, n - 1):
                want = np.zeros((h, rs), np.uint8)
                orc.orc_mirror(mode, P(srcs[f]), rs, P(want), rs, w, h, psize)
                assert (host(outs[f])[:h, :w * psize] == want[:, :w * psize]).all()
        crs = align(nw * psize)
        outs = [guarded(np.full((nh, crs), 7, np.uint8)) for _ in range(n)]
        lib.call("lgpu_letterbox_batch", ptrs(d_s), rs, w, h, ptrs(outs), crs, nw, nh, psize, black, n, None)
        for f in range(n):
            one = guarded(np.full((nh, crs), 7, np.uint8))
            lib.call("lgpu_letterbox", d_s[f].data_ptr(), rs, w, h, one.data_ptr(), crs, nw, nh, psize, black, None)
            assert (host(one) == host(outs[f])).all(), (n, f)
        for f in (0, n - 1):
            want = np.full((nh, crs), 7, np.uint8)
            orc.orc_letterbox(P(srcs[f]), rs, w, h, P(want), crs, nw, nh, psize, black)
            assert (host(outs[f])[:nh] == want).all()
        if psize == 3:
            s2 = [frame(rng, w, h, 3) for _ in range(n)]
            d2 = [dev(x) for x in s2]
            outs = [guarded(np.zeros((h, rs), np.uint8)) for _ in range(n)]
            args = (w, h, 0, 0.35, 0.7, 40, 200, 90)
            lib.call("lgpu_colorkey_batch", ptrs(d_s), rs, ptrs(d2), rs, ptrs(outs), rs, *args, n, None)
            for f in range(n):
                one = guarded(np.zeros((h, rs), np.uint8))
                lib.call("lgpu_colorkey", d_s[f].data_ptr(), rs, d2[f].data_ptr(), rs, one.data_ptr(), rs, *args, None)
                assert (host(one) == host(outs[f])).all(), (n, f)
            for f in (0, n - 1):
                want = np.zeros((h, rs), np.uint8)
                orc.orc_colorkey(P(srcs[f]), rs, P(s2[f]), rs, P(want), rs, w, h, 0, 0.35, 0.7, 40, 200, 90, 0)
                assert (host(outs[f])[:h, :w * 3] == want[:, :w * 3]).all()


@pytest.mark.parametrize("fmt", [0, 2, 4, 5])
def test_rgb_to_yuv_and_back_batch(gpu, fmt):
    """K4 / K3: the planar side as a table of n x 4 plane pointers; every frame against the single-frame entry point (which tests/test_gpu_parity.py and
    tests/test_gpu_golden.py hold against the oracle and the reference's fixtures)"""
    rng = np.random.default_rng(0xBA30 + fmt)
    w, h = 128, 36
    for n in NS:
        srcs = [frame(rng, w, h, 4) for _ in range(n)]
        d_s = [dev(x) for x in srcs]
        rs = srcs[0].strides[0]
        if fmt == 0:
            dims = [(h, align(w * 3))]
        elif fmt == 2:
            dims = [(h, align(w * 2))]
        else:
            dims = [(h, align(w))] + [((h >> 1) if fmt == 4 else h, align(w) >> 1)] * 2
        outs = [[guarded(np.zeros(d, np.uint8)) for d in dims] for _ in range(n)]
        tab = (vp * (4 * n))()
        for f in range(n):
            for k, t in enumerate(outs[f]):
                tab[4 * f + k] = t.data_ptr()
        orow = (ctypes.c_int * 4)(*([d[1] for d in dims] + [0] * (4 - len(dims))))
        lib.call("lgpu_rgb_to_yuv_batch", ptrs(d_s), rs, w, h, 1, 1, tab, orow, fmt, 0, 0, n, None)
        for f in range(n):
            one = [guarded(np.zeros(d, np.uint8)) for d in dims]
            gpu.rgb_to_yuv(d_s[f], [t[:-1] for t in one], w, h, 1, 1, fmt, 0, 0)
            for k in range(len(dims)):
                assert (host(one[k]) == host(outs[f][k])).all(), (fmt, n, f, k)
        if fmt in (0, 2):                      # back: packed 4:4:4 / UYVY -> RGBA32
            backs = [guarded(np.zeros((h, rs), np.uint8)) for _ in range(n)]
            irow = (ctypes.c_int * 4)(dims[0][1], 0, 0, 0)
            lib.call("lgpu_yuv_to_rgb_batch", tab, irow, w, h, fmt, 0, ptrs(backs), rs, 0, 1, 0, n, None)
            for f in range(n):
                one = guarded(np.zeros((h, rs), np.uint8))
                gpu.yuv_to_rgb([outs[f][0][:-1]], one[:-1], w, h, fmt, 0, 0, 1, 0)
                assert (host(one) == host(backs[f])).all(), (fmt, n, f)


def test_batch_forms_refuse_bad_tables(gpu):
    d = dev(np.zeros((8, 64), np.uint8))
    one = ptrs([d])
    assert lib.load().lgpu_swizzle_batch(4, 0, one, 64, one, 64, 16, 8, None, 0, None) == -2          # 0 frames
    assert lib.load().lgpu_swizzle_batch(4, 0, one, 64, one, 64, 16, 8, None, 17, None) == -2         # more than LGPU_FX_MAX_FRAMES
    null = (vp * 2)(d.data_ptr(), None)
    assert lib.load().lgpu_gamma_apply_batch(null, 64, 0, 0, 16, 8, 4, 0, np.zeros(256, np.uint8).ctypes.data, 2, None) == -2
