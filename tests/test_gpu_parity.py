"""GPU parity: every HIP entry point of liblivesgpu.so against the CPU oracle on the same seeded inputs.

Bit-exact (integer / byte work).  Masked pixels are exactly the ones DESIGN.md lists as undefined in the
reference.  Everything goes through the C ABI (lives_amd.lib via lives_amd.ops).
"""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, assert_padding_untouched, assert_same, dev, frame, host

pytestmark = pytest.mark.gpu

P = po.P
SIZES = [(64, 32), (66, 34), (130, 18), (7, 5), (1, 1), (640, 480)]


def lut_for(rng, kind):
    if kind == "none":
        return None
    if kind == "l2s":
        lut = np.zeros(256, np.uint8)
        assert po.oracle().orc_gamma_lut8(1.0, po.GAMMA_LINEAR, po.GAMMA_SRGB, 1.4, P(lut)) == 1
        return lut
    return rng.integers(0, 256, 256, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------- K1
@pytest.mark.parametrize("op", range(13))
@pytest.mark.parametrize("lutkind", ["none", "rand"])
def test_swizzle(gpu, orc, op, lutkind):
    rng = np.random.default_rng(100 + op)
    for (w, h) in SIZES:
        for af in ((0, 1) if op in (po.OPS.index("swap4"), po.OPS.index("swapprepost")) else (0,)):
            ib, ob = po.OP_IBPP[op], po.OP_OBPP[op]
            lut = lut_for(rng, lutkind)
            src = frame(rng, w, h, ib, extra_rows=1)
            want = np.full((h + 1, align(w * ob)), 0xAB, np.uint8)
            orc.orc_swizzle(op, af, P(src), src.strides[0], P(want), want.strides[0], w, h, P(lut))
            d_src, d_dst = dev(src), dev(np.full_like(want, 0xAB))
            gpu.swizzle(op, d_src, d_dst, w, h, alpha_first=af, lut=lut)
            got = host(d_dst)
            assert_same(got, want, w, h, ob, "%s %dx%d af=%d" % (po.OPS[op], w, h, af))
            assert_padding_untouched(got, np.full_like(want, 0xAB), w, h, ob, po.OPS[op])
            if ib == ob:   # in place, as convert_layer_palette_full does when it may (src/colourspace.c:12392)
                d_io = dev(src)
                gpu.swizzle(op, d_io, d_io, w, h, alpha_first=af, lut=lut)
                assert_same(host(d_io), want, w, h, ob, "%s in place" % po.OPS[op])


def test_swizzle_unaligned_rows(gpu, orc):
    """compact (-1 alignment hint) strides and odd base addresses take the byte path"""
    rng = np.random.default_rng(7)
    w, h = 37, 9
    for op in (po.OPS.index("swap3addpost"), po.OPS.index("swap3delpost"), po.OPS.index("swap3"), po.OPS.index("swap3postalpha")):
        ib, ob = po.OP_IBPP[op], po.OP_OBPP[op]
        src = frame(rng, w, h, ib, stride=w * ib + 1)
        want = np.zeros((h, w * ob + 3), np.uint8)
        orc.orc_swizzle(op, 0, P(src), src.strides[0], P(want), want.strides[0], w, h, None)
        d_dst = dev(np.zeros_like(want))
        gpu.swizzle(op, dev(src), d_dst, w, h)
        assert_same(host(d_dst), want, w, h, ob, po.OPS[op] + " unaligned")


# ---------------------------------------------------------------------------------------------- K6
@pytest.mark.parametrize("psize,af", [(3, 0), (4, 0), (4, 1)])
def test_gamma_apply(gpu, orc, psize, af):
    rng = np.random.default_rng(200 + psize + af)
    lut = lut_for(rng, "l2s")
    for (w, h) in SIZES:
        pix = frame(rng, w, h, psize, extra_rows=1)
        want = pix.copy()
        orc.orc_gamma_apply(P(want), want.strides[0], w, h, psize, af, P(lut))
        d = dev(pix)
        gpu.gamma_apply(d, w, h, psize, lut, alpha_first=af)
        got = host(d)
        assert_same(got, want, w, h, psize, "gamma %dx%d" % (w, h))
        assert_padding_untouched(got, pix, w, h, psize, "gamma")


def test_gamma_apply_subrect(gpu, orc):
    rng = np.random.default_rng(201)
    lut = lut_for(rng, "rand")
    w, h = 66, 34
    for psize in (3, 4):
        pix = frame(rng, w, h, psize)
        x, y, rw, rh = 5, 3, 41, 17
        want = pix.copy()
        sub = want[y:, x * psize:]
        orc.orc_gamma_apply(ctypes.c_void_p(want.ctypes.data + y * want.strides[0] + x * psize), want.strides[0], rw, rh, psize, 0, P(lut))
        d = dev(pix)
        gpu.gamma_apply(d, rw, rh, psize, lut, x=x, y=y)
        assert (host(d) == want).all(), "sub-rectangle gamma psize %d" % psize
        del sub


# ---------------------------------------------------------------------------------------------- K9
@pytest.mark.parametrize("af", [0, 1])
@pytest.mark.parametrize("un", [0, 1])
def test_alpha_premult_all_pairs(gpu, orc, af, un):
    """every (alpha, value) pair of the reference's 256x256 tables"""
    w, h = 256, 256
    pix = np.zeros((h, w * 4), np.uint8)
    a = np.arange(256, dtype=np.uint8)
    for c in range(4):
        pix[:, c::4] = a[None, :]
    pix[:, (0 if af else 3)::4] = a[:, None]   # alpha = row index
    want = pix.copy()
    orc.orc_alpha_premult(P(want), want.strides[0], w, h, af, un)
    d = dev(pix)
    gpu.alpha_premult(d, w, h, alpha_first=af, un=un)
    assert_same(host(d), want, w, h, 4, "premult af=%d un=%d" % (af, un))


def test_premult_yuv_tables_match_the_fixture(gpu):
    from tests import golden_util as gu
    g = gu.load("premult_yuv.npz")
    tabs = [np.zeros((256, 256), np.uint8) for _ in range(4)]
    gpu.lib.call("lgpu_premult_yuv_tables", *[t.ctypes.data for t in tabs])
    for name, t in zip(("unalcy", "alcy", "unalcuv", "alcuv"), tabs):
        assert (t == g[name]).all(), name


def test_premult_yuv_tables_as_the_device_evaluates_them(gpu):
    """lgpu_alpha_premult_yuva is table-free since round 4: the integer form of alcy / unalcy and the float product of alcuv / unalcuv, evaluated by the kernel's own
    device functions for all 4 x 65,536 (alpha, value) pairs, against the reference's tables (fixture made from the colourspace.c slice)"""
    from tests import golden_util as gu
    g = gu.load("premult_yuv.npz")
    out = np.zeros((4, 256, 256), np.uint8)
    gpu.lib.call("lgpu_debug_premult_yuv_tables_device", out.ctypes.data)
    for k, name in enumerate(("unalcy", "alcy", "unalcuv", "alcuv")):
        bad = np.argwhere(out[k] != g[name])
        assert len(bad) == 0, "%s: %d entries differ, first (alpha, value) = %s" % (name, len(bad), bad[0].tolist())


@pytest.mark.parametrize("pal", [589, 545])
@pytest.mark.parametrize("clamped", [0, 1])
@pytest.mark.parametrize("un", [0, 1])
def test_alpha_premult_yuva(gpu, orc, pal, clamped, un):
    """alpha_premult on YUVA8888 / YUVA4444P (src/colourspace.c:12005-12047, :12063-12096): every (alpha, value) pair in each of Y, U, V,
    plus a random padded frame"""
    a = np.arange(256, dtype=np.uint8)
    rng = np.random.default_rng(pal + 2 * clamped + un)
    for w, h, pad, sweep in ((256, 256, 0, True), (61, 19, 12, False)):
        if pal == 589:
            fr = rng.integers(0, 256, (h, w * 4 + pad), dtype=np.uint8)
            if sweep:
                for c in range(3):
                    fr[:, c:w * 4:4] = a[None, :]
                fr[:, 3:w * 4:4] = a[:, None]
            planes = [fr]
        else:
            planes = [rng.integers(0, 256, (h, w + pad), dtype=np.uint8) for _ in range(4)]
            if sweep:
                for c in range(3):
                    planes[c][:, :w] = a[None, :]
                planes[3][:, :w] = a[:, None]
        want = [p.copy() for p in planes]
        pp = (ctypes.c_void_p * 4)(*([x.ctypes.data for x in want] + [None] * (4 - len(want))))
        ss = (ctypes.c_int * 4)(*([x.strides[0] for x in want] + [0] * (4 - len(want))))
        orc.orc_alpha_premult_yuva(pp, ss, w, h, pal, clamped, un)
        ds = [dev(p) for p in planes]
        gpu.alpha_premult_yuva(ds, w, h, pal, clamped, un=un)
        for i, (d, wt) in enumerate(zip(ds, want)):
            bw = w * 4 if pal == 589 else w
            assert_same(host(d), wt, bw, h, 1, "premult yuva pal=%d clamped=%d un=%d plane %d" % (pal, clamped, un, i))
            assert (host(d)[:, bw:] == planes[i][:, bw:]).all()       # padding untouched


# ---------------------------------------------------------------------------------------------- K3b
@pytest.mark.parametrize("order,oa", [(0, 0), (0, 1), (1, 0), (1, 1), (2, 1)])
@pytest.mark.parametrize("uncl", [0, 1])
def test_yuv411_to_rgb(gpu, orc, order, oa, uncl):
    """YUV411 -> RGB family against the oracle: ragged widths (1 macropixel too), padded rows, unwritten alpha bytes and padding kept"""
    rng = np.random.default_rng(411 + order * 4 + oa * 2 + uncl)
    ps = 4 if (order == 2 or oa) else 3
    for wm, h, pad in ((1, 1, 0), (2, 5, 4), (67, 9, 8), (480, 270, 0), (257, 3, 16)):
        src = rng.integers(0, 256, (h, wm * 6), dtype=np.uint8)
        init = rng.integers(0, 256, (h, wm * 4 * ps + pad), dtype=np.uint8)
        want = init.copy()
        assert orc.orc_yuv411_to_rgb(P(src), wm, h, P(want), want.strides[0], order, oa, uncl) == 0
        d = dev(init)
        gpu.yuv411_to_rgb(dev(src), d, wm, h, out_order=order, out_alpha=oa, unclamped=uncl)
        assert (host(d) == want).all(), (wm, h, pad)


@pytest.mark.parametrize("order,ia", [(0, 0), (0, 1), (1, 0), (1, 1), (2, 1)])
@pytest.mark.parametrize("uncl", [0, 1])
def test_rgb_to_yuv411(gpu, orc, order, ia, uncl):
    """RGB family -> YUV411 against the oracle; widths that are not multiples of 4 lose their right edge, padded source rows"""
    rng = np.random.default_rng(114 + order * 4 + ia * 2 + uncl)
    ips = 4 if ia else 3
    for w, h, pad in ((4, 1, 0), (7, 5, 4), (270, 9, 8), (1920, 270, 0), (1023, 3, 16)):
        src = rng.integers(0, 256, (h, w * ips + pad), dtype=np.uint8)
        want = np.full((h, (w >> 2) * 6), 0x5A, np.uint8)
        assert orc.orc_rgb_to_yuv411(P(src), src.strides[0], w, h, order, ia, P(want), uncl) == 0
        d = dev(np.full_like(want, 0x5A))
        gpu.rgb_to_yuv411(dev(src), d, w, h, in_order=order, in_alpha=ia, unclamped=uncl)
        assert (host(d) == want).all(), (w, h, pad)


def test_yuv411_round_trip_property(gpu):
    """size-independent property at 1080p: grey RGB -> YUV411 -> RGB returns the grey ramp within the table rounding, and chroma is neutral"""
    w, h = 1920, 1080
    ramp = np.repeat(np.arange(w, dtype=np.int64) * 255 // (w - 1), 3).astype(np.uint8)
    src = np.tile(ramp, (h, 1))
    d411 = dev(np.zeros((h, (w >> 2) * 6), np.uint8))
    gpu.rgb_to_yuv411(dev(src), d411, w, h, unclamped=1)
    m = host(d411).reshape(h, w >> 2, 6)
    assert (np.abs(m[:, :, 0].astype(int) - 128) <= 1).all() and (np.abs(m[:, :, 3].astype(int) - 128) <= 1).all()
    back = dev(np.zeros((h, w * 3), np.uint8))
    gpu.yuv411_to_rgb(d411, back, w >> 2, h, unclamped=1)
    assert np.abs(host(back).astype(int) - src.astype(int)).max() <= 2
    assert (host(back) == host(back)[0]).all()                 # every row identical


# ---------------------------------------------------------------------------------------------- K2
def k2_mask(w, h, is_422):
    m = np.zeros((h, w), bool)
    if not is_422:
        m[0, 1::2] = True        # reference indexes its tables out of bounds here (undefined)
        if h % 2 == 0:
            m[h - 1, 1::2] = True    # never written by the 1-thread reference
    return m


@pytest.mark.parametrize("which", range(4))
@pytest.mark.parametrize("opsize", [3, 4])
@pytest.mark.parametrize("quality", [1, 2, 3])      # 3 = HIGH: the oracle takes the float path of _spc_rnd, the GPU the shift
def test_yuv420p_to_rgb(gpu, orc, which, opsize, quality):
    rng = np.random.default_rng(300 + which * 10 + opsize)
    for (w, h, ys, cs) in [(64, 32, 64, 32), (66, 34, 96, 48), (130, 18, 160, 80), (2, 2, 32, 16), (640, 480, 640, 320)]:
        for lutkind in ("none", "l2s"):
            lut = lut_for(rng, lutkind)
            Y = rng.integers(0, 256, (h, ys), dtype=np.uint8)
            U = rng.integers(0, 256, (h // 2, cs), dtype=np.uint8)
            V = rng.integers(0, 256, (h // 2, cs), dtype=np.uint8)
            orow = align(w * opsize)
            strides = (ctypes.c_int * 3)(ys, cs, cs)
            for fix in (0, 1):
                want = np.full((h, orow), 0xAB, np.uint8)
                orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, U.size, V.size, P(want), orow, w, h, opsize, 0, 0, which, quality, P(lut), fix)
                d = dev(np.full_like(want, 0xAB))
                gpu.yuv420p_to_rgb(dev(Y), dev(U), dev(V), d, w, h, opsize=opsize, which_tables=which, pb_quality=quality, lut=lut,
                                   flags=gpu.lib.YUV_FIX_EDGES if fix else 0)
                got = host(d)
                # the oracle writes the same "intent" values into the undefined pixels, so compare everything
                assert_same(got, want, w, h, opsize, "yuv420p %dx%d which=%d fix=%d" % (w, h, which, fix))
                assert_padding_untouched(got, np.full_like(want, 0xAB), w, h, opsize, "yuv420p")


def test_yuv422p_and_orders(gpu, orc):
    rng = np.random.default_rng(333)
    w, h = 66, 34
    Y = rng.integers(0, 256, (h, 96), dtype=np.uint8)
    U = rng.integers(0, 256, (h, 48), dtype=np.uint8)
    V = rng.integers(0, 256, (h, 48), dtype=np.uint8)
    strides = (ctypes.c_int * 3)(96, 48, 48)
    for order in (0, 1, 2):
        for opsize in (3, 4):
            if order == 2 and opsize == 3:
                continue
            for is422 in (0, 1):
                u, v = (U, V) if is422 else (U[:h // 2], V[:h // 2])
                orow = align(w * opsize)
                want = np.zeros((h, orow), np.uint8)
                orc.orc_yuv420p_to_rgb(P(Y), P(u), P(v), strides, u.size, v.size, P(want), orow, w, h, opsize, order, is422, 0, 2, None, 0)
                d = dev(np.zeros_like(want))
                gpu.yuv420p_to_rgb(dev(Y), dev(u), dev(v), d, w, h, opsize=opsize, out_order=order, is_422=is422)
                assert_same(host(d), want, w, h, opsize, "yuv order=%d ops=%d 422=%d" % (order, opsize, is422))


# ---------------------------------------------------------------------------------------------- K8
@pytest.mark.parametrize("psize", [1, 3, 4])
def test_letterbox(gpu, orc, psize):
    rng = np.random.default_rng(400 + psize)
    black = {1: [16, 0, 0, 0], 3: [0, 0, 0, 0], 4: [0, 0, 0, 255]}[psize]
    for (w, h, nw, nh) in [(64, 32, 64, 40), (50, 30, 66, 34), (64, 32, 80, 32), (1920 // 8, 1080 // 8, 1920 // 8, 1200 // 8), (5, 3, 8, 8)]:
        src = frame(rng, w, h, psize)
        before = np.full((nh + 1, align(nw * psize) + 32), 0x77, np.uint8)
        want = before.copy()
        bp = np.array(black, np.uint8)
        orc.orc_letterbox(P(src), src.strides[0], w, h, P(want), want.strides[0], nw, nh, psize, P(bp))
        d = dev(before)
        gpu.letterbox(dev(src), d, w, h, nw, nh, psize, black)
        got = host(d)
        assert_same(got, want, nw, nh, psize, "letterbox %dx%d in %dx%d" % (w, h, nw, nh))
        assert_padding_untouched(got, before, nw, nh, psize, "letterbox")


# ---------------------------------------------------------------------------------------------- F1..F5
PALS = {1: (3, 0, 0), 2: (3, 1, 0), 3: (4, 0, 0), 4: (4, 1, 0), 5: (4, 2, 1)}   # weed palette -> psize, order, alpha_first


@pytest.mark.parametrize("pal", [1, 2, 3, 4, 5])
def test_blend_chroma(gpu, orc, pal):
    ps, order, af = PALS[pal]
    rng = np.random.default_rng(500 + pal)
    for (w, h) in SIZES:
        for bf in (0, 1, 100, 128, 200, 255):
            s1 = frame(rng, w, h, ps, extra_rows=1, alpha_mix=True)
            s2 = frame(rng, w, h, ps, extra_rows=1, alpha_mix=True, pad_px=1)
            for inplace in (0, 1):
                init = s1.copy() if inplace else np.full_like(s1, 0x5A)
                want = init.copy()
                src1 = want if inplace else s1
                orc.orc_blend_chroma(P(src1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, ps, af, bf)
                d1 = dev(s1)
                dd = d1 if inplace else dev(init)
                gpu.blend_chroma(d1, dev(s2), dd, w, h, ps, bf, alpha_first=af)
                got = host(dd)
                assert_same(got, want, w, h, ps, "chroma pal=%d bf=%d %dx%d inplace=%d" % (pal, bf, w, h, inplace))
                assert_padding_untouched(got, init, w, h, ps, "chroma blend")


@pytest.mark.parametrize("pal", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", [1, 2, 3, 4])
def test_blend_luma(gpu, orc, pal, kind):
    ps, order, af = PALS[pal]
    rng = np.random.default_rng(600 + pal * 10 + kind)
    for (w, h) in SIZES[:4]:
        for thr in (0, 1, 64, 128, 255):
            s1, s2 = frame(rng, w, h, ps), frame(rng, w, h, ps)
            for inplace in (0, 1):
                init = s1.copy() if inplace else np.full_like(s1, 0x5A)
                want = init.copy()
                orc.orc_blend_luma(kind, P(want if inplace else s1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, ps, order, thr, inplace)
                d1 = dev(s1)
                dd = d1 if inplace else dev(init)
                gpu.blend_luma(kind, d1, dev(s2), dd, w, h, ps, order, thr)
                assert_same(host(dd), want, w, h, ps, "luma kind=%d pal=%d thr=%d inplace=%d" % (kind, pal, thr, inplace))


@pytest.mark.parametrize("kind", range(7))
def test_blend_multi(gpu, orc, kind):
    rng = np.random.default_rng(700 + kind)
    for (w, h) in SIZES[:5]:
        for is_bgr in (0, 1):
            for bf in (0, 1, 100, 127, 128, 200, 255):
                s1, s2 = frame(rng, w, h, 3), frame(rng, w, h, 3)
                if w >= 2:
                    s1[0, :6] = [0, 255, 1, 254, 0, 255]
                    s2[0, :6] = [255, 0, 254, 1, 0, 255]
                want = np.full_like(s1, 0x5A)
                orc.orc_blend_multi(kind, P(s1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, is_bgr, bf)
                d = dev(np.full_like(s1, 0x5A))
                gpu.blend_multi(kind, dev(s1), dev(s2), d, w, h, is_bgr, bf)
                assert_same(host(d), want, w, h, 3, "multi kind=%d bgr=%d bf=%d" % (kind, is_bgr, bf))


def test_colorkey(gpu, orc):
    rng = np.random.default_rng(800)
    for (w, h) in SIZES[:4]:
        for is_bgr in (0, 1):
            for delta in (0.0, 0.2, 0.5, 1.0):
                for opac in (0.0, 0.3, 0.77, 1.0):
                    for col in ((0, 0, 255), (10, 200, 30), (128, 128, 128)):
                        s0, s1 = frame(rng, w, h, 3), frame(rng, w, h, 3)
                        want = np.full_like(s0, 0x5A)
                        orc.orc_colorkey(P(s0), s0.strides[0], P(s1), s1.strides[0], P(want), want.strides[0], w, h, is_bgr, delta, opac, col[0], col[1], col[2], 0)
                        d = dev(np.full_like(s0, 0x5A))
                        gpu.colorkey(dev(s0), dev(s1), d, w, h, is_bgr, delta, opac, col)
                        assert_same(host(d), want, w, h, 3, "colorkey d=%s o=%s col=%s" % (delta, opac, col))


@pytest.mark.parametrize("psize", [3, 4])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_mirror(gpu, orc, psize, mode):
    rng = np.random.default_rng(900 + psize + mode)
    for (w, h) in SIZES[:5]:
        s = frame(rng, w, h, psize)
        for inplace in (0, 1):
            want = s.copy() if inplace else np.full_like(s, 0x5A)
            orc.orc_mirror(mode, P(want if inplace else s), s.strides[0], P(want), want.strides[0], w, h, psize)
            d_s = dev(s)
            d = d_s if inplace else dev(np.full_like(s, 0x5A))
            gpu.mirror(mode, d_s, d, w, h, psize)
            assert_same(host(d), want, w, h, psize, "mirror mode=%d %dx%d inplace=%d" % (mode, w, h, inplace))


# ---------------------------------------------------------------------------------------------- K2 with the fused LUT16
@pytest.mark.parametrize("is_422", [0, 1])
def test_yuv420p_to_rgb_lut16(gpu, orc, is_422):
    import torch
    from lives_amd.lib import load
    rng = np.random.default_rng(2100 + is_422)
    for (gf, gt) in ((po.GAMMA_LINEAR, po.GAMMA_SRGB), (1, 2)):
        want16 = np.zeros(65536, np.uint16)
        assert orc.orc_gamma_lut16(1.0, gf, gt, 1.4, P(want16)) == 1
        mine16 = np.zeros(65536, np.uint16)
        assert load().lgpu_gamma_lut16(1.0, gf, gt, 1.4, mine16.ctypes.data) == 1
        assert (mine16 == want16).all()                                       # host LUT builder == oracle (== reference, pinned on CPU)
        d_lut = torch.from_numpy(mine16.view(np.int16)).cuda()
        for (w, h) in [(64, 32), (66, 34), (130, 18)]:
            for which in (0, 1, 3):
                for order, ops in ((0, 4), (1, 3), (2, 4)):
                    ch = h if is_422 else h // 2
                    Y, U, V = frame(rng, w, h, 1), frame(rng, w // 2, ch, 1), frame(rng, w // 2, ch, 1)
                    st = (ctypes.c_int * 3)(Y.strides[0], U.strides[0], V.strides[0])
                    want = np.zeros((h, align(w * ops)), np.uint8)
                    assert orc.orc_yuv420p_to_rgb_lut16(P(Y), P(U), P(V), st, U.size, V.size, P(want), want.strides[0], w, h, ops, order, is_422,
                                                        which, 2, P(want16), 1) == 0
                    d = dev(np.zeros_like(want))
                    gpu.yuv420p_to_rgb_lut16(dev(Y), dev(U), dev(V), d, w, h, d_lut, opsize=ops, out_order=order, is_422=is_422, which_tables=which,
                                             flags=1)
                    assert_same(host(d), want, w, h, ops, "K2 lut16 422=%d which=%d order=%d %dx%d" % (is_422, which, order, w, h))


# ---------------------------------------------------------------------------------------------- K4 / K3 palette matrix
def _planes(p):
    return po.planes_args(p)


@pytest.mark.parametrize("in_order", [0, 1, 2])
@pytest.mark.parametrize("out_fmt", [0, 1, 2, 3, 4, 5])
def test_rgb_to_yuv(gpu, orc, in_order, out_fmt):
    if out_fmt >= 4 and in_order == 2:
        pytest.skip("ARGB32 -> 4:2:0 / 4:2:2: reference-broken, declined")
    rng = np.random.default_rng(1700 + 10 * in_order + out_fmt)
    sizes = [(20, 8), (66, 34), (130, 50), (258, 6)] + ([(21, 7), (1, 3)] if out_fmt <= 1 else [])
    if out_fmt in (2, 3, 5):
        sizes += [(64, 32), (4, 1), (2048, 3), (516, 9)]       # widths the one-row cell kernel takes (k_rgb_to_yuv422_s)
    if out_fmt == 4:
        sizes += [(64, 32), (128, 8), (4, 2), (256, 2), (2048, 4), (516, 10)]       # widths the 4 x 2 cell kernel takes (k_rgb_to_yuv420_s), one chroma row, a partial last workgroup
    for (w, h) in sizes:
        for in_alpha in ((1,) if in_order == 2 else (0, 1)):
            for out_alpha in ((0, 1) if out_fmt <= 1 else (0,)):
                for which in ((0, 1, 2, 3) if out_fmt >= 4 else (0, 1)):
                    ips = 4 if (in_order == 2 or in_alpha) else 3
                    src = frame(rng, w, h, ips)
                    want, dims = po.k4_out_planes(0x5A, w, h, out_fmt, out_alpha, compact=False)
                    wp, ws = _planes(want)
                    assert orc.orc_rgb_to_yuv(P(src), src.strides[0], w, h, in_order, in_alpha, ctypes.addressof(wp), ctypes.addressof(ws),
                                              out_fmt, out_alpha, which) == 0
                    got = [dev(np.full_like(a, 0x5A)) for a in want]
                    gpu.rgb_to_yuv(dev(src), got, w, h, in_order, in_alpha, out_fmt, out_alpha, which)
                    for i, (a, b) in enumerate(dims):
                        assert_same(host(got[i]), want[i], a, b, 1, "rgb_to_yuv order=%d alpha=%d fmt=%d oa=%d which=%d %dx%d plane %d"
                                    % (in_order, in_alpha, out_fmt, out_alpha, which, w, h, i))


@pytest.mark.parametrize("in_fmt", [0, 1, 2, 3])
@pytest.mark.parametrize("out_order", [0, 1, 2])
def test_yuv_to_rgb(gpu, orc, in_fmt, out_order):
    rng = np.random.default_rng(1800 + 10 * in_fmt + out_order)
    sizes = [(20, 8), (66, 34), (130, 50), (258, 6)] + ([(21, 7), (1, 3)] if in_fmt <= 1 else [(64, 32), (4, 1), (2048, 3), (516, 9)])       # in_fmt >= 2: widths the cell kernel takes
    n = 0
    for (w, h) in sizes:
        for in_alpha in ((0, 1) if in_fmt <= 1 else (0,)):
            for out_alpha in ((1,) if out_order == 2 else (0, 1)):
                if in_fmt == 1 and (out_order == 2 or (out_order == 1 and not out_alpha)):
                    continue        # reference-broken, declined
                for which in ((0, 1, 2, 3) if in_fmt == 0 else (0, 1)):
                    if in_fmt == 0:
                        planes = [frame(rng, w, h, 4 if in_alpha else 3)]
                    elif in_fmt == 1:
                        planes = [frame(rng, w, h, 1) for _ in range(4 if in_alpha else 3)]
                    else:
                        planes = [frame(rng, w, h, 2)]
                    ops = 4 if (out_order == 2 or out_alpha) else 3
                    want = np.full((h, align(w * ops)), 0x5A, np.uint8)
                    sp, ss = _planes(planes)
                    assert orc.orc_yuv_to_rgb(ctypes.addressof(sp), ctypes.addressof(ss), w, h, in_fmt, in_alpha, P(want), want.strides[0],
                                              out_order, out_alpha, which) == 0
                    d = dev(np.full_like(want, 0x5A))
                    gpu.yuv_to_rgb([dev(a) for a in planes], d, w, h, in_fmt, in_alpha, out_order, out_alpha, which)
                    assert_same(host(d), want, w, h, ops, "yuv_to_rgb fmt=%d ia=%d order=%d oa=%d which=%d %dx%d" % (in_fmt, in_alpha, out_order, out_alpha, which, w, h))
                    n += 1
    assert n > 0 or (in_fmt == 1 and out_order == 2)


# ---------------------------------------------------------------------------------------------- F7 geometric transitions
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("psize", [3, 4])
def test_transition(gpu, orc, kind, psize):
    rng = np.random.default_rng(2400 + 10 * kind + psize)
    for (w, h) in [(20, 10), (33, 17), (64, 36), (7, 5), (300, 50)]:
        for amt in (0., 0.1, 0.25, 0.5, 0.73, 0.99, 1.0):
            for inplace in ((0, 1) if kind < 2 else (0,)):
                s1, s2 = frame(rng, w, h, psize), frame(rng, w, h, psize)
                want = s1.copy() if inplace else np.full_like(s1, 0x5A)
                orc.orc_transition(kind, P(want) if inplace else P(s1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, psize, amt)
                d1 = dev(s1)
                d = d1 if inplace else dev(np.full_like(s1, 0x5A))
                gpu.transition(kind, d1, dev(s2), d, w, h, psize, amt)
                assert_same(host(d), want, w, h, psize, "transition %d ps=%d %dx%d amount=%s inplace=%d" % (kind, psize, w, h, amt, inplace))


@pytest.mark.parametrize("psize", [3, 4])
def test_slide_over(gpu, orc, psize):
    rng = np.random.default_rng(2500 + psize)
    for (w, h) in [(20, 10), (33, 17), (300, 50), (5, 3)]:
        for dirn in (1, 2, 3, 4):
            for tv in (0, 1, 60, 128, 254, 255):
                for mvl, mvu in ((1, 0), (0, 1), (1, 1), (0, 0)):
                    s1, s2 = frame(rng, w, h, psize), frame(rng, w, h, psize)
                    want = np.full_like(s1, 0x5A)
                    orc.orc_slide_over(P(s1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, psize, tv, dirn, mvl, mvu)
                    d = dev(np.full_like(s1, 0x5A))
                    gpu.slide_over(dev(s1), dev(s2), d, w, h, psize, tv, dirn, mvl, mvu)
                    assert_same(host(d), want, w, h, psize, "slide over ps=%d %dx%d dir=%d amount=%d lower=%d upper=%d" % (psize, w, h, dirn, tv, mvl, mvu))


@pytest.mark.parametrize("palette", [1, 2, 588, 3, 4, 589, 5, 564, 565])
def test_deinterlace(gpu, orc, palette):
    from lives_amd.lib import LgpuError
    rng = np.random.default_rng(2700 + palette)
    ps = 3 if palette in (1, 2, 588) else 4
    for (w, h) in [(12, 9), (30, 16), (33, 17), (300, 51), (64, 3), (9, 4), (31, 10)]:
        for inplace in ((1,) if palette == 5 else (0, 1)):
            s1 = frame(rng, w, h, ps)
            # smooth vertical structure with comb rows so that both branches of the decision occur
            s1[1::2] = (s1[1::2] >> 2) + 160
            s1[0::2] = (s1[0::2] >> 2) + (rng.integers(0, 2, (s1[0::2].shape[0], 1), dtype=np.uint8) * 120)
            n = (w + 2) // 3 * 3 * ps
            if n > s1.strides[0]:
                with pytest.raises(LgpuError):
                    gpu.deinterlace(dev(s1), dev(s1.copy()), w, h, palette)
                continue
            want = s1.copy() if inplace else np.full_like(s1, 0x5A)
            src = want if inplace else s1
            assert orc.orc_deinterlace(P(src), src.strides[0], P(want), want.strides[0], w, h, palette) == 0
            if inplace:
                d = dev(s1)
                gpu.deinterlace(d, d, w, h, palette)
            else:
                d = dev(np.full_like(s1, 0x5A))
                gpu.deinterlace(dev(s1), d, w, h, palette)
            assert (host(d)[:, :n] == want[:, :n]).all(), "deinterlace pal=%d %dx%d inplace=%d" % (palette, w, h, inplace)
    if palette == 5:
        a = frame(rng, 12, 8, 4)
        with pytest.raises(LgpuError):
            gpu.deinterlace(dev(a), dev(a.copy()), 12, 8, 5)


@pytest.mark.parametrize("palette", [1, 2, 588])
def test_rgbdelay_with_changing_parameters(gpu, orc, palette):
    """a longer run than the fixtures: the parameter set changes mid-sequence (the ring grows, shrinks and empties), frames are padded"""
    rng = np.random.default_rng(2900 + palette)
    plans = [({0: (1, 0, 0, 1.0), 3: (0, 1, 0, 0.8), 6: (0, 0, 1, 1.0)}, 12), ({0: (1, 1, 1, 0.5), 1: (1, 1, 1, 0.5), 9: (1, 0, 1, 0.7)}, 12),
             ({0: (0, 1, 1, 0.9)}, 20), ({0: (1, 0, 0, 1.0), 2: (0, 1, 1, 1.0)}, 2), ({0: (1, 1, 1, 1.0), 49: (1, 1, 1, 0.3)}, 50)]
    for inplace, (w, h) in ((0, (70, 24)), (1, (70, 24)), (0, (72, 10)), (1, (132, 7))):     # 3 * width % 4 == 0 takes the four-pixel kernel
        s = orc.orc_rgbdelay_new()
        rd = gpu.RgbDelay()
        for groups, maxcache in plans:
            on, st = gu_params(groups)
            for _ in range(5):
                src = frame(rng, w, h, 3)
                want = src.copy() if inplace else np.full_like(src, 0x5A)
                a = want if inplace else src
                assert orc.orc_rgbdelay_process(s, P(a), a.strides[0], P(want), want.strides[0], w, h, palette, 1, maxcache, on.ctypes.data, st.ctypes.data) == 0
                ds = dev(src)
                d = ds if inplace else dev(np.full_like(src, 0x5A))
                rd.process(ds, d, w, h, palette, maxcache, on, st, yuv_clamped=True)
                assert (host(d) == want).all(), "rgbdelay pal=%d inplace=%d" % (palette, inplace)
        orc.orc_rgbdelay_free(s)
        rd.close()


def gu_params(groups):
    from tests import golden_util as gu
    return gu.rgbdelay_params(groups)


@pytest.mark.parametrize("psize", [3, 4])
def test_byte_luts(gpu, orc, psize):
    rng = np.random.default_rng(3000 + psize)
    for (w, h) in [(64, 8), (67, 5), (1, 1), (641, 33), (1920, 16)]:
        luts = rng.integers(0, 256, (psize, 256), dtype=np.uint8)
        for inplace in (0, 1):
            src = frame(rng, w, h, psize)
            want = src.copy() if inplace else np.full_like(src, 0x5A)
            a = want if inplace else src
            orc.orc_byte_luts(P(a), a.strides[0], P(want), want.strides[0], w, h, psize, luts.ctypes.data)
            ds = dev(src)
            d = ds if inplace else dev(np.full_like(src, 0x5A))
            gpu.byte_luts(ds, d, w, h, psize, luts)
            assert (host(d) == want).all(), "byte_luts ps=%d %dx%d inplace=%d" % (psize, w, h, inplace)
    for kind in (0, 1, 2):
        for pal in (1, 2, 3, 4, 5):
            ref = np.zeros((4, 256), np.uint8)
            n = orc.orc_fx_luts(kind, pal, 3.0, 0.6, 1.9, ref.ctypes.data)
            got = gpu.fx_luts(kind, pal, 3.0, 0.6, 1.9)
            assert (got is None and n == 0) or (got == ref[:n]).all()


@pytest.mark.parametrize("psize", [3, 4])
def test_dissolve(gpu, orc, psize):
    import torch
    rng = np.random.default_rng(3200 + psize)
    for (w, h) in [(20, 10), (333, 47), (1, 1)]:
        seed = 0xC0FFEE + w
        mask = np.zeros(w * h, np.float32)
        orc.orc_dissolve_mask(seed, w, h, mask.ctypes.data)
        gm = gpu.dissolve_mask(seed, w, h)
        assert (gm == mask).all()
        dm = torch.from_numpy(gm).cuda()
        for amt in (0.0, 0.01, 0.37, 0.5, 0.999, 1.0):
            for inplace in (0, 1):
                s1, s2 = frame(rng, w, h, psize), frame(rng, w, h, psize)
                want = s1.copy() if inplace else np.full_like(s1, 0x5A)
                a = want if inplace else s1
                orc.orc_dissolve(P(a), a.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, psize, mask.ctypes.data, amt)
                d1 = dev(s1)
                d = d1 if inplace else dev(np.full_like(s1, 0x5A))
                gpu.dissolve(d1, dev(s2), d, w, h, psize, dm, amt)
                assert_same(host(d), want, w, h, psize, "dissolve ps=%d %dx%d amount=%s inplace=%d" % (psize, w, h, amt, inplace))


def test_triple_split(gpu, orc):
    rng = np.random.default_rng(3100)
    bc = np.array([13, 250, 77], np.int32)
    for (w, h) in [(20, 10), (33, 17), (301, 50), (2, 2)]:
        for is_bgr in (0, 1):
            for (start, sym, end, vert, bw) in [(0.666667, 1, 0.333333, 0, 0.), (0.25, 0, 0.75, 0, 0.04), (0.4, 1, 0.0, 1, 0.07), (0.8, 0, 0.3, 1, 0.5),
                                                (0.0, 0, 1.0, 0, 0.0), (0.5, 0, 0.5, 1, 0.01)]:
                for inplace in (0, 1):
                    s1, s2 = frame(rng, w, h, 3), frame(rng, w, h, 3)
                    want = s1.copy() if inplace else np.full_like(s1, 0x5A)
                    a = want if inplace else s1
                    orc.orc_triple_split(P(a), a.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, is_bgr, start, sym, end, vert, bw, bc.ctypes.data)
                    d1 = dev(s1)
                    d = d1 if inplace else dev(np.full_like(s1, 0x5A))
                    gpu.triple_split(d1, dev(s2), d, w, h, is_bgr, start, sym, end, vert, bw, bc)
                    assert_same(host(d), want, w, h, 3, "triple split %dx%d %r inplace=%d" % (w, h, (start, sym, end, vert, bw), inplace))


# ---------------------------------------------------------------------------------------------- K5b YUV -> YUV repacks
@pytest.mark.parametrize("pair", po.YUV_REPACK_PAIRS, ids=lambda p: "%d-%d" % (p[0], p[1]))
def test_yuv_repack(gpu, orc, pair):
    import ctypes
    ip, op, padok = pair
    rng = np.random.default_rng(2600 + ip * 7 + op)
    subs = {512, 513, 522, 564, 565}
    sizes = [(16, 8), (66, 34), (130, 18), (320, 200)]
    if ip not in subs and op not in subs:
        sizes += [(7, 5), (33, 3)]                     # odd geometry only exists for the 4:4:4 layouts
    for (w, h) in sizes:
        for unc in (0, 1):
            for pad in ((0, 24) if padok else (0,)):
                src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                want = po.yuv_planes(op, w, h, fill=0x5A, pad=pad)
                sp, ss = po.planes_args(src)
                wp, ws = po.planes_args(want)
                assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, unc, 0) == 0
                dst = [dev(np.full_like(a, 0x5A)) for a in want]
                gpu.yuv_repack(ip, op, [dev(a) for a in src], dst, w, h, unc)
                for i, a in enumerate(want):
                    assert (host(dst[i]) == a).all(), "repack %d->%d %dx%d unclamped=%d pad=%d plane %d" % (ip, op, w, h, unc, pad, i)


def test_chroma_average_table(gpu, orc):
    """the clamped chroma average (init_average, src/colourspace.c:190-216) as the kernels compute it -- fmaf(fa[x] + fa[y], 0.4375f, 128.f) on a 256-entry float table
    instead of the reference's float / double chain -- for ALL 65,536 pairs against the oracle's table (itself pinned on the reference's cavgc)"""
    from lives_amd.lib import load
    t = np.zeros(65536, np.uint8)
    assert load().lgpu_chroma_average_table(t.ctypes.data) == 0
    want = np.array([[orc.orc_cavg(1, x, y) for y in range(256)] for x in range(256)], np.uint8)
    assert (t.reshape(256, 256) == want).all()
    from tests import golden_util as gu
    assert (t.reshape(256, 256) == gu.load("cavg.npz")["cavgc"]).all()            # ... and against the reference's own table


@pytest.mark.parametrize("pair", po.YUV411_REPACK_PAIRS, ids=lambda p: "%d-%d" % (p[0], p[1]))
def test_yuv411_repack(gpu, orc, pair):
    """K5c: YUV411 <-> YUV888 / YUVA8888 / YUV444P / YUVA4444P / UYVY / YUYV / YUV422P / YUV420P / YVU420P (:7755-7798, :7973-8033, :8272-8303,
    :8622-9196), compact streams on both sides as in the reference, at sizes from one macropixel per row to 1080p"""
    import ctypes
    ip, op, padok = pair
    rng = np.random.default_rng(2700 + ip * 7 + op)
    sizes = [(4, 2), (8, 6), (64, 34), (132, 18), (320, 200), (1920, 1080)]
    if op in (512, 513):
        sizes += [(16, 5), (16, 1)]                     # odd heights: the last even row has no odd row to fold
    for (w, h) in sizes:
        for unc in (0, 1):
            for pad in ((0, 24) if padok else (0,)):
                src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                want = po.yuv_planes(op, w, h + ((h & 1) if op in (512, 513) else 0), fill=0x5A, pad=0)
                sp, ss = po.planes_args(src)
                wp, ws = po.planes_args(want)
                assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, unc, 0) == 0
                dst = [dev(np.full_like(a, 0x5A)) for a in want]
                gpu.yuv_repack(ip, op, [dev(a) for a in src], dst, w, h, unc)
                for i, a in enumerate(want):
                    assert (host(dst[i]) == a).all(), "repack %d->%d %dx%d unclamped=%d pad=%d plane %d" % (ip, op, w, h, unc, pad, i)


@pytest.mark.parametrize("order,alpha", [(0, 0), (0, 1), (1, 0), (1, 1), (2, 1)])
def test_rgb_to_packed_422_with_gamma_lut16(gpu, orc, order, alpha):
    """K4 with the 16-bit LUT inline (rgb2uyvy_with_gamma / rgb2yuyv_with_gamma): every table sum's top 16 bits index create_gamma_lut's table"""
    import torch
    rng = np.random.default_rng(2900 + order * 2 + alpha)
    ips = 4 if alpha else 3
    for (gf, gt) in ((po.GAMMA_LINEAR, po.GAMMA_SRGB), (po.GAMMA_SRGB, po.GAMMA_LINEAR), (po.GAMMA_SRGB, po.GAMMA_BT709)):
        lut = np.zeros(65536, np.uint16)
        assert orc.orc_gamma_lut16(1.0, gf, gt, 1.4, P(lut)) == 1
        d_lut = torch.from_numpy(lut.view(np.int16)).cuda()
        for (w, h) in [(2, 1), (64, 16), (130, 9), (1920, 1080)]:
            for fmt in (2, 3):
                for unc in (0, 1):
                    src = frame(rng, w, h, ips)
                    want = np.full((h, align(w * 2)), 0x5A, np.uint8)
                    assert orc.orc_rgb_to_yuv_lut16(P(src), src.strides[0], w, h, order, alpha, P(want), want.strides[0], fmt, unc, P(lut)) == 0
                    d = dev(np.full_like(want, 0x5A))
                    gpu.rgb_to_yuv_lut16(dev(src), d, w, h, order, alpha, fmt, unc, d_lut)
                    assert (host(d) == want).all(), "rgb -> %s lut16 %dx%d order=%d alpha=%d unclamped=%d" % ("UYVY" if fmt == 2 else "YUYV", w, h, order, alpha, unc)


@pytest.mark.parametrize("pair", po.CHROMA_UP_PAIRS, ids=lambda p: "%d-%d" % p)
def test_chroma_up_packed(gpu, orc, pair):
    """K5d: YUV420P / YUV422P -> YUV888 / YUVA8888 (convert_quad_chroma_packed / convert_double_chroma_packed, :10715-10873): both chroma sitings, padded and compact planes
    (the clamped out-of-plane read is the oracle's rule too), bytes the reference leaves alone stay 0x5A"""
    import ctypes
    ip, op = pair
    rng = np.random.default_rng(2800 + ip + op)
    for (w, h) in [(4, 2), (16, 8), (66, 34), (130, 18), (320, 200), (1920, 1080)]:
        for unc in (0, 1):
            for sampling in (0, 1):
                for pad in (0, 24):
                    src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                    want = po.yuv_planes(op, w, h, fill=0x5A, pad=pad)
                    sp, ss = po.planes_args(src)
                    wp, ws = po.planes_args(want)
                    assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, unc, sampling) == 0
                    d = dev(np.full_like(want[0], 0x5A))
                    gpu.yuv_repack(ip, op, [dev(a) for a in src], [d], w, h, unc, sampling)
                    assert (host(d) == want[0]).all(), "chroma up %d->%d %dx%d unclamped=%d sampling=%d pad=%d" % (ip, op, w, h, unc, sampling, pad)


def test_yuv_repack_declines(gpu):
    from lives_amd.lib import LgpuError
    w, h = 16, 8
    for (ip, op, pad) in [(512, 564, 8), (544, 564, 8), (564, 512, 8), (522, 512, 0), (544, 522, 0), (512, 544, 0), (589, 544, 0), (588, 545, 0), (512, 545, 0), (522, 544, 0)]:
        src = [dev(a) for a in po.yuv_planes(ip, w, h, fill=1, pad=pad)]
        dst = [dev(a) for a in po.yuv_planes(op, w, h, fill=2, pad=pad)]
        with pytest.raises(LgpuError):
            gpu.yuv_repack(ip, op, src, dst, w, h, 0)


# ---------------------------------------------------------------------------------------------- K5 clamping switch
@pytest.mark.parametrize("palette", [588, 589, 544, 545, 522, 512, 513, 564, 565])
def test_yuv_switch_clamping(gpu, orc, palette):
    rng = np.random.default_rng(2300 + palette)
    for (w, h) in [(20, 8), (66, 34), (130, 6)]:
        for to_unclamped in (0, 1):
            if palette in (588, 589):
                planes = [frame(rng, w, h, 3 if palette == 588 else 4)]
            elif palette in (564, 565):
                planes = [frame(rng, w, h, 2)]
            else:
                Y = frame(rng, w, h, 1)
                cs = Y.strides[0] if palette in (544, 545) else Y.strides[0] >> 1
                ch = h >> 1 if palette in (512, 513) else h
                planes = [Y] + [rng.integers(0, 256, (ch, cs), dtype=np.uint8) for _ in range(2)] + ([frame(rng, w, h, 1)] if palette == 545 else [])
            want = [a.copy() for a in planes]
            wp, ws = po.planes_args(want)
            assert orc.orc_switch_yuv_clamping(ctypes.addressof(wp), ctypes.addressof(ws), palette, h, to_unclamped) == 0
            got = [dev(a) for a in planes]
            gpu.yuv_switch_clamping(got, palette, h, to_unclamped)
            for i in range(len(planes)):
                assert (host(got[i]) == want[i]).all(), (palette, w, h, to_unclamped, i)


# ---------------------------------------------------------------------------------------------- F6 stencils
@pytest.mark.parametrize("shape", ["rb2", "rb4", "tiles"])
@pytest.mark.parametrize("palette", [544, 545, 522, 512, 513])
def test_softlight(gpu, orc, tune, palette, shape):
    """4-aligned planes take the register form k_softlight_s (bands of 2 or 4 rows per wave, strips of 248 columns: widths that end inside a strip, several strips,
    heights that end inside a band), everything else -- and everything when the switch says so -- the LDS tile kernel"""
    if shape == "rb4":
        tune("SOFT_RB", 4)
    elif shape == "tiles":
        tune("SOFT_NO_S", 1)
    rng = np.random.default_rng(1500 + palette)
    for (w, h) in [(20, 9), (64, 16), (66, 34), (130, 50), (4, 3), (258, 33), (256, 40), (1000, 21), (1920, 30), (252, 7), (8, 3), (496, 11), (500, 5)]:
        for unclamped in (0, 1):
            cw = w >> 1 if palette in (512, 513, 522) else w
            ch = h >> 1 if palette in (512, 513) else h
            dims = [(w, h), (cw, ch), (cw, ch)] + ([(w, h)] if palette == 545 else [])
            src = [frame(rng, a, b, 1) for (a, b) in dims]
            want = np.full_like(src[0], 0x5A)
            orc.orc_softlight_y(P(src[0]), src[0].strides[0], P(want), want.strides[0], w, h, unclamped)
            dst = [dev(np.full_like(a, 0x5A)) for a in src]
            gpu.softlight([dev(a) for a in src], dst, w, h, palette, unclamped)
            assert_same(host(dst[0]), want, w, h, 1, "softlight Y pal=%d %dx%d uncl=%d" % (palette, w, h, unclamped))
            for i in range(1, len(dims)):      # chroma / alpha planes are copies (softlight.c:143-151)
                assert_same(host(dst[i]), src[i], dims[i][0], dims[i][1], 1, "softlight plane %d" % i)


@pytest.mark.parametrize("palette", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_edge(gpu, orc, palette, mode):
    rng = np.random.default_rng(1600 + 10 * palette + mode)
    ps = 3 if palette <= 2 else 4
    for (w, h) in [(18, 8), (64, 16), (70, 37), (131, 50), (4, 4), (5, 5), (200, 120)]:
        for inplace in (0, 1):
            s = frame(rng, w, h, ps)
            # smooth structure under the noise so that the histogram is not flat
            yy, xx = np.mgrid[0:h, 0:w]
            for c in range(ps):
                s[:, c:w * ps:ps] = ((s[:, c:w * ps:ps] >> 3) + (96 * ((xx // 9 + yy // 7 + c) % 2)).astype(np.uint8) + 40).astype(np.uint8)
            d0 = s.copy() if inplace else rng.integers(0, 256, s.shape, dtype=np.uint8)
            want = d0.copy()
            m16 = np.zeros(w * h, np.int16)
            orc.orc_edge(P(want) if inplace else P(s), s.strides[0], P(want), want.strides[0], w, h, palette, mode, P(m16), inplace)
            d = dev(d0)
            gpu.edge(d if inplace else dev(s), d, w, h, palette, mode)
            assert_same(host(d), want, w, h, ps, "edge pal=%d mode=%d %dx%d inplace=%d" % (palette, mode, w, h, inplace))


@pytest.mark.parametrize("palette", [3, 4, 5])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_edge_by_quads(gpu, orc, palette, mode, tune):
    """the three-launch form for 4-byte pixels on 16-byte aligned rows (k_edge_map4 / k_edge_reduce_otsu / k_edge_paint4): whole and partial 128 x 32 tiles, more
    tiles than the map's 1,024 workgroups, frames of one quad column group, in place and out of place; a smooth and a noisy frame (few / all histogram bins)"""
    rng = np.random.default_rng(2600 + 10 * palette + mode)
    for (w, h, noisy) in [(8, 5, 0), (128, 32, 0), (132, 33, 1), (260, 70, 0), (516, 131, 1), (1920, 1080, 0), (4096, 1100, 1)]:
        if w >= 1920 and (mode == 1 or palette == 4):
            continue
        for inplace in (0, 1):
            s = frame(rng, w, h, 4)
            if not noisy:
                yy, xx = np.mgrid[0:h, 0:w]
                for c in range(4):
                    s[:, c:w * 4:4] = ((s[:, c:w * 4:4] >> 3) + (96 * ((xx // 9 + yy // 7 + c) % 2)).astype(np.uint8) + 40).astype(np.uint8)
            d0 = s.copy() if inplace else rng.integers(0, 256, s.shape, dtype=np.uint8)
            want = d0.copy()
            m16 = np.zeros(w * h, np.int16)
            orc.orc_edge(P(want) if inplace else P(s), s.strides[0], P(want), want.strides[0], w, h, palette, mode, P(m16), inplace)
            for eh in ((0,) if w >= 1920 else (16, 32)):      # both tile heights of the map kernel (0: the library's choice)
                tune("EDGE_TH", eh)
                d = dev(d0)
                gpu.edge(d if inplace else dev(s), d, w, h, palette, mode)
                assert_same(host(d), want, w, h, 4, "edge by quads pal=%d mode=%d %dx%d inplace=%d tile rows %d" % (palette, mode, w, h, inplace, eh))
                assert_padding_untouched(host(d), d0, w, h, 4, "edge by quads %dx%d" % (w, h))


def test_edge_forms_agree(gpu, tune):
    """the quad form against the general kernels (EDGE_NO_S) on the same frames, state carried over consecutive calls on one stream"""
    import torch
    rng = np.random.default_rng(77)
    for (w, h) in [(640, 360), (1280, 720)]:
        s = dev(frame(rng, w, h, 4))
        outs = []
        for off in (1, 0, 1, 0):
            tune("EDGE_NO_S", off)
            for mode in (0, 2):
                o = torch.zeros_like(s)
                gpu.edge(s, o, w, h, 3, mode)
                outs.append(host(o).copy())
        for i in range(2, len(outs)):
            assert (outs[i] == outs[i % 2]).all(), "edge forms differ (%dx%d, call %d)" % (w, h, i)


def bz_sequence(rng, w, h, n, compact):
    """frames with a bright block moving over a dim noisy background, so that the background subtraction fires"""
    out = []
    for f in range(n):
        a = frame(rng, w, h, 4) if not compact else rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
        a[:, :w * 4] = (a[:, :w * 4] >> 4) + 40
        a[2 + f % 3:6 + f % 3, ((8 + 5 * f) % (w - 20)) * 4:((24 + 5 * f) % (w - 4)) * 4] = 250
        out.append(a)
    return out


@pytest.mark.parametrize("palette", [3, 4])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_blurzoom(gpu, orc, palette, mode):
    rng = np.random.default_rng(2200 + 10 * palette + mode)
    for (w, h) in [(70, 12), (64, 9), (330, 40)]:
        for pattern in (0, 3) if mode else (0, 1, 2, 3):
            seq = bz_sequence(rng, w, h, 7, compact=mode in (1, 2))
            z = orc.orc_blurzoom_new(w, h, palette)
            g = gpu.Blurzoom(w, h, palette)
            for f, a in enumerate(seq):
                want = np.full_like(a, 0x5A)
                assert orc.orc_blurzoom_process(z, P(a), a.strides[0], P(want), want.strides[0], mode, pattern) == 0
                d = dev(np.full_like(a, 0x5A))
                g.process(dev(a), d, mode, pattern)
                assert_same(host(d), want, w, h, 4, "blurzoom pal=%d mode=%d pattern=%d %dx%d frame %d" % (palette, mode, pattern, w, h, f))
            orc.orc_blurzoom_free(z)
            g.close()


# ---------------------------------------------------------------------------------------------- compositor fan-in
@pytest.mark.parametrize("psize", [3, 4])
def test_composite(gpu, orc, psize):
    rng = np.random.default_rng(1900 + psize)
    for (ow, oh) in [(40, 20), (130, 50), (300, 17)]:
        for n in (0, 1, 3, 16):
            layers, keep = [], []
            for z in range(n):
                w, h = int(rng.integers(4, ow + 8)), int(rng.integers(4, oh + 8))
                a = frame(rng, w, h, psize)
                keep.append(a)
                layers.append((a, w, h, int(rng.integers(0, ow)), int(rng.integers(0, oh)), float(rng.choice([0., 0.25, 0.5, 0.7312, 1.]))))
            if n >= 3:
                layers[1] = (None,) + layers[1][1:]                         # a disabled channel
            bg = [int(v) for v in rng.integers(0, 256, 3)]
            for is_bgr in (0, 1):
                for revz in (0, 1):
                    L = (po.CompLayer * max(1, n))()
                    for z, (a, w, h, ox, oy, al) in enumerate(layers):
                        L[z].src = a.ctypes.data if a is not None else None
                        L[z].irow = a.strides[0] if a is not None else 0
                        L[z].width, L[z].height, L[z].offs_x, L[z].offs_y, L[z].alpha = w, h, ox, oy, al
                    want = np.full((oh, align(ow * psize)), 0x5A, np.uint8)
                    orc.orc_composite(P(want), want.strides[0], ow, oh, psize, is_bgr, (ctypes.c_int * 3)(*bg), L, n, revz)
                    d = dev(np.full_like(want, 0x5A))
                    gpu.composite(d, ow, oh, psize, [(dev(a) if a is not None else None, w, h, ox, oy, al) for (a, w, h, ox, oy, al) in layers],
                                  bgcol=bg, is_bgr=is_bgr, revz=revz)
                    assert_same(host(d), want, ow, oh, psize, "composite ps=%d n=%d bgr=%d revz=%d %dx%d" % (psize, n, is_bgr, revz, ow, oh))


# ---------------------------------------------------------------------------------------------- K7 / B1 (own spec)
@pytest.mark.parametrize("psize", [4, 3, 1])
def test_resize(gpu, orc, psize):
    rng = np.random.default_rng(1000 + psize)
    cases = [(128, 64, 64, 32, 3), (200, 120, 100, 60, 3), (260, 44, 130, 22, 3), (130, 70, 64, 36, 3), (64, 32, 128, 64, 3), (100, 60, 37, 23, 3), (128, 64, 64, 32, 2),
             (64, 36, 200, 100, 2), (320, 180, 96, 54, 3), (16, 16, 4, 4, 3)]
    for (sw, sh, dw, dh, interp) in cases:
        src = frame(rng, sw, sh, psize)
        want = np.zeros((dh, align(dw * psize)), np.uint8)
        assert orc.orc_resize(P(src), src.strides[0], sw, sh, P(want), want.strides[0], dw, dh, psize, interp) == 0
        d = dev(np.zeros_like(want))
        gpu.resize(dev(src), d, sw, sh, dw, dh, psize=psize, interp=interp)
        assert_same(host(d), want, dw, dh, psize, "resize %dx%d->%dx%d interp=%d ps=%d" % (sw, sh, dw, dh, interp, psize))


def test_resize_persistent_kernel(gpu, orc, tune):
    """k_sep2p (persistent workgroups, windows and tables prefetched by LDS-DMA; taken by default for large shrinking launches) forced on small
    frames: same bytes as the oracle for shrinking, enlarging and mixed ratios, windows that cross every frame border, partial tiles, one tile
    per workgroup and several, and -- through the chain -- several tracks per launch with the byte swap, the blend and the gamma LUT"""
    tune("SEP2P_FORCE", 1)
    rng = np.random.default_rng(1050)
    cases = [(384, 216, 128, 72, 3), (256, 144, 512, 288, 3), (192, 108, 128, 72, 3), (640, 360, 212, 120, 3), (128, 64, 64, 32, 2), (400, 300, 100, 75, 3),
             (1920, 1080, 1280, 720, 3), (1280, 720, 1920, 1080, 3), (3840, 2160, 1280, 720, 3), (64, 32, 128, 64, 3), (100, 60, 36, 24, 3), (16, 16, 4, 4, 3),
             (3840, 40, 480, 8, 3), (64, 2160, 16, 720, 2), (3840, 2160, 960, 540, 3), (1920, 1080, 480, 270, 3)]
    for (sw, sh, dw, dh, interp) in cases:
        src = frame(rng, sw, sh, 4)
        want = np.zeros((dh, align(dw * 4)), np.uint8)
        assert orc.orc_resize(P(src), src.strides[0], sw, sh, P(want), want.strides[0], dw, dh, 4, interp) == 0
        d = dev(np.zeros_like(want))
        gpu.resize(dev(src), d, sw, sh, dw, dh, psize=4, interp=interp)
        assert_same(host(d), want, dw, dh, 4, "resize (k_sep2p) %dx%d->%dx%d interp=%d" % (sw, sh, dw, dh, interp))
    lut = lut_for(rng, "l2s")
    for (sw, sh, dw, dh, swap, bf, ntr, use_lut) in [(384, 216, 128, 72, 1, 90, 3, 1), (320, 200, 200, 120, 0, 255, 2, 0), (200, 120, 320, 200, 1, 17, 5, 1)]:
        srcs = [frame(rng, sw, sh, 4) for _ in range(ntr)]
        l2s = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(ntr)]
        orow = align(dw * 4)
        wants = []
        for i in range(ntr):
            w_ = np.zeros((dh, orow), np.uint8)
            assert orc.orc_chain(P(srcs[i]), srcs[i].strides[0], sw, sh, P(l2s[i]), l2s[i].strides[0], P(w_), orow, dw, dh, swap, 3, 0, bf, P(lut) if use_lut else None) == 0
            wants.append(w_)
        dd = [dev(np.zeros((dh, orow), np.uint8)) for _ in range(ntr)]
        prm = gpu.chain_params(sw, sh, srcs[0].strides[0], dw, dh, l2s[0].strides[0], orow, swap_rb=swap, interp=3, do_blur=0, bf=bf, lut=lut if use_lut else None)
        gpu.chain(prm, gpu.chain_tracks([dev(s_) for s_ in srcs], [dev(s_) for s_ in l2s], dd))
        for i in range(ntr):
            assert_same(host(dd[i]), wants[i], dw, dh, 4, "chain (k_sep2p) %dx%d->%dx%d track %d" % (sw, sh, dw, dh, i))


@pytest.mark.parametrize("psize", [4, 3, 1])
def test_gauss5(gpu, orc, psize):
    rng = np.random.default_rng(1100 + psize)
    for (w, h) in [(64, 32), (66, 34), (130, 18), (7, 5), (3, 2), (320, 200)]:
        src = frame(rng, w, h, psize)
        want = np.zeros_like(src)
        orc.orc_gauss5(P(src), src.strides[0], P(want), want.strides[0], w, h, psize)
        d = dev(np.zeros_like(src))
        gpu.gauss5(dev(src), d, w, h, psize=psize)
        assert_same(host(d), want, w, h, psize, "gauss5 %dx%d ps=%d" % (w, h, psize))


@pytest.fixture
def yuv_tuning():
    from lives_amd.lib import load
    lib = load()
    yield lambda nc=-1, block=-1, wgs=-1: lib.lgpu_yuv420_tuning(nc, block, wgs)
    lib.lgpu_yuv420_tuning(2, 512, 8)


@pytest.mark.parametrize("nc", [0, 1, 2, 4])
def test_yuv420p_every_cell_width(gpu, orc, yuv_tuning, nc):
    """k_yuv420p_to_rgb_s with cells of 1 / 2 / 4 chroma columns (and switched off: the one-column kernel for everything), three workgroup sizes,
    a grid capped at one group per CU (every thread walks several cells): same bytes as the oracle for the four table sets, the three byte orders,
    with and without the gamma LUT, including the cells handed to the one-column walk (row 0, the trailing row, partial column groups, plane ends)"""
    rng = np.random.default_rng(3400 + nc)
    for (block, wgs) in [(256, 8), (512, 1), (1024, 1)]:
        assert yuv_tuning(nc, block, wgs) == 0
        for which in range(4):
            for order in (0, 1, 2):
                for use_lut in (0, 1):
                    for (w, h, ys, cs) in [(64, 32, 64, 32), (66, 34, 96, 48), (130, 19, 160, 80), (24, 6, 32, 16), (2, 2, 8, 4), (6, 3, 8, 4), (640, 480, 640, 320), (1920, 64, 1920, 960)]:
                        if (which, order, use_lut) != (0, 0, 1) and (w, block) not in ((66, 256), (130, 512), (24, 1024)):
                            continue                    # the full size list once per shape, three sizes for every table set / order / LUT
                        for is422 in (0, 1):             # planar 4:2:2 rides the same kernel (one luma row per cell)
                            lut = lut_for(rng, "l2s") if use_lut else None
                            Y = rng.integers(0, 256, (h, ys), dtype=np.uint8)
                            U = rng.integers(0, 256, (h if is422 else h // 2, cs), dtype=np.uint8)
                            V = rng.integers(0, 256, (h if is422 else h // 2, cs), dtype=np.uint8)
                            orow = align(w * 4)
                            strides = (ctypes.c_int * 3)(ys, cs, cs)
                            want = np.full((h, orow), 0xAB, np.uint8)
                            orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, U.size, V.size, P(want), orow, w, h, 4, order, is422, which, 2, P(lut) if use_lut else None, 0)
                            d = dev(np.full_like(want, 0xAB))
                            gpu.yuv420p_to_rgb(dev(Y), dev(U), dev(V), d, w, h, opsize=4, out_order=order, is_422=is422, which_tables=which, pb_quality=2, lut=lut)
                            assert_same(host(d), want, w, h, 4, "yuv42%dp nc=%d block=%d %dx%d which=%d order=%d lut=%d" % (2 if is422 else 0, nc, block, w, h, which, order, use_lut))
    # extreme samples: every (y, u, v) corner reaches the ends of the clamps
    yuv_tuning(nc, 512, 8)
    w, h = 64, 32
    for which in range(4):
        for yv in (0, 16, 235, 255):
            for uv in (0, 16, 128, 240, 255):
                Y = np.full((h, w), yv, np.uint8)
                U = np.full((h // 2, w // 2), uv, np.uint8)
                V = np.full((h // 2, w // 2), 255 - uv if uv not in (0, 255) else uv, np.uint8)
                strides = (ctypes.c_int * 3)(w, w // 2, w // 2)
                want = np.zeros((h, w * 4), np.uint8)
                orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, U.size, V.size, P(want), w * 4, w, h, 4, 0, 0, which, 2, None, 0)
                d = dev(np.zeros_like(want))
                gpu.yuv420p_to_rgb(dev(Y), dev(U), dev(V), d, w, h, opsize=4, which_tables=which)
                assert_same(host(d), want, w, h, 4, "corner y=%d u=%d which=%d" % (yv, uv, which))
    # a batch in one launch == the same frames one by one through the one-column kernel
    import torch
    w, h, n = 256, 96, 5
    lut = lut_for(rng, "l2s")
    frames = []
    for _ in range(n):
        Y = dev(rng.integers(0, 256, (h, w), dtype=np.uint8))
        U = dev(rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8))
        V = dev(rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8))
        frames.append((Y, U, V, torch.zeros((h, w * 4), dtype=torch.uint8, device="cuda")))
    gpu.yuv420p_to_rgb_batch(frames, w, h, lut=lut)
    torch.cuda.synchronize()
    yuv_tuning(0)
    for (Y, U, V, d) in frames:
        one = torch.zeros_like(d)
        gpu.yuv420p_to_rgb(Y, U, V, one, w, h, lut=lut)
        assert torch.equal(one, d)


def test_yuv420p_batch_equals_single_calls(gpu):
    import torch
    rng = np.random.default_rng(3300)
    lut = lut_for(rng, "l2s")
    for (w, h, is_422) in [(64, 32, 0), (130, 18, 0), (66, 34, 1)]:
        frames, singles = [], []
        for t in range(5):
            Y = dev(rng.integers(0, 256, (h, align(w)), dtype=np.uint8))
            ch = h if is_422 else h // 2
            U, V = (dev(rng.integers(0, 256, (ch, align(w) // 2), dtype=np.uint8)) for _ in range(2))
            d1 = torch.full((h, align(w * 4)), 0xAB, dtype=torch.uint8, device="cuda")
            d2 = d1.clone()
            gpu.yuv420p_to_rgb(Y, U, V, d1, w, h, is_422=is_422, which_tables=1, lut=lut)
            frames.append((Y, U, V, d2))
            singles.append(d1)
        gpu.yuv420p_to_rgb_batch(frames, w, h, is_422=is_422, which_tables=1, lut=lut)
        for (_, _, _, d2), d1 in zip(frames, singles):
            assert torch.equal(d1, d2)


# ---------------------------------------------------------------------------------------------- chain
@pytest.mark.parametrize("do_blur", [0, 1])
def test_chain_matches_oracle_and_unfused(gpu, orc, do_blur):
    rng = np.random.default_rng(1200 + do_blur)
    lut = lut_for(rng, "l2s")
    for (sw, sh, dw, dh) in [(128, 64, 64, 32), (384, 216, 192, 108), (200, 120, 66, 34), (204, 76, 102, 38)]:
        for swap in (1, 0):
            for bf in (0, 128, 255):
                ntr = 3
                srcs = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(ntr)]
                l2s = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(ntr)]
                wants = []
                for i in range(ntr):
                    want = np.zeros((dh, align(dw * 4)), np.uint8)
                    assert orc.orc_chain(P(srcs[i]), srcs[i].strides[0], sw, sh, P(l2s[i]), l2s[i].strides[0], P(want), want.strides[0], dw, dh, swap, 3, do_blur, bf, P(lut)) == 0
                    wants.append(want)
                d_src = [dev(s) for s in srcs]
                d_l2 = [dev(s) for s in l2s]
                d_dst = [dev(np.zeros((dh, align(dw * 4)), np.uint8)) for _ in range(ntr)]
                prm = gpu.chain_params(sw, sh, srcs[0].strides[0], dw, dh, l2s[0].strides[0], d_dst[0].stride(0), swap_rb=swap, interp=3, do_blur=do_blur, bf=bf, lut=lut)
                trk = gpu.chain_tracks(d_src, d_l2, d_dst)
                gpu.chain(prm, trk)
                for i in range(ntr):
                    assert_same(host(d_dst[i]), wants[i], dw, dh, 4, "chain track %d %dx%d->%dx%d swap=%d bf=%d blur=%d" % (i, sw, sh, dw, dh, swap, bf, do_blur))
                # the same thing through the single entry points, in reference order
                i = 0
                conv = dev(np.zeros_like(srcs[i]))
                if swap:
                    gpu.swizzle(gpu.lib.SWAP3POSTALPHA, d_src[i], conv, sw, sh)
                else:
                    conv = d_src[i]
                rs = dev(np.zeros((dh, align(dw * 4)), np.uint8))
                gpu.resize(conv, rs, sw, sh, dw, dh, psize=4, interp=3)
                if do_blur:
                    bl = dev(np.zeros((dh, align(dw * 4)), np.uint8))
                    gpu.gauss5(rs, bl, dw, dh, psize=4)
                    rs = bl
                gpu.blend_chroma(rs, d_l2[i], rs, dw, dh, 4, bf)
                gpu.gamma_apply(rs, dw, dh, 4, lut)
                assert_same(host(rs), wants[i], dw, dh, 4, "unfused chain")


@pytest.mark.parametrize("do_blur", [0, 1])
def test_chain_with_device_param_block(gpu, orc, do_blur):
    """what bench.py times: the blend amount is read by the kernel from the device-resident shared parameter block
    (lgpu_chain_params.param_block_d, int32[4], [0] = blend amount), a different one every step, 16 tracks per launch;
    the kernel-argument bf is deliberately wrong so that a launch that ignores the block cannot pass"""
    import torch
    rng = np.random.default_rng(5100 + do_blur)
    lut = lut_for(rng, "l2s")
    sw, sh, dw, dh, T = 384, 216, 192, 108, 16
    srcs = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(T)]
    l2s = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(T)]
    l2s[3][:, 3::4] = 255                                         # one fully opaque layer 2 (the wave-uniform integer path)
    d_src, d_l2 = [dev(s) for s in srcs], [dev(s) for s in l2s]
    d_dst = [dev(np.zeros((dh, align(dw * 4)), np.uint8)) for _ in range(T)]
    schedule = torch.tensor([[b, 0, 0, 0] for b in (0, 1, 96, 128, 200, 254, 255, 256 + 77)], dtype=torch.int32, device="cuda")
    prm = gpu.chain_params(sw, sh, srcs[0].strides[0], dw, dh, l2s[0].strides[0], d_dst[0].stride(0), swap_rb=1, interp=3, do_blur=do_blur,
                           bf=13, lut=lut, param_block=schedule)
    trk = gpu.chain_tracks(d_src, d_l2, d_dst)
    for s in range(schedule.shape[0]):
        prm.param_block_d = schedule.data_ptr() + 16 * s
        gpu.chain(prm, trk)
        bf = int(schedule[s, 0].item()) & 0xFF                   # the block's low byte is the amount
        for i in (0, 3, 7, 15):
            want = np.zeros((dh, align(dw * 4)), np.uint8)
            assert orc.orc_chain(P(srcs[i]), srcs[i].strides[0], sw, sh, P(l2s[i]), l2s[i].strides[0], P(want), want.strides[0], dw, dh, 1, 3, do_blur, bf, P(lut)) == 0
            assert_same(host(d_dst[i]), want, dw, dh, 4, "param block step %d (bf=%d) track %d blur=%d" % (s, bf, i, do_blur))


def test_chain_with_spare_workgroup_slots(gpu, tune):
    """multi-GPU hosts run the persistent chain kernel with a few workgroup slots left free (LGPU_CHAIN_SPARE_WGS, for RCCL's broadcast): another grid size
    and tile-list stride, the same bytes -- at the bench's geometry, one 4K track"""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    src = torch.randint(0, 256, (sh, sw * 4), dtype=torch.uint8, device="cuda", generator=g)
    l2 = torch.randint(0, 256, (dh, dw * 4), dtype=torch.uint8, device="cuda", generator=g)
    lut = lut_for(np.random.default_rng(1), "l2s")
    outs = []
    for spare in (None, "8", "16", "100"):
        tune("CHAIN_SPARE_WGS", None if spare is None else int(spare))
        d = torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda")
        prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3, do_blur=0, bf=99, lut=lut)
        gpu.chain(prm, gpu.chain_tracks([src], [l2], [d]))
        torch.cuda.synchronize()
        outs.append(d)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("use_lut", [0, 1])
def test_chain_every_alpha_and_colour_pair(gpu, orc, use_lut):
    """the translucent scaling of the chroma blend (simple_blend.c:137-145) inside the fused chain, for every (layer-2 alpha, colour)
    pair against every track colour class: layer 2 sweeps alpha along y and colour along x; the track is a flat frame per call
    (a flat source stays flat through the resize), so one launch covers 256 x 256 (alpha, c2) pairs for a given c1"""
    rng = np.random.default_rng(5200 + use_lut)
    lut = lut_for(rng, "l2s") if use_lut else None
    dw, dh = 256, 256
    sw, sh = 2 * dw, 2 * dh
    l2 = np.zeros((dh, dw * 4), np.uint8)
    xs = np.arange(dw, dtype=np.uint8)
    l2[:, 0::4] = xs[None, :]
    l2[:, 1::4] = (255 - xs)[None, :]
    l2[:, 2::4] = (xs * 7 + 3)[None, :]
    l2[:, 3::4] = np.arange(dh, dtype=np.uint8)[:, None]
    d_l2 = dev(l2)
    for c1 in list(range(0, 256, 5)) + [1, 2, 127, 128, 254, 255]:
        src = np.zeros((sh, sw * 4), np.uint8)
        src[:, 0::4] = c1
        src[:, 1::4] = 255 - c1
        src[:, 2::4] = (c1 * 3 + 1) & 0xFF
        src[:, 3::4] = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        for bf in (255, 0, 128, 77):
            want = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_chain(P(src), sw * 4, sw, sh, P(l2), dw * 4, P(want), dw * 4, dw, dh, 0, 3, 0, bf, P(lut) if use_lut else None) == 0
            d_dst = dev(np.zeros((dh, dw * 4), np.uint8))
            prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=0, interp=3, do_blur=0, bf=bf, lut=lut)
            gpu.chain(prm, gpu.chain_tracks([dev(src)], [d_l2], [d_dst]))
            assert_same(host(d_dst), want, dw, dh, 4, "c1=%d bf=%d lut=%d" % (c1, bf, use_lut))


# ---------------------------------------------------------------------------------------------- host threads
def test_multi_launch_paths_from_two_host_threads(gpu):
    """LiVES calls from several host threads; the multi-launch paths (chain with blur, in-place deinterlace, edge) keep their
    intermediates per (device, stream) and enqueue each sequence atomically: results equal the single-threaded ones"""
    import threading
    import torch
    rng = np.random.default_rng(2800)
    sw, sh, dw, dh = 384, 216, 192, 108
    jobs = []
    for t in range(2):
        src, l2 = frame(rng, sw, sh, 4), frame(rng, dw, dh, 4, alpha_mix=True)
        de = frame(rng, 300, 120, 3)
        jobs.append(dict(src=dev(src), l2=dev(l2), de=dev(de), ed=dev(frame(rng, 320, 200, 4))))

    def run(j, out):
        d = torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda")
        prm = gpu.chain_params(sw, sh, j["src"].stride(0), dw, dh, j["l2"].stride(0), d.stride(0), do_blur=1, bf=100)
        gpu.chain(prm, gpu.chain_tracks([j["src"]], [j["l2"]], [d]))
        x = j["de"].clone()
        gpu.deinterlace(x, x, 300, 120, 1)
        e = torch.zeros_like(j["ed"])
        gpu.edge(j["ed"], e, 320, 200, 3, 2)
        torch.cuda.synchronize()
        out.append((host(d), host(x), host(e)))

    want = []
    for j in jobs:
        o = []
        run(j, o)
        want.append(o[0])
    for _ in range(10):
        outs = [[], []]
        th = [threading.Thread(target=run, args=(jobs[i], outs[i])) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for i in range(2):
            for a, b in zip(outs[i][0], want[i]):
                assert (a == b).all()


def test_chain_is_graph_capturable(gpu):
    """steady-state launches neither allocate, copy from the host nor synchronise (tables / LUT / track pointers travel as kernel arguments):
    the fused chain, with and without the blur stage, can be captured into a HIP graph and replayed on new frame contents"""
    import torch
    rng = np.random.default_rng(3500)
    sw, sh, dw, dh, T = 384, 216, 192, 108, 4
    lut = lut_for(rng, "l2s")
    for blur in (0, 1):
        srcs = [dev(frame(rng, sw, sh, 4, alpha_mix=True)) for _ in range(T)]
        l2s = [dev(frame(rng, dw, dh, 4, alpha_mix=True)) for _ in range(T)]
        dsts = [torch.zeros((dh, align(dw * 4)), dtype=torch.uint8, device="cuda") for _ in range(T)]
        prm = gpu.chain_params(sw, sh, srcs[0].stride(0), dw, dh, l2s[0].stride(0), dsts[0].stride(0), do_blur=blur, bf=77, lut=lut)
        trk = gpu.chain_tracks(srcs, l2s, dsts)
        gpu.chain(prm, trk)                                   # warm-up: filter banks, fragments, scratch are created here, once
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            gpu.chain(prm, trk)                               # the scratch of the blur path is keyed by (device, stream): create it for this stream first
            st.synchronize()
            with torch.cuda.graph(g, stream=st):
                gpu.chain(prm, trk)
        # new contents in the same buffers, direct call = the expected result
        for t in range(T):
            srcs[t].copy_(dev(frame(rng, sw, sh, 4, alpha_mix=True)))
        gpu.chain(prm, trk)
        torch.cuda.synchronize()
        want = [d.clone() for d in dsts]
        for d in dsts:
            d.zero_()
        g.replay()
        torch.cuda.synchronize()
        for t in range(T):
            assert torch.equal(dsts[t], want[t]), "graph replay differs (blur=%d track %d)" % (blur, t)


def test_pinned_allocator_and_pageable_staging(gpu):
    """lgpu_upload / lgpu_download on pageable memory (staged in two pinned chunks per thread, several chunks long) and on lgpu_pinned_calloc memory
    (direct DMA) move the same bytes"""
    import torch
    from lives_amd.lib import load, call
    L = load()
    rng = np.random.default_rng(3600)
    for n in (1000, 300 * 1024, 9 * 1024 * 1024 + 123):
        src = rng.integers(0, 256, n, dtype=np.uint8)
        d = torch.zeros(n, dtype=torch.uint8, device="cuda")
        call("lgpu_upload", d.data_ptr(), src.ctypes.data, n, None)
        back = np.zeros(n, np.uint8)
        call("lgpu_download", back.ctypes.data, d.data_ptr(), n, None)
        call("lgpu_sync", None)
        assert (back == src).all() and (d.cpu().numpy() == src).all()
        p = L.lgpu_pinned_calloc(n)
        assert p
        buf = np.frombuffer((ctypes.c_uint8 * n).from_address(p), np.uint8)
        assert not buf.any()
        buf[:] = src[::-1]
        call("lgpu_upload", d.data_ptr(), p, n, None)
        call("lgpu_sync", None)
        assert (d.cpu().numpy() == src[::-1]).all()
        del buf
        L.lgpu_pinned_free(p)


# ---------------------------------------------------------------------------------------------- multi-GPU C entry points
def test_rccl_entry_points_on_a_one_rank_communicator(gpu):
    """include/lives_gpu.h "multi-GPU exchange": RCCL bound at run time, a communicator from the 128-byte id, the parameter block
    broadcast in place, the status max and the compositing fan-in -- on the one GPU of this box as a world of one rank (the
    two-rank protocol is covered over gloo in tests/test_dist_cpu.py; N > 1 runs on the driver's 8-GPU node through bench.py)"""
    import torch
    from lives_amd import dist as ld
    comm = ld.RcclComm("cuda")
    assert (comm.rank, comm.world) == (0, 1)
    blk = torch.tensor([107, 1, 2, 3], dtype=torch.int32, device="cuda")
    comm.broadcast_params(blk)
    st = torch.tensor([5], dtype=torch.int32, device="cuda")
    comm.status_max(st)
    frames = torch.arange(3 * 64, dtype=torch.int32, device="cuda").to(torch.uint8).reshape(3, 64).contiguous()
    out = torch.zeros((3, 64), dtype=torch.uint8, device="cuda")
    comm.fan_in(frames, 3, 64, out)
    torch.cuda.synchronize()
    assert blk.tolist() == [107, 1, 2, 3] and st.item() == 5 and torch.equal(out, frames)
    # the pipelined parameter block on the C path drives the kernel: same frames as the kernel-argument amount
    rng = np.random.default_rng(6100)
    sw, sh, dw, dh = 256, 144, 128, 72
    src, l2 = dev(frame(rng, sw, sh, 4)), dev(frame(rng, dw, dh, 4, alpha_mix=True))
    pipe = ld.ParamPipeline("cuda", comm=comm)
    pipe.prefetch(0, [41, 0])
    for s in range(4):
        b = pipe.acquire(s)
        if s + 1 < 4:
            pipe.prefetch(s + 1, [41 + 50 * (s + 1), 0])
        d1, d2 = torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda"), torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda")
        prm = gpu.chain_params(sw, sh, src.stride(0), dw, dh, l2.stride(0), dw * 4, bf=13, param_block=b)
        gpu.chain(prm, gpu.chain_tracks([src], [l2], [d1]))
        prm2 = gpu.chain_params(sw, sh, src.stride(0), dw, dh, l2.stride(0), dw * 4, bf=41 + 50 * s)
        gpu.chain(prm2, gpu.chain_tracks([src], [l2], [d2]))
        torch.cuda.synchronize()
        assert torch.equal(d1, d2), s
    comm.close()


@pytest.mark.gpu
def test_stream_ordered_allocation():
    """lgpu_malloc_ordered / lgpu_free_ordered (what the layer seam's resident planes come from): blocks of the device's stream-ordered pool are usable by
    work enqueued after the call, freed blocks come back without a device synchronisation, and lgpu_debug_fail_alloc counts these allocations too"""
    from lives_amd import lib
    L = lib.load()
    assert L.lgpu_init(0) == 0
    ptrs = []
    for i, n in enumerate((1 << 20, 2400000, 8294400)):
        p = ctypes.c_void_p()
        assert L.lgpu_malloc_ordered(ctypes.byref(p), n, None) == 0 and p.value
        assert L.lgpu_fill(p, 17 + i, n, None) == 0
        ptrs.append((p, n, 17 + i))
    for p, n, v in ptrs:
        back = np.zeros(n, np.uint8)
        assert L.lgpu_download(back.ctypes.data, p, n, None) == 0 and L.lgpu_sync(None) == 0
        assert (back == v).all()
        assert L.lgpu_free_ordered(p, None) == 0
    assert L.lgpu_free_ordered(None, None) == 0
    # a freed block is handed out again to later work (no sync in between), and holds whatever that work writes
    for k in range(8):
        p = ctypes.c_void_p()
        assert L.lgpu_malloc_ordered(ctypes.byref(p), 2400000, None) == 0
        assert L.lgpu_fill(p, k, 2400000, None) == 0
        back = np.zeros(16, np.uint8)
        assert L.lgpu_download(back.ctypes.data, ctypes.c_void_p(p.value + 2400000 - 16), 16, None) == 0
        assert L.lgpu_free_ordered(p, None) == 0
        assert L.lgpu_sync(None) == 0 and (back == k).all()
    L.lgpu_debug_fail_alloc.argtypes = [ctypes.c_int]
    L.lgpu_debug_fail_alloc(1)
    p = ctypes.c_void_p(1)
    rc = L.lgpu_malloc_ordered(ctypes.byref(p), 4096, None)
    L.lgpu_debug_fail_alloc(0)
    assert rc == -5 and not p.value
    assert L.lgpu_malloc_ordered(ctypes.byref(p), 4096, None) == 0 and L.lgpu_free_ordered(p, None) == 0


@pytest.mark.gpu
def test_entry_points_run_on_the_stream_they_are_given(gpu):
    """every frame-level entry point takes a stream: work given to a side stream must be complete when THAT stream has been synchronised (nothing may have been
    launched on the null stream instead) and must equal the same calls on the null stream.  A large fill keeps the side stream busy first, so a kernel that went to
    another stream would overtake it and be visible as a mismatch; the ops with internal scratch (3-byte resize, the blur chain, gauss5, K2) are the ones that matter."""
    import torch
    rng = np.random.default_rng(4711)
    sw, sh, dw, dh = 640, 360, 320, 180
    src4, src3 = dev(frame(rng, sw, sh, 4)), dev(frame(rng, sw, sh, 3))
    l2 = dev(frame(rng, dw, dh, 4, alpha_mix=True))
    Y = dev(rng.integers(16, 236, (sh, sw), dtype=np.uint8)); U = dev(rng.integers(16, 241, (sh // 2, sw // 2), dtype=np.uint8)); V = dev(rng.integers(16, 241, (sh // 2, sw // 2), dtype=np.uint8))
    lut = np.arange(255, -1, -1, dtype=np.uint8)
    big = torch.zeros(512 << 20, dtype=torch.uint8, device="cuda")

    def run_all():
        outs = []
        o = torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda"); gpu.resize(src4, o, sw, sh, dw, dh, psize=4, interp=3); outs.append(o)
        o = torch.zeros((dh, align(dw * 3)), dtype=torch.uint8, device="cuda"); gpu.resize(src3, o, sw, sh, dw, dh, psize=3, interp=3); outs.append(o)
        o = torch.zeros((200, 300 * 4), dtype=torch.uint8, device="cuda"); gpu.resize(src4, o, sw, sh, 300, 200, psize=4, interp=2); outs.append(o)
        o = torch.zeros_like(src4); gpu.gauss5(src4, o, sw, sh, psize=4); outs.append(o)
        o = torch.zeros((sh, sw * 4), dtype=torch.uint8, device="cuda"); gpu.yuv420p_to_rgb(Y, U, V, o, sw, sh, lut=lut); outs.append(o)
        for blur in (0, 1):
            o = torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda")
            prm = gpu.chain_params(sw, sh, src4.stride(0), dw, dh, l2.stride(0), dw * 4, swap_rb=1, interp=3, do_blur=blur, bf=77, lut=lut)
            gpu.chain(prm, gpu.chain_tracks([src4], [l2], [o])); outs.append(o)
        o = torch.zeros_like(src4); gpu.mirror(2, src4, o, sw, sh, 4); outs.append(o)
        # effects with device state of their own (edge: map + histogram scratch per stream; deinterlace: snapshot per stream for in-place calls)
        o = torch.zeros_like(src4); gpu.edge(src4, o, sw, sh, 3, 0); outs.append(o)
        o = src4.clone(); gpu.deinterlace(o, o, 636, sh, 3); outs.append(o)          # the reference wants width % 3 == 0 here
        o = torch.zeros_like(src4); gpu.blend_chroma(src4, src4.flip(0).contiguous(), o, sw, sh, 4, 99); outs.append(o)
        o = torch.zeros_like(src4); gpu.slide_over(src4, src4.flip(0).contiguous(), o, sw, sh, 4, 0.4, 2); outs.append(o)
        return outs

    want = run_all()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(4):
            big.fill_(1)                                   # ~0.4 ms of work in front on the side stream
        got = run_all()
    side.synchronize()                                     # this stream only
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("psize", [1, 3, 4])
def test_letterbox_bars_paint_everything_but_the_inner_frame(gpu, psize):
    """lgpu_letterbox_bars (what letterbox_layer uses when the scaler writes the inner frame straight into the canvas): black outside the rectangle, not one byte inside,
    row padding untouched; odd offsets and sizes, rectangles that touch the canvas edges"""
    from lives_amd import lib
    import torch
    rng = np.random.default_rng(60 + psize)
    black = {1: [16, 0, 0, 0], 3: [1, 2, 3, 0], 4: [0, 0, 0, 255]}[psize]
    for (nw, nh, ox, oy, w, h) in ((96, 64, 10, 7, 50, 33), (64, 48, 0, 6, 64, 36), (61, 37, 5, 0, 51, 37), (40, 40, 0, 0, 40, 40), (33, 9, 32, 8, 1, 1)):
        canvas = frame(rng, nw, nh, psize)
        want = canvas.copy()
        px = np.array(black[:psize], np.uint8)
        for y in range(nh):
            for x in range(nw):
                if not (oy <= y < oy + h and ox <= x < ox + w):
                    want[y, x * psize:(x + 1) * psize] = px
        d = dev(canvas)
        b = (ctypes.c_uint8 * 4)(*black)
        lib.call("lgpu_letterbox_bars", d.data_ptr(), d.stride(0), nw, nh, psize, b, ox, oy, w, h, None)
        torch.cuda.synchronize()
        assert (host(d) == want).all(), (psize, nw, nh, ox, oy, w, h)


@pytest.mark.gpu
def test_stream_and_event_entry_points():
    """lgpu_stream_create / lgpu_event_*: two streams of the library, work ordered from one to the other by an event (what the layer seam does at a cross-thread
    hand-over), both the blocking and the non-blocking kind"""
    from lives_amd import lib
    L = lib.load()
    assert L.lgpu_init(0) == 0
    n = 64 << 20
    for nonblocking in (0, 1):
        sa, sb, ev, buf = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert L.lgpu_stream_create(ctypes.byref(sa), nonblocking) == 0 and L.lgpu_stream_create(ctypes.byref(sb), nonblocking) == 0
        assert L.lgpu_event_create(ctypes.byref(ev)) == 0
        assert L.lgpu_malloc_ordered(ctypes.byref(buf), n, sa) == 0
        for v in (1, 2, 3, 4, 5, 6, 7, 8):
            assert L.lgpu_fill(buf, v, n, sa) == 0                      # a queue of fills on stream a, the last one wins
        assert L.lgpu_event_record(ev, sa) == 0 and L.lgpu_stream_wait_event(sb, ev) == 0
        back = np.zeros(4096, np.uint8)
        assert L.lgpu_download(back.ctypes.data, ctypes.c_void_p(buf.value + n - 4096), 4096, sb) == 0      # on stream b, behind the event
        assert L.lgpu_sync(sb) == 0
        assert (back == 8).all()
        assert L.lgpu_free_ordered(buf, sb) == 0 and L.lgpu_sync(sb) == 0 and L.lgpu_sync(sa) == 0
        assert L.lgpu_event_destroy(ev) == 0 and L.lgpu_stream_destroy(sa) == 0 and L.lgpu_stream_destroy(sb) == 0
    assert L.lgpu_event_record(None, None) < 0 and L.lgpu_stream_wait_event(None, None) < 0
