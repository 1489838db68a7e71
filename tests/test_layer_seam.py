"""The weed_layer_t seam (include/lives_gpu_layer.h) on genuine weed plants (reference libweed as the host).

Mirrors how LiVES calls these functions (src/nodemodel.c:1065-1253) on the BASELINE configs:
  C1  640x480 RGB24 -> convert_layer_palette(BGRA32)
  C2  YUV420P -> convert_layer_palette(RGBA32) + gamma_convert_layer(SRGB)
  C3  RGBA32 -> resize_layer (bicubic 0.5x) -> letterbox_layer -> (plugin blend is covered in test_plugin_seam.py)
"""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, frame

needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (reference libweed) not built")
P = po.P
RGB24, BGR24, RGBA32, BGRA32, ARGB32, YUV420P, YVU420P, YUV888 = 1, 2, 3, 4, 5, 512, 513, 588


@pytest.fixture(scope="module")
def seam():
    from lives_amd import lib
    from tests import weedhost
    L = lib.load()
    weedhost.bind(L)
    # this module exercises the swscale body's stand-in ("host built with USE_SWSCALE"): every palette, fused target gamma, the library's own polyphase spec.
    # The seam's default is the pinned gdk-pixbuf body (tests/test_layer_seam_pixbuf.py, tests/test_dropin.py).
    assert L.lives_gpu_set_resize_backend(0) == 0
    yield L, weedhost
    assert L.lives_gpu_set_resize_backend(1) == 0


@needs_ref
def test_host_side_functions_without_gpu(seam):
    L, wh = seam
    n = ctypes.c_int()
    rs = L.lives_gpu_calc_rowstrides(640, RGB24, None, ctypes.byref(n))
    assert n.value == 1 and rs[0] == 1920
    rs = L.lives_gpu_calc_rowstrides(1918, YUV420P, None, ctypes.byref(n))
    assert n.value == 3 and [rs[0], rs[1], rs[2]] == [1920, 960, 960]
    # create_empty_pixel_data: black fill rules of src/colourspace.c:11448-11460
    lay = wh.new_layer(RGBA32, 10, 4, [np.full((4, 64), 7, np.uint8)])
    assert L.lives_gpu_create_empty_pixel_data(lay, 1, 1) == 1
    planes, _, rs = wh.planes_of(lay)
    assert rs == [64] and (planes[0][:, :40].reshape(4, 10, 4) == [0, 0, 0, 255]).all()
    lay = wh.new_layer(YUV420P, 11, 5, [np.zeros((5, 32), np.uint8), np.zeros((2, 16), np.uint8), np.zeros((2, 16), np.uint8)], clamping=0)
    assert L.lives_gpu_create_empty_pixel_data(lay, 1, 1) == 1
    planes, _, rs = wh.planes_of(lay)
    assert wh.geti(lay, "width") == 10 and wh.geti(lay, "height") == 4 and rs == [32, 16, 16]
    assert (planes[0] == 16).all() and (planes[1] == 128).all() and (planes[2] == 128).all()
    # fixed rowstrides of a decoder plugin (src/colourspace.c:11268-11275, :11358-11363): the rowstrides leaf flagged LIVES_FLAG_CONST_VALUE
    # (1 << 16) wins per plane while computed <= fixed < 2 * computed
    W = wh.weed()
    set_flags = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32)(W.fn["weed_leaf_set_flags"])
    get_flags = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_char_p)(W.fn["weed_leaf_get_flags"])
    lay = wh.new_layer(YUV420P, 640, 4, [np.zeros((4, 704), np.uint8), np.zeros((2, 352), np.uint8), np.zeros((2, 2048), np.uint8)], clamping=0)
    rs = L.lives_gpu_calc_rowstrides(0, 0, lay, ctypes.byref(n))
    assert [rs[0], rs[1], rs[2]] == [640, 320, 320]                       # not flagged: the ordinary rule
    set_flags(lay, b"rowstrides", get_flags(lay, b"rowstrides") | (1 << 16))
    rs = L.lives_gpu_calc_rowstrides(0, 0, lay, ctypes.byref(n))
    assert [rs[0], rs[1], rs[2]] == [704, 352, 320]                       # 2048 >= 2 * 320: that plane keeps the computed stride
    assert L.lives_gpu_create_empty_pixel_data(lay, 0, 1) == 1
    assert wh.planes_of(lay)[2] == [704, 352, 320]


@needs_ref
def test_failure_leaves_layer_untouched(seam):
    """no device (this box) or a declined conversion: FALSE, same planes, same leaves"""
    L, wh = seam
    src = np.arange(4 * 64, dtype=np.uint8).reshape(4, 64)
    lay = wh.new_layer(ARGB32, 10, 4, [src], gamma=1)
    before, ptrs, _ = wh.planes_of(lay)
    assert L.lives_gpu_convert_layer_palette(lay, YUV420P, 0) == 0            # ARGB32 -> 4:2:0: reference-broken (K4-d), declined everywhere
    after, ptrs2, _ = wh.planes_of(lay)
    assert ptrs == ptrs2 and (before[0] == after[0]).all() and wh.geti(lay, "current_palette") == ARGB32


@needs_ref
@pytest.mark.gpu
def test_declines_and_device_failures_leave_the_layer_untouched(seam, orc):
    """the memfail: contract (src/colourspace.c:13906-13927) on a real device: (1) conversions this library declines, (2) calls that fail
    AFTER they have started -- a device allocation fails half way (lgpu_debug_fail_alloc) -- return FALSE with the same plane
    pointers, the same bytes and the same leaves; the same call then succeeds"""
    L, wh = seam
    rng = np.random.default_rng(77)
    L.lgpu_debug_fail_alloc.argtypes = [ctypes.c_int]

    def snapshot(lay):
        planes, ptrs, rs = wh.planes_of(lay)
        return ([p.copy() for p in planes], ptrs, rs, wh.geti(lay, "current_palette"), wh.geti(lay, "width"), wh.geti(lay, "height"),
                wh.geti(lay, "gamma_type"), wh.geti(lay, "host_flags"))

    def same(a, b):
        return a[1:] == b[1:] and all((x == y).all() for x, y in zip(a[0], b[0]))

    # (1) a declined pair: ARGB32 -> 4:2:0 (K4-d)
    src = frame(rng, 64, 32, 4)
    lay = wh.new_layer(ARGB32, 64, 32, [src], gamma=1)
    s0 = snapshot(lay)
    assert L.lives_gpu_convert_layer_palette(lay, YUV420P, 0) == 0 and same(s0, snapshot(lay))
    Y, U, V = frame(rng, 64, 32, 1), frame(rng, 32, 32, 1), frame(rng, 32, 32, 1)
    # a subspace change between YUV palettes goes through RGB(A) as in the reference (src/colourspace.c:12248-12262): equal to the two conversions done by hand
    for (pal, planes, outpl, osub) in ((522, [Y, U, V], YUV888, 0), (YUV888, [frame(rng, 64, 32, 3)], 589, 2)):
        lay = wh.new_layer(pal, 64, 32, planes, clamping=0, subspace=1)
        assert L.lives_gpu_convert_layer_palette_full(lay, outpl, 0, 0, osub, 0) == 1
        twin = wh.new_layer(pal, 64, 32, planes, clamping=0, subspace=1)
        assert L.lives_gpu_convert_layer_palette(twin, RGB24, 0) == 1 and L.lives_gpu_convert_layer_palette_full(twin, outpl, 0, 0, osub, 0) == 1
        a, b = snapshot(lay), snapshot(twin)
        assert a[2:] == b[2:] and all((x == y).all() for x, y in zip(a[0], b[0])) and wh.geti(lay, "current_palette") == outpl
    # (2) injected allocation failures inside calls that are served: sizes nobody used before, so every call has to allocate
    for nth, (bw, bh) in ((1, (2303, 1301)), (2, (2603, 1501))):     # larger than any frame so far: both scratch slots have to grow in each round
        big = frame(rng, bw, bh, 4)
        lay = wh.new_layer(RGBA32, bw, bh, [big], gamma=1)
        s0 = snapshot(lay)
        L.lgpu_debug_fail_alloc(nth)
        rc = L.lives_gpu_convert_layer_palette(lay, BGR24, 0)
        L.lgpu_debug_fail_alloc(0)
        assert rc == 0 and same(s0, snapshot(lay)), "failed allocation %d" % nth
        assert L.lives_gpu_convert_layer_palette(lay, BGR24, 0) == 1 and wh.geti(lay, "current_palette") == BGR24
    big = frame(rng, 2811, 1607, 4)
    lay = wh.new_layer(RGBA32, 2811, 1607, [big])
    s0 = snapshot(lay)
    L.lgpu_debug_fail_alloc(1)
    rc = L.lives_gpu_resize_layer(lay, 1404, 802, 3, 0, 0)
    L.lgpu_debug_fail_alloc(0)
    assert rc == 0 and same(s0, snapshot(lay))
    assert L.lives_gpu_resize_layer(lay, 1404, 802, 3, 0, 0) == 1
    # a pinned layer: the failure happens when the new resident plane is taken from the pool
    lay = wh.new_layer(RGBA32, 3011, 1693, [frame(rng, 3011, 1693, 4)], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0
    s0 = snapshot(lay)
    L.lgpu_debug_fail_alloc(1)
    rc = L.lives_gpu_convert_layer_palette(lay, BGR24, 0)
    L.lgpu_debug_fail_alloc(0)
    # FALSE on a pinned layer, whatever the cause: the layer comes back synchronised and unpinned, so that the host's CPU body reads current bytes
    assert rc == 0 and same(s0, snapshot(lay)) and wh.geti(lay, "host_gpu_resident") is None
    assert L.lives_gpu_convert_layer_palette(lay, BGR24, 0) == 1


@needs_ref
@pytest.mark.gpu
def test_c1_rgb24_to_bgra32(seam, orc):
    L, wh = seam
    rng = np.random.default_rng(1)
    w, h = 640, 480
    src = frame(rng, w, h, 3)
    lay = wh.new_layer(RGB24, w, h, [src], gamma=1)
    assert L.lives_gpu_convert_layer_palette(lay, BGRA32, 0) == 1
    planes, _, rs = wh.planes_of(lay)
    assert (wh.geti(lay, "current_palette"), wh.geti(lay, "width"), wh.geti(lay, "height"), rs) == (BGRA32, w, h, [2560])
    want = np.zeros((h, 2560), np.uint8)
    orc.orc_swizzle(po.OPS.index("swap3addpost"), 0, P(src), src.strides[0], P(want), 2560, w, h, None)
    assert (planes[0] == want).all()
    assert wh.geti(lay, "gamma_type") == 1 and wh.geti(lay, "host_flags") == 1      # alpha added -> premult flag (:12298-12301)
    # and every other RGB <-> RGB pair, against the oracle op the selector tree picks
    names = {(RGB24, BGR24): "swap3", (RGB24, RGBA32): "addpost", (BGR24, RGBA32): "swap3addpost", (RGB24, ARGB32): "addpre",
             (BGR24, ARGB32): "swap3addpre", (RGBA32, RGB24): "delpost", (RGBA32, BGR24): "swap3delpost", (RGBA32, BGRA32): "swap3postalpha",
             (ARGB32, RGB24): "delpre", (ARGB32, BGR24): "swap3delpre", (RGBA32, ARGB32): "swapprepost", (ARGB32, RGBA32): "swapprepost",
             (BGRA32, ARGB32): "swap4", (ARGB32, BGRA32): "swap4"}
    psz = {RGB24: 3, BGR24: 3, RGBA32: 4, BGRA32: 4, ARGB32: 4}
    for (ip, op_), name in names.items():
        s = frame(rng, 66, 34, psz[ip])
        lay = wh.new_layer(ip, 66, 34, [s])
        assert L.lives_gpu_convert_layer_palette(lay, op_, 0) == 1, name
        planes, _, rs = wh.planes_of(lay)
        want = np.zeros((34, align(66 * psz[op_])), np.uint8)
        orc.orc_swizzle(po.OPS.index(name), int(ip == ARGB32), P(s), s.strides[0], P(want), want.strides[0], 66, 34, None)
        assert rs == [want.strides[0]] and (planes[0][:, :66 * psz[op_]] == want[:, :66 * psz[op_]]).all(), name


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("size", [(128, 72), (1920, 1080)])
def test_c2_yuv420p_to_rgba_then_gamma(seam, orc, size):
    L, wh = seam
    rng = np.random.default_rng(2)
    w, h = size
    ys, cs = align(w), align(w) >> 1
    Y = rng.integers(16, 236, (h, ys), dtype=np.uint8)
    U = rng.integers(16, 241, (h // 2, cs), dtype=np.uint8)
    V = rng.integers(16, 241, (h // 2, cs), dtype=np.uint8)
    lay = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)       # linear gamma, clamped YCbCr
    assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1
    assert wh.geti(lay, "current_palette") == RGBA32 and wh.geti(lay, "YUV_clamping") is None
    assert L.lives_gpu_gamma_convert_layer(1, lay) == 1                                   # -> WEED_GAMMA_SRGB
    assert wh.geti(lay, "gamma_type") == 1
    planes, _, rs = wh.planes_of(lay)
    orow = align(w * 4)
    want = np.zeros((h, orow), np.uint8)
    strides = (ctypes.c_int * 3)(ys, cs, cs)
    orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, U.size, V.size, P(want), orow, w, h, 4, 0, 0, 0, 2, None, 0)
    lut = np.zeros(256, np.uint8)
    assert orc.orc_gamma_lut8(1.0, -1, 1, 1.4, P(lut)) == 1
    orc.orc_gamma_apply(P(want), orow, w, h, 4, 0, P(lut))
    assert rs == [orow] and (planes[0][:, :w * 4] == want[:, :w * 4]).all()
    # convert_layer_palette_full with a target gamma: the reference fuses the 16-bit LUT into the conversion (:3274-3283)
    lay = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette_full(lay, RGBA32, 0, 0, 1, 1) == 1
    assert wh.geti(lay, "gamma_type") == 1
    planes, _, rs = wh.planes_of(lay)
    lut16 = np.zeros(65536, np.uint16)
    assert orc.orc_gamma_lut16(1.0, -1, 1, 1.4, P(lut16)) == 1
    want = np.zeros((h, orow), np.uint8)
    orc.orc_yuv420p_to_rgb_lut16(P(Y), P(U), P(V), strides, U.size, V.size, P(want), orow, w, h, 4, 0, 0, 0, 2, P(lut16), 0)
    assert (planes[0][:, :w * 4] == want[:, :w * 4]).all()


@needs_ref
@pytest.mark.gpu
def test_c3_resize_then_letterbox(seam, orc):
    L, wh = seam
    rng = np.random.default_rng(3)
    sw, sh, dw, dh, nw, nh = 384, 216, 192, 108, 192, 120
    src = frame(rng, sw, sh, 4)
    lay = wh.new_layer(RGBA32, sw, sh, [src])
    assert L.lives_gpu_letterbox_layer(lay, nw, nh, dw, dh, 3, 0, 0) == 1          # LIVES_INTERP_BEST
    planes, _, rs = wh.planes_of(lay)
    assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (nw, nh)
    rs_want = np.zeros((dh, dw * 4), np.uint8)
    assert orc.orc_resize(P(src), src.strides[0], sw, sh, P(rs_want), dw * 4, dw, dh, 4, 3) == 0
    want = np.zeros((nh, rs[0]), np.uint8)
    orc.orc_letterbox(P(rs_want), dw * 4, dw, dh, P(want), rs[0], nw, nh, 4, P(np.array([0, 0, 0, 255], np.uint8)))
    assert (planes[0] == want).all()
    # planar: YUV420P resize keeps the plane geometry rules
    Y = rng.integers(0, 256, (64, 128), dtype=np.uint8)
    U = rng.integers(0, 256, (32, 64), dtype=np.uint8)
    V = rng.integers(0, 256, (32, 64), dtype=np.uint8)
    lay = wh.new_layer(YUV420P, 128, 64, [Y, U, V], clamping=0)
    assert L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0) == 1
    planes, _, rs = wh.planes_of(lay)
    assert rs == [64, 32, 32] and (wh.geti(lay, "width"), wh.geti(lay, "height")) == (64, 32)
    for pl, s, (pw, ph, qw, qh) in zip(planes, (Y, U, V), ((128, 64, 64, 32), (64, 32, 32, 16), (64, 32, 32, 16))):
        want = np.zeros((qh, pl.shape[1]), np.uint8)
        assert orc.orc_resize(P(s), s.strides[0], pw, ph, P(want), want.strides[0], qw, qh, 1, 3) == 0
        assert (pl[:, :qw] == want[:, :qw]).all()
    # a palette hint "may be ignored" (colourspace.c:14746-14751): the frame is resized in its own palette ...
    src = frame(rng, sw, sh, 4)
    lay = wh.new_layer(RGBA32, sw, sh, [src])
    assert L.lives_gpu_resize_layer(lay, dw, dh, 3, RGB24, 0) == 1
    planes, _, rs = wh.planes_of(lay)
    assert wh.geti(lay, "current_palette") == RGBA32 and (wh.geti(lay, "width"), wh.geti(lay, "height")) == (dw, dh)
    want = np.zeros((dh, rs[0]), np.uint8)
    assert orc.orc_resize(P(src), src.strides[0], sw, sh, P(want), rs[0], dw, dh, 4, 3) == 0
    assert (planes[0][:, :dw * 4] == want[:, :dw * 4]).all()
    # ... and a packed-YUV frame goes to the hinted palette first (UYVY -> RGB24 is served here), then resizes
    uy = frame(rng, 128, 32, 2)
    lay = wh.new_layer(564, 64, 32, [uy], clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer(lay, 64, 16, 3, RGB24, 0) == 1
    planes, _, rs = wh.planes_of(lay)
    assert wh.geti(lay, "current_palette") == RGB24 and (wh.geti(lay, "width"), wh.geti(lay, "height")) == (64, 16)
    rgb = np.zeros((32, 128 * 3), np.uint8)
    sp, ss = po.planes_args([uy])
    assert orc.orc_yuv_to_rgb(ctypes.addressof(sp), ctypes.addressof(ss), 128, 32, 2, 0, P(rgb), rgb.strides[0], 0, 0, 0) == 0
    want = np.zeros((16, rs[0]), np.uint8)
    assert orc.orc_resize(P(rgb), rgb.strides[0], 128, 32, P(want), rs[0], 64, 16, 3, 3) == 0
    assert (planes[0][:, :64 * 3] == want[:, :64 * 3]).all()
    # no usable hint for a packed-YUV frame: FALSE, layer untouched
    lay = wh.new_layer(564, 64, 32, [uy], clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer(lay, 64, 16, 3, 0, 0) == 0 and wh.geti(lay, "current_palette") == 564


@needs_ref
@pytest.mark.gpu
def test_alpha_premult_layer(seam, orc):
    L, wh = seam
    rng = np.random.default_rng(4)
    s = frame(rng, 66, 34, 4)
    for direction, un in ((1, 0), (-1, 1)):
        lay = wh.new_layer(RGBA32, 66, 34, [s], flags=0 if un == 0 else 1)
        L.lives_gpu_alpha_premult(lay, direction)
        planes, _, _ = wh.planes_of(lay)
        want = s.copy()
        orc.orc_alpha_premult(P(want), want.strides[0], 66, 34, 0, un)
        assert (planes[0][:, :66 * 4] == want[:, :66 * 4]).all()
        assert wh.geti(lay, "host_flags") == (1 if direction == 1 else 0)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("pal", [589, 545])
def test_alpha_premult_yuva_layer(seam, orc, pal):
    """YUVA layers: the clamped tables when YUV_clamping is CLAMPED (or missing), al / unal when UNCLAMPED (src/colourspace.c:11982-12096)"""
    L, wh = seam
    rng = np.random.default_rng(pal)
    w, h = 50, 22
    for clamp_leaf in (None, 0, 1):
        for direction, un in ((1, 0), (-1, 1)):
            src = [frame(rng, w, h, 4)] if pal == 589 else [frame(rng, w, h, 1) for _ in range(4)]
            lay = wh.new_layer(pal, w, h, src, clamping=clamp_leaf, flags=0 if un == 0 else 1)
            L.lives_gpu_alpha_premult(lay, direction)
            planes, _, _ = wh.planes_of(lay)
            want = [s.copy() for s in src]
            pp = (ctypes.c_void_p * 4)(*([x.ctypes.data for x in want] + [None] * (4 - len(want))))
            ss = (ctypes.c_int * 4)(*([x.strides[0] for x in want] + [0] * (4 - len(want))))
            orc.orc_alpha_premult_yuva(pp, ss, w, h, pal, 0 if clamp_leaf == 1 else 1, un)
            bw = w * 4 if pal == 589 else w
            for i in range(len(src)):
                assert (planes[i][:, :bw] == want[i][:, :bw]).all(), (pal, clamp_leaf, direction, i)
            assert wh.geti(lay, "host_flags") == (1 if direction == 1 else 0)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("outpl", [RGB24, BGR24, RGBA32, BGRA32, ARGB32])
def test_yuv411_layer_to_rgb(seam, orc, outpl):
    """a YUV411 layer (width leaf in macropixels, src/colourspace.c:13755-13795): new width = 4 * width, YUV leaves deleted; the alpha
    bytes the reference leaves unwritten are those of a zeroed new frame"""
    L, wh = seam
    rng = np.random.default_rng(outpl)
    wm, h = 24, 10
    order = {RGB24: 0, RGBA32: 0, BGR24: 1, BGRA32: 1, ARGB32: 2}[outpl]
    ps = 3 if outpl in (RGB24, BGR24) else 4
    for clamp in (0, 1):
        src = rng.integers(0, 256, (h, wm * 6), dtype=np.uint8)
        lay = wh.new_layer(595, wm, h, [src], clamping=clamp, gamma=1)
        assert L.lives_gpu_convert_layer_palette(lay, outpl, clamp) == 1
        planes, _, rs = wh.planes_of(lay)
        assert wh.geti(lay, "current_palette") == outpl and wh.geti(lay, "width") == wm * 4 and wh.geti(lay, "YUV_clamping") is None
        want = np.zeros((h, rs[0]), np.uint8)
        assert orc.orc_yuv411_to_rgb(P(src), wm, h, P(want), want.strides[0], order, 1 if ps == 4 else 0, clamp) == 0
        assert (planes[0][:, :wm * 4 * ps] == want[:, :wm * 4 * ps]).all(), (outpl, clamp)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("inpl", [RGB24, BGR24, RGBA32, BGRA32, ARGB32])
def test_rgb_layer_to_yuv411(seam, orc, inpl):
    L, wh = seam
    rng = np.random.default_rng(500 + inpl)
    w, h = 52, 9
    order = {RGB24: 0, RGBA32: 0, BGR24: 1, BGRA32: 1, ARGB32: 2}[inpl]
    ips = 3 if inpl in (RGB24, BGR24) else 4
    for clamp in (0, 1):
        src = frame(rng, w, h, ips)
        lay = wh.new_layer(inpl, w, h, [src], gamma=1, flags=1 if ips == 4 else 0)
        assert L.lives_gpu_convert_layer_palette(lay, 595, clamp) == 1
        planes, _, rs = wh.planes_of(lay)
        assert wh.geti(lay, "current_palette") == 595 and wh.geti(lay, "width") == w >> 2 and wh.geti(lay, "YUV_clamping") == clamp
        want = np.zeros((w >> 2) * 6 * h, np.uint8)
        assert orc.orc_rgb_to_yuv411(P(src), src.strides[0], w, h, order, 1 if ips == 4 else 0, P(want), clamp) == 0
        assert (planes[0].reshape(-1)[:want.size] == want).all(), (inpl, clamp)     # compact rows from the start of the plane
        assert wh.geti(lay, "host_flags", 0) == 0


# ---- K4 / K3 on layers: the RGB -> YUV and YUV -> RGB cases of convert_layer_palette_full (src/colourspace.c:12559-13860) ----
K4_FMT = {588: 0, 589: 0, 544: 1, 545: 1, 564: 2, 565: 3, 512: 4, 513: 4, 522: 5}


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("inpl", [RGB24, BGR24, RGBA32, BGRA32, ARGB32])
def test_rgb_layers_to_yuv(seam, orc, inpl):
    L, wh = seam
    rng = np.random.default_rng(40 + inpl)
    w, h = 64, 16          # aligned width: the rowstrides the layer gets are compact, where the reference's 4:2:0 walk is sane
    ips = 3 if inpl in (RGB24, BGR24) else 4
    order = {RGB24: 0, RGBA32: 0, BGR24: 1, BGRA32: 1, ARGB32: 2}[inpl]
    for outpl, fmt in K4_FMT.items():
        for oclamp in (0, 1):
            src = frame(rng, w, h, ips)
            lay = wh.new_layer(inpl, w, h, [src], gamma=1, flags=1 if ips == 4 else 0)
            rc = L.lives_gpu_convert_layer_palette(lay, outpl, oclamp)
            if order == 2 and fmt >= 4:
                assert rc == 0 and wh.geti(lay, "current_palette") == inpl       # declined (reference-broken), layer untouched
                continue
            assert rc == 1, (inpl, outpl)
            planes, _, rs = wh.planes_of(lay)
            out_alpha = 1 if outpl in (589, 545) else 0
            want, dims = po.k4_out_planes(0, w, h, fmt, out_alpha)
            want = [np.zeros((a.shape[0], r), np.uint8) for a, r in zip(want, rs)]
            wp, ws = po.planes_args(want)
            assert orc.orc_rgb_to_yuv(P(src), src.strides[0], w, h, order, int(ips == 4), ctypes.addressof(wp), ctypes.addressof(ws), fmt, out_alpha,
                                      1 if oclamp == 1 else 0) == 0
            if outpl == 513:
                planes = [planes[0], planes[2], planes[1]]                       # YVU420P: chroma pointers swapped (:13890)
            for i, (a, b) in enumerate(dims):
                assert (planes[i][:b, :a] == want[i][:b, :a]).all(), (inpl, outpl, oclamp, i)
            assert wh.geti(lay, "current_palette") == outpl and wh.geti(lay, "YUV_clamping") == oclamp
            assert wh.geti(lay, "width") == (w >> 1 if fmt in (2, 3) else w) and wh.geti(lay, "YUV_subspace") == 1
            if ips == 4 and not out_alpha:
                assert wh.geti(lay, "host_flags") == 0                           # alpha dropped -> premult flag cleared


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("inpl", [588, 589, 544, 545, 564, 565])
def test_yuv_layers_to_rgb(seam, orc, inpl):
    L, wh = seam
    rng = np.random.default_rng(60 + inpl)
    w, h = 64, 12
    fmt = {588: 0, 589: 0, 544: 1, 545: 1, 564: 2, 565: 3}[inpl]
    in_alpha = 1 if inpl in (589, 545) else 0
    for outpl in (RGB24, BGR24, RGBA32, BGRA32, ARGB32):
        for clamp in (0, 1):
            if fmt == 0:
                planes = [frame(rng, w, h, 4 if in_alpha else 3)]
            elif fmt == 1:
                planes = [frame(rng, w, h, 1) for _ in range(4 if in_alpha else 3)]
            else:
                planes = [frame(rng, w, h, 2)]
            lw = w >> 1 if fmt >= 2 else w
            lay = wh.new_layer(inpl, lw, h, planes, clamping=clamp, subspace=1)
            rc = L.lives_gpu_convert_layer_palette(lay, outpl, clamp)
            order = {RGB24: 0, RGBA32: 0, BGR24: 1, BGRA32: 1, ARGB32: 2}[outpl]
            out_alpha = 1 if outpl in (RGBA32, BGRA32, ARGB32) else 0
            if fmt == 1 and (order == 2 or (order == 1 and not out_alpha)):
                assert rc == 0 and wh.geti(lay, "current_palette") == inpl       # declined (reference-broken)
                continue
            assert rc == 1, (inpl, outpl)
            got, _, rs = wh.planes_of(lay)
            ops = 4 if out_alpha else 3
            want = np.zeros((h, rs[0]), np.uint8)
            sp, ss = po.planes_args(planes)
            assert orc.orc_yuv_to_rgb(ctypes.addressof(sp), ctypes.addressof(ss), w, h, fmt, in_alpha, P(want), rs[0], order, out_alpha, clamp) == 0
            assert (got[0][:, :w * ops] == want[:, :w * ops]).all(), (inpl, outpl, clamp)
            assert wh.geti(lay, "current_palette") == outpl and wh.geti(lay, "width") == w and wh.geti(lay, "YUV_clamping") is None


@needs_ref
@pytest.mark.gpu
def test_yuv_layers_repack(seam, orc):
    """the non-RGB half of convert_layer_palette_full (:12937-13750): YUV -> YUV pairs through lgpu_yuv_repack; pairs the
    library declines return FALSE with the layer untouched"""
    L, wh = seam
    rng = np.random.default_rng(88)
    w, h = 64, 12
    for (ip, op, _) in po.YUV_REPACK_PAIRS:
        for clamp in (0, 1):
            planes = po.yuv_planes(ip, w, h, rng=rng)
            lw = w >> 1 if ip in (564, 565) else w
            lay = wh.new_layer(ip, lw, h, [a.copy() for a in planes], clamping=clamp, subspace=1)
            assert L.lives_gpu_convert_layer_palette_full(lay, op, clamp, 0, 1, 0) == 1, (ip, op)
            got, _, rs = wh.planes_of(lay)
            dims = po.YUV_PLANE_DIMS[op](w, h)
            want = [np.zeros((b, r), np.uint8) for (a, b), r in zip(dims, rs)]
            sp, ss = po.planes_args(planes)
            wp, ws = po.planes_args(want)
            assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, clamp, 0) == 0
            for i, (a, b) in enumerate(dims):
                assert (got[i][:b, :a] == want[i][:b, :a]).all(), (ip, op, clamp, i)
            assert wh.geti(lay, "current_palette") == op and wh.geti(lay, "width") == (w >> 1 if op in (564, 565) else w)
            assert wh.geti(lay, "YUV_clamping") == clamp
    for (ip, op) in [(522, 512), (544, 522), (512, 544), (589, 544), (512, 545)]:          # reference functions that overrun / depend on stale bytes
        planes = po.yuv_planes(ip, w, h, rng=rng)
        lay = wh.new_layer(ip, w, h, planes, clamping=0, subspace=1)
        assert L.lives_gpu_convert_layer_palette_full(lay, op, 0, 0, 1, 0) == 0 and wh.geti(lay, "current_palette") == ip


@needs_ref
@pytest.mark.gpu
def test_rgb_layers_to_yuv_change_gamma_on_the_way(seam, orc):
    """src/colourspace.c:12311-12332: an RGB layer of known gamma that goes to YUV becomes SRGB (BT709 for a BT.709 target subspace, or the caller's target);
    UYVY / YUYV take the 16-bit LUT inline (rgb2uyvy_with_gamma), every other YUV palette gets gamma_convert_layer() first"""
    L, wh = seam
    rng = np.random.default_rng(91)
    w, h = 64, 10
    LIN, SRGB, BT709 = -1, 1, 2
    for (inpl, ips, order) in ((RGBA32, 4, 0), (BGR24, 3, 1)):
        for (tgt, osub, want_gamma) in ((0, 1, SRGB), (BT709, 1, BT709), (0, 2, BT709)):      # (tgt_gamma, osubspace (1 YCbCr, 2 BT.709), resulting gamma)
            src = frame(rng, w, h, ips)
            # UYVY: the LUT inline
            lay = wh.new_layer(inpl, w, h, [src.copy()], gamma=LIN, flags=0)
            assert L.lives_gpu_convert_layer_palette_full(lay, 564, 0, 0, osub, tgt) == 1
            got, _, rs = wh.planes_of(lay)
            lut16 = np.zeros(65536, np.uint16)
            assert orc.orc_gamma_lut16(1.0, LIN, want_gamma, 1.4, P(lut16)) == 1
            want = np.zeros((h, rs[0]), np.uint8)
            assert orc.orc_rgb_to_yuv_lut16(P(src), src.strides[0], w, h, order, int(ips == 4), P(want), rs[0], 2, 0, P(lut16)) == 0
            assert (got[0] == want).all(), (inpl, tgt, osub)
            assert wh.geti(lay, "gamma_type") == want_gamma and wh.geti(lay, "YUV_subspace") == (2 if want_gamma == BT709 else 1)
            # YUV888: gamma first (LUT8 over the RGB frame), then the palette
            lay = wh.new_layer(inpl, w, h, [src.copy()], gamma=LIN, flags=0)
            assert L.lives_gpu_convert_layer_palette_full(lay, 588, 0, 0, osub, tgt) == 1
            got, _, rs = wh.planes_of(lay)
            lut8 = np.zeros(256, np.uint8)
            assert orc.orc_gamma_lut8(1.0, LIN, want_gamma, 1.4, P(lut8)) == 1
            g = src.copy()
            orc.orc_gamma_apply(P(g), g.strides[0], w, h, ips, 0, P(lut8))
            want = [np.zeros((h, rs[0]), np.uint8)]
            wp, ws = po.planes_args(want)
            assert orc.orc_rgb_to_yuv(P(g), g.strides[0], w, h, order, int(ips == 4), ctypes.addressof(wp), ctypes.addressof(ws), 0, 0, 0) == 0
            assert (got[0] == want[0]).all(), (inpl, tgt, osub, "888")
            assert wh.geti(lay, "gamma_type") == want_gamma
    # the same on a pinned layer: both steps run on the resident planes, the bytes come home at the sync
    src = frame(rng, w, h, 4)
    for outpl, fmt in ((564, 2), (588, 0)):
        lay = wh.new_layer(RGBA32, w, h, [src.copy()], gamma=LIN, flags=0)
        assert L.lives_gpu_layer_pin(lay) == 0
        assert L.lives_gpu_convert_layer_palette_full(lay, outpl, 0, 0, 1, 0) == 1
        assert wh.geti(lay, "host_gpu_resident") == 1 and L.lives_gpu_layer_unpin(lay) == 0
        got, _, rs = wh.planes_of(lay)
        want = np.zeros((h, rs[0]), np.uint8)
        if fmt == 2:
            lut16 = np.zeros(65536, np.uint16)
            assert orc.orc_gamma_lut16(1.0, LIN, SRGB, 1.4, P(lut16)) == 1
            assert orc.orc_rgb_to_yuv_lut16(P(src), src.strides[0], w, h, 0, 1, P(want), rs[0], 2, 0, P(lut16)) == 0
        else:
            lut8 = np.zeros(256, np.uint8)
            assert orc.orc_gamma_lut8(1.0, LIN, SRGB, 1.4, P(lut8)) == 1
            g = src.copy()
            orc.orc_gamma_apply(P(g), g.strides[0], w, h, 4, 0, P(lut8))
            wl = [want]
            wp, ws = po.planes_args(wl)
            assert orc.orc_rgb_to_yuv(P(g), g.strides[0], w, h, 0, 1, ctypes.addressof(wp), ctypes.addressof(ws), 0, 0, 0) == 0
        assert (got[0] == want).all() and wh.geti(lay, "gamma_type") == SRGB, outpl
    # a layer that is already SRGB: nothing changes on the way (the plain entry points)
    src = frame(rng, w, h, 4)
    lay = wh.new_layer(RGBA32, w, h, [src.copy()], gamma=SRGB, flags=0)
    assert L.lives_gpu_convert_layer_palette_full(lay, 565, 0, 0, 1, 0) == 1
    got, _, rs = wh.planes_of(lay)
    want = [np.zeros((h, rs[0]), np.uint8)]
    wp, ws = po.planes_args(want)
    assert orc.orc_rgb_to_yuv(P(src), src.strides[0], w, h, 0, 1, ctypes.addressof(wp), ctypes.addressof(ws), 3, 0, 0) == 0
    assert (got[0] == want[0]).all() and wh.geti(lay, "gamma_type") == SRGB


@needs_ref
@pytest.mark.gpu
def test_planar_layers_to_packed_444(seam, orc):
    """K5d through the layer seam (:13624-13635, :13731-13742): YUV420P / YVU420P / YUV422P -> YUV888 / YUVA8888, reading the layer's YUV_sampling leaf; the new
    frame starts zeroed as create_empty_pixel_data() leaves it, so the bytes the reference never writes (last odd row's chroma, skipped alpha) are zero"""
    L, wh = seam
    W = wh.weed()
    rng = np.random.default_rng(90)
    w, h = 72, 12
    for ip in (512, 513, 522):
        for op in (588, 589):
            for sampling in (0, 1):
                dims_in = po.YUV_PLANE_DIMS[ip](w, h)
                planes = [rng.integers(0, 256, (dims_in[0][1], align(w)), dtype=np.uint8)] + [rng.integers(0, 256, (b, align(w) >> 1), dtype=np.uint8) for (a, b) in dims_in[1:]]
                lay = wh.new_layer(ip, w, h, [a.copy() for a in planes], clamping=0, subspace=1)
                W.weed_set_int_value(lay, b"YUV_sampling", sampling)
                assert L.lives_gpu_convert_layer_palette_full(lay, op, 0, sampling, 1, 0) == 1, (ip, op)
                got, _, rs = wh.planes_of(lay)
                want = [np.zeros((h, rs[0]), np.uint8)]
                src = planes if ip != 513 else [planes[0], planes[2], planes[1]]          # the dispatcher swaps the chroma pointers of a YVU layer first
                sp, ss = po.planes_args(src)
                wp, ws = po.planes_args(want)
                assert orc.orc_yuv_repack(512 if ip == 513 else ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, 0, sampling) == 0
                assert (got[0] == want[0]).all(), (ip, op, sampling)
                assert wh.geti(lay, "current_palette") == op and wh.geti(lay, "width") == w
    lay = wh.new_layer(512, w, 11, [rng.integers(0, 256, (11, 96), dtype=np.uint8), rng.integers(0, 256, (5, 48), dtype=np.uint8), rng.integers(0, 256, (5, 48), dtype=np.uint8)], clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette_full(lay, 588, 0, 0, 1, 0) == 0 and wh.geti(lay, "current_palette") == 512      # odd height: the reference overruns, declined


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("w", [64, 72])
def test_yuv411_layers_repack(seam, orc, w):
    """K5c through the layer seam (:13024-13029 ..., :13793-13846): YUV411 <-> the other YUV palettes.  The reference allocates the new layer with its ordinary
    aligned rowstrides and then walks it -- and the 4:1:1 side -- as compact streams; with w = 72 neither side's rows are compact, so the rows "drift"
    exactly as they do there (the oracle is handed the same strided, zeroed planes and writes the same stream)"""
    L, wh = seam
    rng = np.random.default_rng(89 + w)
    h = 12
    for (ip, op, _) in po.YUV411_REPACK_PAIRS:
        for clamp in (0, 1):
            dims_in = po.YUV_PLANE_DIMS[ip](w, h)
            planes = [rng.integers(0, 256, (b, align(a)), dtype=np.uint8) for (a, b) in dims_in]
            if ip in (512, 513, 522):
                planes[1] = planes[1][:, :align(dims_in[0][0]) >> 1].copy(); planes[2] = planes[2][:, :align(dims_in[0][0]) >> 1].copy()   # rs[1] = rs[0] >> 1
            lw = w >> 1 if ip in (564, 565) else w >> 2 if ip == 595 else w
            lay = wh.new_layer(ip, lw, h, [a.copy() for a in planes], clamping=clamp, subspace=1)
            pinned = clamp == 1                          # half of the cases on resident planes
            if pinned:
                assert L.lives_gpu_layer_pin(lay) == 0
            assert L.lives_gpu_convert_layer_palette_full(lay, op, clamp, 0, 1, 0) == 1, (ip, op)
            if pinned:
                assert wh.geti(lay, "host_gpu_resident") == 1 and L.lives_gpu_layer_unpin(lay) == 0
            got, _, rs = wh.planes_of(lay)
            dims = po.YUV_PLANE_DIMS[op](w, h)
            want = [np.zeros((b, r), np.uint8) for (a, b), r in zip(dims, rs)]
            sp, ss = po.planes_args(planes)
            wp, ws = po.planes_args(want)
            assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, clamp, 0) == 0
            if op == 513:
                got = [got[0], got[2], got[1]]              # swap_chroma_planes (:13890) after the is_yvu walk
            for i in range(len(dims)):
                assert (got[i] == want[i]).all(), (ip, op, clamp, i)
            assert wh.geti(lay, "current_palette") == op and wh.geti(lay, "YUV_clamping") == clamp
            assert wh.geti(lay, "width") == (w >> 1 if op in (564, 565) else w >> 2 if op == 595 else w) and wh.geti(lay, "height") == h
    lay = wh.new_layer(595, 5, h, [rng.integers(0, 256, (h, 32), dtype=np.uint8)], clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette_full(lay, 512, 0, 0, 1, 0) == 1        # 5 macropixels = 20 pixels: still a multiple of 4
    lay = wh.new_layer(595, 4, 7, [rng.integers(0, 256, (7, 32), dtype=np.uint8)], clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette_full(lay, 512, 0, 0, 1, 0) == 0 and wh.geti(lay, "current_palette") == 595      # odd height: declined


@needs_ref
@pytest.mark.gpu
def test_yuv_layer_clamping_switch(seam, orc):
    """convert_layer_palette_full(layer, same palette, other clamping, same subspace) = in-place range switch (:12241-12247)"""
    L, wh = seam
    rng = np.random.default_rng(77)
    w, h = 64, 8
    for pal in (512, 544, 588, 564):
        if pal in (588, 564):
            planes = [frame(rng, w, h, 3 if pal == 588 else 2)]
        else:
            ch = h >> 1 if pal == 512 else h
            cw = align(w) >> 1 if pal == 512 else align(w)
            planes = [frame(rng, w, h, 1)] + [rng.integers(0, 256, (ch, cw), dtype=np.uint8) for _ in range(2)]
        lw = w >> 1 if pal == 564 else w
        lay = wh.new_layer(pal, lw, h, planes, clamping=0, subspace=1)
        assert L.lives_gpu_convert_layer_palette_full(lay, pal, 1, 0, 1, 0) == 1
        got, _, rs = wh.planes_of(lay)
        want = [a.copy() for a in planes]
        wp, ws = po.planes_args(want)
        assert orc.orc_switch_yuv_clamping(ctypes.addressof(wp), ctypes.addressof(ws), pal, h, 1) == 0
        for i in range(len(planes)):
            assert (got[i] == want[i]).all(), (pal, i)
        assert wh.geti(lay, "YUV_clamping") == 1 and wh.geti(lay, "current_palette") == pal
        # a subspace change goes through RGB as in the reference (:12248-12262): the layer comes back in the same palette with the target's range and subspace
        lay2 = wh.new_layer(pal, lw, h, planes, clamping=0, subspace=1)
        twin = wh.new_layer(pal, lw, h, planes, clamping=0, subspace=1)
        ok = L.lives_gpu_convert_layer_palette_full(lay2, pal, 1, 0, 2, 0)
        assert L.lives_gpu_convert_layer_palette(twin, 1, 0) == 1
        assert ok == L.lives_gpu_convert_layer_palette_full(twin, pal, 1, 0, 2, 0)
        if ok:
            assert (wh.geti(lay2, "YUV_clamping"), wh.geti(lay2, "current_palette")) == (1, pal)
            assert all((a == b).all() for a, b in zip(wh.planes_of(lay2)[0], wh.planes_of(twin)[0]))


@needs_ref
@pytest.mark.gpu
def test_pinned_layer_stays_in_hbm(seam, orc):
    """pin -> convert -> gamma -> resize -> letterbox -> sync: same pixels as the unpinned run, PCIe crossed once per direction"""
    L, wh = seam

    def stats():
        a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        L.lives_gpu_transfer_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    rng = np.random.default_rng(11)
    w, h = 256, 144
    ys, cs = align(w), align(w) >> 1
    Y = rng.integers(16, 236, (h, ys), dtype=np.uint8)
    U = rng.integers(16, 241, (h // 2, cs), dtype=np.uint8)
    V = rng.integers(16, 241, (h // 2, cs), dtype=np.uint8)

    def chain(lay):
        assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1
        assert L.lives_gpu_gamma_convert_layer(1, lay) == 1
        assert L.lives_gpu_resize_layer(lay, 128, 72, 3, 0, 0) == 1
        assert L.lives_gpu_letterbox_layer(lay, 128, 96, 128, 72, 3, 0, 0) == 1

    plain = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    h0, d0 = stats()
    chain(plain)
    h1, d1 = stats()
    want, _, rs = wh.planes_of(plain)
    pinned = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    assert L.lives_gpu_layer_pin(pinned) == 0 and wh.geti(pinned, "host_gpu_resident") == 1
    h2, d2 = stats()
    chain(pinned)
    h3, d3 = stats()
    assert (h3, d3) == (h2, d2), "a pinned layer must not cross PCIe inside the chain"
    assert (wh.geti(pinned, "current_palette"), wh.geti(pinned, "width"), wh.geti(pinned, "height")) == (RGBA32, 128, 96)
    assert L.lives_gpu_layer_sync(pinned) == 0
    h4, d4 = stats()
    got, _, rs2 = wh.planes_of(pinned)
    assert rs2 == rs and (got[0] == want[0]).all()
    in_bytes, out_bytes = Y.nbytes + U.nbytes + V.nbytes, got[0].nbytes
    assert h2 - h1 == in_bytes and d4 - d3 == out_bytes                 # once in, once out
    assert (h1 - h0) > 3 * in_bytes and (d1 - d0) > 3 * out_bytes         # the unpinned chain pays for every step
    assert L.lives_gpu_layer_unpin(pinned) == 0 and wh.geti(pinned, "host_gpu_resident") is None



@needs_ref
@pytest.mark.gpu
def test_plugin_effects_on_pinned_layers_stay_in_hbm(seam, orc):
    """both seams together: convert a pinned layer, run weed filters of livesgpu_fx.so on its planes (the channels carry the layer's host plane
    pointers, as weed_apply_instance hands them over), sync once -- the effect results are right and nothing crossed PCIe in between"""
    import os
    L, wh = seam
    H = po.RefHost()
    OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lives_amd", "livesgpu_fx.so")

    def stats():
        a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        L.lives_gpu_transfer_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def view(layer):            # numpy views ON the layer's host planes (same addresses the channels would carry)
        _, ptrs, rs = wh.planes_of(layer)
        hh = wh.geti(layer, "height")
        return [np.frombuffer((ctypes.c_uint8 * (rs[0] * hh)).from_address(ptrs[0]), np.uint8).reshape(hh, rs[0])]

    rng = np.random.default_rng(21)
    w, h = 64, 16
    a_rgb, b_rgb = frame(rng, w, h, 3), frame(rng, w, h, 4, alpha_mix=True)
    la = wh.new_layer(RGB24, w, h, [a_rgb], gamma=1)
    lb = wh.new_layer(RGBA32, w, h, [b_rgb], gamma=1)
    assert L.lives_gpu_layer_pin(la) == 0 and L.lives_gpu_layer_pin(lb) == 0
    s0 = stats()
    assert L.lives_gpu_convert_layer_palette(la, RGBA32, 0) == 1               # on the device copy
    va, vb = view(la)[0], view(lb)[0]
    stale = va.copy()
    H.run(OURS, "chroma blend", RGBA32, w, h, [va, vb], va, [po.p_int(90)])     # in place on layer a, second input layer b
    H.run(OURS, "negate", RGBA32, w, h, [va], va, [])
    assert stats() == s0, "effects on pinned layers must not cross PCIe"
    assert (va == stale).all(), "the host plane stays stale until the layer is synced"
    assert L.lives_gpu_layer_sync(la) == 0
    got = view(la)[0].copy()
    # expected: the same steps on the oracle
    conv = np.zeros((h, va.shape[1]), np.uint8)
    orc.orc_swizzle(po.OPS.index("addpost"), 0, P(a_rgb), a_rgb.strides[0], P(conv), conv.strides[0], w, h, None)
    orc.orc_blend_chroma(P(conv), conv.strides[0], P(b_rgb), b_rgb.strides[0], P(conv), conv.strides[0], w, h, 4, 0, 90)
    luts = np.zeros((4, 256), np.uint8)
    assert orc.orc_fx_luts(0, RGBA32, 0., 0., 0., luts.ctypes.data) == 4
    orc.orc_byte_luts(P(conv), conv.strides[0], P(conv), conv.strides[0], w, h, 4, luts.ctypes.data)
    assert (got[:, :w * 4] == conv[:, :w * 4]).all()
    assert L.lives_gpu_layer_unpin(la) == 0 and L.lives_gpu_layer_unpin(lb) == 0


@needs_ref
@pytest.mark.gpu
def test_plugin_batch_hook_on_pinned_layers(seam):
    """the plan step of n tracks: n transition instances whose channels are planes of pinned layers go through livesgpu_fx_process_batch as one launch
    on the planes where they live -- nothing crosses PCIe until the layers are synced, and the results are the reference plugin's"""
    import os
    L, wh = seam
    H = po.RefHost()
    OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lives_amd", "livesgpu_fx.so")

    def stats():
        a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        L.lives_gpu_transfer_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def view(layer):
        _, ptrs, rs = wh.planes_of(layer)
        hh = wh.geti(layer, "height")
        return np.frombuffer((ctypes.c_uint8 * (rs[0] * hh)).from_address(ptrs[0]), np.uint8).reshape(hh, rs[0])

    rng = np.random.default_rng(23)
    w, h, n = 128, 36, 6
    fa = [frame(rng, w, h, 4) for _ in range(n)]
    fb = [frame(rng, w, h, 4) for _ in range(n)]
    amounts = [0.05 + 0.15 * i for i in range(n)]
    want = [np.zeros_like(x) for x in fa]
    H.run_batch(po.refplugin("multi_transitions"), "iris circle", RGBA32, w, h, fa, fb, want, amounts)
    la = [wh.new_layer(RGBA32, w, h, [x.copy()], gamma=1) for x in fa]
    lb = [wh.new_layer(RGBA32, w, h, [x.copy()], gamma=1) for x in fb]
    lo = [wh.new_layer(RGBA32, w, h, [np.zeros_like(x)], gamma=1) for x in fa]
    for lay in la + lb + lo:
        assert L.lives_gpu_layer_pin(lay) == 0
    va, vb, vo = [view(x) for x in la], [view(x) for x in lb], [view(x) for x in lo]
    s0 = stats()
    H.run_batch(OURS, "iris circle", RGBA32, w, h, va, vb, vo, amounts, hook="livesgpu_fx_process_batch")
    assert stats() == s0, "a batch on pinned layers must not cross PCIe"
    assert all((v == 0).all() for v in vo), "the host planes stay stale until the layers are synced"
    for lay in lo:
        assert L.lives_gpu_layer_sync(lay) == 0
    for i in range(n):
        assert (view(lo[i])[:, :w * 4] == want[i][:, :w * 4]).all(), i
    for lay in la + lb + lo:
        assert L.lives_gpu_layer_unpin(lay) == 0
