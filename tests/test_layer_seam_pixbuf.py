"""resize_layer / resize_layer_full / letterbox_layer on the PIXBUF backend: the reference's gdk-pixbuf resize body (src/colourspace.c:15262-15322),
checked against the pinned restatement of gdk_pixbuf_scale_simple (oracle/orc_pixbuf.c, itself byte-equal to the runtime library and the committed
fixtures).  Also: every FALSE on a pinned layer leaves it synchronised and unpinned (the failure paths, not only the declines)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, frame

needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (reference libweed) not built")
pytestmark = [needs_ref, pytest.mark.gpu]
P = po.P
RGB24, BGR24, RGBA32, BGRA32, ARGB32, YUV420P, UYVY, YUV888, YUVA8888 = 1, 2, 3, 4, 5, 512, 564, 588, 589
POLYPHASE, PIXBUF = 0, 1


@pytest.fixture(scope="module")
def seam():
    from lives_amd import lib
    from tests import weedhost
    L = lib.load()
    weedhost.bind(L)
    return L, weedhost


@pytest.fixture()
def pixbuf_backend(seam):
    L, _ = seam
    assert L.lives_gpu_get_resize_backend() == PIXBUF, "the pinned body is the seam's default"
    assert L.lives_gpu_set_resize_backend(7) == -1 and L.lives_gpu_get_resize_backend() == PIXBUF
    yield


def want_scaled(orc, src, sw, sh, dw, dh, ch, interp):
    out = np.zeros((dh, dw * ch), np.uint8)
    assert orc.orc_pixbuf_scale(P(src), src.strides[0], sw, sh, P(out), dw * ch, dw, dh, ch, interp) == 0
    return out


@pytest.mark.parametrize("pinned", [0, 1])
def test_resize_layer_follows_the_pixbuf_body(seam, orc, pixbuf_backend, pinned):
    L, wh = seam
    rng = np.random.default_rng(0x9DB3)
    cases = [(RGBA32, 4, 384, 216, 192, 108, 3), (BGRA32, 4, 131, 77, 64, 36, 3), (RGB24, 3, 200, 120, 133, 80, 2), (BGR24, 3, 64, 36, 200, 100, 3),
             (RGBA32, 4, 96, 54, 320, 180, 2), (RGB24, 3, 320, 180, 96, 54, 0), (YUVA8888, 4, 128, 72, 64, 36, 3), (YUV888, 3, 128, 72, 86, 48, 2)]
    for (pal, ch, sw, sh, dw, dh, interp) in cases:
        src = frame(rng, sw, sh, ch, alpha_mix=(ch == 4))
        if ch == 4:
            a = src[:, 3:sw * 4:4]
            a[rng.random(a.shape) < 0.2] = 0
        lay = wh.new_layer(pal, sw, sh, [src], gamma=-1 if pal < 10 else None, clamping=1 if pal > 10 else None)
        if pinned:
            assert L.lives_gpu_layer_pin(lay) == 0
        # oclamp_hint: the layer's own clamping for the YUV frames (UNCLAMPED) -- with another one the frame is first switched to it (:14901-14908, next test)
        assert L.lives_gpu_resize_layer(lay, dw, dh, interp, 0, 1 if pal > 10 else 0) == 1, (pal, sw, sh, dw, dh)
        if pinned:
            assert wh.geti(lay, "host_gpu_resident") == 1 and L.lives_gpu_layer_unpin(lay) == 0
        planes, _, rs = wh.planes_of(lay)
        assert (wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "current_palette")) == (dw, dh, pal)
        assert rs[0] == align(dw * ch, 4), "the pixbuf's rowstride"
        want = want_scaled(orc, src, sw, sh, dw, dh, ch, interp)          # the WHOLE layer, odd sizes included (:15263-15264)
        assert (planes[0][:dh, :dw * ch] == want).all(), (pal, sw, sh, dw, dh, interp)
        if pal < 10:
            assert wh.geti(lay, "gamma_type") == 1, "pixbuf_to_layer tags RGB layers WEED_GAMMA_SRGB (:14378-14379)"


def test_a_clamping_hint_that_differs_is_applied_before_the_body(seam, orc, pixbuf_backend):
    """:14901-14908: `resolved != palette || oclamp_hint != iclamping` -> convert_layer_palette_full(layer, resolved, oclamp_hint, ...) -- an UNCLAMPED YUV888 frame
    with the hint CLAMPED is switched to clamped, and the body switches it back to unclamped before it scales (:15277-15284): both table passes show in the pixels"""
    L, wh = seam
    rng = np.random.default_rng(0x9DB9)
    src = frame(rng, 128, 72, 3)
    lay = wh.new_layer(YUV888, 128, 72, [src], clamping=1, subspace=1)
    assert L.lives_gpu_resize_layer(lay, 64, 36, 3, 0, 0) == 1 and wh.geti(lay, "YUV_clamping") == 1
    twin = wh.new_layer(YUV888, 128, 72, [src], clamping=1, subspace=1)
    # the body's call is convert_layer_palette(layer, YUV888, UNCLAMPED) = subspace argument WEED_YUV_SUBSPACE_YUV (:13931): for a frame tagged YCbCr that is a
    # "subspace change", which the reference takes through RGB24 (:12248-12262) -- kept
    assert L.lives_gpu_convert_layer_palette_full(twin, YUV888, 0, 0, 1, 0) == 1 and L.lives_gpu_convert_layer_palette(twin, YUV888, 1) == 1
    there_and_back = np.ascontiguousarray(wh.planes_of(twin)[0][0])
    assert not (there_and_back == src).all()
    assert (wh.planes_of(lay)[0][0][:36, :192] == want_scaled(orc, there_and_back, 128, 72, 64, 36, 3, 3)).all()


def test_size_rules_of_the_common_prologue(seam, orc, pixbuf_backend):
    """:14854-14868: even source size only for the nothing-to-do test, targets below 4 become 4, odd target heights lose a row"""
    L, wh = seam
    rng = np.random.default_rng(0x9DB4)
    src = frame(rng, 65, 33, 4)
    lay = wh.new_layer(RGBA32, 65, 33, [src], gamma=1)
    assert L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0) == 1                      # (65 >> 1) << 1 == 64: "no resize needed"
    assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (65, 33)
    assert L.lives_gpu_resize_layer(lay, 2, 21, 3, 0, 0) == 1
    assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (4, 20)
    want = want_scaled(orc, src, 65, 33, 4, 20, 4, 3)
    assert (wh.planes_of(lay)[0][0][:20, :16] == want).all()


def yuv420_planes(rng, w, h):
    ys, cs = align(w), align(w) >> 1
    return (rng.integers(16, 236, (h, ys), dtype=np.uint8), rng.integers(16, 241, (h // 2, cs), dtype=np.uint8), rng.integers(16, 241, (h // 2, cs), dtype=np.uint8))


@pytest.mark.parametrize("pinned", [0, 1])
@pytest.mark.parametrize("geom", [(128, 64, 64, 32), (130, 74, 200, 112)])
def test_yuv420p_with_the_hint_res_substep_passes_is_converted_then_scaled(seam, orc, pixbuf_backend, pinned, geom):
    """src/nodemodel.c:1187 calls resize_layer(layer, w, h, interp, opalette, oclamping); for a decoder's YUV420P frame and an RGBA32 hint get_resizable resolves
    RGBA32 (src/colourspace.c:14625-14637), convert_layer_palette_full runs (:14907) and the gdk-pixbuf body scales the result.  Expected side: orc_yuv420p_to_rgb ->
    orc_pixbuf_scale, nothing of the library (quirk A4's undefined pixels of the first stage are taken from a separately converted twin, as in test_dropin.py)"""
    L, wh = seam
    w, h, dw, dh = geom
    rng = np.random.default_rng(0x9DC0 + w)
    Y, U, V = yuv420_planes(rng, w, h)
    lay = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=1, clamping=0, subspace=1)
    if pinned:
        assert L.lives_gpu_layer_pin(lay) == 0
    assert L.lives_gpu_resize_layer(lay, dw, dh, 3, RGBA32, 0) == 1
    if pinned:
        assert wh.geti(lay, "host_gpu_resident") == 1 and L.lives_gpu_layer_unpin(lay) == 0
    got, _, rs = wh.planes_of(lay)
    assert (wh.geti(lay, "current_palette"), wh.geti(lay, "width"), wh.geti(lay, "height")) == (RGBA32, dw, dh) and rs[0] == dw * 4
    assert wh.geti(lay, "YUV_clamping") is None and wh.geti(lay, "gamma_type") == 1
    import ctypes
    strides = (ctypes.c_int * 3)(Y.strides[0], U.strides[0], V.strides[0])
    rgba = np.zeros((h, align(w * 4)), np.uint8)
    orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, U.size, V.size, P(rgba), rgba.strides[0], w, h, 4, 0, 0, 0, 2, None, 0)
    twin = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=1, clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette(twin, RGBA32, 0) == 1
    first = wh.planes_of(twin)[0][0][:, :w * 4].reshape(h, w, 4)
    a4 = np.zeros((h, w), bool)
    a4[0, 1::2] = True
    a4[h - 1, 1::2] = True
    view = rgba[:, :w * 4].reshape(h, w, 4)
    assert (first[~a4] == view[~a4]).all()
    view[a4] = first[a4]
    want = want_scaled(orc, rgba, w, h, dw, dh, 4, 3)
    assert (got[0][:dh, :dw * 4] == want).all()


def scaled_twin(L, wh, orc, pal, w, h, planes, kw, to_pal, clamp, dw, dh, ch, interp):
    """the expected side of a resolved resize: the library's own (separately oracle-tested) conversion of a twin layer, then the ORACLE's gdk-pixbuf scale"""
    twin = wh.new_layer(pal, w, h, planes, **kw)
    assert L.lives_gpu_convert_layer_palette_full(twin, to_pal, clamp, 0, 1, 0) == 1
    tp, _, trs = wh.planes_of(twin)
    tw = wh.geti(twin, "width")
    return want_scaled(orc, np.ascontiguousarray(tp[0]), tw, h, dw, dh, ch, interp), twin


def test_palettes_outside_the_switch_take_the_route_get_resizable_resolves(seam, orc, pixbuf_backend):
    """ARGB32 with an RGBA32 hint -> RGBA32 (get_inter_pal, :14516-14521); without a hint RGB24 when reducing and YUV888 when enlarging (:14539-14557: "if upscaling,
    better to convert yuv / rgb now"); UYVY with an RGB24 hint -> RGB24 in both directions.  Each: TRUE, the resolved palette, the pixels = conversion then scale;
    pinned == unpinned"""
    L, wh = seam
    rng = np.random.default_rng(0x9DC5)
    src = frame(rng, 128, 64, 4, alpha_mix=True)
    uy = frame(rng, 64, 64, 4)                                                            # UYVY: 64 macropixels = 128 pixels
    uy[:] = np.clip(uy, 16, 235)
    for (pal, planes, kw, hint, dw, dh, res, ch, interp) in (
            (ARGB32, [src], dict(gamma=1), RGBA32, 64, 32, RGBA32, 4, 3), (ARGB32, [src], dict(gamma=1), 0, 64, 32, RGB24, 3, 3),
            (ARGB32, [src], dict(gamma=1), RGBA32, 200, 100, RGBA32, 4, 2),
            (UYVY, [uy], dict(clamping=0, subspace=1), RGB24, 64, 32, RGB24, 3, 3), (UYVY, [uy], dict(clamping=0, subspace=1), RGB24, 200, 96, RGB24, 3, 2)):
        w = 128
        want, _ = scaled_twin(L, wh, orc, pal, planes[0].shape[1] // 4 if pal == UYVY else w, 64, planes, kw, res, 0, dw, dh, ch, interp)
        outs = []
        for pinned in (0, 1):
            lay = wh.new_layer(pal, planes[0].shape[1] // 4 if pal == UYVY else w, 64, planes, **kw)
            if pinned:
                assert L.lives_gpu_layer_pin(lay) == 0
            assert L.lives_gpu_resize_layer(lay, dw, dh, interp, hint, 0) == 1, (pal, hint, dw, dh)
            if pinned:
                assert wh.geti(lay, "host_gpu_resident") == 1 and L.lives_gpu_layer_unpin(lay) == 0
            assert (wh.geti(lay, "current_palette"), wh.geti(lay, "width"), wh.geti(lay, "height")) == (res, dw, dh), (pal, hint)
            outs.append(wh.planes_of(lay)[0][0][:dh, :dw * ch])
            assert (outs[-1] == want).all(), (pal, hint, dw, dh, pinned)
    # ARGB32 enlarged without a hint: through YUV888 -- converted with the clamping the caller hinted (CLAMPED), switched to UNCLAMPED inside the body (:15277-15284)
    lay = wh.new_layer(ARGB32, 128, 64, [src], gamma=1)
    assert L.lives_gpu_resize_layer(lay, 200, 100, 3, 0, 0) == 1
    assert (wh.geti(lay, "current_palette"), wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "YUV_clamping")) == (YUV888, 200, 100, 1)
    twin = wh.new_layer(ARGB32, 128, 64, [src], gamma=1)
    assert L.lives_gpu_convert_layer_palette_full(twin, YUV888, 0, 0, 1, 1) == 1 and L.lives_gpu_convert_layer_palette(twin, YUV888, 1) == 1
    want = want_scaled(orc, np.ascontiguousarray(wh.planes_of(twin)[0][0]), 128, 64, 200, 100, 3, 3)
    assert (wh.planes_of(lay)[0][0][:100, :600] == want).all()


def test_where_the_reference_has_no_route_the_call_fails_and_the_layer_is_as_it_came(seam, orc, pixbuf_backend):
    """YUV420P with a YUV420P hint or, reducing, without one: get_inter_pal answers YUV444P, which neither scales nor masquerades -- the reference's LIVES_FATAL
    (:14641-14650); here FALSE, the layer untouched, a pinned one synchronised and unpinned.  The nothing-to-do tests (:14854-14868) still come first."""
    L, wh = seam
    rng = np.random.default_rng(0x9DC6)
    Y, U, V = yuv420_planes(rng, 128, 64)
    for pinned in (0, 1):
        for hint in (YUV420P, 0):
            lay = wh.new_layer(YUV420P, 128, 64, [Y, U, V], clamping=0, subspace=1)
            if pinned:
                assert L.lives_gpu_layer_pin(lay) == 0
            assert L.lives_gpu_resize_layer(lay, 128, 64, 3, hint, 0) == 1                # no resize needed
            assert L.lives_gpu_resize_layer(lay, 64, 32, 3, hint, 0) == 0
            assert wh.geti(lay, "host_gpu_resident") is None
            assert (wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "current_palette")) == (128, 64, YUV420P)
            got = wh.planes_of(lay)[0]
            assert (got[0] == Y).all() and (got[1] == U).all() and (got[2] == V).all()
            assert L.lives_gpu_letterbox_layer(lay, 160, 80, 64, 32, 3, hint, 0) == 0      # the inner resize fails first (:15389)
    # enlarging without a hint has a route: RGB24 first (:14559-14563)
    lay = wh.new_layer(YUV420P, 128, 64, [Y, U, V], clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer(lay, 256, 128, 2, 0, 0) == 1 and wh.geti(lay, "current_palette") == RGB24
    # the opt-in: a host "built with USE_SWSCALE" scales the planes as they are
    assert L.lives_gpu_set_resize_backend(POLYPHASE) == 0
    try:
        lay = wh.new_layer(YUV420P, 128, 64, [Y, U, V], clamping=0, subspace=1)
        assert L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0) == 1 and wh.geti(lay, "current_palette") == YUV420P
    finally:
        assert L.lives_gpu_set_resize_backend(PIXBUF) == 0


def test_quirk_r2_an_rgb_layer_with_an_unclamped_hint_fails_the_post_check(seam, orc, pixbuf_backend):
    """:14916-14923 compares weed_layer_get_yuv_clamping(layer) -- 0 = CLAMPED for a layer without the leaf, i.e. every RGB layer -- with oclamp_hint: an RGB frame
    with the hint UNCLAMPED fails although nothing about it is YUV.  unletterbox_layer passes exactly that hint (:15628): its cut happens, its final resize does not."""
    L, wh = seam
    rng = np.random.default_rng(0x9DC7)
    src = frame(rng, 96, 64, 4)
    lay = wh.new_layer(RGBA32, 96, 64, [src], gamma=1)
    assert L.lives_gpu_resize_layer(lay, 48, 32, 3, RGBA32, 1) == 0 and (wh.geti(lay, "width"), wh.geti(lay, "height")) == (96, 64)
    assert (wh.planes_of(lay)[0][0] == src).all()
    assert L.lives_gpu_resize_layer(lay, 48, 32, 3, RGBA32, 0) == 1
    lay = wh.new_layer(RGBA32, 96, 64, [src], gamma=1)
    assert L.lives_gpu_unletterbox_layer(lay, -1, -1, 6, 10, 8, 12) == 0
    assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (76, 48), "the borders are gone, the frame was not scaled back"
    lay = wh.new_layer(RGBA32, 96, 64, [src], gamma=1)
    assert L.lives_gpu_unletterbox_layer(lay, 0, 0, 6, 10, 8, 12) == 1


def test_the_target_gamma_goes_into_the_pre_conversion(seam, orc, pixbuf_backend):
    """:14890-14907: tgt_gamma is handed to convert_layer_palette_full, which fuses it (LUT16 for 4:2:0 -> RGB); then the body tags the RGB frame SRGB whatever it was"""
    L, wh = seam
    rng = np.random.default_rng(0x9DC8)
    Y, U, V = yuv420_planes(rng, 128, 64)
    lay = wh.new_layer(YUV420P, 128, 64, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer_full(lay, 64, 32, 3, RGBA32, 0, 0, 1, 1) == 1
    twin = wh.new_layer(YUV420P, 128, 64, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette_full(twin, RGBA32, 0, 0, 1, 1) == 1 and wh.geti(twin, "gamma_type") == 1
    want = want_scaled(orc, np.ascontiguousarray(wh.planes_of(twin)[0][0]), 128, 64, 64, 32, 4, 3)
    assert (wh.planes_of(lay)[0][0][:32, :256] == want).all() and wh.geti(lay, "gamma_type") == 1
    plain = wh.new_layer(YUV420P, 128, 64, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer(plain, 64, 32, 3, RGBA32, 0) == 1
    assert not (wh.planes_of(plain)[0][0][:32, :256] == want).all(), "without a target the LINEAR frame is converted as it is"


def test_reductions_past_the_one_step_range_are_declined(seam, pixbuf_backend):
    L, wh = seam
    rng = np.random.default_rng(0x9DB6)
    src = frame(rng, 400, 400, 3)
    lay = wh.new_layer(RGB24, 400, 400, [src], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0
    assert L.lives_gpu_resize_layer(lay, 8, 8, 3, 0, 0) == 0                         # 53 x 53 taps: gdk-pixbuf's two-step scaler, not covered
    assert wh.geti(lay, "host_gpu_resident") is None
    assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (400, 400) and (wh.planes_of(lay)[0][0] == src).all()


def test_letterbox_layer_scales_with_the_pixbuf_body(seam, orc, pixbuf_backend):
    L, wh = seam
    rng = np.random.default_rng(0x9DB7)
    sw, sh, w, h, nw, nh = 384, 216, 171, 96, 192, 120
    src = frame(rng, sw, sh, 4, alpha_mix=True)
    lay = wh.new_layer(RGBA32, sw, sh, [src], gamma=1)
    assert L.lives_gpu_letterbox_layer(lay, nw, nh, w, h, 3, 0, 0) == 1
    planes, _, rs = wh.planes_of(lay)
    assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (nw, nh)
    inner = want_scaled(orc, src, sw, sh, w, h, 4, 3)
    ox, oy = (nw - w + 1) >> 1, (nh - h + 1) >> 1
    want = np.zeros((nh, nw, 4), np.uint8)
    want[..., 3] = 255
    want[oy:oy + h, ox:ox + w] = inner.reshape(h, w, 4)
    assert (planes[0][:nh, :nw * 4].reshape(nh, nw, 4) == want).all()


def test_a_false_on_a_pinned_layer_always_brings_it_home(seam):
    """not only the declines: letterbox_layer on a pinned packed-YUV layer that already has the inner size has nothing it can blit (no pixel size on
    this path) and returns FALSE -- the layer must come back synchronised and unpinned, or the host's CPU body would letterbox stale bytes; the same
    for a call that fails after it started (allocation failure injected)"""
    L, wh = seam
    rng = np.random.default_rng(0x9DB8)
    src = frame(rng, 32, 32, 4)                                                       # UYVY: 64 pixels = 32 macropixels of 4 bytes
    lay = wh.new_layer(UYVY, 32, 32, [src], clamping=0, subspace=1)
    assert L.lives_gpu_layer_pin(lay) == 0
    assert L.lives_gpu_convert_layer_palette_full(lay, UYVY, 1, 0, 1, 0) == 1        # clamped -> unclamped on the device; the host planes are stale now
    ref = wh.new_layer(UYVY, 32, 32, [src], clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette_full(ref, UYVY, 1, 0, 1, 0) == 1
    assert wh.geti(lay, "host_gpu_resident") == 1 and (wh.planes_of(lay)[0][0] == src).all()
    assert L.lives_gpu_letterbox_layer(lay, 80, 40, 64, 32, 3, 0, 0) == 0
    assert wh.geti(lay, "host_gpu_resident") is None
    assert (wh.planes_of(lay)[0][0] == wh.planes_of(ref)[0][0]).all() and not (wh.planes_of(ref)[0][0] == src).all()
    # a failure after the call has started: the second allocation of a resize fails
    src = frame(rng, 128, 64, 4)
    lay = wh.new_layer(RGBA32, 128, 64, [src], gamma=-1)
    assert L.lives_gpu_layer_pin(lay) == 0 and L.lives_gpu_gamma_convert_layer(1, lay) == 1
    want = wh.new_layer(RGBA32, 128, 64, [src], gamma=-1)
    assert L.lives_gpu_gamma_convert_layer(1, want) == 1
    assert L.lgpu_debug_fail_alloc(1) == 0
    rc = L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0)
    L.lgpu_debug_fail_alloc(0)
    if rc == 0:        # (a pooled block can satisfy the request without a new allocation: then the call simply succeeds)
        assert wh.geti(lay, "host_gpu_resident") is None
        assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (128, 64)
        assert (wh.planes_of(lay)[0][0] == wh.planes_of(want)[0][0]).all()
    else:
        assert L.lives_gpu_layer_unpin(lay) == 0
