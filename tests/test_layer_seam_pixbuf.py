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
        assert L.lives_gpu_resize_layer(lay, dw, dh, interp, 0, 0) == 1, (pal, sw, sh, dw, dh)
        if pinned:
            assert wh.geti(lay, "host_gpu_resident") == 1 and L.lives_gpu_layer_unpin(lay) == 0
        planes, _, rs = wh.planes_of(lay)
        assert (wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "current_palette")) == (dw, dh, pal)
        assert rs[0] == align(dw * ch, 4), "the pixbuf's rowstride"
        want = want_scaled(orc, src, sw, sh, dw, dh, ch, interp)          # the WHOLE layer, odd sizes included (:15263-15264)
        assert (planes[0][:dh, :dw * ch] == want).all(), (pal, sw, sh, dw, dh, interp)
        if pal < 10:
            assert wh.geti(lay, "gamma_type") == 1, "pixbuf_to_layer tags RGB layers WEED_GAMMA_SRGB (:14378-14379)"


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


def test_palettes_outside_the_pixbuf_switch_fail_as_the_reference_body_does(seam, orc, pixbuf_backend):
    """src/colourspace.c:15303-15307: "Warning: resizing unknown palette", retval FALSE, the layer as it came (a pinned one synchronised and unpinned); the
    nothing-to-do tests (:14854-14868, :15265-15270) come before the switch and answer TRUE for every palette; the polyphase backend is the opt-in that scales them"""
    L, wh = seam
    rng = np.random.default_rng(0x9DB5)
    src = frame(rng, 128, 64, 4)
    for pinned in (0, 1):
        lay = wh.new_layer(ARGB32, 128, 64, [src], gamma=1)
        if pinned:
            assert L.lives_gpu_layer_pin(lay) == 0
        assert L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0) == 0
        assert wh.geti(lay, "host_gpu_resident") is None
        assert (wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "current_palette")) == (128, 64, ARGB32)
        assert (wh.planes_of(lay)[0][0] == src).all()
        assert L.lives_gpu_resize_layer(lay, 128, 64, 3, 0, 0) == 1                       # no resize needed
        assert L.lives_gpu_letterbox_layer(lay, 160, 80, 64, 32, 3, 0, 0) == 0           # the inner resize fails first (:15389)
        assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (128, 64)
    ys = align(128)
    Y, U, V = frame(rng, 128, 64, 1), frame(rng, 64, 32, 1), frame(rng, 64, 32, 1)
    lay = wh.new_layer(YUV420P, 128, 64, [Y, U, V], clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0) == 0 and (wh.geti(lay, "width"), wh.geti(lay, "height")) == (128, 64)
    uy = frame(rng, 32, 32, 4)                                                            # UYVY: 32 macropixels = 64 pixels
    lay = wh.new_layer(UYVY, 32, 32, [uy], clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0) == 1                            # width in PIXELS equals the layer's: nothing to do
    assert L.lives_gpu_resize_layer(lay, 32, 32, 3, 0, 0) == 0 and wh.geti(lay, "width") == 32
    # the opt-in: a host "built with USE_SWSCALE"
    assert L.lives_gpu_set_resize_backend(POLYPHASE) == 0
    try:
        lay = wh.new_layer(ARGB32, 128, 64, [src], gamma=1)
        assert L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0) == 1
        planes, _, rs = wh.planes_of(lay)
        want = np.zeros((32, rs[0]), np.uint8)
        assert orc.orc_resize(P(src), src.strides[0], 128, 64, P(want), rs[0], 64, 32, 4, 3) == 0
        assert (planes[0][:, :256] == want[:, :256]).all()
    finally:
        assert L.lives_gpu_set_resize_backend(PIXBUF) == 0


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
