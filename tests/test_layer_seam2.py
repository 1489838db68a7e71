"""The rest of the layer-op seam (src/colourspace.h:377-423) on genuine weed plants: resize_layer_full with its fused gamma pass,
unletterbox_layer, compact_rowstrides, weed_layer_clear_pixel_data, the _with_sampling / _variant wrappers, the premultiplied-alpha
bookkeeping with prefs->alpha_post, and the residency rules (a declined call unpins, forget + a new allocation at the same address)."""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, frame

needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (reference libweed) not built")
pytestmark = [needs_ref, pytest.mark.gpu]
P = po.P
RGB24, BGR24, RGBA32, BGRA32, ARGB32, YUV420P, YUV422P, YUV444P, UYVY, YUYV, YUV888, YUVA8888, YUV411 = 1, 2, 3, 4, 5, 512, 522, 544, 564, 565, 588, 589, 595


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


class Prefs(ctypes.Structure):
    _fields_ = [("apply_gamma", ctypes.c_int), ("alpha_post", ctypes.c_int), ("pb_quality", ctypes.c_int), ("screen_gamma", ctypes.c_double), ("device", ctypes.c_int)]


def test_resize_layer_full_fuses_the_target_gamma(seam, orc):
    """tgt_gamma != layer gamma on an RGB layer: the LUT8 of create_gamma_lut8(layer gamma -> target) runs over the scaled frame
    (src/colourspace.c:14718-14720, :15119-15127) and the layer is tagged with the target"""
    L, wh = seam
    rng = np.random.default_rng(41)
    sw, sh, dw, dh = 384, 216, 192, 108
    src = frame(rng, sw, sh, 4)
    lay = wh.new_layer(RGBA32, sw, sh, [src], gamma=-1)                     # WEED_GAMMA_LINEAR
    assert L.lives_gpu_resize_layer_full(lay, dw, dh, 3, RGBA32, 0, 0, 0, 1) == 1      # -> WEED_GAMMA_SRGB
    planes, _, rs = wh.planes_of(lay)
    assert (wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "gamma_type")) == (dw, dh, 1)
    want = np.zeros((dh, rs[0]), np.uint8)
    assert orc.orc_resize(P(src), src.strides[0], sw, sh, P(want), rs[0], dw, dh, 4, 3) == 0
    lut = np.zeros(256, np.uint8)
    assert orc.orc_gamma_lut8(1.0, -1, 1, 1.4, P(lut)) == 1
    orc.orc_gamma_apply(P(want), rs[0], dw, dh, 4, 0, P(lut))
    assert (planes[0][:, :dw * 4] == want[:, :dw * 4]).all()
    # unknown target: the layer's own gamma, no LUT; the six-argument form is the same call
    a = wh.new_layer(RGBA32, sw, sh, [src], gamma=1)
    b = wh.new_layer(RGBA32, sw, sh, [src], gamma=1)
    assert L.lives_gpu_resize_layer_full(a, dw, dh, 3, 0, 0, 0, 0, 0) == 1 and L.lives_gpu_resize_layer(b, dw, dh, 3, 0, 0) == 1
    assert (wh.planes_of(a)[0][0] == wh.planes_of(b)[0][0]).all() and wh.geti(a, "gamma_type") == 1


def test_unletterbox_layer(seam, orc):
    """borders cut off (src/colourspace.c:15570-15628); quirk U1: the reference's row copy moves xwidth BYTES (:15614), the rest of each
    row of the new opaque-black frame stays black"""
    L, wh = seam
    rng = np.random.default_rng(42)
    w, h, top, bottom, left, right = 96, 64, 6, 10, 8, 12
    for pal, ps, black in ((RGBA32, 4, [0, 0, 0, 255]), (RGB24, 3, [0, 0, 0]), (ARGB32, 4, [255, 0, 0, 0])):
        src = frame(rng, w, h, ps)
        lay = wh.new_layer(pal, w, h, [src])
        assert L.lives_gpu_unletterbox_layer(lay, 0, 0, top, bottom, left, right) == 1
        xw, xh = w - left - right, h - top - bottom
        planes, _, rs = wh.planes_of(lay)
        assert (wh.geti(lay, "width"), wh.geti(lay, "height"), rs) == (xw, xh, [align(xw * ps)])
        want = np.zeros((xh, rs[0]), np.uint8)
        want[:, :xw * ps] = np.tile(np.array(black, np.uint8), xw)[None, :]
        want[:, :xw] = src[top:top + xh, left * ps:left * ps + xw]
        assert (planes[0] == want).all(), pal
    # with an output size: the cut frame is resized (LIVES_INTERP_BEST)
    src = frame(rng, w, h, 4)
    lay = wh.new_layer(RGBA32, w, h, [src])
    assert L.lives_gpu_unletterbox_layer(lay, -1, -1, top, bottom, left, right) == 1
    assert (wh.geti(lay, "width"), wh.geti(lay, "height")) == (w, h)
    # nothing to cut: TRUE, same planes
    lay = wh.new_layer(RGBA32, w, h, [src])
    _, ptrs, _ = wh.planes_of(lay)
    assert L.lives_gpu_unletterbox_layer(lay, 0, 0, 0, 0, -3, 0) == 1 and wh.planes_of(lay)[1] == ptrs


def test_compact_rowstrides(seam):
    L, wh = seam
    rng = np.random.default_rng(43)
    src = frame(rng, 50, 20, 3, stride=192)
    lay = wh.new_layer(RGB24, 50, 20, [src])
    assert L.lives_gpu_compact_rowstrides(lay) == 1
    planes, _, rs = wh.planes_of(lay)
    assert rs == [150] and (planes[0] == src[:, :150]).all()
    Y, U, V = frame(rng, 50, 20, 1, stride=64), frame(rng, 25, 10, 1, stride=32), frame(rng, 25, 10, 1, stride=32)
    lay = wh.new_layer(YUV420P, 50, 20, [Y, U, V], clamping=0)
    assert L.lives_gpu_compact_rowstrides(lay) == 1
    planes, _, rs = wh.planes_of(lay)
    assert rs == [50, 25, 25] and (planes[0] == Y[:, :50]).all() and (planes[1] == U[:, :25]).all() and (planes[2] == V[:, :25]).all()
    _, ptrs, _ = wh.planes_of(lay)
    assert L.lives_gpu_compact_rowstrides(lay) == 1 and wh.planes_of(lay)[1] == ptrs      # already compact: untouched


def test_weed_layer_clear_pixel_data(seam):
    """blank_frame (src/colourspace.c:11212-11226) in place: the palette's black over the pixels, row padding untouched; host bytes of an
    ordinary layer, the resident planes of a pinned one"""
    L, wh = seam
    rng = np.random.default_rng(44)
    cases = [(RGB24, 3, [0, 0, 0], None), (RGBA32, 4, [0, 0, 0, 255], None), (ARGB32, 4, [255, 0, 0, 0], None), (YUV888, 3, [16, 128, 128], 0),
             (YUVA8888, 4, [0, 128, 128, 255], 1), (UYVY, 4, [128, 16, 128, 16], 0), (YUV411, 6, [128, 16, 16, 128, 16, 16], 0)]
    for pinned in (0, 1):
        for pal, ps, black, clamping in cases:
            w, h = 40, 12
            src = frame(rng, w, h, ps)
            lay = wh.new_layer(pal, w, h, [src], clamping=clamping)
            if pinned:
                assert L.lives_gpu_layer_pin(lay) == 0
            _, ptrs, _ = wh.planes_of(lay)
            assert L.lives_gpu_weed_layer_clear_pixel_data(lay) == 1
            if pinned:
                assert L.lives_gpu_layer_unpin(lay) == 0
            planes, ptrs2, _ = wh.planes_of(lay)
            want = src.copy()
            want[:, :w * ps] = np.tile(np.array(black, np.uint8), w)[None, :]
            assert ptrs2 == ptrs and (planes[0] == want).all(), (pal, pinned)
        # YUYV: blank_pixel never advances (:11150-11154): only the first macropixel of every row is painted
        src = frame(rng, 40, 12, 4)
        lay = wh.new_layer(YUYV, 40, 12, [src], clamping=0)
        if pinned:
            assert L.lives_gpu_layer_pin(lay) == 0
        assert L.lives_gpu_weed_layer_clear_pixel_data(lay) == 1
        if pinned:
            assert L.lives_gpu_layer_unpin(lay) == 0
        want = src.copy()
        want[:, :4] = [16, 128, 16, 128]
        assert (wh.planes_of(lay)[0][0] == want).all()
        # planar 4:2:0: Y 16 (clamped), chroma 128
        Y, U, V = frame(rng, 64, 16, 1), frame(rng, 32, 8, 1), frame(rng, 32, 8, 1)
        lay = wh.new_layer(YUV420P, 64, 16, [Y, U, V], clamping=0)
        if pinned:
            assert L.lives_gpu_layer_pin(lay) == 0
        assert L.lives_gpu_weed_layer_clear_pixel_data(lay) == 1
        if pinned:
            assert L.lives_gpu_layer_unpin(lay) == 0
        planes, _, _ = wh.planes_of(lay)
        assert (planes[0][:, :64] == 16).all() and (planes[1][:, :32] == 128).all() and (planes[2][:, :32] == 128).all()


def test_wrappers_with_sampling_and_variant(seam, orc):
    L, wh = seam
    rng = np.random.default_rng(45)
    src = frame(rng, 66, 34, 3)
    a, b = wh.new_layer(RGB24, 66, 34, [src]), wh.new_layer(RGB24, 66, 34, [src])
    assert L.lives_gpu_convert_layer_palette_with_sampling(a, YUV888, 0) == 1                 # (unclamped, default subspace, no gamma) :13935
    assert L.lives_gpu_convert_layer_palette_full(b, YUV888, 1, 0, 0, 0) == 1
    assert (wh.planes_of(a)[0][0] == wh.planes_of(b)[0][0]).all() and wh.geti(a, "YUV_clamping") == 1
    # gamma_convert_layer_variant (:14157-14168): the layer is tagged LINEAR first, then converted to the target; the file gamma enters the
    # table only for the target WEED_GAMMA_VARIANT (2048), which leaves the tag alone (:14137-14138)
    src = frame(rng, 66, 34, 4)
    for tgt, fg in ((1, 1.8), (2048, 1.8)):
        lay = wh.new_layer(RGBA32, 66, 34, [src], gamma=1)
        assert L.lives_gpu_gamma_convert_layer_variant(fg, tgt, lay) == 1
        lut = np.zeros(256, np.uint8)
        assert orc.orc_gamma_lut8(fg if tgt == 2048 else 1.0, -1, tgt, 1.4, P(lut)) == 1
        want = src.copy()
        orc.orc_gamma_apply(P(want), want.strides[0], 66, 34, 4, 0, P(lut))
        assert (wh.planes_of(lay)[0][0] == want).all() and wh.geti(lay, "gamma_type") == (1 if tgt == 1 else -1), tgt


def test_a_sampling_only_request_is_a_no_op_as_in_the_reference(seam):
    """src/colourspace.c:12265-12274: switch_yuv_sampling (:10876-10925) sits behind `isampling == osampling` and asks for `isampling != osampling`:
    it is never called.  A 4:2:0 layer asked to change its chroma siting only comes back TRUE, pixels, pointers and the YUV_sampling leaf untouched."""
    L, wh = seam
    W = wh.weed()
    rng = np.random.default_rng(47)
    Y, U, V = frame(rng, 64, 32, 1), frame(rng, 32, 16, 1), frame(rng, 32, 16, 1)
    for pinned in (0, 1):
        for isamp, osamp in ((1, 0), (0, 1)):
            lay = wh.new_layer(YUV420P, 64, 32, [Y, U, V], clamping=0, subspace=1)
            W.weed_set_int_value(lay, b"YUV_sampling", isamp)
            if pinned:
                assert L.lives_gpu_layer_pin(lay) == 0
            _, ptrs, _ = wh.planes_of(lay)
            assert L.lives_gpu_convert_layer_palette_full(lay, YUV420P, 0, osamp, 1, 0) == 1
            if pinned:
                assert wh.geti(lay, "host_gpu_resident") == 1 and L.lives_gpu_layer_unpin(lay) == 0      # not a decline: the layer stayed resident
            planes, ptrs2, _ = wh.planes_of(lay)
            assert ptrs2 == ptrs and wh.geti(lay, "YUV_sampling") == isamp
            assert (planes[0] == Y).all() and (planes[1] == U).all() and (planes[2] == V).all()


def test_premult_bookkeeping_applies_to_every_palette_pair(seam, orc):
    """src/colourspace.c:12290-12306 runs before the palette dispatch: with prefs->alpha_post a PREMULT RGBA layer that loses its alpha
    to a YUV palette is un-premultiplied first; without it, RGB24 -> YUVA8888 gains the PREMULT flag"""
    L, wh = seam
    rng = np.random.default_rng(46)
    L.lives_gpu_set_prefs.argtypes = [ctypes.POINTER(Prefs)]
    src = frame(rng, 64, 32, 4, alpha_mix=True)
    try:
        assert L.lives_gpu_set_prefs(ctypes.byref(Prefs(1, 1, 2, 1.4, 0))) == 0
        lay = wh.new_layer(RGBA32, 64, 32, [src], flags=1)                                  # LIVES_LAYER_ALPHA_PREMULT
        assert L.lives_gpu_convert_layer_palette(lay, YUV888, 0) == 1
        un = src.copy()
        orc.orc_alpha_premult(P(un), un.strides[0], 64, 32, 0, 1)
        want = np.zeros((32, align(64 * 3)), np.uint8)
        dp, ds = po.planes_args([want])
        assert orc.orc_rgb_to_yuv(P(un), un.strides[0], 64, 32, 0, 1, ctypes.addressof(dp), ctypes.addressof(ds), 0, 0, 0) == 0
        assert (wh.planes_of(lay)[0][0][:, :64 * 3] == want[:, :64 * 3]).all() and (wh.geti(lay, "host_flags") or 0) & 1 == 0
    finally:
        assert L.lives_gpu_set_prefs(ctypes.byref(Prefs(1, 0, 2, 1.4, 0))) == 0
    rgb = frame(rng, 64, 32, 3)
    lay = wh.new_layer(RGB24, 64, 32, [rgb])
    assert L.lives_gpu_convert_layer_palette(lay, YUVA8888, 0) == 1 and (wh.geti(lay, "host_flags") or 0) & 1 == 1


def test_a_declined_call_brings_a_pinned_layer_home(seam):
    """FALSE means the caller's CPU body runs next; the host bytes of a pinned layer are stale by contract, so the
    decline synchronises and unpins it first"""
    L, wh = seam
    rng = np.random.default_rng(47)
    src = frame(rng, 64, 32, 4)
    lay = wh.new_layer(RGBA32, 64, 32, [src], gamma=-1)
    assert L.lives_gpu_layer_pin(lay) == 0
    assert L.lives_gpu_gamma_convert_layer(1, lay) == 1                    # device copy changes, host bytes are stale now
    stale, _, _ = wh.planes_of(lay)
    assert (stale[0] == src).all()
    assert L.lives_gpu_convert_layer_palette(lay, YUV420P, 0) == 1          # served: still pinned
    assert wh.geti(lay, "host_gpu_resident") == 1
    lay2 = wh.new_layer(ARGB32, 64, 32, [src], gamma=-1)
    assert L.lives_gpu_layer_pin(lay2) == 0 and L.lives_gpu_gamma_convert_layer(1, lay2) == 1
    want = wh.new_layer(ARGB32, 64, 32, [src], gamma=-1)
    assert L.lives_gpu_gamma_convert_layer(1, want) == 1
    assert L.lives_gpu_convert_layer_palette(lay2, YUV420P, 0) == 0         # ARGB32 -> 4:2:0 is declined ...
    assert wh.geti(lay2, "host_gpu_resident") is None                      # ... and the layer came home first
    assert (wh.planes_of(lay2)[0][0] == wh.planes_of(want)[0][0]).all()
    assert L.lives_gpu_layer_unpin(lay) == 0


def test_forget_drops_the_device_copy(seam):
    """the table of device copies is keyed by host plane pointer: a host that frees a pinned layer's planes itself calls
    lives_gpu_layer_forget() first; a new layer that lands on the same address is then an ordinary layer"""
    L, wh = seam
    rng = np.random.default_rng(48)
    L.lives_gpu_resident_lookup.restype = ctypes.c_void_p
    L.lives_gpu_resident_lookup.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    a = frame(rng, 64, 32, 4)
    lay = wh.new_layer(RGBA32, 64, 32, [a], gamma=-1)
    _, ptrs, _ = wh.planes_of(lay)
    assert L.lives_gpu_layer_pin(lay) == 0 and L.lives_gpu_resident_lookup(ptrs[0], a.nbytes)
    assert L.lives_gpu_layer_forget(lay) == 0
    assert not L.lives_gpu_resident_lookup(ptrs[0], a.nbytes) and wh.geti(lay, "host_gpu_resident") is None
    # the host reuses the memory for another frame (same address, never pinned): the seam must read the HOST bytes
    b = frame(rng, 64, 32, 4)
    ctypes.memmove(ptrs[0], b.ctypes.data, b.nbytes)
    assert L.lives_gpu_convert_layer_palette(lay, BGRA32, 0) == 1
    got = wh.planes_of(lay)[0][0]
    assert (got[:, 0::4] == b[:, 2::4]).all() and (got[:, 2::4] == b[:, 0::4]).all()
    # an unpinned layer never consults the table, even if an entry with its address existed
    lay1 = wh.new_layer(RGBA32, 64, 32, [a], gamma=-1)
    assert L.lives_gpu_layer_pin(lay1) == 0
    _, p1, _ = wh.planes_of(lay1)
    assert L.lives_gpu_gamma_convert_layer(1, lay1) == 1                    # device copy differs from the host bytes
    W = wh.weed()
    lay_alias = W.plant_new(128)                                           # a second, unpinned layer over the same host memory
    for k, v in (("current_palette", RGBA32), ("width", 64), ("height", 32), ("gamma_type", -1)):
        W.weed_set_int_value(lay_alias, k.encode(), v)
    W.weed_set_int_array(lay_alias, b"rowstrides", 1, (ctypes.c_int * 1)(a.strides[0]))
    W.weed_set_voidptr_array(lay_alias, b"pixel_data", 1, (ctypes.c_void_p * 1)(p1[0]))
    ref = wh.new_layer(RGBA32, 64, 32, [a], gamma=-1)
    assert L.lives_gpu_gamma_convert_layer(1, lay_alias) == 1 and L.lives_gpu_gamma_convert_layer(1, ref) == 1
    assert L.lives_gpu_layer_forget(lay1) == 0
    assert (wh.planes_of(lay_alias)[0][0] == wh.planes_of(ref)[0][0]).all()


def _chain_bytes(L, wh, Y, U, V, w, h):
    """convert -> gamma -> resize -> letterbox on one thread, unpinned: the reference result for the threaded runs below"""
    lay = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1
    assert L.lives_gpu_gamma_convert_layer(1, lay) == 1
    assert L.lives_gpu_resize_layer(lay, w // 2, h // 2, 3, 0, 0) == 1
    assert L.lives_gpu_letterbox_layer(lay, w // 2, h // 2 + 24, w // 2, h // 2, 3, 0, 0) == 1
    return wh.planes_of(lay)[0][0].copy()


def test_host_threads_enqueue_on_their_own_streams(seam):
    """LiVES runs plan steps on pool threads (src/threading.c; the seam's threading rule): every host thread enqueues on a stream of its own,
    a resident plane carries an event behind its last use and a call on another thread's stream waits for it.  (1) eight threads, each its own pinned
    layers through the four-call chain; (2) ONE layer handed from thread to thread between the calls (pin | convert | gamma | resize | letterbox | unpin,
    six different threads, no synchronisation in between but the hand-over itself).  Same bytes as the single-threaded unpinned chain."""
    import threading
    L, wh = seam
    w, h = 320, 180
    rng = np.random.default_rng(4242)
    frames = []
    for _ in range(8):
        Y = rng.integers(16, 236, (h, w), dtype=np.uint8)
        U = rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8)
        V = rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8)
        frames.append((Y, U, V, _chain_bytes(L, wh, Y, U, V, w, h)))
    errors = []

    def own_layers(i):
        try:
            Y, U, V, want = frames[i]
            for _ in range(6):
                lay = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
                assert L.lives_gpu_layer_pin(lay) == 0
                assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1
                assert L.lives_gpu_gamma_convert_layer(1, lay) == 1
                assert L.lives_gpu_resize_layer(lay, w // 2, h // 2, 3, 0, 0) == 1
                assert L.lives_gpu_letterbox_layer(lay, w // 2, h // 2 + 24, w // 2, h // 2, 3, 0, 0) == 1
                assert L.lives_gpu_layer_unpin(lay) == 0
                assert (wh.planes_of(lay)[0][0] == want).all(), "thread %d" % i
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    ts = [threading.Thread(target=own_layers, args=(i,)) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors

    # (2) one layer, a different thread per step
    def run_on_new_thread(fn):
        box = []
        t = threading.Thread(target=lambda: box.append(fn()))
        t.start()
        t.join()
        return box[0]

    for Y, U, V, want in frames[:4]:
        lay = wh.new_layer(YUV420P, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
        assert run_on_new_thread(lambda: L.lives_gpu_layer_pin(lay)) == 0
        assert run_on_new_thread(lambda: L.lives_gpu_convert_layer_palette(lay, RGBA32, 0)) == 1
        assert run_on_new_thread(lambda: L.lives_gpu_gamma_convert_layer(1, lay)) == 1
        assert run_on_new_thread(lambda: L.lives_gpu_resize_layer(lay, w // 2, h // 2, 3, 0, 0)) == 1
        assert run_on_new_thread(lambda: L.lives_gpu_letterbox_layer(lay, w // 2, h // 2 + 24, w // 2, h // 2, 3, 0, 0)) == 1
        assert run_on_new_thread(lambda: L.lives_gpu_layer_unpin(lay)) == 0
        assert (wh.planes_of(lay)[0][0] == want).all()


def test_effects_and_seam_calls_from_several_threads(seam):
    """both seams from pool threads: livesgpu_fx.so enqueues its effects on the calling thread's stream as well (lives_gpu_resident_acquire / _release),
    so a plan step = seam call -> effect -> effect -> seam call stays on one stream, and the same layer handed to another thread between the steps is
    ordered by the hand-over events.  Result bytes against the same steps run on the main thread."""
    import os
    import threading
    L, wh = seam
    OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lives_amd", "livesgpu_fx.so")
    w, h = 192, 96
    rng = np.random.default_rng(77)

    def view(layer):
        _, ptrs, rs = wh.planes_of(layer)
        hh = wh.geti(layer, "height")
        return np.frombuffer((ctypes.c_uint8 * (rs[0] * hh)).from_address(ptrs[0]), np.uint8).reshape(hh, rs[0])

    def steps(H, la, lb):
        yield lambda: L.lives_gpu_layer_pin(la) == 0 and L.lives_gpu_layer_pin(lb) == 0
        yield lambda: L.lives_gpu_convert_layer_palette(la, RGBA32, 0) == 1
        yield lambda: H.run(OURS, "chroma blend", RGBA32, w, h, [view(la), view(lb)], view(la), [po.p_int(90)]) is not None     # in place on layer a
        yield lambda: H.run(OURS, "negate", RGBA32, w, h, [view(la)], view(la), []) is not None
        yield lambda: L.lives_gpu_gamma_convert_layer(2, la) == 1
        yield lambda: L.lives_gpu_layer_unpin(la) == 0 and L.lives_gpu_layer_unpin(lb) == 0

    def new_pair(i):
        r = np.random.default_rng(1000 + i)
        return wh.new_layer(RGB24, w, h, [frame(r, w, h, 3)], gamma=1), wh.new_layer(RGBA32, w, h, [frame(r, w, h, 4, alpha_mix=True)], gamma=1)

    def run_all(i, out, hop):
        try:
            H = po.RefHost()
            la, lb = new_pair(i)
            for st in steps(H, la, lb):
                if hop:                          # every step on a thread of its own
                    ok = []
                    t = threading.Thread(target=lambda: ok.append(st()))
                    t.start(); t.join()
                    assert ok and ok[0]
                else:
                    assert st()
            out[i] = view(la)[:, :w * 4].copy()
        except BaseException as e:              # noqa: BLE001
            out[i] = repr(e)

    want = {}
    for i in range(4):
        run_all(i, want, False)
        assert isinstance(want[i], np.ndarray), want[i]
    got = {}
    ts = [threading.Thread(target=run_all, args=(i, got, False)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(4):
        assert isinstance(got[i], np.ndarray), got[i]
        assert (got[i] == want[i]).all(), "thread %d" % i
    hopped = {}
    run_all(0, hopped, True)
    assert isinstance(hopped[0], np.ndarray), hopped[0]
    assert (hopped[0] == want[0]).all()
    del rng


def _planes_for(rng, pal, w, h):
    """random planes of a w x h (pixels) frame in palette pal, rows 32-byte aligned like the reference's allocator; returns (planes, width leaf)"""
    def plane(pw, ph, lo=0, hi=256):
        a = np.zeros((ph, align(pw)), np.uint8)
        a[:, :pw] = rng.integers(lo, hi, (ph, pw), dtype=np.uint8)
        return a
    if pal in (RGB24, BGR24, YUV888):
        return [plane(w * 3, h)], w
    if pal in (RGBA32, BGRA32, ARGB32, YUVA8888):
        return [plane(w * 4, h)], w
    if pal in (UYVY, YUYV):
        return [plane(w * 2, h, 16, 236)], w // 2
    if pal == YUV411:
        return [plane((w // 4) * 6, h, 16, 236)], w // 4
    if pal in (YUV420P, 513):
        return [plane(w, h, 16, 236), plane(w // 2, h // 2, 16, 241), plane(w // 2, h // 2, 16, 241)], w
    if pal == YUV422P:
        return [plane(w, h, 16, 236), plane(w // 2, h, 16, 241), plane(w // 2, h, 16, 241)], w
    if pal == YUV444P:
        return [plane(w, h, 16, 236), plane(w, h, 16, 241), plane(w, h, 16, 241)], w
    if pal == 545:                                                          # YUVA4444P
        return [plane(w, h, 16, 236), plane(w, h, 16, 241), plane(w, h, 16, 241), plane(w, h)], w
    raise AssertionError(pal)


def test_every_palette_pair_pinned_and_unpinned_agree(seam):
    """the whole convert_layer_palette matrix, differential: the same request on an ordinary layer (upload - kernel - download) and on a pinned one (resident
    planes, result brought home by unpin) must return the same value and leave the same leaves and the same bytes -- including the declined pairs, which
    must leave both layers as they were.  16 palettes x 16 targets x 2 clampings, two sizes."""
    L, wh = seam
    pals = [RGB24, BGR24, RGBA32, BGRA32, ARGB32, YUV420P, 513, YUV422P, YUV444P, 545, UYVY, YUYV, YUV888, YUVA8888, YUV411]
    rng = np.random.default_rng(20260928)
    served = declined = 0
    for (w, h) in ((64, 32), (136, 50)):
        for inpl in pals:
            for outpl in pals:
                for oclamp in (0, 1):
                    planes, wl = _planes_for(rng, inpl, w, h)
                    kw = dict(gamma=1) if inpl <= ARGB32 else dict(clamping=0, subspace=1)
                    a = wh.new_layer(inpl, wl, h, planes, **kw)
                    b = wh.new_layer(inpl, wl, h, planes, **kw)
                    ra = L.lives_gpu_convert_layer_palette(a, outpl, oclamp)
                    assert L.lives_gpu_layer_pin(b) == 0
                    rb = L.lives_gpu_convert_layer_palette(b, outpl, oclamp)
                    assert L.lives_gpu_layer_unpin(b) == 0
                    assert ra == rb, (inpl, outpl, oclamp, w, h, ra, rb)
                    pa, _, rsa = wh.planes_of(a)
                    pb, _, rsb = wh.planes_of(b)
                    la = [wh.geti(a, k) for k in ("current_palette", "width", "height", "YUV_clamping", "YUV_subspace", "YUV_sampling", "gamma_type")]
                    lb = [wh.geti(b, k) for k in ("current_palette", "width", "height", "YUV_clamping", "YUV_subspace", "YUV_sampling", "gamma_type")]
                    assert la == lb and rsa == rsb, (inpl, outpl, oclamp, la, lb)
                    for x, y in zip(pa, pb):
                        assert (x == y).all(), (inpl, outpl, oclamp, w, h)
                    if ra:
                        served += 1
                    else:
                        declined += 1
                        assert la[0] == inpl
    assert served > 300 and declined > 0


def test_every_layer_op_pinned_and_unpinned_agree(seam):
    """the other seam entry points, same differential: gamma, premultiply, resize (three interpolations, shrinking and enlarging), letterbox, unletterbox,
    compact_rowstrides, weed_layer_clear_pixel_data on every palette they take -- ordinary layer against pinned layer, return value, leaves and bytes"""
    L, wh = seam
    pals = [RGB24, BGR24, RGBA32, BGRA32, ARGB32, YUV420P, 513, YUV422P, YUV444P, 545, UYVY, YUYV, YUV888, YUVA8888, YUV411]
    rng = np.random.default_rng(5150)
    ops = [("gamma", lambda lay: L.lives_gpu_gamma_convert_layer(2, lay)),
           ("premult", lambda lay: (L.lives_gpu_alpha_premult(lay, 0), 1)[1]),
           ("unpremult", lambda lay: (L.lives_gpu_alpha_premult(lay, 1), 1)[1]),
           ("resize-bicubic-down", lambda lay: L.lives_gpu_resize_layer(lay, 72, 40, 3, 0, 0)),
           ("resize-half", lambda lay: L.lives_gpu_resize_layer(lay, 64, 32, 3, 0, 0)),
           ("resize-bilinear-up", lambda lay: L.lives_gpu_resize_layer(lay, 200, 96, 2, 0, 0)),
           ("resize-nearest", lambda lay: L.lives_gpu_resize_layer(lay, 96, 48, 0, 0, 0)),
           ("letterbox", lambda lay: L.lives_gpu_letterbox_layer(lay, 160, 100, 96, 48, 3, 0, 0)),
           ("unletterbox", lambda lay: L.lives_gpu_unletterbox_layer(lay, 0, 0, 4, 6, 8, 12)),
           ("compact", lambda lay: L.lives_gpu_compact_rowstrides(lay)),
           ("clear", lambda lay: L.lives_gpu_weed_layer_clear_pixel_data(lay))]
    keys = ("current_palette", "width", "height", "YUV_clamping", "YUV_subspace", "gamma_type", "host_flags")
    n = 0
    for pal in pals:
        for name, op in ops:
            w, h = 128, 64
            planes, wl = _planes_for(rng, pal, w, h)
            kw = dict(gamma=1) if pal <= ARGB32 else dict(clamping=0, subspace=1)
            a = wh.new_layer(pal, wl, h, planes, **kw)
            b = wh.new_layer(pal, wl, h, planes, **kw)
            ra = op(a)
            assert L.lives_gpu_layer_pin(b) == 0
            rb = op(b)
            assert L.lives_gpu_layer_unpin(b) == 0
            assert ra == rb, (pal, name, ra, rb)
            pa, _, rsa = wh.planes_of(a)
            pb, _, rsb = wh.planes_of(b)
            assert [wh.geti(a, k) for k in keys] == [wh.geti(b, k) for k in keys] and rsa == rsb, (pal, name)
            for x, y in zip(pa, pb):
                assert (x == y).all(), (pal, name)
            n += 1
    assert n == len(pals) * len(ops)


def test_four_plane_layers_resize_plane_by_plane(seam, orc):
    """YUVA4444P through resize_layer on an ORDINARY layer: four planes in, four planes out, every plane the oracle's resize of that plane (the differential test above
    found the alpha plane's upload landing in the scratch slot that held the new Y plane; scratch slots are now planes-in 0, 1, 2, 7 / planes-out 3 .. 6)"""
    L, wh = seam
    rng = np.random.default_rng(545)
    w, h, dw, dh = 128, 64, 72, 40
    planes, wl = _planes_for(rng, 545, w, h)
    lay = wh.new_layer(545, wl, h, planes, clamping=0, subspace=1)
    assert L.lives_gpu_resize_layer(lay, dw, dh, 3, 0, 0) == 1
    got, _, rs = wh.planes_of(lay)
    assert len(got) == 4 and (wh.geti(lay, "width"), wh.geti(lay, "height")) == (dw, dh)
    for p in range(4):
        want = np.zeros((dh, rs[p]), np.uint8)
        assert orc.orc_resize(P(planes[p]), planes[p].strides[0], w, h, P(want), rs[p], dw, dh, 1, 3) == 0
        assert (got[p][:, :dw] == want[:, :dw]).all(), p


def test_every_filter_class_on_pinned_planes(seam):
    """the plugin seam's residency path, differential over the filter classes of livesgpu_fx.so: channels that carry ordinary host memory (upload - effect - download)
    against channels that carry the planes of pinned layers (effect on the resident copies, on the calling thread's stream, bytes home at unpin).  Default parameters,
    RGBA32 / RGB24 as the class takes them; stateful and random classes run one frame from a fresh instance either way."""
    import os
    L, wh = seam
    H = po.RefHost()
    OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lives_amd", "livesgpu_fx.so")
    H.H.refhost_set_random_seed(12345)
    w, h = 96, 48
    rng = np.random.default_rng(31)
    ran = 0
    for f in H.filters(OURS):
        pal = RGBA32 if RGBA32 in f["palettes"] else RGB24 if RGB24 in f["palettes"] else None
        if pal is None or f["n_in"] < 1 or f["n_in"] > 2 or f["n_out"] != 1 or f["name"] in ("rand replace",):
            continue
        ps = 4 if pal == RGBA32 else 3
        srcs = [frame(rng, w, h, ps, alpha_mix=(ps == 4)) for _ in range(f["n_in"])]
        # ordinary memory, in place on input 0 (what LiVES does for these classes)
        plain = [s.copy() for s in srcs]
        try:
            H.H.refhost_set_random_seed(12345)
            H.run(OURS, f["name"], pal, w, h, plain, plain[0], [])
        except RuntimeError:
            continue                                     # a class that wants parameters / geometry this harness does not give it
        layers = [wh.new_layer(pal, w, h, [s], gamma=1) for s in srcs]
        for lay in layers:
            assert L.lives_gpu_layer_pin(lay) == 0
        views = []
        for lay in layers:
            _, ptrs, rs = wh.planes_of(lay)
            views.append(np.frombuffer((ctypes.c_uint8 * (rs[0] * h)).from_address(ptrs[0]), np.uint8).reshape(h, rs[0]))
        H.H.refhost_set_random_seed(12345)
        H.run(OURS, f["name"], pal, w, h, views, views[0], [])
        for lay in layers:
            assert L.lives_gpu_layer_unpin(lay) == 0
        assert (views[0][:, :w * ps] == plain[0][:, :w * ps]).all(), f["name"]
        ran += 1
    assert ran >= 20, ran


def test_compositor_class_on_pinned_planes(seam):
    """the compositor class (fx_plugin.c: p_compositor) through the residency bridge: in channels and the out channel on planes of PINNED layers -- whose host bytes are
    stale by contract and are scribbled over here after pinning -- give what ordinary memory gives; a mixed call (one in channel in host memory) too; and the next
    effect on the out layer sees the composite, not an older device copy"""
    import os
    L, wh = seam
    H = po.RefHost()
    OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lives_amd", "livesgpu_fx.so")
    rng = np.random.default_rng(77)
    ow, oh, pal, ps = 160, 90, RGBA32, 4
    sizes = [(64, 36), (200, 120), (96, 54)]
    offsx, offsy, scx, scy, alpha = [0.06, 0.44, 0.3], [0.09, 0.33, 0.4], [0.625, 0.5, 0.6], [0.62, 0.53, 0.6], [0.75, 1.0, 0.3]
    srcs = [frame(rng, w_, h_, ps, alpha_mix=True) for (w_, h_) in sizes]
    plain = np.zeros((oh, po.align(ow * ps, 32)), np.uint8)
    H.run_compositor(OURS, pal, [s_.copy() for s_ in srcs], sizes, [0, 0, 0], plain, ow, oh, offsx, offsy, scx, scy, alpha, [12, 200, 99], 0)
    for mixed in (0, 1):
        layers = [wh.new_layer(pal, w_, h_, [s_], gamma=1) for (w_, h_), s_ in zip(sizes, srcs)]
        out_layer = wh.new_layer(pal, ow, oh, [np.zeros_like(plain)], gamma=1)
        pinned = layers[:2] if mixed else layers
        for lay in pinned + [out_layer]:
            assert L.lives_gpu_layer_pin(lay) == 0
        views = []
        for lay, (w_, h_) in zip(layers, sizes):
            _, ptrs, rs = wh.planes_of(lay)
            views.append(np.frombuffer((ctypes.c_uint8 * (rs[0] * h_)).from_address(ptrs[0]), np.uint8).reshape(h_, rs[0]))
        for v in views[:len(pinned)]:
            v[:] = 0x5A                                     # the host bytes of a pinned plane are stale: whoever reads them composites garbage
        _, optrs, ors = wh.planes_of(out_layer)
        oview = np.frombuffer((ctypes.c_uint8 * (ors[0] * oh)).from_address(optrs[0]), np.uint8).reshape(oh, ors[0])
        H.run_compositor(OURS, pal, views, sizes, [0, 0, 0], oview, ow, oh, offsx, offsy, scx, scy, alpha, [12, 200, 99], 0)
        # the next effect on the out layer (negate, in place) must see the composite
        H.run(OURS, "negate", pal, ow, oh, [oview], oview, [])
        for lay in pinned + [out_layer]:
            assert L.lives_gpu_layer_unpin(lay) == 0
        want = plain.copy()
        H.run(OURS, "negate", pal, ow, oh, [want], want, [])
        assert (oview[:, :ow * ps] == want[:, :ow * ps]).all(), "mixed=%d" % mixed


def test_short_lived_threads_do_not_pile_up_device_objects(seam):
    """a host that makes its seam calls from threads that come and go: a finished thread's stream, staging chunks and device scratch go to spare lists and the
    next new thread starts from them -- 300 threads one after the other, each with an un-pinned and a pinned call, leave the device's free memory where it was"""
    import threading
    import torch
    L, wh = seam
    w, h = 640, 360
    rng = np.random.default_rng(9)
    src = frame(rng, w, h, 4)

    def one():
        a = wh.new_layer(RGBA32, w, h, [src], gamma=1)
        assert L.lives_gpu_convert_layer_palette(a, BGR24, 0) == 1
        b = wh.new_layer(RGBA32, w, h, [src], gamma=1)
        assert L.lives_gpu_layer_pin(b) == 0 and L.lives_gpu_convert_layer_palette(b, BGR24, 0) == 1 and L.lives_gpu_layer_unpin(b) == 0
        assert (wh.planes_of(a)[0][0] == wh.planes_of(b)[0][0]).all()

    def run(n):
        errs = []
        for _ in range(n):
            t = threading.Thread(target=lambda: (one(), None) if True else None)
            t.start(); t.join()
        return errs

    run(20)                                              # warm: pools, spare lists
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    run(300)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), "device memory shrank by %d MB over 300 short-lived threads" % ((free0 - free1) >> 20)


def test_freeing_a_contiguous_block_drops_every_plane_inside_it(seam):
    """lives_gpu_pinned_free(base) of a contiguous planar block (planes 1 and 2 are interior pointers of it): no resident entry inside the block may
    survive, or a later block at the same address would inherit a stale device copy"""
    L, wh = seam
    W = wh.weed()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.lives_gpu_pinned_calloc.restype = vp
    L.lives_gpu_pinned_calloc.argtypes = [ctypes.c_size_t]
    L.lives_gpu_pinned_free.argtypes = [vp]
    L.lives_gpu_resident_acquire.restype = vp
    L.lives_gpu_resident_acquire.argtypes = [vp, ctypes.c_size_t, ci]
    L.lives_gpu_resident_release.argtypes = [vp, ci]
    api = wh.WeedApi(W.fn["weed_leaf_get"], W.fn["weed_leaf_set"], W.fn["weed_leaf_num_elements"], W.fn["weed_leaf_delete"],
                     ctypes.cast(L.lives_gpu_pinned_calloc, vp), ctypes.cast(L.lives_gpu_pinned_free, vp))
    assert L.lives_gpu_bind_weed(ctypes.byref(api)) == 0
    try:
        rng = np.random.default_rng(48)
        lay = wh.new_layer(RGBA32, 64, 32, [frame(rng, 64, 32, 4)], gamma=1)
        assert L.lives_gpu_layer_pin(lay) == 0
        assert L.lives_gpu_convert_layer_palette(lay, YUV420P, 0) == 1              # new planes: one page-locked block, three resident planes
        _, ptrs, rs = wh.planes_of(lay)
        assert ptrs[1] == ptrs[0] + rs[0] * 32 and ptrs[2] == ptrs[1] + rs[1] * 16, "one contiguous block (may_contig allocation, colourspace.c:11601-11664)"
        for pl in range(3):
            d = L.lives_gpu_resident_acquire(ptrs[pl], 16, 0)
            assert d, "plane %d is resident" % pl
            L.lives_gpu_resident_release(ptrs[pl], 0)
        L.lives_gpu_pinned_free(ptrs[0])                                            # the host frees the block itself
        for pl in range(3):
            assert not L.lives_gpu_resident_acquire(ptrs[pl], 16, 0), "plane %d outlived its block" % pl
        leaf_delete = ctypes.CFUNCTYPE(ci, vp, ctypes.c_char_p)(W.fn["weed_leaf_delete"])
        leaf_delete(lay, b"pixel_data")
        leaf_delete(lay, b"host_gpu_resident")
    finally:
        wh.bind(L)
