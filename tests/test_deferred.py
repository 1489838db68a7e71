"""Deferred execution on pinned layers (include/lives_gpu_layer.h): the reference's own calls of one track's plan step -- convert_layer_palette ->
resize_layer[_full] -> [letterbox_layer] -> "chroma blend" (livesgpu_fx.so, in place) -> gamma_convert_layer, src/nodemodel.c:1065-1253 +
src/effects-weed.c:1850-2425 -- are recorded on the plane and run as ONE launch of the fused chain kernel, for all tracks of a tick together when the host calls
lives_gpu_layers_flush().  Everything here is compared three ways: deferred == eager (lives_gpu_set_deferred(0)) == the oracle's composition of the single stages
(orc_swizzle -> orc_pixbuf_scale -> orc_letterbox -> orc_blend_chroma -> orc_gamma_apply), leaves included."""
import ctypes
import os
import threading

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, frame

needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (reference libweed) not built")
pytestmark = [needs_ref, pytest.mark.gpu]
P = po.P
RGB24, RGBA32, BGRA32 = 1, 3, 4
OURS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lives_amd", "livesgpu_fx.so")
LEAVES = ("current_palette", "width", "height", "gamma_type", "host_flags", "YUV_clamping")


@pytest.fixture(scope="module")
def seam():
    from lives_amd import lib
    from tests import weedhost
    L = lib.load()
    weedhost.bind(L)
    L.lives_gpu_layers_flush.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    L.lives_gpu_layer_pin_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.lives_gpu_deferred_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    L.lives_gpu_deferred_stats.restype = None
    assert L.lives_gpu_get_resize_backend() == 1
    return L, weedhost, po.RefHost()


@pytest.fixture()
def deferred(seam):
    L = seam[0]
    prev = L.lives_gpu_set_deferred(1)
    assert prev == 1, "deferred execution is the default"
    yield
    L.lives_gpu_set_deferred(1)


def dstats(L):
    a = (ctypes.c_ulonglong * 4)()
    L.lives_gpu_deferred_stats(a)
    return list(a)


def view(wh, layer):
    """a numpy view ON the layer's host plane (the address a channel of weed_apply_instance would carry)"""
    _, ptrs, rs = wh.planes_of(layer)
    hh = wh.geti(layer, "height")
    return np.frombuffer((ctypes.c_uint8 * (rs[0] * hh)).from_address(ptrs[0]), np.uint8).reshape(hh, rs[0])


def plan_step(L, wh, H, lay, l2, dw, dh, canvas, bf, gamma, swap_to=RGBA32, interp=3):
    """the calls of one track, by what the reference's substeps call (src/nodemodel.c:1093, :1187, :1253, :1138)"""
    assert L.lives_gpu_convert_layer_palette(lay, swap_to, 0) == 1
    assert L.lives_gpu_resize_layer(lay, dw, dh, interp, swap_to, 0) == 1
    if canvas:
        assert L.lives_gpu_letterbox_layer(lay, canvas[0], canvas[1], dw, dh, interp, swap_to, 0) == 1
    if l2 is not None:
        v, v2 = view(wh, lay), view(wh, l2)
        H.run(OURS, "chroma blend", swap_to, wh.geti(lay, "width"), wh.geti(lay, "height"), [v, v2], v, [po.p_int(bf)])
    if gamma is not None:
        assert L.lives_gpu_gamma_convert_layer(gamma, lay) == 1


def oracle_step(orc, src, sw, sh, l2, dw, dh, canvas, bf, gamma_lut, swap, interp=3):
    conv = np.zeros((sh, sw * 4), np.uint8)
    if swap:
        orc.orc_swizzle(po.OPS.index("swap3postalpha"), 0, P(src), src.strides[0], P(conv), sw * 4, sw, sh, None)
    else:
        conv[:] = src[:, :sw * 4]
    out = np.zeros((dh, dw * 4), np.uint8)
    assert orc.orc_pixbuf_scale(P(conv), sw * 4, sw, sh, P(out), dw * 4, dw, dh, 4, interp) == 0
    w, h = dw, dh
    if canvas:
        w, h = canvas
        big = np.zeros((h, w * 4), np.uint8)
        orc.orc_letterbox(P(out), dw * 4, dw, dh, P(big), w * 4, w, h, 4, P(np.array([0, 0, 0, 255], np.uint8)))
        out = big
    if l2 is not None:
        orc.orc_blend_chroma(P(out), w * 4, P(l2), l2.strides[0], P(out), w * 4, w, h, 4, 0, bf)
    if gamma_lut is not None:
        orc.orc_gamma_apply(P(out), w * 4, w, h, 4, 0, P(gamma_lut))
    return out


def srgb_to(orc, tgt):
    lut = np.zeros(256, np.uint8)
    assert orc.orc_gamma_lut8(1.0, 1, tgt, 1.4, P(lut)) == 1
    return lut


SHAPES = [
    # sw, sh, dw, dh, canvas, with layer 2, gamma target (the layer is SRGB after the pixbuf body), note
    (256, 144, 128, 72, None, True, 2, "C5 shape: exact 2:1, blend, gamma -> the fused kernel"),
    (256, 144, 128, 72, (128, 96), True, 2, "C3 shape: 2:1 into a letterbox canvas"),
    (262, 150, 128, 72, None, True, -1, "not 2:1: the chain's staged form"),
    (131, 77, 200, 112, (210, 120), True, None, "enlargement, odd canvas offset, no gamma"),
    (256, 144, 128, 72, None, False, 2, "no blend: the chain without a layer 2"),
    (300, 170, 200, 112, (220, 120), False, 2, "no blend, not 2:1, letterbox"),
    (128, 72, 128, 72, None, True, 2, "no resize: swap, blend, gamma"),
    (128, 72, 128, 72, (160, 100), True, 2, "letterbox only, blend, gamma"),
    (128, 72, 128, 72, (160, 100), False, None, "letterbox only"),
]


@pytest.mark.parametrize("shape", SHAPES, ids=[s[-1] for s in SHAPES])
def test_the_stage_by_stage_fallback_gives_the_same_bytes(seam, orc, deferred, tune, shape):
    """the walk that runs a recorded program stage by stage (what is left for launches the one-launch forms decline: more than 64 tracks of a shape, planes of 2 GiB),
    forced with SEAM_STAGED: the oracle's bytes for every shape"""
    L, wh, H = seam
    sw, sh, dw, dh, canvas, with_l2, gamma, _ = shape
    rng = np.random.default_rng(0xDEFD + sw + dw)
    ow, oh = canvas if canvas else (dw, dh)
    src, l2a = frame(rng, sw, sh, 4, alpha_mix=True), frame(rng, ow, oh, 4, alpha_mix=True)
    tune("SEAM_STAGED", 1)
    lay = wh.new_layer(BGRA32, sw, sh, [src], gamma=1)
    l2 = wh.new_layer(RGBA32, ow, oh, [l2a], gamma=1) if with_l2 else None
    assert L.lives_gpu_layer_pin(lay) == 0 and (l2 is None or L.lives_gpu_layer_pin(l2) == 0)
    s0 = dstats(L)
    plan_step(L, wh, H, lay, l2, dw, dh, canvas, 90, gamma)
    assert L.lives_gpu_layer_sync(lay) == 0
    s1 = dstats(L)
    assert (s1[1] - s0[1], s1[3] - s0[3]) == (0, 1), "no chain launch, one program walked"
    want = oracle_step(orc, src, sw, sh, l2a if with_l2 else None, dw, dh, canvas, 90, srgb_to(orc, gamma) if gamma is not None else None, True)
    assert (view(wh, lay)[:, :ow * 4] == want).all()
    assert L.lives_gpu_layer_unpin(lay) == 0 and (l2 is None or L.lives_gpu_layer_unpin(l2) == 0)


@pytest.mark.parametrize("shape", SHAPES, ids=[s[-1] for s in SHAPES])
@pytest.mark.parametrize("src_pal", [BGRA32, RGBA32])
def test_deferred_equals_eager_equals_oracle(seam, orc, deferred, shape, src_pal):
    L, wh, H = seam
    sw, sh, dw, dh, canvas, with_l2, gamma, _ = shape
    rng = np.random.default_rng(0xDEF0 + sw + dw)
    src = frame(rng, sw, sh, 4, alpha_mix=True)
    ow, oh = canvas if canvas else (dw, dh)
    l2a = frame(rng, ow, oh, 4, alpha_mix=True)
    to = RGBA32 if src_pal == BGRA32 else BGRA32
    results = []
    for mode in (1, 0):
        L.lives_gpu_set_deferred(mode)
        lay = wh.new_layer(src_pal, sw, sh, [src], gamma=1)
        l2 = wh.new_layer(to, ow, oh, [l2a], gamma=1) if with_l2 else None
        assert L.lives_gpu_layer_pin(lay) == 0 and (l2 is None or L.lives_gpu_layer_pin(l2) == 0)
        s0 = dstats(L)
        plan_step(L, wh, H, lay, l2, dw, dh, canvas, 90, gamma, swap_to=to)
        s1 = dstats(L)
        leaves = [wh.geti(lay, k) for k in LEAVES] + [wh.planes_of(lay)[2]]
        if mode:
            assert s1[0] - s0[0] == 1 + (1 if (dw, dh) != (sw, sh) else 0) + (1 if canvas else 0) + (1 if with_l2 else 0) + (1 if gamma is not None else 0), "every call of the step was recorded (a resize to the same size is no call at all)"
            assert s1[1:] == s0[1:], "nothing has run yet"
        else:
            assert s1 == s0
        assert L.lives_gpu_layer_sync(lay) == 0
        s2 = dstats(L)
        if mode:
            assert (s2[1] - s1[1], s2[2] - s1[2], s2[3] - s1[3]) == (1, 1, 0), "the program ran as one call of the chain (with or without a layer 2)"
        results.append((leaves, view(wh, lay).copy()))
        assert L.lives_gpu_layer_unpin(lay) == 0 and (l2 is None or L.lives_gpu_layer_unpin(l2) == 0)
    L.lives_gpu_set_deferred(1)
    assert results[0][0] == results[1][0], "the leaves change exactly as in the eager calls"
    assert (results[0][1][:, :ow * 4] == results[1][1][:, :ow * 4]).all(), "deferred == eager"
    want = oracle_step(orc, src, sw, sh, l2a if with_l2 else None, dw, dh, canvas, 90, srgb_to(orc, gamma) if gamma is not None else None, True)
    assert (results[0][1][:, :ow * 4] == want).all(), "deferred == oracle"


def test_the_tracks_of_a_tick_share_one_launch(seam, orc, deferred):
    """lives_gpu_layers_flush(layers, n): n programs of one shape = ONE launch of the fused kernel with n tracks; another shape in the same call gets its own; the
    plan steps themselves run on one host thread per track, as src/nodemodel.c:2027-2101 runs them"""
    L, wh, H = seam
    rng = np.random.default_rng(0xDEF5)
    sw, sh, dw, dh, n = 256, 144, 128, 72, 6
    srcs = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(n + 1)]
    l2s = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(n + 1)]
    lays = [wh.new_layer(BGRA32, sw, sh, [s], gamma=1) for s in srcs]
    l2l = [wh.new_layer(RGBA32, dw, dh, [s], gamma=1) for s in l2s]
    for a in lays + l2l:
        assert L.lives_gpu_layer_pin(a) == 0
    errs = []

    def track(i):
        try:
            plan_step(L, wh, H, lays[i], l2l[i], dw, dh, None, 100 + 9 * i, 2 if i < n else None)     # every track its own blend amount (still one shape); the last has no gamma step: another shape
        except Exception as e:         # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=track, args=(i,)) for i in range(n + 1)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    s0 = dstats(L)
    arr = (ctypes.c_void_p * (n + 1))(*lays)
    assert L.lives_gpu_layers_flush(arr, n + 1) == 0
    s1 = dstats(L)
    assert (s1[1] - s0[1], s1[2] - s0[2], s1[3] - s0[3]) == (2, n + 1, 0), "two shapes, two launches, every track in one of them"
    assert L.lives_gpu_layers_flush(arr, n + 1) == 0 and dstats(L) == s1, "nothing is pending any more"
    lut = srgb_to(orc, 2)
    for i in range(n + 1):
        assert L.lives_gpu_layer_sync(lays[i]) == 0
        want = oracle_step(orc, srcs[i], sw, sh, l2s[i], dw, dh, None, 100 + 9 * i, lut if i < n else None, True)
        assert (view(wh, lays[i])[:, :dw * 4] == want).all(), i
    for a in lays + l2l:
        assert L.lives_gpu_layer_unpin(a) == 0


def test_whoever_needs_the_pixels_gets_them(seam, orc, deferred):
    """a call that is not part of the chain (here: RGBA32 -> RGB24, and alpha_premult) on a plane with a pending program runs the program first; a second scale
    cannot join a program that has one: the first runs, the second is recorded on its result"""
    L, wh, H = seam
    rng = np.random.default_rng(0xDEF6)
    src = frame(rng, 256, 144, 4, alpha_mix=True)
    outs = []
    for mode in (1, 0):
        L.lives_gpu_set_deferred(mode)
        lay = wh.new_layer(BGRA32, 256, 144, [src], gamma=1)
        assert L.lives_gpu_layer_pin(lay) == 0
        assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1 and L.lives_gpu_resize_layer(lay, 128, 72, 3, RGBA32, 0) == 1
        assert L.lives_gpu_resize_layer(lay, 96, 54, 2, RGBA32, 0) == 1
        L.lives_gpu_alpha_premult(lay, 1)
        assert L.lives_gpu_convert_layer_palette(lay, RGB24, 0) == 1
        assert L.lives_gpu_layer_unpin(lay) == 0
        outs.append(([wh.geti(lay, k) for k in LEAVES], wh.planes_of(lay)[0][0]))
    L.lives_gpu_set_deferred(1)
    assert outs[0][0] == outs[1][0] and (outs[0][1] == outs[1][1]).all()
    a = oracle_step(orc, src, 256, 144, None, 128, 72, None, 0, None, True)
    b = np.zeros((54, 96 * 4), np.uint8)
    assert orc.orc_pixbuf_scale(P(a), 128 * 4, 128, 72, P(b), 96 * 4, 96, 54, 4, 2) == 0
    orc.orc_alpha_premult(P(b), 96 * 4, 96, 54, 0, 0)
    c = np.zeros((54, align(96 * 3)), np.uint8)
    orc.orc_swizzle(po.OPS.index("delpost"), 0, P(b), 96 * 4, P(c), c.strides[0], 96, 54, None)
    assert (outs[0][1][:, :96 * 3] == c[:, :96 * 3]).all()


def test_layer_2_is_read_as_it_was_when_the_blend_was_called(seam, orc, deferred):
    """a recorded blend reads its second layer LATER -- so anything that would change that layer first (here a gamma pass on it, and its release) makes the program
    run before; a layer 2 that is itself pending runs when it is taken as an input"""
    L, wh, H = seam
    rng = np.random.default_rng(0xDEF7)
    src, l2a = frame(rng, 256, 144, 4, alpha_mix=True), frame(rng, 256, 144, 4, alpha_mix=True)
    lay = wh.new_layer(BGRA32, 256, 144, [src], gamma=1)
    l2 = wh.new_layer(BGRA32, 256, 144, [l2a], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0 and L.lives_gpu_layer_pin(l2) == 0
    assert L.lives_gpu_convert_layer_palette(l2, RGBA32, 0) == 1 and L.lives_gpu_resize_layer(l2, 128, 72, 3, RGBA32, 0) == 1        # layer 2: pending itself
    plan_step(L, wh, H, lay, l2, 128, 72, None, 140, None)
    assert L.lives_gpu_gamma_convert_layer(2, l2) == 1                     # changes layer 2 AFTER the blend was called
    assert L.lives_gpu_layer_unpin(l2) == 0
    assert L.lives_gpu_layer_sync(lay) == 0
    l2_then = oracle_step(orc, l2a, 256, 144, None, 128, 72, None, 0, None, True)
    want = oracle_step(orc, src, 256, 144, l2_then, 128, 72, None, 140, None, True)
    assert (view(wh, lay)[:, :512] == want).all()
    l2_now = l2_then.copy()
    orc.orc_gamma_apply(P(l2_now), 512, 128, 72, 4, 0, P(srgb_to(orc, 2)))
    assert (wh.planes_of(l2)[0][0][:, :512] == l2_now).all()
    assert L.lives_gpu_layer_unpin(lay) == 0


def test_a_refusal_on_a_pending_plane_brings_the_layer_home_as_it_is(seam, orc, deferred):
    """400 x 400 -> 8 x 8 is past the scaler's one-step range: FALSE as in the eager call, and the layer comes home with what the recorded calls made of it"""
    L, wh, H = seam
    rng = np.random.default_rng(0xDEF8)
    src = frame(rng, 400, 400, 4, alpha_mix=True)
    lay = wh.new_layer(BGRA32, 400, 400, [src], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0
    assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1
    assert L.lives_gpu_resize_layer(lay, 8, 8, 3, RGBA32, 0) == 0
    assert wh.geti(lay, "host_gpu_resident") is None and (wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "current_palette")) == (400, 400, RGBA32)
    want = np.zeros((400, 1600), np.uint8)
    orc.orc_swizzle(po.OPS.index("swap3postalpha"), 0, P(src), src.strides[0], P(want), 1600, 400, 400, None)
    assert (wh.planes_of(lay)[0][0][:, :1600] == want).all()


def test_forgetting_a_layer_with_a_pending_program_runs_nothing(seam, deferred):
    L, wh, H = seam
    rng = np.random.default_rng(0xDEF9)
    lay = wh.new_layer(BGRA32, 256, 144, [frame(rng, 256, 144, 4)], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0
    assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1 and L.lives_gpu_resize_layer(lay, 128, 72, 3, RGBA32, 0) == 1
    s0 = dstats(L)
    assert L.lives_gpu_layer_forget(lay) == 0 and wh.geti(lay, "host_gpu_resident") is None
    assert dstats(L)[1:] == s0[1:]


def test_frames_that_already_are_in_hbm(seam, orc, deferred, gpu):
    """lives_gpu_layer_pin_device: the layer's plane IS a caller-owned device buffer (a decoder surface): no upload; the chain reads it where it lies, the buffer is
    neither written nor freed, and the same buffer serves the next tick's layer"""
    import torch
    L, wh, H = seam
    from tests.util import dev, host
    rng = np.random.default_rng(0xDEFA)
    sw, sh, dw, dh = 256, 144, 128, 72
    src, l2a = frame(rng, sw, sh, 4, alpha_mix=True), frame(rng, dw, dh, 4, alpha_mix=True)
    d_src, d_l2 = dev(src), dev(l2a)
    torch.cuda.synchronize()
    h2d0 = ctypes.c_ulonglong()
    L.lives_gpu_transfer_stats(ctypes.byref(h2d0), None)
    want = oracle_step(orc, src, sw, sh, l2a, dw, dh, None, 128, srgb_to(orc, 2), True)
    for tick in range(3):
        lay = wh.new_layer(BGRA32, sw, sh, [np.zeros_like(src)], gamma=1)          # the host plane holds nothing: the pixels are on the device
        l2 = wh.new_layer(RGBA32, dw, dh, [np.zeros_like(l2a)], gamma=1)
        for (la, t) in ((lay, d_src), (l2, d_l2)):
            pl = (ctypes.c_void_p * 1)(t.data_ptr())
            assert L.lives_gpu_layer_pin_device(la, pl, 1, None, 1) == 0 and wh.geti(la, "host_gpu_resident") == 1
        plan_step(L, wh, H, lay, l2, dw, dh, None, 128, 2)
        arr = (ctypes.c_void_p * 1)(lay)
        assert L.lives_gpu_layers_flush(arr, 1) == 0
        assert L.lives_gpu_layer_sync(lay) == 0
        assert (view(wh, lay)[:, :dw * 4] == want).all(), tick
        assert L.lives_gpu_layer_forget(lay) == 0 and L.lives_gpu_layer_forget(l2) == 0
    h2d1 = ctypes.c_ulonglong()
    L.lives_gpu_transfer_stats(ctypes.byref(h2d1), None)
    assert h2d1.value == h2d0.value, "nothing was uploaded"
    assert (host(d_src) == src).all() and (host(d_l2) == l2a).all(), "the caller's buffers are as they were"


def test_the_plugins_batch_hook_records_too(seam, orc, deferred):
    """livesgpu_fx_process_batch(n chroma blend instances) on planes that are pending programs: every blend joins its program (nothing is launched), and the flush
    that follows is one launch of the chain for all of them"""
    L, wh, H = seam
    rng = np.random.default_rng(0xDEFB)
    sw, sh, dw, dh, n = 256, 144, 128, 72, 5
    srcs = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(n)]
    l2s = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(n)]
    lays = [wh.new_layer(BGRA32, sw, sh, [s], gamma=1) for s in srcs]
    l2l = [wh.new_layer(RGBA32, dw, dh, [s], gamma=1) for s in l2s]
    for a in lays + l2l:
        assert L.lives_gpu_layer_pin(a) == 0
    for lay in lays:
        assert L.lives_gpu_convert_layer_palette(lay, RGBA32, 0) == 1 and L.lives_gpu_resize_layer(lay, dw, dh, 3, RGBA32, 0) == 1
    vs, v2 = [view(wh, a) for a in lays], [view(wh, a) for a in l2l]
    amounts = [40 + 30 * i for i in range(n)]
    s0 = dstats(L)
    H.run_batch(OURS, "chroma blend", RGBA32, dw, dh, vs, v2, vs, amounts, hook="livesgpu_fx_process_batch", int_param=True)
    s1 = dstats(L)
    assert s1[0] - s0[0] == n and s1[1:] == s0[1:], "five blends recorded, nothing launched"
    for lay in lays:
        assert L.lives_gpu_gamma_convert_layer(2, lay) == 1
    arr = (ctypes.c_void_p * n)(*lays)
    assert L.lives_gpu_layers_flush(arr, n) == 0
    s2 = dstats(L)
    assert (s2[1] - s1[1], s2[2] - s1[2], s2[3] - s1[3]) == (1, n, 0), "five blend amounts, ONE launch (lgpu_chain_amounts): the amount is per track"
    lut = srgb_to(orc, 2)
    for i in range(n):
        assert L.lives_gpu_layer_sync(lays[i]) == 0
        want = oracle_step(orc, srcs[i], sw, sh, l2s[i], dw, dh, None, amounts[i], lut, True)
        assert (view(wh, lays[i])[:, :dw * 4] == want).all(), i
    for a in lays + l2l:
        assert L.lives_gpu_layer_unpin(a) == 0


def test_a_deep_copy_of_a_pinned_layer_gets_the_pixels_on_the_device(seam, orc, deferred):
    """lives_gpu_layer_copy(dlayer, slayer) behind the host's weed_layer_copy(NULL, slayer): the copy becomes a pinned layer with device copies of the planes -- of what
    the source's recorded calls make of it (they run first) -- and lives on when the source changes or goes"""
    L, wh, H = seam
    rng = np.random.default_rng(0xDEFC)
    sw, sh, dw, dh = 256, 144, 128, 72
    src = frame(rng, sw, sh, 4, alpha_mix=True)
    lay = wh.new_layer(BGRA32, sw, sh, [src], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0
    plan_step(L, wh, H, lay, None, dw, dh, None, 0, None)                       # convert + resize recorded, pending
    rs = wh.planes_of(lay)[2][0]
    cp = wh.new_layer(RGBA32, dw, dh, [np.zeros((dh, rs), np.uint8)], gamma=wh.geti(lay, "gamma_type"))      # what weed_layer_copy makes: same leaves, stale bytes
    assert wh.planes_of(cp)[2][0] == rs
    assert L.lives_gpu_layer_copy(cp, lay) == 0
    assert L.lives_gpu_gamma_convert_layer(2, lay) == 1                        # the source moves on
    assert L.lives_gpu_layer_sync(cp) == 0 and L.lives_gpu_layer_sync(lay) == 0
    want = oracle_step(orc, src, sw, sh, None, dw, dh, None, 0, None, True)
    assert (view(wh, cp)[:, :dw * 4] == want).all(), "the copy holds the frame as it was when it was copied"
    want2 = oracle_step(orc, src, sw, sh, None, dw, dh, None, 0, srgb_to(orc, 2), True)
    assert (view(wh, lay)[:, :dw * 4] == want2).all()
    assert L.lives_gpu_layer_unpin(lay) == 0
    assert L.lives_gpu_convert_layer_palette(cp, BGRA32, 0) == 1 and L.lives_gpu_layer_sync(cp) == 0      # the copy is an ordinary pinned layer
    back = np.zeros((dh, dw * 4), np.uint8)
    orc.orc_swizzle(po.OPS.index("swap3postalpha"), 0, P(want), dw * 4, P(back), dw * 4, dw, dh, None)
    assert (view(wh, cp)[:, :dw * 4] == back).all()
    assert L.lives_gpu_layer_unpin(cp) == 0
    # an unpinned source: nothing to do; mismatched geometry: refused
    a, b = wh.new_layer(RGBA32, 64, 32, [frame(rng, 64, 32, 4)], gamma=1), wh.new_layer(RGBA32, 64, 32, [frame(rng, 64, 32, 4)], gamma=1)
    assert L.lives_gpu_layer_copy(b, a) == 0 and wh.geti(b, "host_gpu_resident") is None
    assert L.lives_gpu_layer_pin(a) == 0
    c = wh.new_layer(RGBA32, 32, 32, [frame(rng, 32, 32, 4)], gamma=1)
    assert L.lives_gpu_layer_copy(c, a) != 0
    assert L.lives_gpu_layer_unpin(a) == 0


@pytest.mark.parametrize("geom", [(128, 72, 200, 112, None), (128, 72, 200, 112, (220, 120)), (300, 170, 200, 112, None), (256, 144, 128, 72, None)])
@pytest.mark.parametrize("mode", [1, 0], ids=["deferred", "eager"])
def test_the_hosts_word_that_a_frame_is_opaque(seam, orc, deferred, geom, mode):
    """lives_gpu_layer_set_opaque(layer, 1) on a frame whose alpha is 255 everywhere: the scalers' all-opaque instantiations run (enlargement, pair kernel; the 2:1
    chain has none) -- the bytes are the oracle's, the word stays on the layer through the step and goes when the host takes it back"""
    L, wh, H = seam
    sw, sh, dw, dh, canvas = geom
    rng = np.random.default_rng(0xDEFE + sw + dw)
    ow, oh = canvas if canvas else (dw, dh)
    src = frame(rng, sw, sh, 4)
    src[:, 3::4] = 255
    l2a = frame(rng, ow, oh, 4, alpha_mix=True)
    L.lives_gpu_set_deferred(mode)
    lay = wh.new_layer(BGRA32, sw, sh, [src], gamma=1)
    l2 = wh.new_layer(RGBA32, ow, oh, [l2a], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0 and L.lives_gpu_layer_pin(l2) == 0
    assert L.lives_gpu_layer_set_opaque(lay, 1) == 0 and wh.geti(lay, "host_gpu_opaque") == 1
    plan_step(L, wh, H, lay, l2, dw, dh, canvas, 77, 2)
    assert L.lives_gpu_layer_sync(lay) == 0
    want = oracle_step(orc, src, sw, sh, l2a, dw, dh, canvas, 77, srgb_to(orc, 2), True)
    assert (view(wh, lay)[:, :ow * 4] == want).all()
    assert wh.geti(lay, "host_gpu_opaque") == 1
    assert L.lives_gpu_layer_set_opaque(lay, 0) == 0 and wh.geti(lay, "host_gpu_opaque") is None
    assert L.lives_gpu_layer_unpin(lay) == 0 and L.lives_gpu_layer_unpin(l2) == 0
    L.lives_gpu_set_deferred(1)
