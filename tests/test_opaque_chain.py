"""LGPU_INTERP_OPAQUE: the caller's word that every source pixel has alpha 255 lets the gaussian chain run its lighter instantiation (pixbuf.hip, pb_half_hrow<.., OPAQUE>):
the bytes must be those of the general kernel and of the oracle's chain (orc_chain: swizzle -> gdk-pixbuf scale -> gauss5 -> chroma blend -> LUT), HYPER and BILINEAR,
with and without the R <-> B swap, on small frames with edge strips and at BASELINE's 3840 x 2160."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import dev, host

pytestmark = pytest.mark.gpu
P = po.P
PIXBUF, OPAQUE = 0x100, 0x200


@pytest.mark.parametrize("interp", [3, 2])
@pytest.mark.parametrize("swap", [1, 0])
@pytest.mark.parametrize("geom", [(256, 144), (1000, 600), (3840, 2160)])
def test_opaque_chain_equals_the_general_kernel_and_the_oracle(gpu, orc, geom, swap, interp):
    sw, sh = geom
    dw, dh = sw // 2, sh // 2
    rng = np.random.default_rng(0x0FA0 + sw + swap * 2 + interp)
    n = 1 if sw > 2000 else 3
    srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(n)]
    for s in srcs:
        s[:, 3::4] = 255
    l2s = [rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8) for _ in range(n)]
    for t in l2s:
        a = t[:, 3::4]
        a[rng.random(a.shape) < 0.5] = 255
    lut = rng.permutation(256).astype(np.uint8)
    d_s, d_l = [dev(a) for a in srcs], [dev(a) for a in l2s]
    outs = {}
    for flag in (OPAQUE, 0):
        d_o = [dev(np.zeros((dh, dw * 4), np.uint8)) for _ in range(n)]
        prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=swap, interp=interp | PIXBUF | flag, do_blur=1, bf=113, lut=lut)
        gpu.chain(prm, gpu.chain_tracks(d_s, d_l, d_o))
        outs[flag] = [host(t) for t in d_o]
    for i in range(n):
        assert (outs[OPAQUE][i] == outs[0][i]).all(), "track %d: the all-opaque instantiation differs from the general kernel" % i
    if sw <= 1000:
        for i in range(n):
            want = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_chain(P(srcs[i]), sw * 4, sw, sh, P(l2s[i]), dw * 4, P(want), dw * 4, dw, dh, swap, interp | PIXBUF, 1, 113, P(lut)) == 0
            assert (outs[OPAQUE][i] == want).all(), i


def test_the_flag_changes_nothing_where_there_is_no_lighter_form(gpu):
    """without the gaussian, or off the exact 2:1 case, LGPU_INTERP_OPAQUE is ignored: same bytes, translucent sources included"""
    rng = np.random.default_rng(0x0FA9)
    for (sw, sh, dw, dh, blur) in ((256, 144, 128, 72, 0), (262, 150, 128, 72, 1)):
        src, l2 = rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8), rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8)
        if blur:
            src[:, 3::4] = 255
        res = []
        for flag in (OPAQUE, 0):
            d_o = dev(np.zeros((dh, dw * 4), np.uint8))
            prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | PIXBUF | flag, do_blur=blur, bf=90, lut=None)
            gpu.chain(prm, gpu.chain_tracks([dev(src)], [dev(l2)], [d_o]))
            res.append(host(d_o))
        assert (res[0] == res[1]).all()


@pytest.mark.parametrize("interp", [3, 2])
@pytest.mark.parametrize("geom", [(384, 216, 171, 96), (200, 120, 133, 80), (128, 72, 200, 112), (96, 54, 320, 180), (1920, 1080, 1280, 720), (300, 200, 100, 50), (384, 216, 128, 72), (131, 77, 64, 36)])
def test_opaque_scaler_equals_the_general_kernels_and_the_oracle(gpu, orc, geom, interp):
    """lgpu_pixbuf_scale[_batch] with interp | LGPU_INTERP_OPAQUE on frames whose alpha is 255 everywhere: the pair kernel's lighter form (and, where another kernel
    serves the ratio, the flag ignored) -- the bytes of the general path and of the oracle's gdk-pixbuf restatement, alpha 255 out"""
    sw, sh, dw, dh = geom
    rng = np.random.default_rng(0x0FB0 + sw + dw + interp)
    n = 3
    srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(n)]
    for s in srcs:
        s[:, 3::4] = 255
    d_s = [dev(a) for a in srcs]
    outs = {}
    for flag in (OPAQUE, 0):
        d_o = [dev(np.full((dh + 1, dw * 4), 0xA5, np.uint8)) for _ in range(n)]
        gpu.pixbuf_scale_batch(d_s, d_o, sw, sh, dw, dh, channels=4, interp=interp | flag)
        one = dev(np.full((dh + 1, dw * 4), 0xA5, np.uint8))
        gpu.pixbuf_scale(d_s[1], one, sw, sh, dw, dh, channels=4, interp=interp | flag)
        outs[flag] = [host(t) for t in d_o]
        assert (host(one) == outs[flag][1]).all()
    for i in range(n):
        assert (outs[OPAQUE][i] == outs[0][i]).all(), i
        assert (outs[OPAQUE][i][dh] == 0xA5).all()
    if sw <= 400:
        want = np.zeros((dh, dw * 4), np.uint8)
        assert orc.orc_pixbuf_scale(P(srcs[0]), sw * 4, sw, sh, P(want), dw * 4, dw, dh, 4, interp) == 0
        assert (outs[OPAQUE][0][:dh] == want).all()


@pytest.mark.parametrize("case", [(256, 144, 128, 72, 0, 0, None), (256, 144, 128, 72, 1, 0, None), (256, 144, 128, 72, 1, OPAQUE, None), (200, 120, 133, 80, 0, 0, None),
                                  (256, 144, 128, 72, 0, 0, (160, 100, 16, 14)), (200, 120, 133, 80, 0, 0, (150, 90, 9, 5))])
def test_a_blend_amount_per_track(gpu, orc, case):
    """lgpu_chain_amounts: one launch, every track blended by its own amount -- each track's bytes are those of lgpu_chain[_canvas] on that track alone with
    params->bf = its amount (and, without a canvas, the oracle's chain)"""
    sw, sh, dw, dh, blur, flag, canvas = case
    rng = np.random.default_rng(0x0FC0 + sw + dw + blur)
    n = 5
    amounts = [0, 255, 17, 128, 201]
    cw, ch = (canvas[0], canvas[1]) if canvas else (dw, dh)
    srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(n)]
    if flag:
        for s in srcs:
            s[:, 3::4] = 255
    l2s = [rng.integers(0, 256, (ch, cw * 4), dtype=np.uint8) for _ in range(n)]
    lut = rng.permutation(256).astype(np.uint8)
    d_s, d_l = [dev(a) for a in srcs], [dev(a) for a in l2s]
    d_o = [dev(np.zeros((ch, cw * 4), np.uint8)) for _ in range(n)]
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, cw * 4, cw * 4, swap_rb=1, interp=3 | PIXBUF | flag, do_blur=blur, bf=99, lut=lut)
    gpu.chain_amounts(prm, gpu.chain_tracks(d_s, d_l, d_o), amounts, canvas)
    got = [host(t) for t in d_o]
    for i in range(n):
        one = dev(np.zeros((ch, cw * 4), np.uint8))
        p1 = gpu.chain_params(sw, sh, sw * 4, dw, dh, cw * 4, cw * 4, swap_rb=1, interp=3 | PIXBUF | flag, do_blur=blur, bf=amounts[i], lut=lut)
        tr = gpu.chain_tracks([d_s[i]], [d_l[i]], [one])
        if canvas:
            gpu.chain_canvas(p1, tr, *canvas)
        else:
            gpu.chain(p1, tr)
        assert (got[i] == host(one)).all(), "track %d (amount %d)" % (i, amounts[i])
        if not canvas:
            want = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_chain(P(srcs[i]), sw * 4, sw, sh, P(l2s[i]), dw * 4, P(want), dw * 4, dw, dh, 1, 3 | PIXBUF, blur, amounts[i], P(lut)) == 0
            assert (got[i] == want).all(), i


NOBLEND = 0x400


@pytest.mark.parametrize("case", [(256, 144, 128, 72, None), (200, 120, 133, 80, None), (96, 54, 200, 112, None), (384, 216, 128, 72, None),
                                  (256, 144, 128, 72, (160, 100, 16, 14)), (200, 120, 133, 80, (150, 90, 9, 5)), (96, 54, 200, 112, (210, 120, 5, 4))])
@pytest.mark.parametrize("interp", [3, 2, 0])
def test_a_track_without_a_layer_2(gpu, orc, case, interp):
    """lgpu_chain_amounts with LGPU_INTERP_NOBLEND: [R <-> B] -> gdk-pixbuf scale [-> letterbox] -> gamma LUT in one launch (every ratio: the exact 2:1, the pair
    kernel's, an enlargement, 3:1; NEAREST goes the staged way) == the oracle's stages one after the other"""
    sw, sh, dw, dh, canvas = case
    rng = np.random.default_rng(0x0FD0 + sw + dw + interp)
    n = 3
    cw, ch = (canvas[0], canvas[1]) if canvas else (dw, dh)
    srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(n)]
    lut = rng.permutation(256).astype(np.uint8)
    d_s = [dev(a) for a in srcs]
    for swap, use_lut in ((1, True), (0, False), (0, True)):
        d_o = [dev(np.full((ch, cw * 4), 0x5A, np.uint8)) for _ in range(n)]
        prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, cw * 4, cw * 4, swap_rb=swap, interp=interp | PIXBUF | NOBLEND, do_blur=0, bf=0, lut=lut if use_lut else None)
        gpu.chain_amounts(prm, gpu.chain_tracks(d_s, None, d_o), None, canvas)
        for i in range(n):
            conv = np.zeros((sh, sw * 4), np.uint8)
            if swap:
                orc.orc_swizzle(po.OPS.index("swap3postalpha"), 0, P(srcs[i]), sw * 4, P(conv), sw * 4, sw, sh, None)
            else:
                conv[:] = srcs[i]
            want = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_pixbuf_scale(P(conv), sw * 4, sw, sh, P(want), dw * 4, dw, dh, 4, interp) == 0
            if canvas:
                big = np.zeros((ch, cw * 4), np.uint8)
                inner = np.zeros((ch, cw * 4), np.uint8)
                inner[:, 3::4] = 255                                            # opaque black bars
                inner[canvas[3]:canvas[3] + dh, canvas[2] * 4:(canvas[2] + dw) * 4] = want
                big[:] = inner
                want = big
            if use_lut:
                orc.orc_gamma_apply(P(want), want.strides[0], want.shape[1] // 4, want.shape[0], 4, 0, P(lut))
            got = host(d_o[i])
            assert (got == want).all(), "track %d swap %d lut %s: %d bytes differ" % (i, swap, use_lut, int((got != want).sum()))


@pytest.mark.parametrize("case", [(200, 120, 133, 80, None, 0), (200, 120, 133, 80, (150, 90, 9, 5), 0), (200, 120, 133, 80, None, NOBLEND), (256, 144, 128, 72, (160, 100, 16, 14), NOBLEND)])
def test_the_staged_groups_equal_the_one_launch_forms(gpu, tune, case):
    """LGPU_PB_CHAIN_GROUP = g: the chain's stages apart, g tracks at a time through lgpu_pixbuf_scale_batch and the batched last kernel (what the gaussian and the
    polyphase backend keep) -- the bytes of the one-launch forms, for groups that divide the tracks and groups that do not"""
    sw, sh, dw, dh, canvas, flag = case
    rng = np.random.default_rng(0x0FE0 + sw + dw + flag)
    n = 5
    amounts = [3, 250, 77, 128, 40]
    cw, ch = (canvas[0], canvas[1]) if canvas else (dw, dh)
    srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(n)]
    l2s = [rng.integers(0, 256, (ch, cw * 4), dtype=np.uint8) for _ in range(n)]
    lut = rng.permutation(256).astype(np.uint8)
    d_s, d_l = [dev(a) for a in srcs], [dev(a) for a in l2s]
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, cw * 4, cw * 4, swap_rb=1, interp=3 | PIXBUF | flag, do_blur=0, bf=99, lut=lut)
    res = []
    for group in (0, 1, 2, 64):
        tune("PB_CHAIN_GROUP", group if group else -1)
        d_o = [dev(np.full((ch, cw * 4), 0x33, np.uint8)) for _ in range(n)]
        gpu.chain_amounts(prm, gpu.chain_tracks(d_s, None if flag else d_l, d_o), None if flag else amounts, canvas)
        res.append([host(t) for t in d_o])
    for g in range(1, len(res)):
        for i in range(n):
            assert (res[g][i] == res[0][i]).all(), "group setting %d, track %d" % (g, i)


def test_bad_arguments_are_refused(gpu):
    """lgpu_chain_amounts: no amounts with a layer 2, a polyphase request, the gaussian without a resize stage, dst == src -- LGPU_E_BADARG / UNSUPPORTED, nothing launched"""
    from lives_amd.lib import LgpuError
    a, b, o = dev(np.zeros((72, 128 * 4), np.uint8)), dev(np.zeros((72, 128 * 4), np.uint8)), dev(np.zeros((72, 128 * 4), np.uint8))
    big = dev(np.zeros((144, 256 * 4), np.uint8))
    ok = gpu.chain_params(256, 144, 256 * 4, 128, 72, 128 * 4, 128 * 4, swap_rb=1, interp=3 | PIXBUF, do_blur=0, bf=9, lut=None)
    with pytest.raises(LgpuError):
        gpu.chain_amounts(ok, gpu.chain_tracks([big], [b], [o]), None)                       # a layer 2 but no amounts
    poly = gpu.chain_params(256, 144, 256 * 4, 128, 72, 128 * 4, 128 * 4, swap_rb=1, interp=3, do_blur=0, bf=9, lut=None)
    with pytest.raises(LgpuError):
        gpu.chain_amounts(poly, gpu.chain_tracks([big], [b], [o]), [9])
    flat_blur = gpu.chain_params(128, 72, 128 * 4, 128, 72, 128 * 4, 128 * 4, swap_rb=1, interp=3 | PIXBUF, do_blur=1, bf=9, lut=None)
    with pytest.raises(LgpuError):
        gpu.chain_amounts(flat_blur, gpu.chain_tracks([a], [b], [o]), [9])
    flat = gpu.chain_params(128, 72, 128 * 4, 128, 72, 128 * 4, 128 * 4, swap_rb=1, interp=3 | PIXBUF, do_blur=0, bf=9, lut=None)
    with pytest.raises(LgpuError):
        gpu.chain_amounts(flat, gpu.chain_tracks([a], [b], [a]), [9])                         # in place
    gpu.chain_amounts(flat, gpu.chain_tracks([a], [b], [o]), [9])                             # ... and the same call with a destination of its own is fine
