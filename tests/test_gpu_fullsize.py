"""GPU parity at the sizes BASELINE.json quotes (configs C1 .. C5): the oracle is fast enough for a full compare of single frames;
the 16-track 4K batch is checked on two tracks against the oracle plus size-independent properties over all of them (identical inputs
give identical outputs whatever the track slot; a permuted batch gives the permuted result; involutions come back to the source)."""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, assert_same, dev, frame, host

pytestmark = pytest.mark.gpu
P = po.P


def l2s_lut():
    lut = np.zeros(256, np.uint8)
    assert po.oracle().orc_gamma_lut8(1.0, po.GAMMA_LINEAR, po.GAMMA_SRGB, 1.4, P(lut)) == 1
    return lut


def test_c1_rgb24_to_bgra32_640x480(gpu, orc):
    rng = np.random.default_rng(4001)
    w, h = 640, 480
    src = frame(rng, w, h, 3, stride=1920)
    want = np.zeros((h, 2560), np.uint8)
    op = po.OPS.index("swap3addpost")
    orc.orc_swizzle(op, 0, P(src), 1920, P(want), 2560, w, h, None)
    d = dev(np.zeros_like(want))
    gpu.swizzle(op, dev(src), d, w, h)
    assert (host(d) == want).all()


def test_c2_yuv420p_1080p_to_rgba_with_gamma(gpu, orc):
    rng = np.random.default_rng(4002)
    w, h = 1920, 1080
    Y, U, V = (rng.integers(0, 256, s, dtype=np.uint8) for s in ((h, w), (h // 2, w // 2), (h // 2, w // 2)))
    lut = l2s_lut()
    strides = (ctypes.c_int * 3)(w, w // 2, w // 2)
    want = np.zeros((h, w * 4), np.uint8)
    orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, U.size, V.size, P(want), w * 4, w, h, 4, 0, 0, 0, 2, P(lut), 0)
    d = dev(np.zeros_like(want))
    gpu.yuv420p_to_rgb(dev(Y), dev(U), dev(V), d, w, h, lut=lut)
    assert_same(host(d), want, w, h, 4, "C2")
    # the batched launch (16 tracks) gives every track the single-frame result
    frames = [(dev(Y), dev(U), dev(V), dev(np.zeros_like(want))) for _ in range(16)]
    gpu.yuv420p_to_rgb_batch(frames, w, h, lut=lut)
    for f in frames:
        assert_same(host(f[3]), want, w, h, 4, "C2 batch")


def test_c3_resize_letterbox_blend_4k(gpu, orc):
    rng = np.random.default_rng(4003)
    sw, sh, dw, dh, nw, nh = 3840, 2160, 1920, 1080, 1920, 1200
    src = frame(rng, sw, sh, 4, alpha_mix=True)
    rs = np.zeros((dh, dw * 4), np.uint8)
    assert orc.orc_resize(P(src), src.strides[0], sw, sh, P(rs), dw * 4, dw, dh, 4, 3) == 0
    d_rs = dev(np.zeros_like(rs))
    gpu.resize(dev(src), d_rs, sw, sh, dw, dh, psize=4, interp=3)
    assert_same(host(d_rs), rs, dw, dh, 4, "C3 resize")
    lb = np.zeros((nh, nw * 4), np.uint8)
    black = np.array([0, 0, 0, 255], np.uint8)
    orc.orc_letterbox(P(rs), dw * 4, dw, dh, P(lb), nw * 4, nw, nh, 4, P(black))
    d_lb = dev(np.zeros_like(lb))
    gpu.letterbox(d_rs, d_lb, dw, dh, nw, nh, 4, black)
    assert (host(d_lb) == lb).all(), "C3 letterbox"
    l2 = frame(rng, nw, nh, 4, alpha_mix=True)
    want = lb.copy()
    orc.orc_blend_chroma(P(want), nw * 4, P(l2), l2.strides[0], P(want), nw * 4, nw, nh, 4, 0, 100)
    gpu.blend_chroma(d_lb, dev(l2), d_lb, nw, nh, 4, 100)
    assert_same(host(d_lb), want, nw, nh, 4, "C3 blend")


def test_c4_gauss5_and_colorkey_4k(gpu, orc):
    rng = np.random.default_rng(4004)
    w, h = 3840, 2160
    src = frame(rng, w, h, 4)
    want = np.zeros_like(src)
    orc.orc_gauss5(P(src), src.strides[0], P(want), want.strides[0], w, h, 4)
    d = dev(np.zeros_like(src))
    gpu.gauss5(dev(src), d, w, h, psize=4)
    assert_same(host(d), want, w, h, 4, "C4 gauss5")
    s0, s1 = frame(rng, w, h, 3), frame(rng, w, h, 3)
    s1[:, :3 * 700] = np.tile(np.array([8, 250, 12], np.uint8), 700 * (align(w * 3) // (3 * 700)) + 1)[:3 * 700]      # a keyed region
    wk = np.zeros_like(s0)
    orc.orc_colorkey(P(s0), s0.strides[0], P(s1), s1.strides[0], P(wk), wk.strides[0], w, h, 0, 0.2, 0.8, 10, 255, 10, 0)
    dk = dev(np.zeros_like(s0))
    gpu.colorkey(dev(s0), dev(s1), dk, w, h, 0, 0.2, 0.8, (10, 255, 10))
    assert_same(host(dk), wk, w, h, 3, "C4 colour key")


@pytest.mark.parametrize("do_blur", [0, 1])
def test_c5_sixteen_track_4k_chain(gpu, orc, do_blur):
    import torch
    rng = np.random.default_rng(4005 + do_blur)
    sw, sh, dw, dh, T = 3840, 2160, 1920, 1080, 16
    lut = l2s_lut()
    base = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(2)]
    l2b = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(2)]
    # tracks 0 / 1 distinct, the other 14 alternate between the two: same input -> same output whatever the slot
    src_d = [dev(base[t & 1]) for t in range(T)]
    l2_d = [dev(l2b[t & 1]) for t in range(T)]
    dst_d = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3, do_blur=do_blur, bf=107, lut=lut)
    gpu.chain(prm, gpu.chain_tracks(src_d, l2_d, dst_d))
    for i in range(2):
        want = np.zeros((dh, dw * 4), np.uint8)
        assert orc.orc_chain(P(base[i]), sw * 4, sw, sh, P(l2b[i]), dw * 4, P(want), dw * 4, dw, dh, 1, 3, do_blur, 107, P(lut)) == 0
        assert_same(host(dst_d[i]), want, dw, dh, 4, "C5 track %d blur=%d" % (i, do_blur))
    for t in range(2, T):
        assert torch.equal(dst_d[t], dst_d[t & 1]), "track %d differs from track %d with the same input" % (t, t & 1)
    # a permuted batch gives the permuted result
    perm = [(5 * t + 3) % T for t in range(T)]
    dst2 = [torch.zeros_like(d) for d in dst_d]
    gpu.chain(prm, gpu.chain_tracks([src_d[p] for p in perm], [l2_d[p] for p in perm], dst2))
    for t in range(T):
        assert torch.equal(dst2[t], dst_d[perm[t]])


def test_involutions_at_4k(gpu):
    """size-independent properties: swapping R and B twice, mirroring twice, negating twice, clamped -> unclamped -> clamped on legal values"""
    import torch
    rng = np.random.default_rng(4006)
    w, h = 3840, 2160
    src = dev(frame(rng, w, h, 4))
    a, b = torch.zeros_like(src), torch.zeros_like(src)
    gpu.swizzle(po.OPS.index("swap4"), src, a, w, h)
    gpu.swizzle(po.OPS.index("swap4"), a, b, w, h)
    assert torch.equal(b, src)
    gpu.mirror(2, src, a, w, h, 4)          # mirror x, mirror y: the top-left quadrant of the source survives both
    assert torch.equal(a[:h // 2, :(w // 2) * 4], src[:h // 2, :(w // 2) * 4])
    neg = gpu.fx_luts(0, 3)
    gpu.byte_luts(src, a, w, h, 4, neg)
    gpu.byte_luts(a, b, w, h, 4, neg)
    assert torch.equal(b, src)


def test_yuva_premult_4k_properties(gpu, orc):
    """alpha_premult on a 3840x2160 YUVA4444P layer: full compare with the oracle (clamped tables), and on the unclamped tables the
    size-independent facts alpha 255 -> unchanged in both directions (the tables' ratio is 1 there; what alpha 0 gives is the reference's
    inf / NaN conversion, left to the oracle compare)"""
    rng = np.random.default_rng(4010)
    w, h = 3840, 2160
    planes = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(4)]
    planes[3][: h // 2] = 255
    planes[3][h // 2: h // 2 + 64] = 0
    want = [p.copy() for p in planes]
    pp = (ctypes.c_void_p * 4)(*[x.ctypes.data for x in want])
    ss = (ctypes.c_int * 4)(*[x.strides[0] for x in want])
    orc.orc_alpha_premult_yuva(pp, ss, w, h, 545, 1, 0)
    ds = [dev(p) for p in planes]
    gpu.alpha_premult_yuva(ds, w, h, 545, 1, un=0)
    for i in range(4):
        assert (host(ds[i]) == want[i]).all(), "clamped plane %d" % i
    ds = [dev(p) for p in planes]
    gpu.alpha_premult_yuva(ds, w, h, 545, 0, un=0)
    for i in range(3):
        got = host(ds[i])
        assert (got[: h // 2] == planes[i][: h // 2]).all()
    gpu.alpha_premult_yuva(ds, w, h, 545, 0, un=1)
    for i in range(3):
        assert (host(ds[i])[: h // 2] == planes[i][: h // 2]).all()
    assert (host(ds[3]) == planes[3]).all()                      # the alpha plane is only read
