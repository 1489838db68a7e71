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


PIXBUF = 0x100      # LGPU_INTERP_PIXBUF: the resize stage on gdk-pixbuf's arithmetic (the pinned one, and what bench.py launches by default)


def _c5_batch(rng, T):
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    base = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(2)]
    l2b = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(2)]
    # tracks 0 / 1 distinct, the others alternate between the two: same input -> same output whatever the slot
    return base, l2b, [dev(base[t & 1]) for t in range(T)], [dev(l2b[t & 1]) for t in range(T)]


@pytest.mark.parametrize("interp", [3, 3 | PIXBUF], ids=["polyphase", "pixbuf"])
@pytest.mark.parametrize("do_blur", [0, 1])
def test_c5_sixteen_track_4k_chain(gpu, orc, do_blur, interp):
    """the launch bench.py times -- 16 x 3840x2160 BGRA32 tracks, swap + scale 0.5x + [blur] + blend + LUT in ONE launch -- on both resize backends.  With the
    pixbuf backend this is k_pb_half<1, 1, BLUR> at full device size: do_blur = 1 reaches the 24-row bands pb_half_geometry() only picks for >= 8 4K tracks.
    Reference: src/colourspace.c:15262-15322 (the gdk-pixbuf body of resize_layer_full), simple_blend.c:117-146"""
    import torch
    rng = np.random.default_rng(4005 + do_blur)
    sw, sh, dw, dh, T = 3840, 2160, 1920, 1080, 16
    lut = l2s_lut()
    base, l2b, src_d, l2_d = _c5_batch(rng, T)
    dst_d = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=interp, do_blur=do_blur, bf=107, lut=lut)
    gpu.chain(prm, gpu.chain_tracks(src_d, l2_d, dst_d))
    for i in range(2):
        want = np.zeros((dh, dw * 4), np.uint8)
        assert orc.orc_chain(P(base[i]), sw * 4, sw, sh, P(l2b[i]), dw * 4, P(want), dw * 4, dw, dh, 1, interp, do_blur, 107, P(lut)) == 0
        assert_same(host(dst_d[i]), want, dw, dh, 4, "C5 track %d blur=%d interp=%#x" % (i, do_blur, interp))
    for t in range(2, T):
        assert torch.equal(dst_d[t], dst_d[t & 1]), "track %d differs from track %d with the same input" % (t, t & 1)
    # a permuted batch gives the permuted result
    perm = [(5 * t + 3) % T for t in range(T)]
    dst2 = [torch.zeros_like(d) for d in dst_d]
    gpu.chain(prm, gpu.chain_tracks([src_d[p] for p in perm], [l2_d[p] for p in perm], dst2))
    for t in range(T):
        assert torch.equal(dst2[t], dst_d[perm[t]])


@pytest.mark.parametrize("geom", [(3840, 2160, 1706, 960), (1920, 1080, 1280, 720), (1280, 720, 1920, 1080), (3840, 2160, 1280, 720)], ids=["4k_to_1706x960", "1080p_to_720p", "720p_to_1080p", "4k_to_720p"])
def test_chain_off_2to1_at_size_in_one_launch(gpu, orc, geom):
    """the chain on eight tracks at ratios other than 2:1 (the scaler of the ratio -- pair, enlargement, 3:1 gather kernel -- with the chain's last stages in its
    store), a blend amount per track: tracks 0 / 1 against the oracle's chain, the rest against the track with the same input"""
    import torch
    sw, sh, dw, dh = geom
    T = 8
    rng = np.random.default_rng(4105 + dw)
    lut = l2s_lut()
    base = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(2)]
    l2b = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(2)]
    src_d, l2_d = [dev(base[t & 1]) for t in range(T)], [dev(l2b[t & 1]) for t in range(T)]
    dst_d = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    amounts = [107 if t < 2 or (t & 1) == 0 else 31 for t in range(T)]          # tracks 0, 1, 2, 4, 6: 107; 3, 5, 7: 31
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | PIXBUF, do_blur=0, bf=5, lut=lut)
    gpu.chain_amounts(prm, gpu.chain_tracks(src_d, l2_d, dst_d), amounts)
    for i in range(2):
        want = np.zeros((dh, dw * 4), np.uint8)
        assert orc.orc_chain(P(base[i]), sw * 4, sw, sh, P(l2b[i]), dw * 4, P(want), dw * 4, dw, dh, 1, 3 | PIXBUF, 0, 107, P(lut)) == 0
        assert_same(host(dst_d[i]), want, dw, dh, 4, "chain %s track %d" % (geom, i))
    want31 = np.zeros((dh, dw * 4), np.uint8)
    assert orc.orc_chain(P(base[1]), sw * 4, sw, sh, P(l2b[1]), dw * 4, P(want31), dw * 4, dw, dh, 1, 3 | PIXBUF, 0, 31, P(lut)) == 0
    assert_same(host(dst_d[3]), want31, dw, dh, 4, "chain %s track 3 (its own amount)" % (geom,))
    for t in range(2, T):
        assert torch.equal(dst_d[t], dst_d[0] if (t & 1) == 0 else dst_d[3]), "track %d" % t


def test_a_4k_track_without_a_layer_2_at_size(gpu, orc):
    """LGPU_INTERP_NOBLEND at the headline geometry (k_pb_half<2, ..>): BGRA32 4K -> R <-> B -> 1920 x 1080 -> gamma LUT, four tracks in one launch"""
    import torch
    sw, sh, dw, dh, T = 3840, 2160, 1920, 1080, 4
    rng = np.random.default_rng(4207)
    lut = l2s_lut()
    base = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(2)]
    src_d = [dev(base[t & 1]) for t in range(T)]
    dst_d = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | PIXBUF | 0x400, do_blur=0, bf=0, lut=lut)
    gpu.chain_amounts(prm, gpu.chain_tracks(src_d, None, dst_d), None)
    for i in range(2):
        conv = np.zeros((sh, sw * 4), np.uint8)
        orc.orc_swizzle(po.OPS.index("swap3postalpha"), 0, P(base[i]), sw * 4, P(conv), sw * 4, sw, sh, None)
        want = np.zeros((dh, dw * 4), np.uint8)
        assert orc.orc_pixbuf_scale(P(conv), sw * 4, sw, sh, P(want), dw * 4, dw, dh, 4, 3) == 0
        orc.orc_gamma_apply(P(want), dw * 4, dw, dh, 4, 0, P(lut))
        assert_same(host(dst_d[i]), want, dw, dh, 4, "no layer 2, track %d" % i)
    for t in range(2, T):
        assert torch.equal(dst_d[t], dst_d[t & 1])


@pytest.mark.parametrize("shape", ["feeder_lanes", "th8", "xcd_runs", "bands_one_by_one", "groups_of_5", "eight_per_cu"])
def test_c5_bench_launch_in_its_other_shapes(gpu, orc, tune, shape):
    """the same 16-track launch in the shapes that ship behind the launch-shape switches (strips with feeder lanes instead of 64 storing lanes, another band height, the
    contiguous-run work order the blur chain keeps, other band groupings, eight workgroups per CU): every shape must give the bytes of the default shape, which
    test_c5_sixteen_track_4k_chain compares with the oracle -- and track 0 against the oracle here as well"""
    import torch
    rng = np.random.default_rng(4015)
    sw, sh, dw, dh, T = 3840, 2160, 1920, 1080, 16
    lut = l2s_lut()
    base, l2b, src_d, l2_d = _c5_batch(rng, T)
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | PIXBUF, do_blur=0, bf=31, lut=lut)
    ref = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    gpu.chain(prm, gpu.chain_tracks(src_d, l2_d, ref))
    if shape == "feeder_lanes":
        tune("PBH_ALIGNED", 0)
    elif shape == "th8":
        tune("PBH_TH", 8)
    elif shape == "xcd_runs":           # column groups fastest, every XCD a contiguous run of the sequence (what the blur chain keeps)
        tune("PBH_ORDER", 1)
    elif shape == "bands_one_by_one":   # all XCDs on one track, the bands dealt one by one instead of by eighths
        tune("PBH_GROUP", 1)
    elif shape == "groups_of_5":        # a group size that leaves some XCDs a turn short (216 bands = 44 groups: the padding slots of the grid must stay idle)
        tune("PBH_GROUP", 5)
    else:                               # eight_per_cu: no dynamic-LDS cap on the workgroups per CU
        tune("PBH_OCC", 0)
    got = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
    gpu.chain(prm, gpu.chain_tracks(src_d, l2_d, got))
    for t in range(T):
        assert torch.equal(got[t], ref[t]), "track %d: shape %s differs from the default shape" % (t, shape)
    want = np.zeros((dh, dw * 4), np.uint8)
    assert orc.orc_chain(P(base[0]), sw * 4, sw, sh, P(l2b[0]), dw * 4, P(want), dw * 4, dw, dh, 1, 3 | PIXBUF, 0, 31, P(lut)) == 0
    assert_same(host(got[0]), want, dw, dh, 4, "C5 %s" % shape)


@pytest.mark.parametrize("do_blur", [0, 1])
def test_c5_through_the_c_stepper_with_a_device_parameter_block(gpu, orc, do_blur):
    """lgpu_chain_step (the per-step host path of the N > 1 bench, dist.cpp): 16 tracks per step, the blend amount read by the kernel from the stepper's device
    block -- a different value every step, and params.bf deliberately wrong"""
    import torch
    from lives_amd import dist as ld
    rng = np.random.default_rng(4025 + do_blur)
    sw, sh, dw, dh, T = 3840, 2160, 1920, 1080, 16
    lut = l2s_lut()
    base, l2b, src_d, l2_d = _c5_batch(rng, T)
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | PIXBUF, do_blur=do_blur, bf=5, lut=lut)
    bfs = [201, 17, 128]
    outs = [[torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(T)] for _ in bfs]
    st = ld.Stepper(None, [bfs[0], 0, 0, 0])
    for s, o in enumerate(outs):
        st.step([bfs[s + 1], 0, 0, 0] if s + 1 < len(bfs) else None, prm, gpu.chain_tracks(src_d, l2_d, o))
    torch.cuda.synchronize()
    st.close()
    for s, bf in enumerate(bfs):
        for i in range(2):
            want = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_chain(P(base[i]), sw * 4, sw, sh, P(l2b[i]), dw * 4, P(want), dw * 4, dw, dh, 1, 3 | PIXBUF, do_blur, bf, P(lut)) == 0
            assert_same(host(outs[s][i]), want, dw, dh, 4, "C5 step %d (bf %d) track %d blur=%d" % (s, bf, i, do_blur))
        for t in range(2, T):
            assert torch.equal(outs[s][t], outs[s][t & 1])


@pytest.mark.parametrize("th", [24, 7, 1])
def test_blur_chain_band_seams_at_forced_heights(gpu, orc, tune, th):
    """k_pb_half<.., BLUR> with the band height forced (24 rows = what full-device launches take; 7: bands that end inside the frame; 1: every row a seam) on frames
    small enough for a full compare on every run: several bands, bands walking up and down, the frame's last band shorter than the others"""
    tune("PBH_TH", th)
    rng = np.random.default_rng(4040 + th)
    lut = l2s_lut()
    for (sw, sh, dw, dh, interp, ntr) in [(512, 200, 256, 100, 3, 2), (1000, 132, 500, 66, 2, 1), (3840, 160, 1920, 80, 3, 1), (248, 1000, 124, 500, 3, 1)]:
        srcs = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(ntr)]
        l2s = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(ntr)]
        irow, orow = srcs[0].strides[0], l2s[0].strides[0]
        dd = [dev(np.zeros((dh, orow), np.uint8)) for _ in range(ntr)]
        for blur in (1, 0):
            prm = gpu.chain_params(sw, sh, irow, dw, dh, orow, orow, swap_rb=1, interp=interp | PIXBUF, do_blur=blur, bf=77, lut=lut)
            gpu.chain(prm, gpu.chain_tracks([dev(s_) for s_ in srcs], [dev(s_) for s_ in l2s], dd))
            for i in range(ntr):
                want = np.zeros((dh, orow), np.uint8)
                assert orc.orc_chain(P(srcs[i]), irow, sw, sh, P(l2s[i]), orow, P(want), orow, dw, dh, 1, interp | PIXBUF, blur, 77, P(lut)) == 0
                assert_same(host(dd[i]), want, dw, dh, 4, "th=%d %dx%d blur=%d track %d" % (th, sw, sh, blur, i))


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
