"""lgpu_fx_batch: ONE launch for the instances of one filter on the live tracks of a tick (src/effects-weed.c:1850-2425 runs weed_apply_instance once per track).
Every frame of a batch against the oracle (softlight.c:62-141, multi_transitions.c:86-233, src/colourspace.c:8305-8620) and against the single-frame entry point;
slot order shuffled; guard bytes around every output."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, dev, host

pytestmark = pytest.mark.gpu
P = po.P


def guarded(rows, stride, fill=0xA5):
    return dev(np.full((rows + 1, stride), fill, np.uint8))


@pytest.mark.parametrize("palette,w,h", [(512, 128, 64), (545, 96, 40), (522, 130, 33), (544, 64, 36)])
def test_softlight_batch(gpu, orc, palette, w, h):
    rng = np.random.default_rng(0xB001 + palette)
    nplanes = 4 if palette == 545 else 3
    cw = w >> 1 if palette in (512, 522) else w
    ch = h >> 1 if palette == 512 else h
    dims = [(h, w)] + [(ch, cw)] * 2 + ([(h, w)] if nplanes == 4 else [])
    for n in (1, 3, 16):
        srcs = [[rng.integers(0, 256, (r, align(c, 16)), dtype=np.uint8) for (r, c) in dims] for _ in range(n)]
        d_srcs = [[dev(p_) for p_ in fr] for fr in srcs]
        d_outs = [[guarded(r, align(c, 16)) for (r, c) in dims] for _ in range(n)]
        order = list(rng.permutation(n))
        gpu.fx_batch(gpu.FX_SOFTLIGHT, [d_srcs[i] for i in order], [d_outs[i] for i in order], w, h, palette=palette, ip=(0,))
        for f in range(n):
            want = np.zeros((h, align(w, 16)), np.uint8)
            orc.orc_softlight_y(P(srcs[f][0]), srcs[f][0].strides[0], P(want), want.strides[0], w, h, 0)
            got = [host(t) for t in d_outs[f]]
            assert (got[0][:h, :w] == want[:h, :w]).all(), "luma of frame %d of %d" % (f, n)
            for k in range(1, nplanes):
                r, c = dims[k]
                assert (got[k][:r, :c] == srcs[f][k][:r, :c]).all(), "copied plane %d of frame %d" % (k, f)
            for k in range(nplanes):
                r, c = dims[k]
                assert (got[k][r] == 0xA5).all() and (got[k][:r, c:] == 0xA5).all(), "guard bytes of plane %d, frame %d" % (k, f)
            # the single-frame entry point says the same
            single = [guarded(r, align(c, 16)) for (r, c) in dims]
            gpu.softlight(d_srcs[f], single, w, h, palette, 0)
            for k in range(nplanes):
                assert (host(single[k]) == got[k]).all()


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("psize", [3, 4])
def test_transition_batch(gpu, orc, kind, psize):
    rng = np.random.default_rng(0xB100 + kind * 8 + psize)
    w, h = 150, 70
    for n, amt in ((1, 0.3), (4, 0.5), (16, 0.85)):
        stride = align(w * psize, 16)
        s1 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
        s2 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
        d1, d2 = [dev(a) for a in s1], [dev(a) for a in s2]
        outs = [guarded(h, stride) for _ in range(n)]
        order = list(rng.permutation(n))
        gpu.fx_batch(gpu.FX_TRANSITION, [[d1[i]] for i in order], [[outs[i]] for i in order], w, h, ins1=[[d2[i]] for i in order], ip=(kind, psize), dp=(amt,))
        for f in range(n):
            want = np.full((h, stride), 0xA5, np.uint8)
            orc.orc_transition(kind, P(s1[f]), stride, P(s2[f]), stride, P(want), stride, w, h, psize, amt)
            got = host(outs[f])
            assert (got[:h, :w * psize] == want[:, :w * psize]).all(), "frame %d of %d" % (f, n)
            assert (got[h] == 0xA5).all() and (got[:h, w * psize:] == 0xA5).all()
    # an amount per frame (frame_dp0): what the plugin's batch hook passes for the instances of a plan step
    n = 9
    amts = [0., 1.] + list(rng.random(n - 2))
    stride = align(w * psize, 16)
    s1 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
    s2 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
    outs = [guarded(h, stride) for _ in range(n)]
    gpu.fx_batch(gpu.FX_TRANSITION, [[dev(a)] for a in s1], [[o] for o in outs], w, h, ins1=[[dev(a)] for a in s2], ip=(kind, psize), dp=(0.123,), frame_dp0=amts)
    for f in range(n):
        want = np.full((h, stride), 0xA5, np.uint8)
        orc.orc_transition(kind, P(s1[f]), stride, P(s2[f]), stride, P(want), stride, w, h, psize, amts[f])
        assert (host(outs[f])[:h] == want).all(), "frame %d, amount %r" % (f, amts[f])


@pytest.mark.parametrize("fam", ["chroma3", "chroma4", "luma", "multi"])
def test_blend_batches(gpu, orc, fam):
    """the two-input blends of simple_blend.c / multi_blends.c, n frames with an amount each in one launch (k_pixel2's frame table), out of place and in place"""
    rng = np.random.default_rng(0xB300 + len(fam))
    w, h = 133, 37
    cases = {"chroma3": [(gpu.FX_BLEND_CHROMA, 3, (3, 0))], "chroma4": [(gpu.FX_BLEND_CHROMA, 4, (4, 0))],
             "luma": [(gpu.FX_BLEND_LUMA, ps, (t, ps, o)) for t in (1, 2, 3, 4) for ps, o in ((3, 0), (4, 1))],
             "multi": [(gpu.FX_BLEND_MULTI, 3, (t, t & 1)) for t in range(7)]}[fam]
    for op, ps, ip in cases:
        for n, inplace in ((1, False), (5, False), (16, True)):
            stride = align(w * ps, 16)
            s1 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
            s2 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
            if ps == 4:
                for a in s2:
                    a[:, 3::4][rng.random((h, stride // 4)) < 0.5] = 255
            amts = [int(v) for v in rng.integers(0, 256, n)]
            amts[0] = 255
            pre = [rng.integers(0, 256, (h + 1, stride), dtype=np.uint8) for _ in range(n)]      # the chroma blend leaves the destination's alpha byte as it is
            d1, d2 = [dev(a) for a in s1], [dev(a) for a in s2]
            outs = d1 if inplace else [dev(a) for a in pre]
            gpu.fx_batch(op, [[t] for t in d1], [[t] for t in outs], w, h, ins1=[[t] for t in d2], ip=ip, dp=(77.,), frame_dp0=amts)
            for f in range(n):
                want = s1[f].copy() if inplace else pre[f][:h].copy()
                src1 = want if inplace else s1[f]
                if op == gpu.FX_BLEND_CHROMA:
                    orc.orc_blend_chroma(P(src1), stride, P(s2[f]), stride, P(want), stride, w, h, ps, 0, amts[f])
                elif op == gpu.FX_BLEND_LUMA:
                    orc.orc_blend_luma(ip[0], P(src1), stride, P(s2[f]), stride, P(want), stride, w, h, ps, ip[2], amts[f], int(inplace))
                else:
                    orc.orc_blend_multi(ip[0], P(src1), stride, P(s2[f]), stride, P(want), stride, w, h, ip[1], amts[f])
                got = host(outs[f])
                assert (got[:h, :w * ps] == want[:, :w * ps]).all(), (fam, ip, n, inplace, f)
                if not inplace:
                    assert (got[h] == pre[f][h]).all() and (got[:h, w * ps:] == pre[f][:h, w * ps:]).all()


@pytest.mark.parametrize("order,oa", [(0, 0), (0, 1), (1, 1), (2, 0)])
def test_yuv411_to_rgb_batch(gpu, orc, order, oa):
    rng = np.random.default_rng(0xB200 + order * 2 + oa)
    wm, h = 40, 24
    ps = 4 if (order == 2 or oa) else 3
    for n in (1, 5, 16):
        srcs = [rng.integers(16, 236, (h, wm * 6), dtype=np.uint8) for _ in range(n)]
        d_srcs = [dev(a) for a in srcs]
        orow = align(wm * 4 * ps, 16)
        pre = [rng.integers(0, 256, (h + 1, orow), dtype=np.uint8) for _ in range(n)]        # the reference leaves some alpha bytes as they were (quirk K3b): the outputs start from known bytes
        outs = [dev(a) for a in pre]
        perm = list(rng.permutation(n))
        gpu.fx_batch(gpu.FX_YUV411_TO_RGB, [[d_srcs[i]] for i in perm], [[outs[i]] for i in perm], wm, h, ip=(order, oa, 0))
        for f in range(n):
            want = pre[f].copy()
            assert orc.orc_yuv411_to_rgb(P(srcs[f]), wm, h, P(want), want.strides[0], order, oa, 0) == 0
            single = dev(pre[f])
            gpu.yuv411_to_rgb(d_srcs[f], single, wm, h, out_order=order, out_alpha=oa, unclamped=0)
            got = host(outs[f])
            assert (got == host(single)).all(), "batch frame %d of %d differs from the single call" % (f, n)
            assert (got[h] == pre[f][h]).all() and (got[:h, wm * 4 * ps:] == pre[f][:h, wm * 4 * ps:]).all()


def test_fx_batch_refuses_bad_arguments(gpu):
    from lives_amd import lib
    t = dev(np.zeros((8, 64), np.uint8))
    with pytest.raises(lib.LgpuError):
        gpu.fx_batch(99, [[t]], [[t]], 8, 8)
    with pytest.raises(lib.LgpuError):
        gpu.fx_batch(gpu.FX_TRANSITION, [[t]] * 17, [[t]] * 17, 8, 8, ins1=[[t]] * 17, ip=(0, 4), dp=(0.5,))
    with pytest.raises(lib.LgpuError):
        gpu.fx_batch(gpu.FX_TRANSITION, [[t]], [[t]], 8, 8, ins1=[[t]], ip=(2, 4), dp=(0.5,))          # 4 way split in place
    with pytest.raises(lib.LgpuError):
        gpu.fx_batch(gpu.FX_SOFTLIGHT, [[t, t, t]], [[t, t, t]], 8, 8, palette=544, frame_dp0=[0.5])   # a value per frame: transitions only
