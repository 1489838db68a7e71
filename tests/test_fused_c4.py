"""BASELINE config 4 in one launch: lgpu_gauss5_colorkey == the oracle's gaussian followed by its colour key (colorkey.script's arithmetic; the 4-byte form is
the repo's extension), bit for bit -- small frames with every border case, and the 3840x2160 frames of the config."""
import numpy as np
import pytest

from oracle import pyoracle as po
from util import align, dev, host

P = po.P
pytestmark = pytest.mark.gpu


def want_c4(orc, a, b, w, h, ps, is_bgr, delta, opac, col):
    bl = np.zeros_like(a)
    orc.orc_gauss5(P(a), a.strides[0], P(bl), bl.strides[0], w, h, ps)
    out = np.zeros_like(a)
    if ps == 3:
        orc.orc_colorkey(P(bl), bl.strides[0], P(b), b.strides[0], P(out), out.strides[0], w, h, is_bgr, delta, opac, col[0], col[1], col[2], 0)
    else:
        orc.orc_colorkey4(P(bl), bl.strides[0], P(b), b.strides[0], P(out), out.strides[0], w, h, is_bgr, delta, opac, col[0], col[1], col[2])
    return out


@pytest.mark.parametrize("ps", [3, 4])
def test_gauss5_colorkey_one_launch(gpu, orc, ps):
    rng = np.random.default_rng(0xC4 + ps)
    for (w, h, is_bgr, delta, opac, col) in [(256, 40, 0, 0.3, 0.8, (128, 128, 128)), (248, 33, 1, 0.5, 0.35, (120, 140, 100)), (4, 3, 0, 1.0, 1.0, (128, 128, 128)),
                                              (252, 5, 0, 0.2, 0.5, (130, 125, 128)), (1000, 70, 1, 0.4, 0.66, (128, 120, 135)), (3840, 24, 0, 0.3, 0.9, (128, 128, 128))]:
        rs = align(w * ps, 16)
        a = rng.integers(0, 256, (h, rs), dtype=np.uint8)
        b = rng.integers(0, 256, (h, rs), dtype=np.uint8)
        a[:, : (w // 3) * ps] = rng.integers(96, 160, (h, (w // 3) * ps), dtype=np.uint8)       # a region that blurs into the key box
        want = want_c4(orc, a, b, w, h, ps, is_bgr, delta, opac, col)
        d = dev(np.full_like(a, 0x5A))
        gpu.gauss5_colorkey(dev(a), dev(b), d, w, h, ps, is_bgr, delta, opac, col)
        got = host(d)
        bad = np.argwhere(got[:, :w * ps] != want[:, :w * ps])
        assert len(bad) == 0, "%dx%d ps %d: %d bytes differ, first %s" % (w, h, ps, len(bad), bad[0].tolist())
        assert (got[:, w * ps:] == 0x5A).all(), "row padding written"
        keyed = int((want[:, :w * ps] != orc_blur(orc, a, w, h, ps)[:, :w * ps]).sum())
        assert keyed > 0 or w < 8, "the test frame never matched the key"


def orc_blur(orc, a, w, h, ps):
    bl = np.zeros_like(a)
    orc.orc_gauss5(P(a), a.strides[0], P(bl), bl.strides[0], w, h, ps)
    return bl


@pytest.mark.parametrize("ps", [3, 4])
def test_c4_at_size(gpu, orc, ps):
    rng = np.random.default_rng(0xC40 + ps)
    w, h = 3840, 2160
    a = rng.integers(0, 256, (h, w * ps), dtype=np.uint8)
    b = rng.integers(0, 256, (h, w * ps), dtype=np.uint8)
    a[:, :1200 * ps] = rng.integers(100, 156, (h, 1200 * ps), dtype=np.uint8)
    want = want_c4(orc, a, b, w, h, ps, 0, 0.3, 0.8, (128, 128, 128))
    d = dev(np.zeros_like(a))
    gpu.gauss5_colorkey(dev(a), dev(b), d, w, h, ps, 0, 0.3, 0.8, (128, 128, 128))
    assert (host(d) == want).all()


def test_unaligned_frames_are_refused(gpu):
    from lives_amd import lib
    a = dev(np.zeros((8, 48), np.uint8))
    with pytest.raises(lib.LgpuError):
        gpu.gauss5_colorkey(a, a, dev(np.zeros((8, 48), np.uint8)), 10, 8, 4, 0, 0.3, 0.8, (1, 2, 3))      # width % 4 != 0


@pytest.mark.parametrize("ps", [3, 4])
def test_gauss5_colorkey_batch(gpu, orc, ps):
    """config 4 for the frames of several tracks in ONE launch (lgpu_fx_batch, LGPU_FX_GAUSS5_COLORKEY): every frame against the oracle, slots shuffled"""
    rng = np.random.default_rng(0xC40 + ps)
    w, h, is_bgr, delta, opac, col = 504, 37, 1, 0.4, 0.7, (128, 120, 135)
    rs = align(w * ps, 16)
    for n in (1, 3, 16):
        a = [rng.integers(64, 192, (h, rs), dtype=np.uint8) for _ in range(n)]
        b = [rng.integers(0, 256, (h, rs), dtype=np.uint8) for _ in range(n)]
        da, db = [dev(x) for x in a], [dev(x) for x in b]
        outs = [dev(np.full((h + 1, rs), 0x5A, np.uint8)) for _ in range(n)]
        order = list(rng.permutation(n))
        gpu.fx_batch(gpu.FX_GAUSS5_COLORKEY, [[da[i]] for i in order], [[outs[i]] for i in order], w, h, ins1=[[db[i]] for i in order],
                     ip=(ps, is_bgr, col[0] | (col[1] << 8) | (col[2] << 16)), dp=(delta, opac))
        for f in range(n):
            want = want_c4(orc, a[f], b[f], w, h, ps, is_bgr, delta, opac, col)
            got = host(outs[f])
            assert (got[:h, :w * ps] == want[:, :w * ps]).all(), "frame %d of %d" % (f, n)
            assert (got[h] == 0x5A).all() and (got[:h, w * ps:] == 0x5A).all()
