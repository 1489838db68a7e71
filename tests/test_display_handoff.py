"""the display / sink hand-off of load_frame_image() on device-resident layers.

Two sequences of the reference, each as ONE resident chain (pin -> seam calls -> sync), checked against the oracle's functions composed the same way,
with the PCIe byte counters (one upload, one download of exactly what the consumer reads):

* v1 playback plugin with a YUV palette -- the yuv4mpeg stream sink (lives-plugins/plugins/playback/video/yuv4mpeg_stream.c:77-95: YUV420P,
  clamped or unclamped; :166 y4m_write_frame of the three planes): src/player.c:1355-1356 rowstride_alignment_hint = -1,
  :1358-1369 convert_layer_palette_full(frame_layer, vpp->palette, vpp->YUV_clamping, vpp->YUV_sampling, vpp->YUV_subspace, WEED_GAMMA_UNKNOWN),
  :1372-1379 compact_rowstrides().  The bytes a y4m FRAME carries are Y, then U, then V, rows compact.
* screen display: src/player.c:1502-1508 gamma_convert_layer(WEED_GAMMA_MONITOR, frame_layer) on the RGB frame.
"""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, frame

needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (reference libweed) not built")
pytestmark = [needs_ref, pytest.mark.gpu]
P = po.P
RGB24, RGBA32, BGRA32, YUV420P = 1, 3, 4, 512


@pytest.fixture(scope="module")
def seam():
    from lives_amd import lib
    from tests import weedhost
    L = lib.load()
    weedhost.bind(L)
    L.lives_gpu_set_rowstride_alignment_hint.argtypes = [ctypes.c_int]
    return L, weedhost


def stats(L):
    a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    L.lives_gpu_transfer_stats(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def y4m_payload(planes, w, h):
    """what y4m_write_frame() puts after the FRAME line: the three planes, rows compact"""
    return np.concatenate([planes[0][:h, :w].ravel(), planes[1][:h >> 1, :w >> 1].ravel(), planes[2][:h >> 1, :w >> 1].ravel()])


@pytest.mark.parametrize("size", [(200, 72), (1920, 1080)])
@pytest.mark.parametrize("inpl,clamping", [(RGBA32, 0), (BGRA32, 1), (RGB24, 0)])
def test_frame_to_y4m_sink(seam, orc, size, inpl, clamping):
    L, wh = seam
    w, h = size
    rng = np.random.default_rng(w + inpl + clamping)
    ips = 3 if inpl == RGB24 else 4
    order = 1 if inpl == BGRA32 else 0
    src = frame(rng, w, h, ips)
    # what the reference computes: the conversion into COMPACT planes (hint -1), which compact_rowstrides() then leaves alone
    want, dims = po.k4_out_planes(0, w, h, 4, 0)
    wp, ws = po.planes_args(want)
    assert orc.orc_rgb_to_yuv(P(src), src.strides[0], w, h, order, int(ips == 4), ctypes.addressof(wp), ctypes.addressof(ws), 4, 0, 1 if clamping == 1 else 0) == 0
    for hint in (-1, 0):
        # hint 0: a v2 plugin / no hint -- 32-byte rows out of the conversion.  The reference's 4:2:0 walk is only sane on compact rows (quirk K4-c), so
        # the aligned variant is compared where both agree: widths whose aligned rows are compact anyway
        if hint == 0 and align(w) != w:
            continue
        lay = wh.new_layer(inpl, w, h, [src], gamma=1, flags=0)
        assert L.lives_gpu_layer_pin(lay) == 0
        h0, d0 = stats(L)
        L.lives_gpu_set_rowstride_alignment_hint(hint)                                    # player.c:1355-1356
        assert L.lives_gpu_convert_layer_palette_full(lay, YUV420P, clamping, 0, 1, 0) == 1   # :1364
        L.lives_gpu_set_rowstride_alignment_hint(0)                                       # :1366 / :1421 restore
        assert L.lives_gpu_compact_rowstrides(lay) == 1                                   # :1376
        assert stats(L) == (h0, d0), "the hand-off chain of a pinned layer must not cross PCIe"
        assert L.lives_gpu_layer_sync(lay) == 0
        h1, d1 = stats(L)
        planes, _, rs = wh.planes_of(lay)
        assert rs == [w, w >> 1, w >> 1]
        assert (wh.geti(lay, "current_palette"), wh.geti(lay, "YUV_clamping"), wh.geti(lay, "width"), wh.geti(lay, "height")) == (YUV420P, clamping, w, h)
        assert d1 - d0 == w * h * 3 // 2 and h1 == h0                                     # the sink's bytes, nothing else
        assert (y4m_payload(planes, w, h) == y4m_payload(want, w, h)).all(), (size, inpl, clamping, hint)
        assert L.lives_gpu_layer_unpin(lay) == 0


@pytest.mark.parametrize("size", [(200, 72), (1920, 1080)])
@pytest.mark.parametrize("pal", [RGB24, RGBA32, BGRA32])
def test_frame_to_screen_gamma(seam, orc, size, pal):
    L, wh = seam
    w, h = size
    rng = np.random.default_rng(w + pal)
    ps = 3 if pal == RGB24 else 4
    src = frame(rng, w, h, ps)
    lut = np.zeros(256, np.uint8)
    assert orc.orc_gamma_lut8(1.0, po.GAMMA_SRGB, po.GAMMA_MONITOR, 1.4, P(lut)) == 1
    want = src.copy()
    orc.orc_gamma_apply(P(want), want.strides[0], w, h, ps, 0, P(lut))
    lay = wh.new_layer(pal, w, h, [src], gamma=1)
    assert L.lives_gpu_layer_pin(lay) == 0
    h0, d0 = stats(L)
    assert L.lives_gpu_gamma_convert_layer(po.GAMMA_MONITOR, lay) == 1                    # player.c:1508
    assert stats(L) == (h0, d0)
    assert L.lives_gpu_layer_sync(lay) == 0
    planes, _, rs = wh.planes_of(lay)
    assert wh.geti(lay, "gamma_type") == po.GAMMA_MONITOR and rs == [src.strides[0]]
    assert (planes[0][:, :w * ps] == want[:, :w * ps]).all()
    assert L.lives_gpu_layer_unpin(lay) == 0
