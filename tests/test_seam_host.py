"""tools/seam_host.c: the C render host of bench.py's seam_chain leg -- one host thread per track calling convert_layer_palette -> resize_layer -> the "chroma blend"
process_func -> gamma_convert_layer by the reference's names (liblivesgpu_dropin.so, livesgpu_fx.so under tools/miniweed.c as the weed host), one
lives_gpu_layers_flush per tick.  Here on small frames against the oracle's chain, with the launch counters: every tick is ONE launch of the fused kernel."""
import ctypes
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import dev, frame, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = po.P
vp, ci = ctypes.c_void_p, ctypes.c_int


def test_seam_host_exports_load_without_a_device():
    so = os.path.join(ROOT, "tools", "libseam_host.so")
    assert os.path.exists(so), "tools/libseam_host.so is missing: run __graft_entry__.build()"
    H = ctypes.CDLL(so)
    for n in ("seam_host_init", "seam_host_run", "seam_host_release", "seam_host_layer_info"):
        assert hasattr(H, n)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 0])
def test_ticks_through_the_reference_names_equal_the_oracle_chain(gpu, orc, threads):
    import torch
    from lives_amd import lib
    L = lib.load()
    Hs = ctypes.CDLL(os.path.join(ROOT, "tools", "libseam_host.so"))
    Hs.seam_host_run.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(vp), ctypes.POINTER(ci)]
    Hs.seam_host_layer_info.argtypes = [ci, ctypes.POINTER(ci)]
    assert Hs.seam_host_init(os.path.join(ROOT, "lives_amd", "livesgpu_fx.so").encode()) == 0
    L.lives_gpu_deferred_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    L.lives_gpu_deferred_stats.restype = None
    rng = np.random.default_rng(0x5EA0)
    T, sw, sh, dw, dh, ticks, warm = 5, 256, 144, 128, 72, 7, 2
    srcs = [frame(rng, sw, sh, 4, alpha_mix=True) for _ in range(T)]
    l2s = [frame(rng, dw, dh, 4, alpha_mix=True) for _ in range(T)]
    d_src, d_l2 = [dev(a) for a in srcs], [dev(a) for a in l2s]
    torch.cuda.synchronize()
    sp, lp = (vp * T)(*[t.data_ptr() for t in d_src]), (vp * T)(*[t.data_ptr() for t in d_l2])
    st0, st1 = (ctypes.c_ulonglong * 4)(), (ctypes.c_ulonglong * 4)()
    ms, outs, orow = ctypes.c_double(), (vp * T)(), ci()
    L.lives_gpu_deferred_stats(st0)
    try:
        assert Hs.seam_host_run(T, sw, sh, dw, dh, sp, lp, 97, 2, ticks, warm, threads, ctypes.byref(ms), outs, ctypes.byref(orow)) == 0
        L.lives_gpu_deferred_stats(st1)
        assert (st1[1] - st0[1], st1[2] - st0[2], st1[3] - st0[3]) == (ticks + warm, (ticks + warm) * T, 0), "one fused launch per tick, every track in it"
        assert st1[0] - st0[0] == (ticks + warm) * T * 4, "four recorded calls per track and tick"
        lut = np.zeros(256, np.uint8)
        assert orc.orc_gamma_lut8(1.0, 1, 2, 1.4, P(lut)) == 1
        info = (ci * 4)()
        for t in range(T):
            assert Hs.seam_host_layer_info(t, info) == 0 and list(info) == [3, dw, dh, 2]       # RGBA32, the target size, tagged BT709
            want = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_chain(P(srcs[t]), sw * 4, sw, sh, P(l2s[t]), dw * 4, P(want), dw * 4, dw, dh, 1, 3 | 0x100, 0, 97, P(lut)) == 0
            got = torch.zeros((dh, orow.value), dtype=torch.uint8, device="cuda")
            lib.call("lgpu_copy_rows", got.data_ptr(), orow.value, outs[t], orow.value, dw * 4, dh, None)
            torch.cuda.synchronize()
            assert (host(got)[:, :dw * 4] == want).all(), t
        for t in range(T):
            assert (host(d_src[t]) == srcs[t]).all(), "the frames in HBM are read, never written"
    finally:
        Hs.seam_host_release()
        from tests import weedhost          # the layer seam goes back to the reference's libweed for the modules that follow
        if po.have_ref():
            weedhost.bind(L)
