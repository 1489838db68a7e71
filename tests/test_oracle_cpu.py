"""CPU: (1) the oracle against the LIVE reference builds under oracle/_ref (randomised, skipped where the
reference was never built), (2) self-consistency of the oracle's own-spec parts (resize, gauss5, chain)."""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po

P = po.P
needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
def test_live_reference_plugins(orc):
    H = po.RefHost()
    rng = np.random.default_rng(42)
    pals = {1: (3, 0, 0), 2: (3, 1, 0), 3: (4, 0, 0), 4: (4, 1, 0), 5: (4, 2, 1)}
    for (w, h) in ((66, 34), (31, 9)):
        for pal, (ps, order, af) in pals.items():
            for bf in (0, 77, 255):
                s1 = po.make_frame(rng, w, h, ps, extra_rows=1, alpha_mix=True)
                s2 = po.make_frame(rng, w, h, ps, extra_rows=1, alpha_mix=True, pad_px=1)
                d = s1.copy()
                H.run(po.refplugin("simple_blend"), "chroma blend", pal, w, h, [s1, s2], d, [po.p_int(bf)])
                m = s1.copy()
                orc.orc_blend_chroma(P(s1), s1.strides[0], P(s2), s2.strides[0], P(m), m.strides[0], w, h, ps, af, bf)
                assert (d[:h, :w * ps] == m[:h, :w * ps]).all(), ("chroma", pal, bf)
        for t, fn in enumerate(("blend_multiply", "blend_screen", "blend_darken", "blend_lighten", "blend_overlay", "blend_dodge", "blend_burn")):
            s1, s2 = po.make_frame(rng, w, h, 3), po.make_frame(rng, w, h, 3)
            d, m = np.zeros_like(s1), np.zeros_like(s1)
            H.run(po.refplugin("multi_blends"), fn, 2, w, h, [s1, s2], d, [po.p_int(99)])
            orc.orc_blend_multi(t, P(s1), s1.strides[0], P(s2), s2.strides[0], P(m), m.strides[0], w, h, 1, 99)
            assert (d[:h, :w * 3] == m[:h, :w * 3]).all(), fn


@needs_ref
def test_reference_slicing_protocol_gives_same_pixels(orc):
    """process_func_threaded-style row slices (offset / height[2] / pre-offset pixel_data) == one call"""
    H = po.RefHost()
    rng = np.random.default_rng(43)
    w, h = 40, 24
    s1, s2 = po.make_frame(rng, w, h, 4, extra_rows=1, alpha_mix=True), po.make_frame(rng, w, h, 4, extra_rows=1, alpha_mix=True)
    a, b = s1.copy(), s1.copy()
    H.run(po.refplugin("simple_blend"), "chroma blend", 3, w, h, [s1, s2], a, [po.p_int(140)], nslices=1)
    H.run(po.refplugin("simple_blend"), "chroma blend", 3, w, h, [s1, s2], b, [po.p_int(140)], nslices=3)
    assert (a == b).all()


@needs_ref
def test_live_reference_k2(orc):
    R = po.csref()
    rng = np.random.default_rng(44)
    for (w, h, ys, cs) in ((64, 32, 64, 32), (66, 34, 96, 48), (130, 18, 160, 80)):
        for which in (0, 2):
            for quality in (1, 2, 3):
                Y = rng.integers(0, 256, (h, ys), dtype=np.uint8)
                U = rng.integers(0, 256, (h // 2 * cs + 1,), dtype=np.uint8)
                V = rng.integers(0, 256, (h // 2 * cs + 1,), dtype=np.uint8)
                U[-1], V[-1] = U[-2], V[-2]
                orow = po.align(w * 4)
                R.csref_set_prefs(quality, 1, 1.4)
                ref = np.zeros((h + 1, orow), np.uint8)
                strides = (ctypes.c_int * 3)(ys, cs, cs)
                R.csref_yuv420p_to_rgb(P(Y), P(U), P(V), w, h, strides, orow, P(ref), 1, 0, which & 1, 2 if which & 2 else 1, None)
                got = np.zeros((h, orow), np.uint8)
                orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, h // 2 * cs, h // 2 * cs, P(got), orow, w, h, 4, 0, 0, which, quality, None, 0)
                diff = (got[:, :w * 4].reshape(h, w, 4) != ref[:h, :w * 4].reshape(h, w, 4)).any(axis=2)
                diff[0, 1::2] = False
                diff[h - 1, 1::2] = False
                assert not diff.any(), (w, h, which, quality, np.argwhere(diff)[:3])
    R.csref_set_prefs(2, 1, 1.4)


# ---- own-spec parts -----------------------------------------------------------------------------------------------
def test_threaded_chain_equals_serial(orc):
    rng = np.random.default_rng(3)
    for (sw, sh, dw, dh) in [(128, 64, 64, 32), (200, 120, 66, 34), (384, 216, 192, 108)]:
        for blur in (0, 1):
            for nt in (1, 2, 3, 8):
                src = rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8)
                l2 = rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8)
                lut = rng.integers(0, 256, 256, dtype=np.uint8)
                a, b = np.zeros((dh, dw * 4), np.uint8), np.zeros((dh, dw * 4), np.uint8)
                assert orc.orc_chain(P(src), sw * 4, sw, sh, P(l2), dw * 4, P(a), dw * 4, dw, dh, 1, 3, blur, 100, P(lut)) == 0
                assert orc.orc_chain_threaded(P(src), sw * 4, sw, sh, P(l2), dw * 4, P(b), dw * 4, dw, dh, 1, 3, blur, 100, P(lut), nt) == 0
                assert (a == b).all(), (sw, sh, blur, nt)
                # the same on the pinned resize arithmetic (LGPU_INTERP_PIXBUF = 0x100): what bench.py's cpu_baseline times by default
                src[:, 3::4][rng.random((sh, sw)) < 0.4] = 255
                assert orc.orc_chain(P(src), sw * 4, sw, sh, P(l2), dw * 4, P(a), dw * 4, dw, dh, 1, 3 | 0x100, blur, 100, P(lut)) == 0
                assert orc.orc_chain_threaded(P(src), sw * 4, sw, sh, P(l2), dw * 4, P(b), dw * 4, dw, dh, 1, 3 | 0x100, blur, 100, P(lut), nt) == 0
                assert (a == b).all(), ("pixbuf", sw, sh, blur, nt)


def test_resize_spec_properties(orc):
    """size-independent properties of lgpu-polyphase-v1: flat images stay flat (taps sum to 1.0), identity is exact,
    channels are independent, horizontal and vertical axes behave alike"""
    rng = np.random.default_rng(5)
    for (sw, sh, dw, dh, interp) in [(128, 64, 64, 32, 3), (64, 32, 128, 64, 3), (100, 60, 37, 23, 2), (50, 50, 50, 50, 3)]:
        for val in (0, 1, 127, 255):
            src = np.full((sh, sw * 4), val, np.uint8)
            dst = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_resize(P(src), sw * 4, sw, sh, P(dst), dw * 4, dw, dh, 4, interp) == 0
            assert (dst == val).all(), (sw, sh, dw, dh, val)
    src = rng.integers(0, 256, (40, 56 * 3), dtype=np.uint8)
    dst = np.zeros_like(src)
    orc.orc_resize(P(src), 56 * 3, 56, 40, P(dst), 56 * 3, 56, 40, 3, 3)
    assert (dst == src).all(), "same-size resize must be the identity"
    # transposing input transposes output (single channel)
    a = rng.integers(0, 256, (48, 80), dtype=np.uint8)
    o1 = np.zeros((24, 40), np.uint8)
    orc.orc_resize(P(a), 80, 80, 48, P(o1), 40, 40, 24, 1, 3)
    at = np.ascontiguousarray(a.T)
    o2 = np.zeros((40, 24), np.uint8)
    orc.orc_resize(P(at), 48, 48, 80, P(o2), 24, 24, 40, 1, 3)
    assert np.abs(o1.astype(int) - o2.T.astype(int)).max() <= 1     # pass order differs -> at most one rounding step


def test_gauss5_spec_properties(orc):
    rng = np.random.default_rng(6)
    for val in (0, 3, 255):
        src = np.full((20, 30 * 4), val, np.uint8)
        dst = np.zeros_like(src)
        orc.orc_gauss5(P(src), 120, P(dst), 120, 30, 20, 4)
        assert (dst == val).all()
    # impulse response = outer([1 4 6 4 1], [1 4 6 4 1]) / 256 rounded
    src = np.zeros((11, 11), np.uint8)
    src[5, 5] = 255
    dst = np.zeros_like(src)
    orc.orc_gauss5(P(src), 11, P(dst), 11, 11, 11, 1)
    k = np.array([1, 4, 6, 4, 1])
    want = (np.outer(k, k) * 255 + 128) >> 8
    assert (dst[3:8, 3:8] == want).all() and dst.sum() == want.sum()


def test_spc_rnd_high_equals_med_once_clamped():
    """_spc_rnd (src/colourspace.c:832-835): pb_quality HIGH returns (int)((float)val / 65536.), everything else val >> 16.
    After any of the clamps that follow it (0..255, 16..235, 16..240) the two agree for every 32-bit sum the tables can make."""
    for lo in range(-2 ** 26, 2 ** 26, 2 ** 22):
        v = np.arange(lo, lo + 2 ** 22, dtype=np.int64)
        med = v >> 16
        hi = np.trunc(v.astype(np.float32).astype(np.float64) / 65536.0).astype(np.int64)
        for (a, b) in ((0, 255), (16, 235), (16, 240)):
            assert np.array_equal(np.clip(med, a, b), np.clip(hi, a, b))


@needs_ref
def test_reference_slices_agree_between_pb_quality_high_and_med():
    R = po.csref()
    P = po.P

    def run(pbq):
        R.csref_set_prefs(pbq, 1, 1.4)
        rng = np.random.default_rng(31)
        outs = []
        w, h = 64, 16
        for in_order in (0, 1):
            for out_fmt in range(6):
                for which in ((0, 1, 2, 3) if out_fmt >= 4 else (0, 1)):
                    src = rng.integers(0, 256, (h, w * 3), dtype=np.uint8)
                    src[0, :6] = [0, 0, 0, 255, 255, 255]
                    out, _ = po.k4_out_planes(0x5A, w, h, out_fmt, 0)
                    op, os_ = po.planes_args(out)
                    assert R.csref_k4(in_order, 0, out_fmt, 0, P(src), src.strides[0], w, h, ctypes.addressof(op), ctypes.addressof(os_), which & 1, which >> 1) == 0
                    outs += [a.copy() for a in out]
        for which in range(4):
            planes = [rng.integers(0, 256, (h + 1, w * 3), dtype=np.uint8)[:h]]
            sp, ss = po.planes_args(planes)
            dst = np.zeros((h, w * 4), np.uint8)
            assert R.csref_k3(0, 0, 0, 1, ctypes.addressof(sp), ctypes.addressof(ss), w, h, P(dst), dst.strides[0], which & 1, which >> 1) == 0
            outs.append(dst)
            y = rng.integers(0, 256, (h, w), dtype=np.uint8)
            u, v = (rng.integers(0, 256, (h // 2 + 1, w // 2), dtype=np.uint8)[:h // 2] for _ in range(2))
            ist = (ctypes.c_int * 3)(w, w // 2, w // 2)
            dst = np.zeros((h, w * 4), np.uint8)
            R.csref_yuv420p_to_rgb(P(y), P(u), P(v), w, h, ist, dst.strides[0], P(dst), 1, 0, which & 1, 2 if which & 2 else 1, None)
            outs.append(dst)
        return outs

    med, high = run(2), run(3)
    R.csref_set_prefs(2, 1, 1.4)
    assert len(med) == len(high) and all(np.array_equal(a, b) for a, b in zip(med, high))
