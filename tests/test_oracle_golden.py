"""CPU: the oracle (oracle/lives_oracle.c) against the committed reference-generated fixtures (tests/golden).

This is what pins the oracle: every fixture holds inputs and the bytes the REFERENCE's own code produced
for them (oracle/ref/gen_golden.py).  Runs anywhere (no /root/reference, no GPU).
"""
import ctypes

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import golden_util as gu

P = po.P


def test_conversion_tables(orc):
    g = gu.load("tables.npz")
    for which in range(4):
        a = np.zeros((9, 256), np.int32)
        b = np.zeros((5, 256), np.int32)
        orc.orc_tables(which, P(a), P(b))
        assert (a == g["rgb2yuv_%d" % which]).all(), which
        assert (b == g["yuv2rgb_%d" % which]).all(), which


def test_alpha_tables(orc):
    g = gu.load("tables.npz")
    orc.orc_unal.restype = ctypes.c_int
    orc.orc_al.restype = ctypes.c_int
    un = np.array([[orc.orc_unal(i, j) for j in range(256)] for i in range(256)], np.uint8)
    al = np.array([[orc.orc_al(i, j) for j in range(256)] for i in range(256)], np.uint8)
    assert (un == g["unal"]).all()
    assert (al == g["al"]).all()


def test_premult_yuv_tables(orc):
    """init_unal's clamped-YUV tables (src/colourspace.c:1141-1160), as the reference slice produced them"""
    g = gu.load("premult_yuv.npz")
    tabs = [np.zeros((256, 256), np.uint8) for _ in range(4)]
    orc.orc_premult_yuv_tables(*[P(t) for t in tabs])
    for name, t in zip(("unalcy", "alcy", "unalcuv", "alcuv"), tabs):
        assert (t == g[name]).all(), name


def test_yuv411_to_rgb(orc):
    """convert_yuv411_to_rgb_frame / _bgr_frame / _argb_frame (src/colourspace.c:8305-8620)"""
    g = gu.load("yuv411.npz")
    for n, (wm, h, order, oa, uncl, _pad) in enumerate(g["cases"].tolist()):
        got = g["init%d" % n].copy()
        assert orc.orc_yuv411_to_rgb(P(np.ascontiguousarray(g["src%d" % n])), wm, h, P(got), got.strides[0], order, oa, uncl) == 0
        assert (got == g["out%d" % n]).all(), (n, wm, h, order, oa, uncl)      # includes the alpha bytes the reference never writes


def test_rgb_to_yuv411(orc):
    """convert_rgb_to_yuv411_frame / _bgr_ / _argb_ (src/colourspace.c:6499-6615)"""
    g = gu.load("rgb_to_yuv411.npz")
    for n, (w, h, order, ia, uncl, _pad) in enumerate(g["cases"].tolist()):
        src = np.ascontiguousarray(g["src%d" % n])
        got = np.full_like(g["out%d" % n], 0xA5)
        assert orc.orc_rgb_to_yuv411(P(src), src.strides[0], w, h, order, ia, P(got), uncl) == 0
        assert (got == g["out%d" % n]).all(), (n, w, h, order, ia, uncl)


def test_gamma_luts(orc):
    g = gu.load("luts.npz")
    n = 0
    for key in g.files:
        parts = key.split("_")
        if key.startswith("lut8v_"):
            fileg, gfrom, gto = float(parts[1].replace("p", ".")), int(parts[2]), 2048
        else:
            fileg, gfrom, gto = 1.0, int(parts[1]), int(parts[2])
        want_ok, want = int(g[key][0]), g[key][1:]
        lut = np.zeros(256, np.uint8)
        ok = orc.orc_gamma_lut8(fileg, gfrom, gto, 1.4, P(lut))
        assert ok == want_ok, key
        if ok:
            assert (lut == want).all(), key
        n += 1
    assert n == 18


def test_k1_swizzles(orc):
    g = gu.load("k1_swizzle.npz")
    w, h = map(int, g["geom"])
    lut = g["lut"]
    for rec in g["records"]:
        name, lutflag, _ = rec.split("_")
        op = po.OPS.index(name)
        src, want = g[rec + "_in"], g[rec + "_out"]
        ob = po.OP_OBPP[op]
        got = np.zeros((h, po.align(w * ob)), np.uint8)
        assert orc.orc_swizzle(op, 0, P(src), src.strides[0], P(got), got.strides[0], w, h, P(lut) if lutflag == "lut1" else None) == 0
        assert (got[:, :w * ob] == want).all(), rec


def test_k2_yuv420p(orc):
    g = gu.load("k2_yuv420p.npz")
    for rec in g["records"]:
        w, h, ys, cs, which, opsize, quality, is422, orow = map(int, g[rec + "_geom"])
        Y, U, V = g[rec + "_y"], g[rec + "_u"], g[rec + "_v"]
        chh = h if is422 else h // 2
        strides = (ctypes.c_int * 3)(ys, cs, cs)
        got = np.zeros((h, orow), np.uint8)
        assert orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, chh * cs, chh * cs, P(got), orow, w, h, opsize, 0, is422, which, quality, None, 0) == 0
        want = gu.k2_reference_pixels(g[rec + "_out"], w, h, which, opsize, is422, orow)
        diff = (got[:, :w * opsize].reshape(h, w, opsize) != want).any(axis=2) & ~gu.k2_mask(w, h, is422)
        if (which & 1) and not is422:
            # unclamped 4:2:0: neighbouring reference rows overwrite one byte of each other at the seam (row shift quirk)
            diff[:, w - 1] = False
            diff[1, 0] = False      # its first byte sits where row 0's last pixel is written later
            diff[0, :] &= False
            diff[h - 1, :] &= False
        assert not diff.any(), "%s: %d pixels differ, first %s" % (rec, diff.sum(), np.argwhere(diff)[0])


def test_k6_gamma_apply(orc):
    g = gu.load("k6_gamma_apply.npz")
    lut = g["lut"]
    for (psize, af) in ((3, 0), (4, 0), (4, 1)):
        pix = g["p%d_a%d_in" % (psize, af)].copy()
        orc.orc_gamma_apply(P(pix), pix.strides[0], 22, 10, psize, af, P(lut))
        assert (pix == g["p%d_a%d_out" % (psize, af)]).all()


PALS = {1: (3, 0, 0), 2: (3, 1, 0), 3: (4, 0, 0), 4: (4, 1, 0), 5: (4, 2, 1)}
LUMA = {"luma overlay": 1, "luma underlay": 2, "negative luma overlay": 3, "averaged luma overlay": 4}
MULTI = ["blend_multiply", "blend_screen", "blend_darken", "blend_lighten", "blend_overlay", "blend_dodge", "blend_burn"]


def run_oracle_plugin_record(orc, g, rec):
    """oracle result for one plugins.npz record -> (got, want, valid_bytes_per_row, rows)"""
    f = rec.split("|")
    if f[0] == "sb":
        fn, pal, prm = f[1], int(f[2]), int(f[3])
        ps, order, af = PALS[pal]
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        got = a.copy()
        if fn == "chroma blend":
            orc.orc_blend_chroma(P(a), a.strides[0], P(b), b.strides[0], P(got), got.strides[0], 18, 8, ps, af, prm)
        else:
            orc.orc_blend_luma(LUMA[fn], P(a), a.strides[0], P(b), b.strides[0], P(got), got.strides[0], 18, 8, ps, order, prm, 0)
        return got, want, 18 * ps, 8
    if f[0] == "mb":
        fn, pal, prm = f[1], int(f[2]), int(f[3])
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        got = np.zeros_like(a)
        orc.orc_blend_multi(MULTI.index(fn), P(a), a.strides[0], P(b), b.strides[0], P(got), got.strides[0], 18, 8, int(pal == 2), prm)
        return got, want, 18 * 3, 8
    if f[0] == "ck":
        pal, delta, opac = int(f[1]), float(f[2]), float(f[3])
        col = list(map(int, f[4].split(",")))
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        got = np.zeros_like(a)
        orc.orc_colorkey(P(a), a.strides[0], P(b), b.strides[0], P(got), got.strides[0], 18, 8, int(pal == 2), delta, opac, col[0], col[1], col[2], 0)
        return got, want, 18 * 3, 8
    fn, pal, mw, mh = f[1], int(f[2]), int(f[3]), int(f[4])
    ps = 3 if pal == 1 else 4
    a, want = g[rec + "|a"], g[rec + "|o"]
    got = a.copy()
    orc.orc_mirror(["mirrorx", "mirrory", "mirrorxy"].index(fn), P(got), got.strides[0], P(got), got.strides[0], mw, mh, ps)
    return got, want, mw * ps, mh


def test_weed_plugins(orc):
    g = gu.load("plugins.npz")
    n = 0
    for rec in g["records"]:
        got, want, nbytes, rows = run_oracle_plugin_record(orc, g, str(rec))
        assert (got[:rows, :nbytes] == want[:rows, :nbytes]).all(), rec
        n += 1
    assert n == 243


def stencil_dims(pal, w, h):
    cw = w >> 1 if pal in (512, 513, 522) else w
    ch = h >> 1 if pal in (512, 513) else h
    return [(w, h), (cw, ch), (cw, ch)] + ([(w, h)] if pal == 545 else [])


def test_stencil_plugins(orc):
    """softlight.c / edge.c fixtures (tests/golden/stencils.npz)"""
    g = gu.load("stencils.npz")
    n = 0
    for rec in map(str, g["records"]):
        f = rec.split("|")
        if f[0] == "sl":
            pal, w, h, uncl = map(int, f[1:])
            src, want = g[rec + "|i0"], g[rec + "|o0"]
            got = np.full_like(src, 0x5A)
            orc.orc_softlight_y(P(src), src.strides[0], P(got), got.strides[0], w, h, uncl)
            assert (got[:h, :w] == want[:h, :w]).all(), rec
            for i, (cw, ch) in enumerate(stencil_dims(pal, w, h)[1:], 1):
                assert (g[rec + "|o%d" % i][:ch, :cw] == g[rec + "|i%d" % i][:ch, :cw]).all(), rec
        else:
            pal, mode, inplace, w, h = map(int, f[1:])
            ps = 3 if pal <= 2 else 4
            a, d0, want = g[rec + "|a"], g[rec + "|d"], g[rec + "|o"]
            got = d0.copy()
            m16 = np.zeros(w * h, np.int16)
            orc.orc_edge(P(got) if inplace else P(a), a.strides[0], P(got), got.strides[0], w, h, pal, mode, P(m16), inplace)
            assert (got[:h, :w * ps] == want[:h, :w * ps]).all(), rec
        n += 1
    assert n == 50


def test_k34_palette_matrix(orc):
    """RGB <-> YUV conversions of the reference (tests/golden/k34_palette.npz)"""
    import ctypes
    g = gu.load("k34_palette.npz")
    n = 0
    for rec in map(str, g["records"]):
        f = rec.split("|")
        a = list(map(int, f[1:]))
        if f[0] == "k4":
            in_order, in_alpha, out_fmt, out_alpha, which, w, h = a
            src = g[rec + "|in"]
            got, dims = po.k4_out_planes(0x5A, w, h, out_fmt, out_alpha)
            gp, gs = po.planes_args(got)
            assert orc.orc_rgb_to_yuv(P(src), src.strides[0], w, h, in_order, in_alpha, ctypes.addressof(gp), ctypes.addressof(gs), out_fmt,
                                      out_alpha, which) == 0
            for i in range(len(got)):
                assert (got[i] == g[rec + "|o%d" % i]).all(), (rec, i)
        else:
            in_fmt, in_alpha, out_order, out_alpha, which, w, h = a
            npl = (4 if in_alpha else 3) if in_fmt == 1 else 1
            planes = [g[rec + "|i%d" % i] for i in range(npl)]
            want = g[rec + "|out"]
            got = np.full_like(want, 0x5A)
            sp, ss = po.planes_args(planes)
            assert orc.orc_yuv_to_rgb(ctypes.addressof(sp), ctypes.addressof(ss), w, h, in_fmt, in_alpha, P(got), got.strides[0], out_order,
                                      out_alpha, which) == 0
            assert (got == want).all(), rec
        n += 1
    assert n == len(g["records"]) and n > 150


def comp_layers(g, rec, ptr_of):
    geo, alphas = g[rec + "|geo"], g[rec + "|alpha"]
    L = (po.CompLayer * 4)()
    keep = []
    for z in range(4):
        a = g[rec + "|l%d" % z]
        keep.append(a)
        L[z].src, L[z].irow = ptr_of(a), a.strides[0]
        L[z].width, L[z].height, L[z].offs_x, L[z].offs_y = [int(v) for v in geo[z]]
        L[z].alpha = float(alphas[z])
    return L, keep


def test_compositor(orc):
    import ctypes
    g = gu.load("comp.npz")
    for rec in map(str, g["records"]):
        ps, is_bgr, revz, ow, oh = map(int, rec.split("|")[1:])
        L, keep = comp_layers(g, rec, lambda a: a.ctypes.data)
        want = g[rec + "|o"]
        got = np.full_like(want, 0x5A)
        orc.orc_composite(P(got), got.strides[0], ow, oh, ps, is_bgr, (ctypes.c_int * 3)(*[int(v) for v in g[rec + "|bg"]]), L, 4, revz)
        assert (got[:, :ow * ps] == want[:, :ow * ps]).all(), rec


def test_lut16_and_fused_k2(orc):
    """create_gamma_lut tables and convert_yuv420p_to_rgb_frame with a LUT16 fused (tests/golden/lut16.npz)"""
    g = gu.load("lut16.npz")
    for name in map(str, g["luts"]):
        f, t = map(int, name.split("_"))
        mine = np.zeros(65536, np.uint16)
        assert orc.orc_gamma_lut16(1.0, f, t, 1.4, P(mine)) == 1
        assert (mine == gu.lut16(g, name)).all(), name
    lut = np.ascontiguousarray(gu.lut16(g, "-1_1"))
    for rec in g["records"]:
        w, h, ys, cs, which, opsize, quality, is422, orow = map(int, g[rec + "_geom"])
        Y, U, V = g[rec + "_y"], g[rec + "_u"], g[rec + "_v"]
        chh = h if is422 else h // 2
        strides = (ctypes.c_int * 3)(ys, cs, cs)
        got = np.zeros((h, orow), np.uint8)
        assert orc.orc_yuv420p_to_rgb_lut16(P(Y), P(U), P(V), strides, chh * cs, chh * cs, P(got), orow, w, h, opsize, 0, is422, which, quality, P(lut), 0) == 0
        want = gu.k2_reference_pixels(g[rec + "_out"], w, h, which, opsize, is422, orow)
        diff = (got[:, :w * opsize].reshape(h, w, opsize) != want).any(axis=2) & ~gu.k2_mask(w, h, is422)
        if (which & 1) and not is422:
            diff[:, w - 1] = False
            diff[1, 0] = False
            diff[0, :] &= False
            diff[h - 1, :] &= False
        assert not diff.any(), "%s: %d pixels differ, first %s" % (rec, diff.sum(), np.argwhere(diff)[0])


def test_blurzoom_sequences(orc):
    g = gu.load("blurzoom.npz")
    for rec in map(str, g["records"]):
        pal, mode, pattern, w, h, n = map(int, rec.split("|")[1:])
        src, want = g[rec + "|in"], g[rec + "|out"]
        z = orc.orc_blurzoom_new(w, h, pal)
        for f in range(n):
            a = np.ascontiguousarray(src[f])
            got = np.full_like(a, 0x5A)
            assert orc.orc_blurzoom_process(z, P(a), a.strides[0], P(got), got.strides[0], mode, pattern) == 0
            assert (got[:, :w * 4] == want[f][:, :w * 4]).all(), (rec, f)
        orc.orc_blurzoom_free(z)


def test_yuv_yuv_tables(orc):
    g = gu.load("yuvyuv.npz")
    t = [np.zeros(256, np.uint8) for _ in range(4)]
    orc.orc_yuv_yuv_tables(*[P(x) for x in t])
    for a, k in zip(t, ("yc2u", "uvc2u", "yu2c", "uvu2c")):
        assert (a == g[k]).all(), k


def test_transitions(orc):
    g = gu.load("transitions.npz")
    for rec in map(str, g["records"]):
        f = rec.split("|")
        t, pal, amt, w, h = int(f[1]), int(f[2]), float(f[3]), int(f[4]), int(f[5])
        ps = 3 if pal <= 2 else 4
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        got = np.full_like(a, 0x5A)
        orc.orc_transition(t, P(a), a.strides[0], P(b), b.strides[0], P(got), got.strides[0], w, h, ps, amt)
        assert (got[:, :w * ps] == want[:, :w * ps]).all(), rec



def test_slide_over(orc):
    g = gu.load("slide_over.npz")
    for rec in map(str, g["records"]):
        _, dirn, pal, tv, mvl, mvu, w, h = rec.split("|")
        w, h, ps = int(w), int(h), (3 if int(pal) <= 2 else 4)
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        got = np.full_like(a, 0x5A)
        orc.orc_slide_over(P(a), a.strides[0], P(b), b.strides[0], P(got), got.strides[0], w, h, ps, int(tv), int(dirn), int(mvl), int(mvu))
        assert (got[:, :w * ps] == want[:, :w * ps]).all(), rec


def test_yuv_repack(orc):
    import ctypes
    g = gu.load("yuv_repack.npz")
    for rec in map(str, g["records"]):
        _, ip, op, unc, pad, w, h = rec.split("|")
        ip, op, unc, w, h = int(ip), int(op), int(unc), int(w), int(h)
        nin, nout = len(po.YUV_PLANE_DIMS[ip](w, h)), len(po.YUV_PLANE_DIMS[op](w, h))
        src = [np.ascontiguousarray(g[rec + "|i%d" % i]) for i in range(nin)]
        want = [g[rec + "|o%d" % i] for i in range(nout)]
        got = [np.full_like(a, 0x5A) for a in want]
        sp, ss = po.planes_args(src)
        gp, gs = po.planes_args(got)
        assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(gp), ctypes.addressof(gs), w, h, unc, 0) == 0
        for i, a in enumerate(want):
            assert (got[i] == a).all(), "%s plane %d" % (rec, i)


def test_yuv411_repack(orc):
    """YUV411 <-> the other YUV palettes against the reference slice fixtures (:7755-7798, :7973-8033, :8272-8303, :8622-9196)"""
    import ctypes
    g = gu.load("yuv411_repack.npz")
    for rec in map(str, g["records"]):
        _, ip, op, unc, pad, w, h = rec.split("|")
        ip, op, unc, w, h = int(ip), int(op), int(unc), int(w), int(h)
        nin, nout = len(po.YUV_PLANE_DIMS[ip](w, h)), len(po.YUV_PLANE_DIMS[op](w, h))
        src = [np.ascontiguousarray(g[rec + "|i%d" % i]) for i in range(nin)]
        want = [g[rec + "|o%d" % i] for i in range(nout)]
        got = [np.full_like(a, 0x5A) for a in want]
        sp, ss = po.planes_args(src)
        gp, gs = po.planes_args(got)
        assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(gp), ctypes.addressof(gs), w, h, unc, 0) == 0
        for i, a in enumerate(want):
            assert (got[i] == a).all(), "%s plane %d" % (rec, i)


def test_chroma_average_tables(orc):
    """orc_cavg against the reference's cavgc / cavgu (init_average, src/colourspace.c:190-216), all 2 x 65,536 entries"""
    g = gu.load("cavg.npz")
    for cl, key in ((1, "cavgc"), (0, "cavgu")):
        got = np.array([[orc.orc_cavg(cl, x, y) for y in range(256)] for x in range(256)], np.uint8)
        assert (got == g[key]).all(), key


def test_chroma_up_packed(orc):
    """4:2:0 / 4:2:2 planar -> YUV888 / YUVA8888 against the reference slice fixtures (:10715-10873); m = 0 where the reference reads past a compact plane"""
    import ctypes
    g = gu.load("chroma_up.npz")
    for rec in map(str, g["records"]):
        _, ip, op, unc, sampling, pad, w, h = rec.split("|")
        ip, op, unc, sampling, w, h = int(ip), int(op), int(unc), int(sampling), int(w), int(h)
        src = [np.ascontiguousarray(g[rec + "|i%d" % i]) for i in range(3)]
        want, mask = g[rec + "|o0"], g[rec + "|m"]
        got = [np.full_like(want, 0x5A)]
        sp, ss = po.planes_args(src)
        gp, gs = po.planes_args(got)
        assert orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(gp), ctypes.addressof(gs), w, h, unc, sampling) == 0
        assert (got[0] * mask == want).all(), rec


def test_k4_lut16(orc):
    """RGB family -> UYVY / YUYV with the 16-bit gamma LUT inline against the reference slice fixtures (rgb2uyvy_with_gamma / rgb2yuyv_with_gamma)"""
    g, L = gu.load("k4_lut16.npz"), gu.load("lut16.npz")
    for rec in map(str, g["records"]):
        _, lname, order, alpha, fmt, unc, w, h, pad = rec.split("|")
        order, alpha, fmt, unc, w, h = int(order), int(alpha), int(fmt), int(unc), int(w), int(h)
        lut = np.ascontiguousarray(gu.lut16(L, lname))
        src, want = np.ascontiguousarray(g[rec + "|i"]), g[rec + "|o"]
        got = np.full_like(want, 0x5A)
        assert orc.orc_rgb_to_yuv_lut16(P(src), src.strides[0], w, h, order, alpha, P(got), got.strides[0], fmt, unc, P(lut)) == 0
        assert (got == want).all(), rec


def test_deinterlace(orc):
    g = gu.load("deinterlace.npz")
    for rec in map(str, g["records"]):
        _, pal, inplace, w, h = rec.split("|")
        a, want = g[rec + "|a"], g[rec + "|o"]
        got = a.copy() if inplace == "1" else np.full_like(a, 0x5A)
        src = got if inplace == "1" else a
        assert orc.orc_deinterlace(P(src), src.strides[0], P(got), got.strides[0], int(w), int(h), int(pal)) == 0
        assert (got == want).all(), rec


def test_rgbdelay_sequences(orc):
    g = gu.load("rgbdelay.npz")
    assert sorted(map(str, g["records"])) == sorted(gu.RGBDELAY_CASES)
    for name, (fn, pal, clamp, maxcache, groups, inplace) in gu.RGBDELAY_CASES.items():
        on, st = gu.rgbdelay_params(groups)
        fin, fout = g[name + "|in"], g[name + "|out"]
        s = orc.orc_rgbdelay_new()
        for i in range(fin.shape[0]):
            src = np.ascontiguousarray(fin[i])
            got = src.copy() if inplace else np.full_like(src, 0x5A)
            a = got if inplace else src
            assert orc.orc_rgbdelay_process(s, P(a), a.strides[0], P(got), got.strides[0], 10, 6, pal, 1 if clamp == 0 else 0, maxcache,
                                            on.ctypes.data, st.ctypes.data) == 0
            assert (got == fout[i]).all(), (name, i)
        orc.orc_rgbdelay_free(s)


def test_script_effects(orc):
    g = gu.load("scriptfx.npz")
    for rec in map(str, g["records"]):
        _, kind, pal, prm, inplace = rec.split("|")
        kind, pal = int(kind), int(pal)
        p = [float(v) for v in prm.split(",")]
        ps = 3 if pal <= 2 else 4
        a, want = g[rec + "|a"], g[rec + "|o"]
        luts = np.zeros((4, 256), np.uint8)
        assert orc.orc_fx_luts(kind, pal, p[0], p[1], p[2], luts.ctypes.data) == ps
        got = a.copy() if inplace == "1" else np.full_like(a, 0x5A)
        src = got if inplace == "1" else a
        orc.orc_byte_luts(P(src), src.strides[0], P(got), got.strides[0], 13, 5, ps, luts.ctypes.data)
        assert (got == want).all(), rec


def _tsplit_cases(g):
    for rec in map(str, g["records"]):
        _, pal, start, sym, end, vert, bw, inplace = rec.split("|")
        yield rec, int(pal), float(start), int(sym), float(end), int(vert), float(bw), int(inplace)


def test_triple_split(orc):
    g = gu.load("triple_split.npz")
    bc = np.array([200, 100, 50], np.int32)
    for rec, pal, start, sym, end, vert, bw, inplace in _tsplit_cases(g):
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        got = a.copy() if inplace else np.full_like(a, 0x5A)
        s1 = got if inplace else a
        orc.orc_triple_split(P(s1), s1.strides[0], P(b), b.strides[0], P(got), got.strides[0], 21, 12, pal == 2, start, sym, end, vert, bw, bc.ctypes.data)
        assert (got == want).all(), rec


def test_dissolve(orc):
    g = gu.load("dissolve.npz")
    for rec in map(str, g["records"]):
        _, pal, amt, seed, inplace = rec.split("|")
        ps = 3 if int(pal) <= 2 else 4
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        mask = np.zeros(17 * 9, np.float32)
        orc.orc_dissolve_mask(int(seed), 17, 9, mask.ctypes.data)
        got = a.copy() if inplace == "1" else np.full_like(a, 0x5A)
        s1 = got if inplace == "1" else a
        orc.orc_dissolve(P(s1), s1.strides[0], P(b), b.strides[0], P(got), got.strides[0], 17, 9, ps, mask.ctypes.data, float(amt))
        assert (got == want).all(), rec
