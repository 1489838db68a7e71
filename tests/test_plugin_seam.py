"""The weed plugin seam: livesgpu_fx.so driven by a real weed host (the reference's libweed + oracle/ref/refhost.c).

CPU part: bootstrap works, the filter classes mirror the reference plugins' (names, palettes, channel and
parameter counts), and without a GPU init_func refuses loudly.  GPU part: every golden plugin record through
weed_setup()/process_func, compared with the bytes the REFERENCE plugin produced.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "lives_amd", "livesgpu_fx.so")
needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (reference libweed host) not built")
PSIZE = {1: 3, 2: 3, 3: 4, 4: 4, 5: 4}


@needs_ref
def test_filter_classes_mirror_the_reference():
    H = po.RefHost()
    ours = {f["name"]: f for f in H.filters(OURS)}
    assert len(ours) == 33
    c = ours["compositor"]          # gdk/compositor.c:295-351 (not buildable here: gdk headers); templates restated, the scaler pinned on the library itself
    assert (c["n_in"], c["n_out"], c["n_params"], c["palettes"]) == (1, 1, 7, [1, 2, 3, 4]) and (c["flags"] & 256)
    for plug in ("simple_blend", "multi_blends", "colorkey", "mirrors", "edge", "softlight", "blurzoom", "slide_over", "deinterlace", "RGBdelay", "negate", "posterise",
                 "ccorrect", "layout_blends"):
        for rf in H.filters(po.refplugin(plug)):
            o = ours[rf["name"]]
            assert (o["n_in"], o["n_out"], o["n_params"]) == (rf["n_in"], rf["n_out"], rf["n_params"]), rf["name"]
            if rf["name"].endswith("luma overlay") or rf["name"] == "luma underlay":
                assert o["palettes"] == [1, 2, 3, 4]          # ARGB32 luma blends declined
            elif rf["name"] == "deinterlace":
                assert o["palettes"] == [p for p in rf["palettes"] if p not in (544, 545, 512, 513, 522)]   # planar: the reference does nothing / crashes
            else:
                assert o["palettes"] == rf["palettes"], rf["name"]
            assert not (o["flags"] & 64), "a GPU filter must not advertise WEED_FILTER_HINT_MAY_THREAD"
            assert (o["flags"] & 8) == (rf["flags"] & 8), "PREF_LINEAR_GAMMA must match: " + rf["name"]


@needs_ref
def test_refuses_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    H = po.RefHost()
    a = np.zeros((8, 64), np.uint8)
    with pytest.raises(RuntimeError, match="returned 64"):       # WEED_ERROR_PLUGIN_INVALID from init_func
        H.run(OURS, "chroma blend", 3, 16, 8, [a, a.copy()], a.copy(), [po.p_int(128)])


@needs_ref
@pytest.mark.gpu
def test_golden_records_through_the_plugin():
    H = po.RefHost()
    g = gu.load("plugins.npz")
    n = 0
    for rec in g["records"]:
        rec = str(rec)
        f = rec.split("|")
        if f[0] == "sb":
            fn, pal, prm = f[1], int(f[2]), int(f[3])
            if fn != "chroma blend" and pal == 5:
                continue
            a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
            d = a.copy()
            H.run(OURS, fn, pal, 18, 8, [a, b], d, [po.p_int(prm)])
            nbytes, rows = 18 * PSIZE[pal], 8
        elif f[0] == "mb":
            fn, pal, prm = f[1], int(f[2]), int(f[3])
            a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
            d = np.zeros_like(a)
            H.run(OURS, fn, pal, 18, 8, [a, b], d, [po.p_int(prm)])
            nbytes, rows = 54, 8
        elif f[0] == "ck":
            pal, delta, opac = int(f[1]), float(f[2]), float(f[3])
            col = list(map(int, f[4].split(",")))
            a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
            d = np.zeros_like(a)
            H.run(OURS, "colorkey", pal, 18, 8, [a, b], d, [po.p_double(delta), po.p_double(opac), po.p_rgb(*col)])
            nbytes, rows = 54, 8
        else:
            fn, pal, mw, mh = f[1], int(f[2]), int(f[3]), int(f[4])
            a, want = g[rec + "|a"], g[rec + "|o"]
            d = a.copy()
            H.run(OURS, fn, pal, mw, mh, [d], d, [])
            nbytes, rows = mw * PSIZE[pal], mh
        assert (d[:rows, :nbytes] == want[:rows, :nbytes]).all(), rec
        n += 1
    assert n > 200


@needs_ref
@pytest.mark.gpu
def test_sliced_calls_equal_one_call():
    """a host that slices anyway (process_func_threaded protocol) gets the same pixels"""
    H = po.RefHost()
    rng = np.random.default_rng(9)
    w, h = 40, 24
    s1 = po.make_frame(rng, w, h, 4, extra_rows=1, alpha_mix=True)
    s2 = po.make_frame(rng, w, h, 4, extra_rows=1, alpha_mix=True)
    a, b = s1.copy(), s1.copy()
    H.run(OURS, "chroma blend", 3, w, h, [s1, s2], a, [po.p_int(140)], nslices=1)
    H.run(OURS, "chroma blend", 3, w, h, [s1, s2], b, [po.p_int(140)], nslices=3)
    assert (a == b).all()


@needs_ref
@pytest.mark.gpu
def test_stencil_records_through_the_plugin():
    """softlight.c / edge.c fixtures through weed_setup() / process_func of livesgpu_fx.so"""
    H = po.RefHost()
    g = gu.load("stencils.npz")
    n = 0
    for rec in map(str, g["records"]):
        f = rec.split("|")
        if f[0] == "sl":
            pal, w, h, uncl = map(int, f[1:])
            npl = 4 if pal == 545 else 3
            src = [g[rec + "|i%d" % i].copy() for i in range(npl)]
            dst = [np.full_like(a, 0x5A) for a in src]
            H.run_planar(OURS, "softlight", pal, w, h, src, dst, uncl)
            cw = w >> 1 if pal in (512, 513, 522) else w
            ch = h >> 1 if pal in (512, 513) else h
            dims = [(w, h), (cw, ch), (cw, ch), (w, h)]
            for i in range(npl):
                assert (dst[i][:dims[i][1], :dims[i][0]] == g[rec + "|o%d" % i][:dims[i][1], :dims[i][0]]).all(), (rec, i)
        else:
            pal, mode, inplace, w, h = map(int, f[1:])
            ps = PSIZE[pal]
            a, d = g[rec + "|a"].copy(), g[rec + "|d"].copy()
            H.run(OURS, "edge detect", pal, w, h, [d if inplace else a], d, [po.p_int(mode)])
            assert (d[:h, :w * ps] == g[rec + "|o"][:h, :w * ps]).all(), rec
        n += 1
    assert n == 50


@needs_ref
@pytest.mark.gpu
def test_blurzoom_sequences_through_the_plugin():
    """one filter instance over a frame sequence (stateful): weed_setup / init_func / process_func x n / deinit_func"""
    H = po.RefHost()
    g = gu.load("blurzoom.npz")
    for rec in map(str, g["records"]):
        pal, mode, pattern, w, h, n = map(int, rec.split("|")[1:])
        srcs = [np.ascontiguousarray(a) for a in g[rec + "|in"]]
        dsts = [np.full_like(a, 0x5A) for a in srcs]
        H.run_seq(OURS, "blurzoom", pal, w, h, srcs, dsts, [po.p_int(mode), po.p_int(pattern)])
        for f in range(n):
            assert (dsts[f][:, :w * 4] == g[rec + "|out"][f][:, :w * 4]).all(), (rec, f)


@needs_ref
@pytest.mark.gpu
def test_transition_records_through_the_plugin():
    H = po.RefHost()
    g = gu.load("transitions.npz")
    names = ["iris rectangle", "iris circle", "4 way split"]
    for rec in map(str, g["records"]):
        f = rec.split("|")
        t, pal, amt, w, h = int(f[1]), int(f[2]), float(f[3]), int(f[4]), int(f[5])
        d = np.full_like(g[rec + "|o"], 0x5A)
        H.run(OURS, names[t], pal, w, h, [g[rec + "|a"].copy(), g[rec + "|b"].copy()], d, [po.p_double(amt)])
        assert (d[:, :w * PSIZE[pal]] == g[rec + "|o"][:, :w * PSIZE[pal]]).all(), rec



@needs_ref
@pytest.mark.gpu
def test_batch_hook_one_launch_for_the_instances_of_a_plan_step():
    """livesgpu_fx_process_batch (this plugin's extension): n instances of one transition class, an amount each, in ONE launch -- the golden records of a class
    as one batch, then 7 and 17 (> LGPU_FX_MAX_FRAMES: per instance behind the same call) random instances against the REFERENCE plugin run one by one"""
    H = po.RefHost()
    g = gu.load("transitions.npz")
    names = ["iris rectangle", "iris circle", "4 way split"]
    groups = {}
    for rec in map(str, g["records"]):
        f = rec.split("|")
        groups.setdefault((int(f[1]), int(f[2]), int(f[4]), int(f[5])), []).append((rec, float(f[3])))
    for (t, pal, w, h), recs in groups.items():
        a = [g[r + "|a"].copy() for r, _ in recs]
        b = [g[r + "|b"].copy() for r, _ in recs]
        d = [np.full_like(g[r + "|o"], 0x5A) for r, _ in recs]
        H.run_batch(OURS, names[t], pal, w, h, a, b, d, [amt for _, amt in recs], hook="livesgpu_fx_process_batch")
        for (r, _), got in zip(recs, d):
            assert (got[:, :w * PSIZE[pal]] == g[r + "|o"][:, :w * PSIZE[pal]]).all(), r
    rng = np.random.default_rng(77)
    ref = po.refplugin("multi_transitions")
    for t, pal, w, h, n, inplace in ((0, 1, 640, 360, 7, False), (1, 3, 640, 360, 7, True), (2, 3, 322, 121, 7, False), (1, 1, 97, 33, 17, False), (0, 4, 64, 40, 1, False)):
        ps = PSIZE[pal]
        a = [po.make_frame(rng, w, h, ps) for _ in range(n)]
        b = [po.make_frame(rng, w, h, ps) for _ in range(n)]
        amounts = list(rng.random(n))
        amounts[0], amounts[-1] = 0., 1.
        want = [x.copy() if inplace else np.full_like(x, 0x33) for x in a]
        H.run_batch(ref, names[t], pal, w, h, want if inplace else a, b, want, amounts)
        got = [x.copy() if inplace else np.full_like(x, 0x33) for x in a]
        H.run_batch(OURS, names[t], pal, w, h, got if inplace else a, b, got, amounts, hook="livesgpu_fx_process_batch")
        for i in range(n):
            assert (got[i] == want[i]).all(), (t, pal, w, h, i)          # the row padding too: it stays as it was
    # a chain inside the batch (instance 1 reads what instance 0 writes): the hook keeps the sequence of the per-instance loop
    a = [po.make_frame(rng, 96, 20, 4) for _ in range(3)]
    b = [po.make_frame(rng, 96, 20, 4) for _ in range(3)]
    outs = [np.zeros_like(a[0]) for _ in range(3)]
    H.run_batch(OURS, "iris circle", 3, 96, 20, [a[0], outs[0], a[2]], b, outs, [0.3, 0.6, 0.9], hook="livesgpu_fx_process_batch")
    want = [np.zeros_like(a[0]) for _ in range(3)]
    H.run_batch(ref, "iris circle", 3, 96, 20, [a[0], want[0], a[2]], b, want, [0.3, 0.6, 0.9])
    assert all((outs[i][:, :96 * 4] == want[i][:, :96 * 4]).all() for i in range(3))
    # "softlight": planar frames, n instances in one launch, against the reference plugin run per instance
    for pal, w, h, n, uncl in ((512, 96, 40, 5, 0), (544, 70, 33, 16, 1), (545, 64, 20, 3, 0), (522, 128, 18, 2, 1), (512, 96, 40, 17, 0)):
        npl = 4 if pal == 545 else 3
        cw = w >> 1 if pal in (512, 513, 522) else w
        ch = h >> 1 if pal in (512, 513) else h
        dims = [(w, h), (cw, ch), (cw, ch), (w, h)][:npl]
        srcs = [[rng.integers(0, 256, (dh_, (dw_ + 15) // 16 * 16), dtype=np.uint8) for dw_, dh_ in dims] for _ in range(n)]
        want = [[np.full_like(a, 0x5A) for a in fr] for fr in srcs]
        H.run_planar_batch(po.refplugin("softlight"), "softlight", pal, w, h, srcs, want, uncl)
        got = [[np.full_like(a, 0x5A) for a in fr] for fr in srcs]
        H.run_planar_batch(OURS, "softlight", pal, w, h, srcs, got, uncl, hook="livesgpu_fx_process_batch")
        for i in range(n):
            for k, (dw_, dh_) in enumerate(dims):
                assert (got[i][k][:dh_, :dw_] == want[i][k][:dh_, :dw_]).all(), (pal, i, k)
                assert (got[i][k][:, dw_:] == 0x5A).all(), "row padding stays as it was"
    # the blends of simple_blend.c / multi_blends.c: an integer amount per instance
    for name, plug, pal, n, inplace in (("chroma blend", "simple_blend", 3, 6, True), ("chroma blend", "simple_blend", 1, 4, False), ("luma overlay", "simple_blend", 4, 5, False),
                                        ("averaged luma overlay", "simple_blend", 2, 3, True), ("blend_screen", "multi_blends", 1, 7, False), ("blend_burn", "multi_blends", 2, 16, True),
                                        ("chroma blend", "simple_blend", 5, 3, False)):          # ARGB32: per instance behind the same call
        ps, w, h = PSIZE[pal], 200, 44
        names = {f["name"] for f in H.filters(OURS)}
        assert name in names, sorted(names)
        a = [po.make_frame(rng, w, h, ps, alpha_mix=True) for _ in range(n)]
        b = [po.make_frame(rng, w, h, ps, alpha_mix=True, pad_px=1) for _ in range(n)]
        amounts = [int(v) for v in rng.integers(0, 256, n)]
        want = [x.copy() if inplace else np.full_like(x, 0x33) for x in a]
        H.run_batch(po.refplugin(plug), name, pal, w, h, want if inplace else a, b, want, amounts, int_param=True)
        got = [x.copy() if inplace else np.full_like(x, 0x33) for x in a]
        H.run_batch(OURS, name, pal, w, h, got if inplace else a, b, got, amounts, hook="livesgpu_fx_process_batch", int_param=True)
        for i in range(n):
            assert (got[i][:, :w * ps] == want[i][:, :w * ps]).all(), (name, pal, i)


@needs_ref
@pytest.mark.gpu
def test_slide_over_records_through_the_plugin():
    H = po.RefHost()
    g = gu.load("slide_over.npz")
    for rec in map(str, g["records"]):
        _, dirn, pal, tv, mvl, mvu, w, h = rec.split("|")
        dirn, pal, w, h = int(dirn), int(pal), int(w), int(h)
        d = np.full_like(g[rec + "|o"], 0x5A)
        radios = [po.p_bool(False)] + [po.p_bool(dirn == k) for k in (1, 2, 3)] + [po.p_bool(False)]
        H.run(OURS, "slide over", pal, w, h, [g[rec + "|a"].copy(), g[rec + "|b"].copy()], d,
              [po.p_int(int(tv))] + radios + [po.p_bool(int(mvl)), po.p_bool(int(mvu))])
        assert (d[:, :w * PSIZE[pal]] == g[rec + "|o"][:, :w * PSIZE[pal]]).all(), rec


@needs_ref
@pytest.mark.gpu
def test_deinterlace_records_through_the_plugin():
    H = po.RefHost()
    g = gu.load("deinterlace.npz")
    for rec in map(str, g["records"]):
        _, pal, inplace, w, h = rec.split("|")
        a, want = g[rec + "|a"], g[rec + "|o"]
        if inplace == "1":
            d = a.copy()
            H.run(OURS, "deinterlace", int(pal), int(w), int(h), [d], d, [])
        else:
            d = np.full_like(a, 0x5A)
            H.run(OURS, "deinterlace", int(pal), int(w), int(h), [a.copy()], d, [])
        n = (int(w) + 2) // 3 * 3 * (3 if int(pal) in (1, 2, 588) else 4)
        assert (d[:, :n] == want[:, :n]).all(), rec


@needs_ref
@pytest.mark.gpu
def test_rgbdelay_sequences_through_the_plugin():
    H = po.RefHost()
    g = gu.load("rgbdelay.npz")
    for name, (fn, pal, clamp, maxcache, groups, inplace) in gu.RGBDELAY_CASES.items():
        on, st = gu.rgbdelay_params(groups)
        params = [po.p_int(maxcache)]
        for j in range(51):
            params += [po.p_bool(on[3 * j]), po.p_bool(on[3 * j + 1]), po.p_bool(on[3 * j + 2]), po.p_double(st[j])]
        fin, fout = g[name + "|in"], g[name + "|out"]
        frames = [np.ascontiguousarray(fin[i]) for i in range(fin.shape[0])]
        H.H.refhost_set_yuv_clamping(clamp)
        try:
            if inplace:
                out = [f.copy() for f in frames]
                H.run_seq(OURS, fn, pal, 10, 6, out, out, params)
            else:
                out = [np.full_like(f, 0x5A) for f in frames]
                H.run_seq(OURS, fn, pal, 10, 6, frames, out, params)
        finally:
            H.H.refhost_set_yuv_clamping(-1)
        for i in range(len(frames)):
            assert (out[i] == fout[i]).all(), (name, i)


@needs_ref
@pytest.mark.gpu
def test_script_effect_records_through_the_plugin():
    H = po.RefHost()
    g = gu.load("scriptfx.npz")
    names = ["negate", "posterise", "ccorrect"]
    for rec in map(str, g["records"]):
        _, kind, pal, prm, inplace = rec.split("|")
        kind, pal = int(kind), int(pal)
        p = [float(v) for v in prm.split(",")]
        params = [] if kind == 0 else [po.p_int(int(p[0]))] if kind == 1 else [po.p_double(v) for v in p]
        a, want = g[rec + "|a"], g[rec + "|o"]
        if inplace == "1":
            d = a.copy()
            H.run(OURS, names[kind], pal, 13, 5, [d], d, params)
        else:
            d = np.full_like(a, 0x5A)
            H.run(OURS, names[kind], pal, 13, 5, [a.copy()], d, params)
        assert (d == want).all(), rec


@needs_ref
@pytest.mark.gpu
def test_triple_split_records_through_the_plugin():
    H = po.RefHost()
    g = gu.load("triple_split.npz")
    for rec in map(str, g["records"]):
        _, pal, start, sym, end, vert, bw, inplace = rec.split("|")
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        prm = [po.p_double(float(start)), po.p_bool(int(sym)), po.p_bool(not int(sym)), po.p_double(float(end)), po.p_bool(int(vert)), po.p_double(float(bw)),
               po.p_rgb(200, 100, 50)]
        d = a.copy() if inplace == "1" else np.full_like(a, 0x5A)
        H.run(OURS, "triple split", int(pal), 21, 12, [d if inplace == "1" else a.copy(), b.copy()], d, prm)
        assert (d == want).all(), rec


@needs_ref
@pytest.mark.gpu
def test_rand_replace_through_the_plugin():
    """"rand replace" (multi_transitions.c:96-112, :213-220): one uniform draw per frame picks the second input (draw < amount) or the first.  The reference draws from a
    time-seeded global generator, so the sequence cannot be pinned; the two ends of the parameter are deterministic in both, and the reference plugin run on the same
    frames gives the same bytes there; in between the share of frames taken from the second input follows the amount"""
    H = po.RefHost()
    rng = np.random.default_rng(77)
    for pal in (1, 3, 4):
        ps = PSIZE[pal]
        a, b = rng.integers(0, 256, (9, po.align(17 * ps)), dtype=np.uint8), rng.integers(0, 256, (9, po.align(17 * ps)), dtype=np.uint8)
        for amt, src in ((0.0, a), (1.0, b)):
            for inplace in (0, 1):
                outs = []
                for so in (OURS, po.refplugin("multi_transitions")):
                    d = a.copy() if inplace else np.full_like(a, 0x5A)
                    H.run(so, "rand replace", pal, 17, 9, [d if inplace else a.copy(), b.copy()], d, [po.p_double(amt)])
                    outs.append(d)
                assert (outs[0][:, :17 * ps] == src[:, :17 * ps]).all() and (outs[0][:, :17 * ps] == outs[1][:, :17 * ps]).all(), (pal, amt, inplace)
    a, b = np.zeros((9, po.align(17 * 4)), np.uint8), np.full((9, po.align(17 * 4)), 200, np.uint8)
    n = 0
    for _ in range(400):
        d = np.zeros_like(a)
        H.run(OURS, "rand replace", 3, 17, 9, [a.copy(), b.copy()], d, [po.p_double(0.3)])
        n += int(d[0, 0] == 200)
    assert 70 <= n <= 170, n          # 400 draws at 0.3: 120 +- 5 sigma


@needs_ref
@pytest.mark.gpu
def test_dissolve_records_through_the_plugin():
    H = po.RefHost()
    g = gu.load("dissolve.npz")
    for rec in map(str, g["records"]):
        _, pal, amt, seed, inplace = rec.split("|")
        a, b, want = g[rec + "|a"], g[rec + "|b"], g[rec + "|o"]
        d = a.copy() if inplace == "1" else np.full_like(a, 0x5A)
        H.H.refhost_set_random_seed(int(seed))
        try:
            H.run(OURS, "dissolve", int(pal), 17, 9, [d if inplace == "1" else a.copy(), b.copy()], d, [po.p_double(float(amt))])
        finally:
            H.H.refhost_set_random_seed(0)
        assert (d == want).all(), rec


@needs_ref
@pytest.mark.gpu
def test_compositor_class_scales_and_paints_as_the_reference(orc):
    """the "compositor" class (gdk/compositor.c:127-292) through weed_setup() / process_func: in channels of different sizes, per-channel offset / scale / alpha
    arrays, a disabled channel, background colour, both z orders; against the pinned restatement of gdk_pixbuf_scale_simple (oracle/orc_pixbuf.c) + the paint loop
    slice (orc_composite), and against the live library where it loads"""
    import ctypes
    from oracle.ref import pixbuf_ref as pr
    H = po.RefHost()
    rng = np.random.default_rng(0xC0)
    for pal, ps in ((1, 3), (2, 3), (3, 4), (4, 4)):
        ow, oh = 160, 90
        sizes = [(64, 36), (200, 120), (50, 40), (96, 54), (33, 21)]
        offsx, offsy = [0.06, 0.44, 0.0, 0.37, 0.2], [0.09, 0.33, 0.22, 0.4, 0.0]
        scx, scy = [0.625, 0.5, 0.31, 0.6, 0.05], [0.62, 0.53, 0.78, 0.6, 0.03]
        alpha = [0.75, 1.0, 0.5, 0.3, 1.0]
        disabled = [0, 0, 0, 1, 0]                 # channel 3 is switched off by the host; channel 4 scales to less than 16 pixels and is skipped (:221)
        srcs = [rng.integers(0, 256, (h, po.align(w * ps, 4)), dtype=np.uint8) for (w, h) in sizes]
        bg = [12, 200, 99]
        for revz in (0, 1):
            dst = np.zeros((oh, po.align(ow * ps, 4)), np.uint8)
            H.run_compositor(OURS, pal, srcs, sizes, disabled, dst, ow, oh, offsx, offsy, scx, scy, alpha, bg, revz)
            L = (po.CompLayer * len(sizes))()
            keep = []
            for z, ((w, h), a) in enumerate(zip(sizes, srcs)):
                outw, outh = (int(ow * scx[z] + 1.) >> 1) << 1, (int(oh * scy[z] + 1.) >> 1) << 1
                if disabled[z] or outw * outh < 16:
                    continue
                interp = 3 if (outw > w or outh > h) else 2
                sc = np.zeros((outh, po.align(outw * ps, 4)), np.uint8)
                assert orc.orc_pixbuf_scale(po.P(a), a.strides[0], w, h, po.P(sc), sc.strides[0], outw, outh, ps, interp) == 0
                if pr.available():
                    assert (pr.scale_simple(a, w, ps, outw, outh, interp) == sc[:, :outw * ps]).all()
                keep.append(sc)
                L[z].src, L[z].irow, L[z].width, L[z].height = sc.ctypes.data, sc.strides[0], outw, outh
                L[z].offs_x, L[z].offs_y, L[z].alpha = int(offsx[z] * ow), int(offsy[z] * oh), alpha[z]
            want = np.zeros_like(dst)
            orc.orc_composite(po.P(want), want.strides[0], ow, oh, ps, 1 if pal in (2, 4) else 0, (ctypes.c_int * 3)(*bg), L, len(sizes), revz)
            assert (dst[:, :ow * ps] == want[:, :ow * ps]).all(), (pal, revz)
