#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE builds under oracle/_ref (run oracle/ref/build_ref.sh first).

TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.  The fixtures are DATA: inputs and the
outputs the reference's own code produced for them (tables, LUTs, frames).  No reference source text goes
into the repo.  Every record carries the seed / parameters it was made with; `manifest.json` lists the
reference files (and line ranges) each group exercises and the prefs the slices were run with
(pb_quality = MED, screen_gamma = 1.4, nfx_threads noted per record).
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
P = po.P
SEED = 0x11FE5


def lut_in_fresh_process(fileg, gfrom, gto):
    """create_gamma_lut8 caches under the wrong key -> one process per LUT"""
    code = ("import ctypes,sys,numpy as np;R=ctypes.CDLL(%r);R.csref_gamma_lut8.argtypes=[ctypes.c_double,ctypes.c_int,ctypes.c_int,ctypes.c_void_p];"
            "R.csref_set_prefs.argtypes=[ctypes.c_int,ctypes.c_int,ctypes.c_double];R.csref_set_prefs(2,1,1.4);a=np.zeros(256,np.uint8);"
            "r=R.csref_gamma_lut8(%r,%d,%d,a.ctypes.data);sys.stdout.write(str(r)+' '+' '.join(map(str,a.tolist())))"
            % (os.path.join(po.REFDIR, "libcsref.so"), fileg, gfrom, gto))
    out = subprocess.check_output([sys.executable, "-c", code]).decode().split()
    return int(out[0]), np.array(list(map(int, out[1:])), np.uint8)


def main():
    assert po.have_ref(), "run oracle/ref/build_ref.sh first"
    os.makedirs(OUT, exist_ok=True)
    R = po.csref()
    R.csref_set_prefs(2, 1, 1.4)
    manifest = {"reference": "salsaman/LiVES @ /root/reference", "prefs": {"pb_quality": "PB_QUALITY_MED", "screen_gamma": 1.4},
                "groups": {},
                "unpinned": {
                 "lgpu_swizzle ops swap4, swapprepost, delpre without LUT": "reference-broken in every mode (DESIGN.md K1-b); checked GPU == oracle only",
                 "lgpu_resize / lgpu_chain resize stage": "libswscale is not in the image; own spec lgpu-polyphase-v1 (DESIGN.md section 5), GPU == oracle + properties; tests/test_swscale_info.py prints PSNR against a libswscale found at run time, never gating",
                 "lgpu_gauss5": "no reference loop exists (BASELINE config 4 names the op only); own spec, GPU == oracle + properties"
                }}

    # ---- conversion tables, alpha tables -----------------------------------------------------------------
    tabs = {}
    for which in range(4):
        a = np.zeros((9, 256), np.int32)
        b = np.zeros((5, 256), np.int32)
        R.csref_tables(which, P(a), P(b))
        tabs["rgb2yuv_%d" % which] = a
        tabs["yuv2rgb_%d" % which] = b
    un = np.zeros((256, 256), np.int32)
    al = np.zeros((256, 256), np.int32)
    R.csref_unal(P(un), P(al))
    tabs["unal"] = un.astype(np.uint8)
    tabs["al"] = al.astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "tables.npz"), **tabs)
    manifest["groups"]["tables.npz"] = "src/colourspace.c:851-1105 (init_RGB_to_YUV_tables, init_YUV_to_RGB_tables), :1141-1160 (init_unal); which: bit0 unclamped, bit1 BT.709"

    # ---- gamma LUTs (fresh process each) ---------------------------------------------------------------------
    luts = {}
    ids = [po.GAMMA_LINEAR, po.GAMMA_SRGB, po.GAMMA_BT709, po.GAMMA_MONITOR]
    for f in ids:
        for t in ids:
            ok, lut = lut_in_fresh_process(1.0, f, t)
            luts["lut8_%d_%d" % (f, t)] = np.concatenate([[ok], lut]).astype(np.uint8)
    for (fg, f, t) in ((2.2, po.GAMMA_SRGB, 2048), (0.45, po.GAMMA_LINEAR, 2048)):
        ok, lut = lut_in_fresh_process(fg, f, t)
        luts["lut8v_%s_%d" % (str(fg).replace(".", "p"), f)] = np.concatenate([[ok], lut]).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "luts.npz"), **luts)
    manifest["groups"]["luts.npz"] = "src/colourspace.c:655-736 create_gamma_lut8(fileg, from, to); key lut8_<from>_<to>; element 0 = 1 if a LUT was returned"

    # ---- K1 swizzles: (op, alpha_first, lut?, mode) where the reference produces the intended permutation -----
    rng = np.random.default_rng(SEED)
    k1 = {}
    lut = rng.integers(0, 256, 256, dtype=np.uint8)
    k1["lut"] = lut
    # canonical call modes: see DESIGN.md "K1" -- (op name, in place?, nfx_threads)
    canon = [("swap3", 1, 2), ("swap3addpost", 0, 1), ("swap3addpost", 0, 2), ("swap3addpre", 0, 2), ("swap3postalpha", 1, 2),
             ("swap3prealpha", 1, 2), ("addpost", 0, 2), ("addpre", 0, 2), ("swap3delpost", 0, 2), ("delpost", 0, 2),
             ("swap3delpre", 0, 2)]
    w, h = 22, 12     # 12 rows slice cleanly for nfx_threads = 2 (8 + 4); tiny heights make slice 0 swallow the frame (:9279)
    recs = []
    for (name, inplace, nthr) in canon:
        op = po.OPS.index(name)
        ib, ob = po.OP_IBPP[op], po.OP_OBPP[op]
        for use_lut in (0, 1):
            src = po.make_frame(rng, w, h, ib)
            R.csref_set_prefs(2, nthr, 1.4)
            if inplace:
                buf = src.copy()
                R.csref_k1(op, P(buf), w, h, buf.strides[0], buf.strides[0], P(buf), P(lut) if use_lut else None, 0)
                dst = buf
            else:
                dst = np.zeros((h, po.align(w * ob)), np.uint8)
                R.csref_k1(op, P(src.copy()), w, h, src.strides[0], dst.strides[0], P(dst), P(lut) if use_lut else None, 0)
            key = "%s_lut%d_t%d" % (name, use_lut, nthr)
            k1[key + "_in"] = src
            k1[key + "_out"] = dst[:, :w * ob].copy()
            recs.append(key)
    # delpre: the no-LUT body copies the same pixel (reference-broken); LUT body is sound
    op = po.OPS.index("delpre")
    src = po.make_frame(rng, w, h, 4)
    dst = np.zeros((h, po.align(w * 3)), np.uint8)
    R.csref_set_prefs(2, 2, 1.4)
    R.csref_k1(op, P(src.copy()), w, h, src.strides[0], dst.strides[0], P(dst), P(lut), 0)
    k1["delpre_lut1_t2_in"] = src
    k1["delpre_lut1_t2_out"] = dst[:, :w * 3].copy()
    recs.append("delpre_lut1_t2")
    k1["records"] = np.array(recs)
    k1["geom"] = np.array([w, h])
    np.savez_compressed(os.path.join(OUT, "k1_swizzle.npz"), **k1)
    R.csref_set_prefs(2, 1, 1.4)
    manifest["groups"]["k1_swizzle.npz"] = ("src/colourspace.c:9259-10577; record <op>_lut<0|1>_t<nfx_threads>; swap3/swap3postalpha/swap3prealpha in place "
                                            "(as convert_layer_palette_full calls them); swap4, swapprepost and delpre-without-LUT are broken in the "
                                            "reference in every mode and have no fixture: UNPINNED (GPU == oracle == the evident permutation; no reference output exists to compare with)")

    # ---- K2 yuv420p / 422p -> rgb --------------------------------------------------------------------------------
    k2 = {}
    recs = []
    for (w, h, ys, cs) in ((32, 12, 32, 16), (34, 14, 64, 32)):
        for which in range(4):
            for opsize in (3, 4):
                for quality in (1, 2):
                    for is422 in (0, 1):
                        if is422 and (quality == 1 or opsize == 3 or which >= 2):
                            continue
                        chh = h if is422 else h // 2
                        Y = rng.integers(0, 256, (h, ys), dtype=np.uint8)
                        U = rng.integers(0, 256, (chh * cs + 1,), dtype=np.uint8)
                        V = rng.integers(0, 256, (chh * cs + 1,), dtype=np.uint8)
                        U[-1] = U[-2]
                        V[-1] = V[-2]     # the reference reads one sample past the plane on the last pair (defined here)
                        orow = po.align(w * opsize)
                        R.csref_set_prefs(quality, 1, 1.4)
                        # one spare row in front: the unclamped path writes one byte before each row (quirk K2-f)
                        buf = np.full((h + 2) * orow, 0xAB, np.uint8)
                        strides = (ctypes.c_int * 3)(ys, cs, cs)
                        R.csref_yuv420p_to_rgb(P(Y), P(U), P(V), w, h, strides, orow, ctypes.c_void_p(buf.ctypes.data + orow), int(opsize == 4),
                                               is422, which & 1, 2 if which & 2 else 1, None)
                        key = "w%d_h%d_t%d_o%d_q%d_s%d" % (w, h, which, opsize, quality, is422)
                        k2[key + "_y"] = Y
                        k2[key + "_u"] = U
                        k2[key + "_v"] = V
                        k2[key + "_out"] = buf.reshape(h + 2, orow)
                        k2[key + "_geom"] = np.array([w, h, ys, cs, which, opsize, quality, is422, orow])
                        recs.append(key)
    k2["records"] = np.array(recs)
    np.savez_compressed(os.path.join(OUT, "k2_yuv420p.npz"), **k2)
    R.csref_set_prefs(2, 1, 1.4)
    manifest["groups"]["k2_yuv420p.npz"] = ("src/colourspace.c:3260-3904 convert_yuv420p_to_rgb_frame, nfx_threads = 1; out has one spare row before and after the "
                                            "frame; masked pixels (reference undefined): row 0 odd x, last row odd x; unclamped (which & 1) 4:2:0 rows 1..h-2 "
                                            "are written one byte early by the reference (:3707 `or = orowstride * i - y_delta`)")

    # ---- K6 gamma apply -----------------------------------------------------------------------------------------------
    k6 = {"lut": lut}
    for (psize, af) in ((3, 0), (4, 0), (4, 1)):
        pix = po.make_frame(rng, 22, 10, psize)
        out = pix.copy()
        R.csref_gamma_apply(P(out), 22, 10, out.strides[0], psize, af, 0, P(lut))
        k6["p%d_a%d_in" % (psize, af)] = pix
        k6["p%d_a%d_out" % (psize, af)] = out
    np.savez_compressed(os.path.join(OUT, "k6_gamma_apply.npz"), **k6)
    manifest["groups"]["k6_gamma_apply.npz"] = "src/colourspace.c:14034-14060 gamma_convert_layer_thread, 22x10"

    # ---- weed plugins ------------------------------------------------------------------------------------------------------
    H = po.RefHost()
    pl = {}
    recs = []
    w, h = 18, 8
    pals = {1: 3, 2: 3, 3: 4, 4: 4, 5: 4}
    for fn in ("chroma blend", "luma overlay", "luma underlay", "negative luma overlay", "averaged luma overlay"):
        for pal, ps in pals.items():
            for prm in (0, 1, 100, 128, 255):
                s1 = po.make_frame(rng, w, h, ps, extra_rows=1, alpha_mix=True)
                s2 = po.make_frame(rng, w, h, ps, extra_rows=1, alpha_mix=True, pad_px=1)
                d = s1.copy()          # dst preset with layer 1: bytes the plugin never writes compare equal
                H.run(po.refplugin("simple_blend"), fn, pal, w, h, [s1, s2], d, [po.p_int(prm)])
                key = "sb|%s|%d|%d" % (fn, pal, prm)
                pl[key + "|a"] = s1
                pl[key + "|b"] = s2
                pl[key + "|o"] = d
                recs.append(key)
    for t, fn in enumerate(("blend_multiply", "blend_screen", "blend_darken", "blend_lighten", "blend_overlay", "blend_dodge", "blend_burn")):
        for pal in (1, 2):
            for prm in (0, 100, 127, 128, 255):
                s1, s2 = po.make_frame(rng, w, h, 3), po.make_frame(rng, w, h, 3)
                s1[0, :6] = [0, 255, 1, 254, 0, 255]
                s2[0, :6] = [255, 0, 254, 1, 0, 255]
                d = np.zeros_like(s1)
                H.run(po.refplugin("multi_blends"), fn, pal, w, h, [s1, s2], d, [po.p_int(prm)])
                key = "mb|%s|%d|%d" % (fn, pal, prm)
                pl[key + "|a"] = s1
                pl[key + "|b"] = s2
                pl[key + "|o"] = d
                recs.append(key)
    for pal in (1, 2):
        for delta in (0.0, 0.2, 1.0):
            for opac in (0.0, 0.3, 1.0):
                for col in ((0, 0, 255), (10, 200, 30)):
                    s1, s2 = po.make_frame(rng, w, h, 3), po.make_frame(rng, w, h, 3)
                    d = np.zeros_like(s1)
                    H.run(po.refplugin("colorkey"), "colorkey", pal, w, h, [s1, s2], d, [po.p_double(delta), po.p_double(opac), po.p_rgb(*col)])
                    key = "ck|%d|%s|%s|%d,%d,%d" % (pal, delta, opac, col[0], col[1], col[2])
                    pl[key + "|a"] = s1
                    pl[key + "|b"] = s2
                    pl[key + "|o"] = d
                    recs.append(key)
    for pal, ps in ((1, 3), (3, 4)):
        for mode, fn in enumerate(("mirrorx", "mirrory", "mirrorxy")):
            for (mw, mh) in ((18, 8), (17, 7)):
                s = po.make_frame(rng, mw, mh, ps, extra_rows=2, pad_px=1)   # spare row / pixel take the reference's stray writes
                d = s.copy()
                H.run(po.refplugin("mirrors"), fn, pal, mw, mh, [d], d, [])
                key = "mr|%s|%d|%d|%d" % (fn, pal, mw, mh)
                pl[key + "|a"] = s
                pl[key + "|o"] = d
                recs.append(key)
    pl["records"] = np.array(recs)
    np.savez_compressed(os.path.join(OUT, "plugins.npz"), **pl)
    manifest["groups"]["plugins.npz"] = ("reference plugins built unmodified: lives-plugins/weed-plugins/simple_blend.c (sb), multi_blends.c (mb), mirrors.c (mr, in place), "
                                         "scripts/colorkey.script via build-weed-plugin-C (ck); one process_func call, no threading; 18x8 frames")

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    import gen_golden_stencils          # softlight.c / edge.c fixtures (own seed stream, own file)
    gen_golden_stencils.main()
    import gen_golden_palette           # K3 / K4 conversions (own seed stream, own file)
    gen_golden_palette.main()
    import gen_golden_comp              # compositor paint loop
    gen_golden_comp.main()
    import gen_golden_lut16             # create_gamma_lut + K2 with the LUT16 fused
    gen_golden_lut16.main()
    import gen_golden_blurzoom          # stateful blurzoom over frame sequences
    gen_golden_blurzoom.main()
    import gen_golden_repack            # YUV -> YUV repacks (K5b)
    gen_golden_repack.main()
    import gen_golden_rgbdelay          # stateful RGBdelay / YUVdelay over frame sequences
    gen_golden_rgbdelay.main()
    import gen_golden_scriptfx          # negate / posterise / ccorrect (script-generated plugins)
    gen_golden_scriptfx.main()
    import gen_golden_premult_yuv       # clamped-YUV premultiply tables
    gen_golden_premult_yuv.main()
    import gen_golden_yuv411            # YUV411 -> RGB family
    gen_golden_yuv411.main()
    import gen_golden_rgb411            # RGB family -> YUV411
    gen_golden_rgb411.main()
    tot = sum(os.path.getsize(os.path.join(OUT, x)) for x in os.listdir(OUT))
    print("wrote", sorted(os.listdir(OUT)), "total %d KB" % (tot // 1024))


if __name__ == "__main__":
    main()
