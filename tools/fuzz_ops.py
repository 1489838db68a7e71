#!/usr/bin/env python3
"""tools/fuzz_ops.py [iterations] [seed] -- randomized differential run: entry points whose geometry / stride / alignment handling has several
code paths (resize, chain, gauss5, swizzle, K2, repacks, deinterlace, letterbox) on random sizes, strides and parameters, GPU against the CPU
oracle, bit for bit.  TEST / DEBUG TOOL (uses oracle/); prints the first mismatch with everything needed to replay it."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None        # optional: restrict the run to these kinds
    import torch
    from lives_amd import ops
    from lives_amd.lib import LgpuError, load
    lib = load()
    from oracle import pyoracle as po
    ops.init(0)
    orc = po.oracle()
    P = po.P
    rng = np.random.default_rng(seed)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()

    def fr(w, h, ps, extra=None):
        al = int(rng.choice([1, 4, 8, 16, 32]))
        stride = (w * ps + al - 1) // al * al + (int(rng.integers(0, 3)) * al if extra is None else extra)
        a = rng.integers(0, 256, (h, stride), dtype=np.uint8)
        if ps == 4 and rng.random() < 0.5:
            a[:, 3::4][rng.random((h, a[:, 3::4].shape[1])) < 0.6] = 255
        return a

    def same(got, want, nbytes, rows, what):
        if not (got[:rows, :nbytes] == want[:rows, :nbytes]).all():
            idx = np.argwhere(got[:rows, :nbytes] != want[:rows, :nbytes])[0]
            print("MISMATCH", what, "first at", idx.tolist(), int(got[tuple(idx)]), int(want[tuple(idx)]), flush=True)
            return False
        return True

    lutl = np.zeros(256, np.uint8)
    orc.orc_gamma_lut8(1.0, po.GAMMA_LINEAR, po.GAMMA_SRGB, 1.4, P(lutl))
    bad = 0
    counts = {}
    for it in range(iters):
        kind = str(rng.choice(["resize", "pbbatch", "fxbatch", "chain", "gauss5", "swizzle", "k2", "repack", "deint", "letterbox", "edge", "softlight", "blend", "mirror", "rgb2yuv", "yuv2rgb", "transition", "premult", "premultyuva", "yuv411", "rgb411", "luma", "multi", "colorkey", "gamma", "bytelut", "slide", "tsplit", "repack411", "pixbuf", "pixbuf", "chainpb", "chainpb", "canvas", "c4"]))
        if only and kind not in only:
            continue
        counts[kind] = counts.get(kind, 0) + 1
        ops.tuning("PBH_ALIGNED", 1 if rng.random() < 0.5 else 0)          # both strip forms of k_pb_half at every size
        ops.tuning("PBH_ORDER", int(rng.choice([1, 2])))                    # the two work orders of k_pb_half
        ops.tuning("PBH_GROUP", int(rng.choice([1, 2, 3, 5, 8])) if rng.random() < 0.5 else None)      # bands an XCD takes per turn (None: an eighth of the track's when that is whole)
        ops.tuning("PBH_OCC", int(rng.choice([0, 2, 5])) if rng.random() < 0.3 else None)               # workgroups per CU
        ops.tuning("PBH_TH", int(rng.choice([1, 2, 3, 5, 6, 7, 12, 24])) if rng.random() < 0.5 else None)       # forced band heights
        try:
            if kind == "resize":
                ps = int(rng.choice([1, 3, 4]))
                sw, sh = int(rng.integers(4, 400)), int(rng.integers(4, 200))
                if rng.random() < 0.4:
                    dw, dh = max(2, sw // 2), max(2, sh // 2)
                    sw, sh = dw * 2, dh * 2
                else:
                    dw, dh = int(rng.integers(2, 400)), int(rng.integers(2, 200))
                interp = int(rng.choice([1, 2, 3]))
                src = fr(sw, sh, ps)
                want = np.zeros((dh, (dw * ps + 31) // 32 * 32), np.uint8)
                if orc.orc_resize(P(src), src.strides[0], sw, sh, P(want), want.strides[0], dw, dh, ps, interp) != 0:
                    continue
                d = dev(np.zeros_like(want))
                forced = rng.random() < 0.35            # the persistent general-ratio kernel (k_sep2p) on small frames too
                if forced:
                    ops.tuning("SEP2P_FORCE", 1)
                    counts["resize(k_sep2p forced)"] = counts.get("resize(k_sep2p forced)", 0) + 1
                try:
                    ops.resize(dev(src), d, sw, sh, dw, dh, psize=ps, interp=interp)
                finally:
                    ops.tuning("SEP2P_FORCE", None)
                ok = same(host(d), want, dw * ps, dh, "resize %dx%d->%dx%d ps=%d interp=%d stride=%d forced=%d" % (sw, sh, dw, dh, ps, interp, src.strides[0], forced))
            elif kind == "chain":
                dw, dh = int(rng.integers(2, 200)), int(rng.integers(2, 120))
                if rng.random() < 0.6:
                    sw, sh = 2 * dw, 2 * dh
                else:
                    sw, sh = int(rng.integers(4, 400)), int(rng.integers(4, 240))
                swap, blur, bf, ntr = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 256)), int(rng.integers(1, 4))
                use_lut = rng.random() < 0.7
                srcs = [fr(sw, sh, 4, extra=0) for _ in range(ntr)]
                srcs = [s if s.strides[0] == srcs[0].strides[0] else np.ascontiguousarray(np.pad(s[:, :sw * 4], ((0, 0), (0, srcs[0].strides[0] - sw * 4)))) for s in srcs]
                l2s = [fr(dw, dh, 4, extra=0) for _ in range(ntr)]
                l2s = [s if s.strides[0] == l2s[0].strides[0] else np.ascontiguousarray(np.pad(s[:, :dw * 4], ((0, 0), (0, l2s[0].strides[0] - dw * 4)))) for s in l2s]
                orow = (dw * 4 + 31) // 32 * 32
                lut = lutl if use_lut else None
                wants = []
                for i in range(ntr):
                    w_ = np.zeros((dh, orow), np.uint8)
                    if orc.orc_chain(P(srcs[i]), srcs[i].strides[0], sw, sh, P(l2s[i]), l2s[i].strides[0], P(w_), orow, dw, dh, swap, 3, blur, bf, P(lut) if use_lut else None) != 0:
                        wants = None
                        break
                    wants.append(w_)
                if wants is None:
                    continue
                dd = [dev(np.zeros((dh, orow), np.uint8)) for _ in range(ntr)]
                # half of the cases hand the blend amount over the way bench.py does: in a device-resident parameter block (int32[4], [0] = bf) that the
                # kernel reads, with a decoy value in the kernarg field
                via_block = rng.random() < 0.5
                if via_block:
                    counts["chain(param_block)"] = counts.get("chain(param_block)", 0) + 1
                pb = torch.tensor([bf, 0, 0, 0], dtype=torch.int32, device="cuda") if via_block else None
                prm = ops.chain_params(sw, sh, srcs[0].strides[0], dw, dh, l2s[0].strides[0], orow, swap_rb=swap, interp=3, do_blur=blur,
                                       bf=(bf * 7 + 13) % 256 if via_block else bf, lut=lut, param_block=pb)
                forced = rng.random() < 0.35
                if forced:
                    ops.tuning("SEP2P_FORCE", 1)
                try:
                    ops.chain(prm, ops.chain_tracks([dev(s) for s in srcs], [dev(s) for s in l2s], dd))
                finally:
                    ops.tuning("SEP2P_FORCE", None)
                ok = all(same(host(dd[i]), wants[i], dw * 4, dh, "chain %dx%d->%dx%d swap=%d blur=%d bf=%d lut=%d strides=%d/%d track %d param_block=%d" %
                              (sw, sh, dw, dh, swap, blur, bf, use_lut, srcs[0].strides[0], l2s[0].strides[0], i, via_block)) for i in range(ntr))
            elif kind == "pixbuf":
                # gdk-pixbuf arithmetic (pinned): every interp, 3 / 4 channels, ratios from 1:12 to 12:1, exact 2:1 (k_pb_half) in a third of the cases
                ch, interp = int(rng.choice([3, 4])), int(rng.choice([0, 2, 3]))
                sw, sh = int(rng.integers(1, 500)), int(rng.integers(1, 260))
                u = rng.random()
                if u < 0.33:
                    dw, dh = max(1, sw // 2), max(1, sh // 2)
                    sw, sh = 2 * dw, 2 * dh
                elif u < 0.5:                     # exact 1:2 (k_pb_double)
                    sw, sh = max(2, (sw // 2) & ~1), max(1, sh // 2)
                    dw, dh = 2 * sw, 2 * sh
                elif u < 0.65:                    # integer reductions (k_pb_gather)
                    fx, fy = int(rng.integers(2, 7)), int(rng.integers(2, 7))
                    dw, dh = max(1, sw // fx), max(1, sh // fy)
                    sw, sh = fx * dw, fy * dh
                else:
                    dw, dh = int(rng.integers(1, 600)), int(rng.integers(1, 300))
                al = 16 if rng.random() < 0.5 else 4
                irow = (sw * ch + al - 1) // al * al
                src = rng.integers(0, 256, (sh, irow), dtype=np.uint8)
                if ch == 4:
                    am = rng.random()
                    a = src[:, 3:sw * 4:4]
                    if am < 0.3:
                        a[:] = 255
                    elif am < 0.7:
                        a[rng.random(a.shape) < 0.4] = 0
                        a[rng.random(a.shape) < 0.3] = 255
                orow = (dw * ch + al - 1) // al * al
                want = np.zeros((dh, orow), np.uint8)
                rc = orc.orc_pixbuf_scale(P(src), irow, sw, sh, P(want), orow, dw, dh, ch, interp)
                if rc == -2:
                    continue
                d = dev(np.zeros((dh, orow), np.uint8))
                if rng.random() < 0.15:
                    ops.tuning("PB_NO_PAIRS", 1)
                try:
                    ops.pixbuf_scale(dev(src), d, sw, sh, dw, dh, channels=ch, interp=interp)
                finally:
                    ops.tuning("PB_NO_PAIRS", None)
                ok = same(host(d), want, dw * ch, dh, "pixbuf %dx%d->%dx%d ch=%d interp=%d strides %d/%d" % (sw, sh, dw, dh, ch, interp, irow, orow))
            elif kind in ("chainpb", "canvas"):
                # the chain on the pinned arithmetic: fused (exact aligned 2:1: k_pb_half, with and without the blur stage, with a letterbox canvas) and staged
                dw, dh = int(rng.integers(2, 260)) & ~1, int(rng.integers(2, 140))
                dw = max(dw, 2)
                if rng.random() < 0.7:
                    sw, sh = 2 * dw, 2 * dh
                else:
                    sw, sh = int(rng.integers(4, 500)), int(rng.integers(4, 260))
                interp = int(rng.choice([2, 3])) | 0x100
                swap, bf, ntr = int(rng.integers(0, 2)), int(rng.integers(0, 256)), int(rng.integers(1, 4))
                use_lut = rng.random() < 0.7
                blur = int(rng.integers(0, 2)) if kind == "chainpb" else 0
                if kind == "canvas":
                    nw, nh = dw + int(rng.integers(0, 40)), dh + int(rng.integers(0, 30))
                    ox, oy = int(rng.integers(0, nw - dw + 1)), int(rng.integers(0, nh - dh + 1))
                    if rng.random() < 0.7:
                        ox &= ~1
                else:
                    nw, nh, ox, oy = dw, dh, 0, 0
                al = 16 if rng.random() < 0.7 else 4
                irow, crow = (sw * 4 + al - 1) // al * al, (nw * 4 + al - 1) // al * al
                srcs = [rng.integers(0, 256, (sh, irow), dtype=np.uint8) for _ in range(ntr)]
                for s_ in srcs:
                    a = s_[:, 3:sw * 4:4]
                    a[rng.random(a.shape) < 0.4] = 255
                    a[rng.random(a.shape) < 0.1] = 0
                l2s = [rng.integers(0, 256, (nh, crow), dtype=np.uint8) for _ in range(ntr)]
                for l_ in l2s:
                    l_[:, 3:nw * 4:4][rng.random((nh, nw)) < 0.5] = 255
                lut = lutl if use_lut else None
                wants = []
                for i in range(ntr):
                    if kind == "chainpb":
                        w_ = np.zeros((dh, crow), np.uint8)
                        if orc.orc_chain(P(srcs[i]), irow, sw, sh, P(l2s[i]), crow, P(w_), crow, dw, dh, swap, interp, blur, bf, P(lut) if use_lut else None) != 0:
                            wants = None            # a reduction past gdk-pixbuf's one-step range
                            break
                    else:
                        conv = np.zeros((sh, sw * 4), np.uint8)
                        if swap:
                            orc.orc_swizzle(4, 0, P(srcs[i]), irow, P(conv), sw * 4, sw, sh, None)
                        else:
                            conv[:] = srcs[i][:, :sw * 4]
                        rs = np.zeros((dh, dw * 4), np.uint8)
                        if orc.orc_pixbuf_scale(P(conv), sw * 4, sw, sh, P(rs), dw * 4, dw, dh, 4, interp & 0xFF) != 0:
                            wants = None
                            break
                        cv = np.zeros((nh, nw, 4), np.uint8)
                        cv[..., 3] = 255
                        cv[oy:oy + dh, ox:ox + dw] = rs.reshape(dh, dw, 4)
                        w_ = np.zeros((nh, crow), np.uint8)
                        w_[:, :nw * 4] = cv.reshape(nh, nw * 4)
                        orc.orc_blend_chroma(P(w_), crow, P(l2s[i]), crow, P(w_), crow, nw, nh, 4, 0, bf)
                        if use_lut:
                            orc.orc_gamma_apply(P(w_), crow, nw, nh, 4, 0, P(lut))
                    wants.append(w_)
                if wants is None:
                    continue
                dd = [dev(np.zeros((nh, crow), np.uint8)) for _ in range(ntr)]
                via_block = rng.random() < 0.4
                pb = torch.tensor([bf, 0, 0, 0], dtype=torch.int32, device="cuda") if via_block else None
                prm = ops.chain_params(sw, sh, irow, dw, dh, crow, crow, swap_rb=swap, interp=interp, do_blur=blur, bf=(bf * 7 + 13) % 256 if via_block else bf, lut=lut, param_block=pb)
                trk = ops.chain_tracks([dev(s_) for s_ in srcs], [dev(s_) for s_ in l2s], dd)
                if kind == "canvas":
                    ops.chain_canvas(prm, trk, nw, nh, ox, oy)
                else:
                    ops.chain(prm, trk)
                ok = all(same(host(dd[i]), wants[i], nw * 4, nh, "%s %dx%d->%dx%d in %dx%d at (%d,%d) interp=%x swap=%d blur=%d bf=%d lut=%d strides %d/%d track %d block=%d" %
                              (kind, sw, sh, dw, dh, nw, nh, ox, oy, interp, swap, blur, bf, use_lut, irow, crow, i, via_block)) for i in range(ntr))
            elif kind == "c4":
                ps = int(rng.choice([3, 4]))
                w, h = 4 * int(rng.integers(1, 130)), int(rng.integers(1, 120))
                rs_ = (w * ps + 15) // 16 * 16
                a = rng.integers(0, 256, (h, rs_), dtype=np.uint8)
                b = rng.integers(0, 256, (h, rs_), dtype=np.uint8)
                a[:, :(w // 2) * ps] = rng.integers(90, 170, (h, (w // 2) * ps), dtype=np.uint8)
                is_bgr, delta, opac = int(rng.integers(0, 2)), float(rng.random()), float(rng.random())
                col = [int(v) for v in rng.integers(100, 160, 3)]
                bl = np.zeros_like(a)
                orc.orc_gauss5(P(a), rs_, P(bl), rs_, w, h, ps)
                want = np.zeros_like(a)
                if ps == 3:
                    orc.orc_colorkey(P(bl), rs_, P(b), rs_, P(want), rs_, w, h, is_bgr, delta, opac, col[0], col[1], col[2], 0)
                else:
                    orc.orc_colorkey4(P(bl), rs_, P(b), rs_, P(want), rs_, w, h, is_bgr, delta, opac, col[0], col[1], col[2])
                d = dev(np.zeros_like(a))
                ops.gauss5_colorkey(dev(a), dev(b), d, w, h, ps, is_bgr, delta, opac, col)
                ok = same(host(d), want, w * ps, h, "c4 %dx%d ps=%d bgr=%d delta=%r opac=%r col=%s" % (w, h, ps, is_bgr, delta, opac, col))
            elif kind == "gauss5":
                ps = int(rng.choice([1, 3, 4]))
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 150))
                src = fr(w, h, ps)
                want = np.zeros_like(src)
                orc.orc_gauss5(P(src), src.strides[0], P(want), want.strides[0], w, h, ps)
                d = dev(np.zeros_like(src))
                ops.gauss5(dev(src), d, w, h, psize=ps)
                ok = same(host(d), want, w * ps, h, "gauss5 %dx%d ps=%d stride=%d" % (w, h, ps, src.strides[0]))
            elif kind == "swizzle":
                op = int(rng.integers(0, 13))
                ib, ob = po.OP_IBPP[op], po.OP_OBPP[op]
                w, h = int(rng.integers(1, 500)), int(rng.integers(1, 60))
                src = fr(w, h, ib)
                want = np.full((h, (w * ob + 3) // 4 * 4 + int(rng.integers(0, 3)) * 4), 0xAB, np.uint8)
                af = int(rng.integers(0, 2)) if op in (po.OPS.index("swap4"), po.OPS.index("swapprepost")) else 0
                lut = rng.integers(0, 256, 256, dtype=np.uint8) if rng.random() < 0.5 else None
                orc.orc_swizzle(op, af, P(src), src.strides[0], P(want), want.strides[0], w, h, P(lut) if lut is not None else None)
                d = dev(np.full_like(want, 0xAB))
                ops.swizzle(op, dev(src), d, w, h, alpha_first=af, lut=lut)
                ok = same(host(d), want, want.shape[1], h, "swizzle op=%d %dx%d af=%d lut=%d strides %d->%d" % (op, w, h, af, lut is not None, src.strides[0], want.strides[0]))
            elif kind == "k2":
                w, h = 2 * int(rng.integers(1, 200)), int(rng.integers(1, 100))
                is422 = int(rng.integers(0, 2))
                ys = (w + 7) // 8 * 8 + 8 * int(rng.integers(0, 3))
                cs = ys // 2
                ch = h if is422 else max(1, h // 2)
                if not is422 and h < 2:
                    continue
                Y = rng.integers(0, 256, (h, ys), dtype=np.uint8)
                U, V = (rng.integers(0, 256, (ch, cs), dtype=np.uint8) for _ in range(2))
                opsz, which, q = int(rng.choice([3, 4])), int(rng.integers(0, 4)), int(rng.choice([1, 2, 3]))
                lib.lgpu_yuv420_tuning(int(rng.choice([0, 1, 2, 4])), int(rng.choice([256, 512, 1024])), int(rng.choice([1, 8])))
                orow = (w * opsz + 15) // 16 * 16
                st = (ctypes.c_int * 3)(ys, cs, cs)
                want = np.full((h, orow), 0xAB, np.uint8)
                orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), st, U.size, V.size, P(want), orow, w, h, opsz, 0, is422, which, q, P(lutl), 0)
                d = dev(np.full_like(want, 0xAB))
                ops.yuv420p_to_rgb(dev(Y), dev(U), dev(V), d, w, h, opsize=opsz, is_422=is422, which_tables=which, pb_quality=q, lut=lutl)
                ok = same(host(d), want, w * opsz, h, "k2 %dx%d 422=%d ops=%d which=%d q=%d ys=%d" % (w, h, is422, opsz, which, q, ys))
            elif kind == "repack":
                ip, op, padok = po.YUV_REPACK_PAIRS[int(rng.integers(0, len(po.YUV_REPACK_PAIRS)))]
                w, h = 2 * int(rng.integers(1, 150)), 2 * int(rng.integers(1, 60))
                pad = int(rng.choice([0, 4, 24])) if padok else 0
                unc = int(rng.integers(0, 2))
                src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                want = po.yuv_planes(op, w, h, fill=0x5A, pad=pad)
                sp, ss = po.planes_args(src)
                wp, ws = po.planes_args(want)
                if orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, unc, 0) != 0:
                    continue
                dst = [dev(np.full_like(a, 0x5A)) for a in want]
                ops.yuv_repack(ip, op, [dev(a) for a in src], dst, w, h, unc)
                ok = all(same(host(dst[i]), want[i], want[i].shape[1], want[i].shape[0], "repack %d->%d %dx%d pad=%d unc=%d plane %d" % (ip, op, w, h, pad, unc, i)) for i in range(len(want)))
            elif kind == "repack411":
                ip, op, padok = po.YUV411_REPACK_PAIRS[int(rng.integers(0, len(po.YUV411_REPACK_PAIRS)))]
                w, h = 4 * int(rng.integers(1, 120)), int(rng.integers(1, 80))
                if ip in (512, 513):
                    h += h & 1
                pad = int(rng.choice([0, 4, 24])) if padok else 0
                unc = int(rng.integers(0, 2))
                src = po.yuv_planes(ip, w, h, rng=rng, pad=pad)
                want = po.yuv_planes(op, w, h + ((h & 1) if op in (512, 513) else 0), fill=0x5A, pad=0)
                sp, ss = po.planes_args(src)
                wp, ws = po.planes_args(want)
                if orc.orc_yuv_repack(ip, op, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(wp), ctypes.addressof(ws), w, h, unc, 0) != 0:
                    continue
                dst = [dev(np.full_like(a, 0x5A)) for a in want]
                ops.yuv_repack(ip, op, [dev(a) for a in src], dst, w, h, unc)
                ok = all(same(host(dst[i]), want[i], want[i].shape[1], want[i].shape[0], "repack411 %d->%d %dx%d pad=%d unc=%d plane %d" % (ip, op, w, h, pad, unc, i)) for i in range(len(want)))
            elif kind == "deint":
                pal = int(rng.choice([1, 2, 588, 3, 4, 589, 564, 565]))
                ps = 3 if pal in (1, 2, 588) else 4
                w, h = int(rng.integers(1, 200)), int(rng.integers(1, 80))
                src = fr(w, h, ps)
                inplace = int(rng.integers(0, 2))
                want = src.copy() if inplace else np.full_like(src, 0x5A)
                a = want if inplace else src
                if orc.orc_deinterlace(P(a), a.strides[0], P(want), want.strides[0], w, h, pal) != 0:
                    continue
                ds = dev(src)
                d = ds if inplace else dev(np.full_like(src, 0x5A))
                ops.deinterlace(ds, d, w, h, pal)
                n = (w + 2) // 3 * 3 * ps
                ok = same(host(d), want, n, h, "deinterlace pal=%d %dx%d inplace=%d stride=%d" % (pal, w, h, inplace, src.strides[0]))
            elif kind == "edge":
                pal, mode = int(rng.integers(1, 6)), int(rng.integers(0, 3))
                ps = 3 if pal <= 2 else 4
                w, h = int(rng.integers(4, 260)), int(rng.integers(4, 140))
                if rng.random() < 0.5:        # whole quads, sometimes several 128 x 32 tiles: with a 16-byte multiple as rowstride these take the quad form (k_edge_map4 / _reduce_otsu / _paint4)
                    w, h = 4 * int(rng.integers(2, 175)), int(rng.integers(4, 300))
                    ops.tuning("EDGE_TH", int(rng.choice([16, 32])))
                inplace = int(rng.integers(0, 2))
                sfr = fr(w, h, ps)
                yy, xx = np.mgrid[0:h, 0:w]
                for c in range(ps):
                    sfr[:, c:w * ps:ps] = ((sfr[:, c:w * ps:ps] >> 3) + (96 * ((xx // 9 + yy // 7 + c) % 2)).astype(np.uint8) + 40).astype(np.uint8)
                d0 = sfr.copy() if inplace else rng.integers(0, 256, sfr.shape, dtype=np.uint8)
                want = d0.copy()
                m16 = np.zeros(w * h, np.int16)
                orc.orc_edge(P(want) if inplace else P(sfr), sfr.strides[0], P(want), want.strides[0], w, h, pal, mode, P(m16), inplace)
                d = dev(d0)
                ops.edge(d if inplace else dev(sfr), d, w, h, pal, mode)
                ok = same(host(d), want, w * ps, h, "edge pal=%d mode=%d %dx%d inplace=%d stride=%d" % (pal, mode, w, h, inplace, sfr.strides[0]))
            elif kind == "softlight":
                pal = int(rng.choice([512, 513, 522, 544, 545]))
                w, h = 2 * int(rng.integers(2, 150)), 2 * int(rng.integers(2, 70))
                unc = int(rng.integers(0, 2))
                cw = w >> 1 if pal in (512, 513, 522) else w
                ch = h >> 1 if pal in (512, 513) else h
                dims = [(w, h), (cw, ch), (cw, ch)] + ([(w, h)] if pal == 545 else [])
                src = [fr(a, b, 1) for (a, b) in dims]
                want = np.full_like(src[0], 0x5A)
                orc.orc_softlight_y(P(src[0]), src[0].strides[0], P(want), want.strides[0], w, h, unc)
                dst = [dev(np.full_like(a, 0x5A)) for a in src]
                ops.softlight([dev(a) for a in src], dst, w, h, pal, unc)
                ok = same(host(dst[0]), want, w, h, "softlight pal=%d %dx%d unc=%d" % (pal, w, h, unc))
                for i in range(1, len(dims)):
                    ok = ok and same(host(dst[i]), src[i], dims[i][0], dims[i][1], "softlight plane %d" % i)
            elif kind == "blend":
                pal = int(rng.integers(1, 6))
                ps = 3 if pal <= 2 else 4
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                bf = int(rng.integers(0, 256))
                inplace = int(rng.integers(0, 2))
                s1, s2 = fr(w, h, ps, extra=0), fr(w, h, ps, extra=32)
                init = s1.copy() if inplace else np.full_like(s1, 0x5A)
                want = init.copy()
                a = want if inplace else s1
                orc.orc_blend_chroma(P(a), a.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, ps, int(pal == 5), bf)
                d1 = dev(s1)
                d = d1 if inplace else dev(init)
                ops.blend_chroma(d1, dev(s2), d, w, h, ps, bf, alpha_first=int(pal == 5))
                ok = same(host(d), want, w * ps, h, "blend_chroma pal=%d %dx%d bf=%d inplace=%d" % (pal, w, h, bf, inplace))
            elif kind == "rgb2yuv":
                in_order, out_fmt = int(rng.integers(0, 3)), int(rng.integers(0, 6))
                if out_fmt >= 4 and in_order == 2:
                    continue
                in_alpha = 1 if in_order == 2 else int(rng.integers(0, 2))
                out_alpha = int(rng.integers(0, 2)) if out_fmt <= 1 else 0
                which = int(rng.integers(0, 4)) if out_fmt >= 4 else int(rng.integers(0, 2))
                w, h = int(rng.integers(1, 200)), int(rng.integers(1, 80))
                if out_fmt >= 2:
                    w, h = 2 * max(1, w // 2), 2 * max(1, h // 2)
                src = fr(w, h, 4 if in_alpha else 3)
                want, dims = po.k4_out_planes(0x5A, w, h, out_fmt, out_alpha, compact=bool(rng.integers(0, 2)))
                wp, ws = po.planes_args(want)
                if orc.orc_rgb_to_yuv(P(src), src.strides[0], w, h, in_order, in_alpha, ctypes.addressof(wp), ctypes.addressof(ws), out_fmt, out_alpha, which) != 0:
                    continue
                got = [dev(np.full_like(a, 0x5A)) for a in want]
                ops.rgb_to_yuv(dev(src), got, w, h, in_order, in_alpha, out_fmt, out_alpha, which)
                ok = all(same(host(got[i]), want[i], a, b, "rgb2yuv order=%d ia=%d fmt=%d oa=%d which=%d %dx%d plane %d" % (in_order, in_alpha, out_fmt, out_alpha, which, w, h, i))
                         for i, (a, b) in enumerate(dims))
            elif kind == "yuv2rgb":
                in_fmt, out_order = int(rng.integers(0, 4)), int(rng.integers(0, 3))
                in_alpha = int(rng.integers(0, 2)) if in_fmt <= 1 else 0
                out_alpha = 1 if out_order == 2 else int(rng.integers(0, 2))
                if in_fmt == 1 and (out_order == 2 or (out_order == 1 and not out_alpha)):
                    continue
                which = int(rng.integers(0, 4)) if in_fmt == 0 else int(rng.integers(0, 2))
                w, h = int(rng.integers(1, 200)), int(rng.integers(1, 80))
                if in_fmt >= 2:
                    w = 2 * max(1, w // 2)
                if in_fmt == 0:
                    planes = [fr(w, h, 4 if in_alpha else 3)]
                elif in_fmt == 1:
                    planes = [fr(w, h, 1, extra=0) for _ in range(4 if in_alpha else 3)]
                    planes = [np.ascontiguousarray(np.pad(a[:, :w], ((0, 0), (0, planes[0].strides[0] - w)))) for a in planes]
                else:
                    planes = [fr(w, h, 2)]
                opsz = 4 if (out_order == 2 or out_alpha) else 3
                want = np.full((h, (w * opsz + 31) // 32 * 32), 0x5A, np.uint8)
                sp, ss = po.planes_args(planes)
                if orc.orc_yuv_to_rgb(ctypes.addressof(sp), ctypes.addressof(ss), w, h, in_fmt, in_alpha, P(want), want.strides[0], out_order, out_alpha, which) != 0:
                    continue
                d = dev(np.full_like(want, 0x5A))
                ops.yuv_to_rgb([dev(a) for a in planes], d, w, h, in_fmt, in_alpha, out_order, out_alpha, which)
                ok = same(host(d), want, w * opsz, h, "yuv2rgb fmt=%d ia=%d order=%d oa=%d which=%d %dx%d" % (in_fmt, in_alpha, out_order, out_alpha, which, w, h))
            elif kind == "transition":
                t, ps = int(rng.integers(0, 3)), int(rng.choice([3, 4]))
                w, h = int(rng.integers(2, 300)), int(rng.integers(2, 120))
                amt = float(rng.choice([0., 1., float(rng.random())]))
                s1, s2 = fr(w, h, ps), fr(w, h, ps)
                inplace = int(rng.integers(0, 2)) if t < 2 else 0
                want = s1.copy() if inplace else np.full_like(s1, 0x5A)
                a = want if inplace else s1
                orc.orc_transition(t, P(a), a.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, ps, amt)
                d1 = dev(s1)
                d = d1 if inplace else dev(np.full_like(s1, 0x5A))
                ops.transition(t, d1, dev(s2), d, w, h, ps, amt)
                ok = same(host(d), want, w * ps, h, "transition %d ps=%d %dx%d amount=%r inplace=%d" % (t, ps, w, h, amt, inplace))
            elif kind == "pbbatch":
                # lgpu_pixbuf_scale_batch: 1 .. 6 frames of one random geometry in one launch, every frame against the oracle
                ch, interp, n = int(rng.choice([3, 4])), int(rng.choice([0, 2, 3])), int(rng.integers(1, 7))
                sw, sh, dw, dh = int(rng.integers(1, 300)), int(rng.integers(1, 160)), int(rng.integers(1, 400)), int(rng.integers(1, 200))
                if rng.random() < 0.3:
                    dw, dh = max(1, sw // 2), max(1, sh // 2)
                    sw, sh = 2 * dw, 2 * dh
                irow, orow = (sw * ch + 15) // 16 * 16, (dw * ch + 15) // 16 * 16
                srcs = [rng.integers(0, 256, (sh, irow), dtype=np.uint8) for _ in range(n)]
                wants, skip = [], False
                for s_ in srcs:
                    w_ = np.zeros((dh, dw * ch), np.uint8)
                    if orc.orc_pixbuf_scale(P(s_), irow, sw, sh, P(w_), dw * ch, dw, dh, ch, interp) != 0:
                        skip = True
                    wants.append(w_)
                ok = True
                if not skip:
                    dd = [dev(np.full((dh, orow), 0x5A, np.uint8)) for _ in range(n)]
                    ops.pixbuf_scale_batch([dev(s_) for s_ in srcs], dd, sw, sh, dw, dh, channels=ch, interp=interp)
                    for f in range(n):
                        ok = ok and same(host(dd[f]), wants[f], dw * ch, dh, "pbbatch frame %d of %d: %dch interp %d %dx%d->%dx%d" % (f, n, ch, interp, sw, sh, dw, dh))
            elif kind == "fxbatch":
                # lgpu_fx_batch, transitions: 1 .. 6 frames in one launch against the oracle
                t, ps, n = int(rng.integers(0, 3)), int(rng.choice([3, 4])), int(rng.integers(1, 7))
                w, h = int(rng.integers(2, 300)), int(rng.integers(2, 120))
                amt = float(rng.choice([0., 1., float(rng.random())]))
                stride = (w * ps + 15) // 16 * 16
                s1 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
                s2 = [rng.integers(0, 256, (h, stride), dtype=np.uint8) for _ in range(n)]
                dd = [dev(np.full((h, stride), 0x5A, np.uint8)) for _ in range(n)]
                ops.fx_batch(ops.FX_TRANSITION, [[dev(a)] for a in s1], [[d_] for d_ in dd], w, h, ins1=[[dev(a)] for a in s2], ip=(t, ps), dp=(amt,))
                ok = True
                for f in range(n):
                    want = np.full((h, stride), 0x5A, np.uint8)
                    orc.orc_transition(t, P(s1[f]), stride, P(s2[f]), stride, P(want), stride, w, h, ps, amt)
                    ok = ok and same(host(dd[f]), want, w * ps, h, "fxbatch transition %d ps=%d %dx%d amount=%r frame %d of %d" % (t, ps, w, h, amt, f, n))
            elif kind == "premult":
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                af, un = int(rng.integers(0, 2)), int(rng.integers(0, 2))
                sfr = fr(w, h, 4, extra=0)
                if sfr.strides[0] & 3:
                    continue
                want = sfr.copy()
                orc.orc_alpha_premult(P(want), want.strides[0], w, h, af, un)
                d = dev(sfr)
                ops.alpha_premult(d, w, h, alpha_first=af, un=un)
                ok = same(host(d), want, w * 4, h, "premult %dx%d af=%d un=%d" % (w, h, af, un))
            elif kind == "premultyuva":
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                pal, clamped, un = int(rng.choice([589, 545])), int(rng.integers(0, 2)), int(rng.integers(0, 2))
                planes = [fr(w, h, 4)] if pal == 589 else [fr(w, h, 1) for _ in range(4)]
                if pal == 589 and planes[0].strides[0] & 3:
                    continue
                want = [x.copy() for x in planes]
                pp = (ctypes.c_void_p * 4)(*([x.ctypes.data for x in want] + [None] * (4 - len(want))))
                ss = (ctypes.c_int * 4)(*([x.strides[0] for x in want] + [0] * (4 - len(want))))
                orc.orc_alpha_premult_yuva(pp, ss, w, h, pal, clamped, un)
                ds = [dev(x) for x in planes]
                ops.alpha_premult_yuva(ds, w, h, pal, clamped, un=un)
                ok = all(same(host(d), wt, w * (4 if pal == 589 else 1), h, "premult yuva %d %dx%d clamped=%d un=%d" % (pal, w, h, clamped, un))
                         for d, wt in zip(ds, want))
            elif kind == "yuv411":
                wm, h = int(rng.integers(1, 200)), int(rng.integers(1, 60))
                order, uncl = int(rng.integers(0, 3)), int(rng.integers(0, 2))
                oa = 1 if order == 2 else int(rng.integers(0, 2))
                ps = 4 if oa else 3
                src = rng.integers(0, 256, (h, wm * 6), dtype=np.uint8)
                init = rng.integers(0, 256, (h, wm * 4 * ps + int(rng.integers(0, 5)) * 4), dtype=np.uint8)
                want = init.copy()
                orc.orc_yuv411_to_rgb(P(src), wm, h, P(want), want.strides[0], order, oa, uncl)
                d = dev(init)
                ops.yuv411_to_rgb(dev(src), d, wm, h, out_order=order, out_alpha=oa, unclamped=uncl)
                ok = same(host(d), want, want.shape[1], h, "yuv411 %dx%d order=%d alpha=%d unclamped=%d" % (wm, h, order, oa, uncl))
            elif kind == "rgb411":
                w, h = int(rng.integers(4, 700)), int(rng.integers(1, 60))
                order, uncl = int(rng.integers(0, 3)), int(rng.integers(0, 2))
                ia = 1 if order == 2 else int(rng.integers(0, 2))
                src = rng.integers(0, 256, (h, w * (4 if ia else 3) + int(rng.integers(0, 9))), dtype=np.uint8)
                want = np.full((h, (w >> 2) * 6), 0x5A, np.uint8)
                orc.orc_rgb_to_yuv411(P(src), src.strides[0], w, h, order, ia, P(want), uncl)
                d = dev(np.full_like(want, 0x5A))
                ops.rgb_to_yuv411(dev(src), d, w, h, in_order=order, in_alpha=ia, unclamped=uncl)
                ok = same(host(d), want, want.shape[1], h, "rgb411 %dx%d order=%d alpha=%d unclamped=%d" % (w, h, order, ia, uncl))
            elif kind == "luma":
                pal = int(rng.integers(1, 5))
                ps, order = (3 if pal <= 2 else 4), (0 if pal in (1, 3) else 1)
                k2, thr, inplace = int(rng.integers(1, 5)), int(rng.integers(0, 256)), int(rng.integers(0, 2))
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                s1, s2 = fr(w, h, ps), fr(w, h, ps)
                init = s1.copy() if inplace else np.full_like(s1, 0x5A)
                want = init.copy()
                orc.orc_blend_luma(k2, P(want if inplace else s1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, ps, order, thr, inplace)
                d1 = dev(s1)
                d = d1 if inplace else dev(init)
                ops.blend_luma(k2, d1, dev(s2), d, w, h, ps, order, thr)
                ok = same(host(d), want, w * ps, h, "blend_luma kind=%d pal=%d %dx%d thr=%d inplace=%d" % (k2, pal, w, h, thr, inplace))
            elif kind == "multi":
                k2, is_bgr, bf = int(rng.integers(0, 7)), int(rng.integers(0, 2)), int(rng.integers(0, 256))
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                s1, s2 = fr(w, h, 3), fr(w, h, 3)
                want = np.full_like(s1, 0x5A)
                orc.orc_blend_multi(k2, P(s1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, is_bgr, bf)
                d = dev(np.full_like(s1, 0x5A))
                ops.blend_multi(k2, dev(s1), dev(s2), d, w, h, is_bgr, bf)
                ok = same(host(d), want, w * 3, h, "blend_multi kind=%d bgr=%d bf=%d %dx%d" % (k2, is_bgr, bf, w, h))
            elif kind == "colorkey":
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                is_bgr, delta, opac = int(rng.integers(0, 2)), float(rng.random()), float(rng.random())
                col = [int(v) for v in rng.integers(0, 256, 3)]
                s0, s1 = fr(w, h, 3), fr(w, h, 3)
                s1[:, :w * 3] = np.where(rng.random((h, w * 3)) < 0.5, np.tile(np.array(col, np.uint8), w)[None, :], s1[:, :w * 3])
                want = np.full_like(s0, 0x5A)
                orc.orc_colorkey(P(s0), s0.strides[0], P(s1), s1.strides[0], P(want), want.strides[0], w, h, is_bgr, delta, opac, col[0], col[1], col[2], 0)
                d = dev(np.full_like(s0, 0x5A))
                ops.colorkey(dev(s0), dev(s1), d, w, h, is_bgr, delta, opac, col)
                ok = same(host(d), want, w * 3, h, "colorkey %dx%d bgr=%d delta=%r opac=%r col=%s" % (w, h, is_bgr, delta, opac, col))
            elif kind == "gamma":
                ps = int(rng.choice([3, 4]))
                w, h = int(rng.integers(2, 300)), int(rng.integers(2, 100))
                x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
                rw, rh = int(rng.integers(1, w - x + 1)), int(rng.integers(1, h - y + 1))
                af = int(rng.integers(0, 2)) if ps == 4 else 0
                lut = rng.integers(0, 256, 256, dtype=np.uint8)
                pix = fr(w, h, ps)
                want = pix.copy()
                orc.orc_gamma_apply(ctypes.c_void_p(want.ctypes.data + y * want.strides[0] + x * ps), want.strides[0], rw, rh, ps, af, P(lut))
                d = dev(pix)
                ops.gamma_apply(d, rw, rh, ps, lut, alpha_first=af, x=x, y=y)
                ok = same(host(d), want, want.shape[1], h, "gamma %dx%d ps=%d sub=(%d,%d,%d,%d) af=%d stride=%d" % (w, h, ps, x, y, rw, rh, af, pix.strides[0]))
            elif kind == "bytelut":
                ps = int(rng.choice([3, 4]))
                w, h = int(rng.integers(1, 400)), int(rng.integers(1, 80))
                luts = rng.integers(0, 256, (ps, 256), dtype=np.uint8)
                inplace = int(rng.integers(0, 2))
                sfr = fr(w, h, ps)
                want = sfr.copy() if inplace else np.full_like(sfr, 0x5A)
                a = want if inplace else sfr
                orc.orc_byte_luts(P(a), a.strides[0], P(want), want.strides[0], w, h, ps, luts.ctypes.data)
                ds = dev(sfr)
                d = ds if inplace else dev(np.full_like(sfr, 0x5A))
                ops.byte_luts(ds, d, w, h, ps, luts)
                ok = same(host(d), want, want.shape[1], h, "byte_luts ps=%d %dx%d inplace=%d stride=%d" % (ps, w, h, inplace, sfr.strides[0]))
            elif kind == "slide":
                ps = int(rng.choice([3, 4]))
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                tv, dirn, mvl, mvu = int(rng.integers(0, 256)), int(rng.integers(1, 5)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
                s1, s2 = fr(w, h, ps), fr(w, h, ps)
                want = np.full_like(s1, 0x5A)
                orc.orc_slide_over(P(s1), s1.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, ps, tv, dirn, mvl, mvu)
                d = dev(np.full_like(s1, 0x5A))
                ops.slide_over(dev(s1), dev(s2), d, w, h, ps, tv, dirn, mvl, mvu)
                ok = same(host(d), want, w * ps, h, "slide ps=%d %dx%d amount=%d dir=%d lower=%d upper=%d" % (ps, w, h, tv, dirn, mvl, mvu))
            elif kind == "tsplit":
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 100))
                start, end, bw = float(rng.random()), float(rng.random()), float(rng.random() * 0.5 * (rng.random() < 0.7))
                sym, vert, is_bgr, inplace = (int(v) for v in rng.integers(0, 2, 4))
                bc = np.array(rng.integers(0, 256, 3), np.int32)
                s1, s2 = fr(w, h, 3), fr(w, h, 3)
                want = s1.copy() if inplace else np.full_like(s1, 0x5A)
                a = want if inplace else s1
                orc.orc_triple_split(P(a), a.strides[0], P(s2), s2.strides[0], P(want), want.strides[0], w, h, is_bgr, start, sym, end, vert, bw, bc.ctypes.data)
                d1 = dev(s1)
                d = d1 if inplace else dev(np.full_like(s1, 0x5A))
                ops.triple_split(d1, dev(s2), d, w, h, is_bgr, start, sym, end, vert, bw, bc)
                ok = same(host(d), want, w * 3, h, "tsplit %dx%d %r sym=%d %r vert=%d bw=%r bgr=%d inplace=%d" % (w, h, start, sym, end, vert, bw, is_bgr, inplace))
            elif kind == "mirror":
                ps, mode = int(rng.choice([3, 4])), int(rng.integers(0, 3))
                w, h = int(rng.integers(1, 300)), int(rng.integers(1, 120))
                inplace = int(rng.integers(0, 2))
                sfr = fr(w, h, ps)
                want = sfr.copy() if inplace else np.full_like(sfr, 0x5A)
                a = want if inplace else sfr
                orc.orc_mirror(mode, P(a), a.strides[0], P(want), want.strides[0], w, h, ps)
                ds = dev(sfr)
                d = ds if inplace else dev(np.full_like(sfr, 0x5A))
                ops.mirror(mode, ds, d, w, h, ps)
                ok = same(host(d), want, w * ps, h, "mirror mode=%d ps=%d %dx%d inplace=%d" % (mode, ps, w, h, inplace))
            else:
                ps = int(rng.choice([1, 3, 4]))
                w, h = int(rng.integers(1, 200)), int(rng.integers(1, 100))
                nw, nh = w + 2 * int(rng.integers(0, 40)), h + 2 * int(rng.integers(0, 40))
                src = fr(w, h, ps)
                black = rng.integers(0, 256, 4, dtype=np.uint8)
                want = np.full((nh, (nw * ps + 31) // 32 * 32), 0x77, np.uint8)
                orc.orc_letterbox(P(src), src.strides[0], w, h, P(want), want.strides[0], nw, nh, ps, P(black))
                d = dev(np.full_like(want, 0x77))
                ops.letterbox(dev(src), d, w, h, nw, nh, ps, black)
                ok = same(host(d), want, want.shape[1], nh, "letterbox %dx%d->%dx%d ps=%d" % (w, h, nw, nh, ps))
        except LgpuError as e:
            print("DECLINED", kind, str(e)[:160], flush=True)
            ok = True
        if not ok:
            bad += 1
            if bad >= 10:
                break
    print("fuzz: %d iterations, %d mismatching cases; per kind %s" % (it + 1, bad, counts))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
