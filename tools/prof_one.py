#!/usr/bin/env python3
"""tools/prof_one.py <op> -- one entry point in a loop for rocprofv3 (--kernel-trace --stats / --pmc); tools/bench_one.py times the same cases by HIP-graph replay.
ops: edge softlight yuv411 premult_yuva composite k2 c3 c4rgb24 c4rgba pb:SWxSH:DWxDH:interp"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from lives_amd import ops   # noqa: E402

NB = 6        # rotating buffer sets: consecutive launches do not read what the one before left in the caches
COLD = False  # tools/bench_one.py --cold: NB is then as many sets as make 1.6 GB


def case(op):
    """-> (fn(i), algorithmic bytes per launch): fn launches the op on buffer set i % NB"""
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    w, h = 1920, 1080

    def rnd(rows, cols):
        return [torch.randint(0, 256, (rows, cols), dtype=torch.uint8, device="cuda", generator=g) for _ in range(NB)]
    if op.startswith("copy"):          # copyN: a plain copy of a 1920 x 1080 frame of N-byte pixels -- the floor of a single-frame launch by this measurement
        n = int(op[4:] or 1)
        from lives_amd import lib
        a_, b_ = rnd(h, w * n), rnd(h, w * n)
        return (lambda i: lib.call("lgpu_copy_rows", b_[i % NB].data_ptr(), w * n, a_[i % NB].data_ptr(), w * n, w * n, h, ops.stream_ptr())), 2 * w * h * n
    if op == "edge":
        src, dst = rnd(h, w * 4), rnd(h, w * 4)
        return (lambda i: ops.edge(src[i % NB], dst[i % NB], w, h, 3, 0)), w * h * 16
    # ---- single-frame entry points without a batch form (1080p): s:<name>
    if op.startswith("s:"):
        k = op[2:]
        if k == "gauss5":
            src, dst = rnd(h, w * 4), rnd(h, w * 4)
            return (lambda i: ops.gauss5(src[i % NB], dst[i % NB], w, h, 4)), w * h * 8
        if k == "resize":                # the polyphase spec (lgpu_resize), 4K -> 1080p BICUBIC class
            src, dst = rnd(2160, 3840 * 4), rnd(h, w * 4)
            return (lambda i: ops.resize(src[i % NB], dst[i % NB], 3840, 2160, w, h, 4, 3)), 3840 * 2160 * 4 + w * h * 4
        if k == "deint":
            src, dst = rnd(h, w * 4), rnd(h, w * 4)
            return (lambda i: ops.deinterlace(src[i % NB], dst[i % NB], w, h, 3)), w * h * 8
        if k == "deint420":
            src, dst = rnd(h, w), rnd(h, w)
            return (lambda i: ops.deinterlace(src[i % NB], dst[i % NB], w, h, 512)), w * h * 2
        if k in ("tsplit", "dissolve", "slide", "transition", "chroma", "luma", "multi", "colorkey"):
            ps = 3 if k in ("tsplit", "multi", "colorkey") else 4
            a_, b_, o = rnd(h, w * ps), rnd(h, w * ps), rnd(h, w * ps)
            nbytes = w * h * ps * 3
            if k == "tsplit":
                return (lambda i: ops.triple_split(a_[i % NB], b_[i % NB], o[i % NB], w, h, 0, 0.3, True, 0.7, False, 0.02, (255, 0, 0))), nbytes
            if k == "dissolve":
                mask = torch.from_numpy(ops.dissolve_mask(1234, w, h)).cuda()
                return (lambda i, keep=mask: ops.dissolve(a_[i % NB], b_[i % NB], o[i % NB], w, h, ps, mask, 0.5)), nbytes + w * h * 4
            if k == "slide":
                return (lambda i: ops.slide_over(a_[i % NB], b_[i % NB], o[i % NB], w, h, ps, 128, 1)), nbytes
            if k == "transition":
                return (lambda i: ops.transition(1, a_[i % NB], b_[i % NB], o[i % NB], w, h, ps, 0.5)), nbytes
            if k == "chroma":
                return (lambda i: ops.blend_chroma(a_[i % NB], b_[i % NB], o[i % NB], w, h, ps, 128)), nbytes
            if k == "luma":
                return (lambda i: ops.blend_luma(1, a_[i % NB], b_[i % NB], o[i % NB], w, h, ps, 0, 128)), nbytes
            if k == "multi":
                return (lambda i: ops.blend_multi(1, a_[i % NB], b_[i % NB], o[i % NB], w, h, 0, 128)), nbytes
            return (lambda i: ops.colorkey(a_[i % NB], b_[i % NB], o[i % NB], w, h, 0, 0.3, 0.8, (128, 128, 128))), nbytes
        if k.startswith("repack"):        # s:repack:<in palette>:<out palette>   (WEED_PALETTE numbers)
            _, ip_, op_ = k.split(":")
            ip_, op_ = int(ip_), int(op_)

            def planes(pal):
                if pal in (512, 513):
                    return [rnd(h, w), rnd(h // 2, w // 2), rnd(h // 2, w // 2)], w * h * 3 // 2
                if pal == 522:
                    return [rnd(h, w), rnd(h, w // 2), rnd(h, w // 2)], w * h * 2
                if pal == 544:
                    return [rnd(h, w), rnd(h, w), rnd(h, w)], w * h * 3
                if pal in (564, 565):
                    return [rnd(h, w * 2)], w * h * 2
                if pal == 588:
                    return [rnd(h, w * 3)], w * h * 3
                if pal == 589:
                    return [rnd(h, w * 4)], w * h * 4
                if pal == 595:
                    return [rnd(h, (w >> 2) * 6)], w * h * 6 // 4
                raise SystemExit("palette " + str(pal))
            sp, sb = planes(ip_)
            dp, db = planes(op_)
            return (lambda i: ops.yuv_repack(ip_, op_, [pl[i % NB] for pl in sp], [pl[i % NB] for pl in dp], w, h)), sb + db
        if k == "clamp":
            ya = rnd(h, w * 2)
            return (lambda i: ops.yuv_switch_clamping([ya[i % NB]], 564, h, 1)), w * h * 4
        if k == "r2y411":
            src, dst = rnd(h, w * 4), rnd(h, (w >> 2) * 6)
            return (lambda i: ops.rgb_to_yuv411(src[i % NB], dst[i % NB], w, h, 0, 1, 0)), w * h * 4 + w * h * 6 // 4
        if k == "r2y444p":
            src, d0, d1, d2 = rnd(h, w * 4), rnd(h, w), rnd(h, w), rnd(h, w)
            return (lambda i: ops.rgb_to_yuv(src[i % NB], [d0[i % NB], d1[i % NB], d2[i % NB]], w, h, 0, 1, 1, 0, 0)), w * h * 7
        if k == "y444p2rgb":
            s0, s1, s2, dst = rnd(h, w), rnd(h, w), rnd(h, w), rnd(h, w * 4)
            return (lambda i: ops.yuv_to_rgb([s0[i % NB], s1[i % NB], s2[i % NB]], dst[i % NB], w, h, 1, 0, 0, 1, 0)), w * h * 7
        if k == "lb":
            src, dst = rnd(h, w * 4), rnd(1200, w * 4)
            return (lambda i: ops.letterbox(src[i % NB], dst[i % NB], w, h, w, 1200, 4, (0, 0, 0, 255))), w * h * 4 + w * 1200 * 4
        raise SystemExit("unknown single op " + op)
    if op == "softlight":
        pl = [[a, b, c] for a, b, c in zip(rnd(h, w), rnd(h // 2, w // 2), rnd(h // 2, w // 2))]
        dl = [[torch.zeros_like(t) for t in p] for p in pl]
        return (lambda i: ops.softlight(pl[i % NB], dl[i % NB], w, h, 512, 0)), w * h * 3
    if op == "yuv411":
        m, o = rnd(h, (w >> 2) * 6), rnd(h, w * 4)
        return (lambda i: ops.yuv411_to_rgb(m[i % NB], o[i % NB], w >> 2, h, out_order=0, out_alpha=1)), w * h * 6 // 4 + w * h * 4
    if op == "premult_yuva":
        ya = rnd(h, w * 4)
        return (lambda i: ops.alpha_premult_yuva([ya[i % NB]], w, h, 589, 1, un=0)), w * h * 8
    if op == "composite":
        lay = [torch.randint(0, 256, (540, 960 * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(8)]
        out = rnd(h, w * 4)
        layers = [(lay[z], 960, 540, 120 * z, 60 * z, 0.75) for z in range(8)]
        return (lambda i: ops.composite(out[i % NB], w, h, 4, layers)), 8 * 960 * 540 * 4 + w * h * 4
    if op == "k2":
        Y, U, V, o = rnd(h, w), rnd(h // 2, w // 2), rnd(h // 2, w // 2), rnd(h, w * 4)
        lut = np.arange(256, dtype=np.uint8)[::-1].copy()
        return (lambda i: ops.yuv420p_to_rgb(Y[i % NB], U[i % NB], V[i % NB], o[i % NB], w, h, lut=lut)), w * h * 3 // 2 + w * h * 4
    if op.startswith("k2b"):           # k2bN: N 1080p frames per launch (lgpu_yuv420p_to_rgb_batch)
        n = int(op[3:])
        nb = 2 if not COLD else max(2, NB // n + 1)
        lut = np.arange(256, dtype=np.uint8)[::-1].copy()
        sets = []
        for _ in range(nb):
            sets.append([(torch.randint(0, 256, (h, w), dtype=torch.uint8, device="cuda", generator=g), torch.randint(0, 256, (h // 2, w // 2), dtype=torch.uint8, device="cuda", generator=g),
                          torch.randint(0, 256, (h // 2, w // 2), dtype=torch.uint8, device="cuda", generator=g), torch.zeros((h, w * 4), dtype=torch.uint8, device="cuda")) for _ in range(n)])
        return (lambda i: ops.yuv420p_to_rgb_batch(sets[i % nb], w, h, lut=lut)), n * (w * h * 3 // 2 + w * h * 4)
    if op in ("c3", "c4rgb24", "c4rgba"):
        W, H = 3840, 2160
        if op == "c3":
            src, l2 = rnd(H, W * 4), rnd(1200, 1920 * 4)
            d = [torch.zeros_like(t) for t in l2]
            prm = ops.chain_params(W, H, W * 4, 1920, 1080, 1920 * 4, 1920 * 4, swap_rb=int(os.environ.get('C3_SWAP', '0')), interp=3 | 0x100, do_blur=0, bf=128, lut=None)
            trk = [ops.chain_tracks([src[i]], [l2[i]], [d[i]]) for i in range(NB)]
            return (lambda i, keep=(src, l2, d): ops.chain_canvas(prm, trk[i % NB], 1920, 1200, 0, 60)), W * H * 4 + 2 * 1920 * 1200 * 4      # keep: the tracks hold raw pointers
        ps = 3 if op == "c4rgb24" else 4
        a, b = rnd(H, W * ps), rnd(H, W * ps)
        o = [torch.zeros_like(t) for t in a]
        delta = float(os.environ.get("C4_DELTA", "0.3"))
        return (lambda i: ops.gauss5_colorkey(a[i % NB], b[i % NB], o[i % NB], W, H, ps, 0, delta, 0.8, (128, 128, 128))), W * H * ps * 3
    if op.startswith("chg"):           # chgN:SWxSH:DWxDH -- the chain (R <-> B, gdk-pixbuf HYPER scale, chroma blend, gamma LUT) on N tracks of any geometry: the staged form off 2:1
        parts = op.split(":")            # chgN:SWxSH:DWxDH[:CWxCH][:nb]: with a letterbox canvas the scaled frame is centred on it; nb: no layer 2 (LGPU_INTERP_NOBLEND)
        nbl = parts[-1] == "nb"
        if nbl:
            parts = parts[:-1]
        head, a_, b_ = parts[:3]
        n = int(head[3:])
        sw, sh = (int(v) for v in a_.split("x"))
        dw, dh = (int(v) for v in b_.split("x"))
        cw, ch = (int(v) for v in parts[3].split("x")) if len(parts) > 3 else (dw, dh)
        nb = 2 if not COLD else max(2, NB // n + 1)
        lut = np.arange(256, dtype=np.uint8)[::-1].copy()
        prm = ops.chain_params(sw, sh, sw * 4, dw, dh, cw * 4, cw * 4, swap_rb=1, interp=3 | 0x100 | (0x400 if nbl else 0), do_blur=0, bf=128, lut=lut)
        sets = []
        for _ in range(nb):
            srcs = [torch.randint(0, 256, (sh, sw * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)]
            l2s = [torch.randint(0, 256, (ch, cw * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)]
            ds = [torch.zeros_like(t) for t in l2s]
            sets.append((ops.chain_tracks(srcs, l2s, ds), srcs, l2s, ds))
        if nbl:
            cvs = (cw, ch, (cw - dw + 1) >> 1, (ch - dh + 1) >> 1) if len(parts) > 3 else None
            return (lambda i: ops.chain_amounts(prm, sets[i % nb][0], None, cvs)), n * (sw * sh * 4 + cw * ch * 4)
        if len(parts) > 3:
            return (lambda i: ops.chain_canvas(prm, sets[i % nb][0], cw, ch, (cw - dw + 1) >> 1, (ch - dh + 1) >> 1)), n * (sw * sh * 4 + 2 * cw * ch * 4)
        return (lambda i: ops.chain(prm, sets[i % nb][0])), n * (sw * sh * 4 + 2 * dw * dh * 4)
    if op.startswith("chain"):         # chainN: the headline chain on N tracks per launch; chainblurN: with config 5's gaussian; chainbluropN: that, all-opaque sources + LGPU_INTERP_OPAQUE
        blur = op.startswith("chainblur")
        opaque = op.startswith("chainblurop")
        n = int(op[len("chainblurop" if opaque else "chainblur" if blur else "chain"):] or 1)
        W, H = 3840, 2160
        nb = NB if COLD else 2             # sets of n tracks
        sets = []
        for _ in range(nb):
            srcs = [torch.randint(0, 256, (H, W * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)]
            if opaque:
                for t_ in srcs:
                    t_[:, 3::4] = 255
            l2s = [torch.randint(0, 256, (1080, 1920 * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)]
            ds = [torch.zeros_like(t) for t in l2s]
            sets.append((srcs, l2s, ds, ops.chain_tracks(srcs, l2s, ds)))
        prm = ops.chain_params(W, H, W * 4, 1920, 1080, 1920 * 4, 1920 * 4, swap_rb=int(os.environ.get('C3_SWAP', '1')), interp=3 | 0x100 | (0x200 if opaque else 0), do_blur=1 if blur else 0, bf=128,
                               lut=np.arange(256, dtype=np.uint8))
        return (lambda i: ops.chain(prm, sets[i % nb][3])), n * (W * H * 4 + 2 * 1920 * 1080 * 4)
    if op.startswith("fx") and ":" in op:       # fxN:softlight | fxN:yuv411 | fxN:transition | fxN:chroma | fxN:luma | fxN:multi -- N frames per launch through lgpu_fx_batch
        n = int(op[2:op.index(":")])
        kind = op.split(":")[1]
        nb = 2 if not COLD else max(2, NB // n + 1)

        def frames(rows, cols):
            return [[torch.randint(0, 256, (rows, cols), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)] for _ in range(nb)]
        if kind == "softlight":
            Y, U, V = frames(h, w), frames(h // 2, w // 2), frames(h // 2, w // 2)
            ins = [[[Y[s][f], U[s][f], V[s][f]] for f in range(n)] for s in range(nb)]
            outs = [[[torch.zeros_like(t) for t in fr] for fr in st_] for st_ in ins]
            return (lambda i: ops.fx_batch(ops.FX_SOFTLIGHT, ins[i % nb], outs[i % nb], w, h, palette=512, ip=(0,))), n * w * h * 3
        if kind == "yuv411":
            m, o = frames(h, (w >> 2) * 6), frames(h, w * 4)
            return (lambda i: ops.fx_batch(ops.FX_YUV411_TO_RGB, [[t] for t in m[i % nb]], [[t] for t in o[i % nb]], w >> 2, h, ip=(0, 1, 0))), n * (w * h * 6 // 4 + w * h * 4)
        if kind == "transition":
            a_, b_, o = frames(h, w * 4), frames(h, w * 4), frames(h, w * 4)
            return (lambda i: ops.fx_batch(ops.FX_TRANSITION, [[t] for t in a_[i % nb]], [[t] for t in o[i % nb]], w, h, ins1=[[t] for t in b_[i % nb]], ip=(1, 4), dp=(0.5,))), n * w * h * 8
        inplace = kind.endswith("_ip")          # chroma_ip / luma_ip: out channel = in channel 0 (what a host does with CHANNEL_CAN_DO_INPLACE): two reads, one write;
        if inplace:                              # out of place the 4-byte blends also READ the destination, whose alpha byte the reference never writes
            kind = kind[:-3]
        if kind in ("chroma", "luma", "multi"):        # the two-input blends: bytes = two frames read, one written
            ps = 3 if kind == "multi" else 4
            a_, b_, o = frames(h, w * ps), frames(h, w * ps), frames(h, w * ps)
            if inplace:
                o = a_
            op, ip = {"chroma": (ops.FX_BLEND_CHROMA, (4, 0)), "luma": (ops.FX_BLEND_LUMA, (1, 4, 0)), "multi": (ops.FX_BLEND_MULTI, (1, 0))}[kind]
            amts = [40 + 10 * f for f in range(n)]
            return (lambda i: ops.fx_batch(op, [[t] for t in a_[i % nb]], [[t] for t in o[i % nb]], w, h, ins1=[[t] for t in b_[i % nb]], ip=ip, frame_dp0=amts)), n * w * h * ps * 3
        if kind in ("c4rgba", "c4rgb24"):
            W, H, ps = 3840, 2160, (4 if kind == "c4rgba" else 3)
            a_ = [[torch.randint(0, 256, (H, W * ps), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)] for _ in range(nb)]
            b_ = [[torch.randint(0, 256, (H, W * ps), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)] for _ in range(nb)]
            o = [[torch.zeros_like(t) for t in st_] for st_ in a_]
            return (lambda i: ops.fx_batch(ops.FX_GAUSS5_COLORKEY, [[t] for t in a_[i % nb]], [[t] for t in o[i % nb]], W, H, ins1=[[t] for t in b_[i % nb]],
                                           ip=(ps, 0, 128 | (128 << 8) | (128 << 16)), dp=(0.3, 0.8))), n * W * H * ps * 3
        raise SystemExit("unknown fx batch " + op)
    if op.startswith("b") and ":" in op and op[1:op.index(":")].isdigit():
        # bN:<kernel> -- N frames of 1920 x 1080 per launch through the batch forms of the CONVERT-step kernels and single-plane effects (include/lives_gpu.h, round 6):
        # swz (BGRA32 -> RGBA32), swz34 (RGB24 -> BGRA32), gamma, premult, mirror, letterbox (-> 1920 x 1200), colorkey, r2y420 / r2uyvy / r2y888 (RGBA32 -> ...), y8882rgb / uyvy2rgb (-> RGBA32)
        import ctypes
        from lives_amd import lib
        n = int(op[1:op.index(":")])
        kind = op.split(":")[1]
        nb = 2 if not COLD else max(2, NB // n + 1)
        vp = ctypes.c_void_p

        def frames(rows, cols):
            return [[torch.randint(0, 256, (rows, cols), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)] for _ in range(nb)]

        def tab(sets):
            return [(vp * n)(*[t.data_ptr() for t in st_]) for st_ in sets]
        lut = np.arange(256, dtype=np.uint8)[::-1].copy()
        sp = ops.stream_ptr
        if kind in ("swz", "swz34"):
            ib = 4 if kind == "swz" else 3
            a_, o = frames(h, w * ib), frames(h, w * 4)
            ta, to = tab(a_), tab(o)
            opi = 4 if kind == "swz" else 2          # LGPU_SWAP3POSTALPHA / LGPU_SWAP3ADDPOST
            return (lambda i, keep=(a_, o): lib.call("lgpu_swizzle_batch", opi, 0, ta[i % nb], w * ib, to[i % nb], w * 4, w, h, None, n, sp())), n * w * h * (ib + 4)
        if kind == "gamma":
            a_ = frames(h, w * 4)
            ta = tab(a_)
            return (lambda i, keep=a_: lib.call("lgpu_gamma_apply_batch", ta[i % nb], w * 4, 0, 0, w, h, 4, 0, lut.ctypes.data, n, sp())), n * w * h * 8
        if kind == "premult":
            a_ = frames(h, w * 4)
            ta = tab(a_)
            return (lambda i, keep=a_: lib.call("lgpu_alpha_premult_batch", ta[i % nb], w * 4, w, h, 0, 0, n, sp())), n * w * h * 8
        if kind == "mirror":
            a_, o = frames(h, w * 4), frames(h, w * 4)
            ta, to = tab(a_), tab(o)
            return (lambda i, keep=(a_, o): lib.call("lgpu_mirror_batch", 2, ta[i % nb], w * 4, to[i % nb], w * 4, w, h, 4, n, sp())), n * w * h * 8
        if kind == "letterbox":
            a_, o = frames(h, w * 4), frames(1200, w * 4)
            ta, to = tab(a_), tab(o)
            black = (ctypes.c_uint8 * 4)(0, 0, 0, 255)
            return (lambda i, keep=(a_, o): lib.call("lgpu_letterbox_batch", ta[i % nb], w * 4, w, h, to[i % nb], w * 4, w, 1200, 4, black, n, sp())), n * (w * h * 4 + w * 1200 * 4)
        if kind == "colorkey":
            a_, b_, o = frames(h, w * 3), frames(h, w * 3), frames(h, w * 3)
            ta, tb, to = tab(a_), tab(b_), tab(o)
            return (lambda i, keep=(a_, b_, o): lib.call("lgpu_colorkey_batch", ta[i % nb], w * 3, tb[i % nb], w * 3, to[i % nb], w * 3, w, h, 0, 0.3, 0.8, 128, 128, 128, n, sp())), n * w * h * 9
        if kind in ("r2y420", "r2uyvy", "r2y888"):
            fmt = {"r2y420": 4, "r2uyvy": 2, "r2y888": 0}[kind]
            a_ = frames(h, w * 4)
            dims = {4: [(h, w), (h // 2, w // 2), (h // 2, w // 2)], 2: [(h, w * 2)], 0: [(h, w * 3)]}[fmt]
            o = [[[torch.zeros(d, dtype=torch.uint8, device="cuda") for d in dims] for _ in range(n)] for _ in range(nb)]
            ta = tab(a_)
            to = []
            for st_ in o:
                t_ = (vp * (4 * n))()
                for f in range(n):
                    for k, pl in enumerate(st_[f]):
                        t_[4 * f + k] = pl.data_ptr()
                to.append(t_)
            orow = (ctypes.c_int * 4)(*([d[1] for d in dims] + [0] * (4 - len(dims))))
            ob = sum(d[0] * d[1] for d in dims)
            return (lambda i, keep=(a_, o): lib.call("lgpu_rgb_to_yuv_batch", ta[i % nb], w * 4, w, h, 0, 1, to[i % nb], orow, fmt, 0, 0, n, sp())), n * (w * h * 4 + ob)
        if kind in ("y8882rgb", "uyvy2rgb"):
            fmt = 0 if kind == "y8882rgb" else 2
            ib = 3 if fmt == 0 else 2
            a_, o = frames(h, w * ib), frames(h, w * 4)
            ta = []
            for st_ in a_:
                t_ = (vp * (4 * n))()
                for f in range(n):
                    t_[4 * f] = st_[f].data_ptr()
                ta.append(t_)
            to = tab(o)
            irow = (ctypes.c_int * 4)(w * ib, 0, 0, 0)
            return (lambda i, keep=(a_, o): lib.call("lgpu_yuv_to_rgb_batch", ta[i % nb], irow, w, h, fmt, 0, to[i % nb], w * 4, 0, 1, 0, n, sp())), n * w * h * (ib + 4)
        raise SystemExit("unknown batch form " + op)
    if op.startswith("pb") and op[2:op.index(":")].isdigit():      # pbN:SWxSH:DWxDH:interp -- N frames of one geometry per launch (lgpu_pixbuf_scale_batch)
        head, a_, b_, it = op.split(":")
        opq = it.endswith("o")          # pbN:...:3o -- sources whose alpha is 255 everywhere, interp | LGPU_INTERP_OPAQUE
        it = str(int(it.rstrip("o")) | (0x200 if opq else 0))
        n = int(head[2:])
        sw, sh = (int(v) for v in a_.split("x"))
        dw, dh = (int(v) for v in b_.split("x"))
        nb = NB if COLD else 2
        if COLD:
            nb = max(2, NB // n + 1)
            nb += nb % 2
        sets = [([torch.randint(0, 256, (sh, sw * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(n)],
                 [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in range(n)]) for _ in range(nb)]
        if opq:
            for st_ in sets:
                for t_ in st_[0]:
                    t_[:, 3::4] = 255
        return (lambda i: ops.pixbuf_scale_batch(sets[i % nb][0], sets[i % nb][1], sw, sh, dw, dh, channels=4, interp=int(it))), n * (sw * sh * 4 + dw * dh * 4)
    if op.startswith("pb:"):           # pb:SWxSH:DWxDH:interp  -- one gdk-pixbuf ratio
        _, a_, b_, it = op.split(":")
        opq = it.endswith("o")
        it = str(int(it.rstrip("o")) | (0x200 if opq else 0))
        sw, sh = (int(v) for v in a_.split("x"))
        dw, dh = (int(v) for v in b_.split("x"))
        src, dst = rnd(sh, sw * 4), rnd(dh, dw * 4)
        if opq:
            for t_ in src:
                t_[:, 3::4] = 255
        return (lambda i: ops.pixbuf_scale(src[i % NB], dst[i % NB], sw, sh, dw, dh, channels=4, interp=int(it))), sw * sh * 4 + dw * dh * 4
    raise SystemExit("unknown op " + op)


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else "edge"
    ops.init(0)
    fn, _ = case(op)
    for i in range(200):
        fn(i)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
