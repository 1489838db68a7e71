#!/usr/bin/env python3
"""tools/fuzz_chain.py [cases] [seed] -- random geometry through lgpu_chain_amounts (every one-launch form: 2:1, other ratios, enlargements, integer reductions, a letterbox
canvas, no layer 2, no resize) against the oracle's stages run one after the other.  GPU box only; prints the first mismatch and exits 1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import torch                # noqa: E402
from lives_amd import ops   # noqa: E402
from oracle import pyoracle as po   # noqa: E402

P = po.P


def run(cases, seed):
    """-> (cases launched, tracks compared, mismatching cases); a geometry the scaler declines (the library's two-step range) is skipped"""
    from lives_amd.lib import LgpuError
    rng = np.random.default_rng(seed)
    ops.init(0)
    orc = po.oracle()
    bad = 0
    ran = tracks = 0
    for c in range(cases):
        kind = rng.integers(0, 6)
        sw, sh = int(rng.integers(4, 200)), int(rng.integers(2, 120))
        if kind == 0:
            sw, sh = 2 * int(rng.integers(2, 100)), 2 * int(rng.integers(1, 60)); dw, dh = sw // 2, sh // 2
        elif kind == 1:
            dw, dh = sw, sh
        elif kind == 2:
            f = int(rng.integers(3, 5)); dw, dh = max(1, sw // f), max(1, sh // f); sw, sh = dw * f, dh * f
        else:
            dw, dh = int(rng.integers(1, 260)), int(rng.integers(1, 160))
        if (dw, dh) == (sw, sh) and kind != 1:
            continue
        canvas = None
        if rng.random() < 0.4:
            cw, ch = dw + int(rng.integers(0, 20)), dh + int(rng.integers(0, 12))
            canvas = (cw, ch, int(rng.integers(0, cw - dw + 1)), int(rng.integers(0, ch - dh + 1)))
        cw, ch = (canvas[0], canvas[1]) if canvas else (dw, dh)
        noblend = rng.random() < 0.35
        interp = int(rng.choice([3, 3, 2, 0]))
        swap, use_lut = int(rng.integers(0, 2)), rng.random() < 0.6
        n = int(rng.integers(1, 4))
        srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(n)]
        l2s = [rng.integers(0, 256, (ch, cw * 4), dtype=np.uint8) for _ in range(n)]
        for t in l2s:
            a = t[:, 3::4]
            a[rng.random(a.shape) < 0.5] = 255
        amounts = [int(v) for v in rng.integers(0, 256, n)]
        lut = rng.permutation(256).astype(np.uint8)
        d_s, d_l = [torch.from_numpy(a).cuda() for a in srcs], [torch.from_numpy(a).cuda() for a in l2s]
        d_o = [torch.full((ch, cw * 4), 0x77, dtype=torch.uint8, device="cuda") for _ in range(n)]
        prm = ops.chain_params(sw, sh, sw * 4, dw, dh, cw * 4, cw * 4, swap_rb=swap, interp=interp | 0x100 | (0x400 if noblend else 0), do_blur=0, bf=0, lut=lut if use_lut else None)
        try:
            ops.chain_amounts(prm, ops.chain_tracks(d_s, None if noblend else d_l, d_o), None if noblend else amounts, canvas)
        except LgpuError as e:
            if "(-3)" in str(e):
                continue
            raise
        torch.cuda.synchronize()
        ran += 1
        for i in range(n):
            conv = srcs[i].copy()
            if swap:
                orc.orc_swizzle(po.OPS.index("swap3postalpha"), 0, P(srcs[i]), sw * 4, P(conv), sw * 4, sw, sh, None)
            out = np.zeros((dh, dw * 4), np.uint8)
            if (dw, dh) == (sw, sh):
                out[:] = conv
            else:
                assert orc.orc_pixbuf_scale(P(conv), sw * 4, sw, sh, P(out), dw * 4, dw, dh, 4, interp) == 0
            if canvas:
                big = np.zeros((ch, cw * 4), np.uint8)
                big[:, 3::4] = 255
                big[canvas[3]:canvas[3] + dh, canvas[2] * 4:(canvas[2] + dw) * 4] = out
                out = big
            if not noblend:
                orc.orc_blend_chroma(P(out), cw * 4, P(l2s[i]), cw * 4, P(out), cw * 4, cw, ch, 4, 0, amounts[i])
            if use_lut:
                orc.orc_gamma_apply(P(out), cw * 4, cw, ch, 4, 0, P(lut))
            got = d_o[i].cpu().numpy()
            tracks += 1
            if not (got == out).all():
                w = np.argwhere(got != out)
                print("MISMATCH case %d: %dx%d -> %dx%d canvas %s interp %d swap %d lut %s noblend %s track %d/%d: %d bytes, first %s (got %d want %d)" %
                      (c, sw, sh, dw, dh, canvas, interp, swap, use_lut, noblend, i, n, len(w), w[0].tolist(), got[tuple(w[0])], out[tuple(w[0])]))
                bad += 1
                break
        if bad >= 5:
            break
    return ran, tracks, bad


def main():
    ran, tracks, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 20260929)
    print("fuzz_chain: %d cases launched, %d tracks compared with the oracle, %d mismatching" % (ran, tracks, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
