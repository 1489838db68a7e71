"""CPU: the product's host-side code (liblivesgpu.so without a GPU): ABI surface, table builders, filters.

No compute entry point is called here except to check that it fails loudly without a device.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from lives_amd import lib
from oracle import pyoracle as po
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = po.P


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lives_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    L = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "include/lives_gpu.h declares %s but liblivesgpu.so does not export it" % s
    for s in syms:
        assert s in lib.PROTOTYPES or s == "lgpu_last_error", "no ctypes prototype for " + s
    assert L.lgpu_abi_version() == 1


def test_layer_seam_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "lives_gpu_layer.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(lives_gpu_[a-z0-9_]+)\s*\(", text)))
    assert len(syms) >= 15
    L = lib.load()
    for s in syms:
        assert hasattr(L, s), "include/lives_gpu_layer.h declares %s but liblivesgpu.so does not export it" % s


def test_plugin_exports_what_its_header_declares():
    """include/livesgpu_fx.h: weed_setup and the batch hook are real exported symbols of livesgpu_fx.so, and the header compiles as C beside the layer header"""
    import ctypes
    import subprocess
    import tempfile
    text = open(os.path.join(ROOT, "include", "livesgpu_fx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    syms = sorted(set(re.findall(r"\b(weed_setup|livesgpu_fx_[a-z0-9_]+)\s*\(", text)))
    assert syms == ["livesgpu_fx_process_batch", "weed_setup"]
    so = os.path.join(ROOT, "lives_amd", "livesgpu_fx.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(syms) <= exported, exported
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "lives_gpu_layer.h"\n#include "livesgpu_fx.h"\nint main(void) { return (int)sizeof(&livesgpu_fx_process_batch) == 0; }\n')
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src], check=True)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.load()
    buf = np.zeros(64, np.uint8)
    rc = L.lgpu_swizzle(lib.SWAP3ADDPOST, 0, P(buf), 12, P(buf), 16, 4, 1, None, None)
    assert rc == -1, "compute entry points must fail with LGPU_E_NODEVICE on a box without a GPU"
    assert b"no CPU fallback" in L.lgpu_last_error()
    with pytest.raises(lib.LgpuError):
        lib.call("lgpu_init", 0)


def test_conversion_tables_match_reference_fixture():
    L = lib.load()
    g = gu.load("tables.npz")
    for which in range(4):
        a = np.zeros((9, 256), np.int32)
        b = np.zeros((5, 256), np.int32)
        assert L.lgpu_conversion_tables(which, P(a), P(b)) == 0
        assert (a == g["rgb2yuv_%d" % which]).all()
        assert (b == g["yuv2rgb_%d" % which]).all()


def test_gamma_lut_builder_matches_reference_fixture():
    L = lib.load()
    g = gu.load("luts.npz")
    for key in g.files:
        parts = key.split("_")
        if key.startswith("lut8v_"):
            fileg, gfrom, gto = float(parts[1].replace("p", ".")), int(parts[2]), 2048
        else:
            fileg, gfrom, gto = 1.0, int(parts[1]), int(parts[2])
        lut = np.zeros(256, np.uint8)
        ok = L.lgpu_gamma_lut8(fileg, gfrom, gto, 1.4, P(lut))
        assert ok == int(g[key][0]), key
        if ok:
            assert (lut == g[key][1:]).all(), key
    # the headline LUT: linear -> sRGB tops out at 246 in the reference
    lut = np.zeros(256, np.uint8)
    L.lgpu_gamma_lut8(1.0, -1, 1, 1.4, P(lut))
    assert lut[[1, 10, 50, 128, 200, 254, 255]].tolist() == [11, 53, 118, 181, 221, 246, 246]


def test_filter_bank_matches_oracle_and_sums_to_one(orc):
    L = lib.load()
    for (srcn, dstn, kernel) in [(3840, 1920, 1), (2160, 1080, 1), (64, 128, 2), (100, 37, 1), (128, 64, 0), (7, 5, 0), (1920, 1280, 1)]:
        nt_a, nt_b = ctypes.c_int(), ctypes.c_int()
        pa, pb = np.zeros(dstn, np.int32), np.zeros(dstn, np.int32)
        ca, cb = np.zeros(dstn * 256, np.int16), np.zeros(dstn * 256, np.int16)
        assert L.lgpu_make_filter(srcn, dstn, kernel, ctypes.byref(nt_a), P(pa), P(ca), 256) == 0
        assert orc.orc_make_filter(srcn, dstn, kernel, ctypes.byref(nt_b), P(pb), P(cb), 256) == 0
        assert nt_a.value == nt_b.value
        n = nt_a.value
        assert (pa == pb).all() and (ca[:dstn * n] == cb[:dstn * n]).all()
        assert (ca[:dstn * n].reshape(dstn, n).astype(np.int64).sum(axis=1) == 16384).all()
    # exact 2:1 bicubic: one phase, first tap at 2i - 3
    nt = ctypes.c_int()
    pos, co = np.zeros(1920, np.int32), np.zeros(1920 * 256, np.int16)
    L.lgpu_make_filter(3840, 1920, 1, ctypes.byref(nt), P(pos), P(co), 256)
    assert nt.value == 8 and (pos == 2 * np.arange(1920) - 3).all()
    taps = co[:8]
    assert (co[:1920 * 8].reshape(1920, 8) == taps).all() and (taps == taps[::-1]).all()


def test_calc_rowstrides_rule():
    L = lib.load()
    rs = (ctypes.c_int * 4)()
    # src/colourspace.c:11252: ALIGN_CEIL(width * psize, 32); chroma strides rs[0] >> 1
    assert L.lgpu_calc_rowstrides(640, 1, 0, rs) == 1 and rs[0] == 1920
    assert L.lgpu_calc_rowstrides(640, 4, 0, rs) == 1 and rs[0] == 2560
    assert L.lgpu_calc_rowstrides(1921, 3, 0, rs) == 1 and rs[0] == 7712          # RGBA32: ALIGN_CEIL(7684, 32)
    assert L.lgpu_calc_rowstrides(1921, 1, 0, rs) == 1 and rs[0] == 5792          # RGB24: ALIGN_CEIL(5763, 32)
    assert L.lgpu_calc_rowstrides(1920, 512, 0, rs) == 3 and list(rs)[:3] == [1920, 960, 960]
    assert L.lgpu_calc_rowstrides(1918, 512, 0, rs) == 3 and list(rs)[:3] == [1920, 960, 960]
    assert L.lgpu_calc_rowstrides(101, 1, -1, rs) == 1 and rs[0] == 303          # compact
    assert L.lgpu_calc_rowstrides(101, 1, 16, rs) == 1 and rs[0] == 304          # resize hint (:14989)
    assert L.lgpu_calc_rowstrides(100, 545, 0, rs) == 4 and list(rs) == [128, 128, 128, 128]
    assert L.lgpu_calc_rowstrides(100, 77777, 0, rs) == 0


def test_weed_abi_constants_match_reference_headers():
    """include/lives_gpu_weed_abi.h restates ids of the public weed ABI; check them against the real headers"""
    ref = "/root/reference/libweed"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present")
    ours = open(os.path.join(ROOT, "include", "lives_gpu_weed_abi.h")).read()
    theirs = "".join(open(os.path.join(ref, f)).read() for f in ("weed.h", "weed-palettes.h", "weed-effects.h"))
    theirs += open("/root/reference/src/colourspace.h").read()

    def defs(text):
        out = {}
        for m in re.finditer(r"^\s*#define\s+(WEED_[A-Z0-9_]+)\s+(\(?-?[0-9]+\)?|\"[^\"]*\"|\(1\s*<<\s*[0-9]+\))\s*(?:/[/*].*)?$", text, re.M):
            out[m.group(1)] = m.group(2).replace(" ", "").strip("()")
        return out
    mine, real = defs(ours), defs(theirs)
    for m in re.finditer(r"^\s*#define\s+(WEED_[A-Z0-9_]+)\s+(WEED_[A-Z0-9_]+)\s*$", theirs, re.M):   # aliases, e.g. WEED_PALETTE_END
        if m.group(2) in real and m.group(1) not in real:
            real[m.group(1)] = real[m.group(2)]
    checked = 0
    for k, v in mine.items():
        if k in ("WEED_API_VERSION_MIN", "WEED_TRUE", "WEED_FALSE"):   # ours; (weed_boolean_t) casts of 1 / 0 in weed.h:90-105
            continue
        assert k in real, "%s is not a weed ABI name" % k
        assert real[k] == v, "%s: ours %s, reference %s" % (k, v, real[k])
        checked += 1
    assert checked > 80


def test_alpha_scalers_equal_the_reference_float_expression():
    """lgpu_alpha_scalers: (c * k2[a]) >> 16 == (uint8_t)((float)c * alpha) and (c * k1[a]) >> 16 == (uint8_t)((float)c * inv_alpha)
    with alpha = (float)a / 255., inv_alpha = 1. - alpha (simple_blend.c:137-145), for every byte c and every translucent alpha;
    checked here against an independent IEEE float32 evaluation (numpy) and against the oracle's chroma blend"""
    L = lib.load()
    k2 = np.zeros(256, np.uint32)
    k1 = np.zeros(256, np.uint32)
    assert L.lgpu_alpha_scalers(P(k2), P(k1)) == 0
    assert k2[255] == 65536 and k1[255] == 65536          # opaque pixels are not scaled (simple_blend.c:128-131)
    a = np.arange(256)
    alpha = (a.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    inv = (1.0 - alpha.astype(np.float64)).astype(np.float32)
    c = np.arange(256, dtype=np.float32)
    t2 = np.floor(c[None, :] * alpha[:, None]).astype(np.int64)        # float32 * float32 -> float32, truncated
    t1 = np.floor(c[None, :] * inv[:, None]).astype(np.int64)
    ci = np.arange(256, dtype=np.int64)
    g2 = (ci[None, :] * k2.astype(np.int64)[:, None]) >> 16
    g1 = (ci[None, :] * k1.astype(np.int64)[:, None]) >> 16
    assert (g2[:255] == t2[:255]).all() and (g1[:255] == t1[:255]).all()
    assert (k2 < (1 << 24)).all() and (k1 < (1 << 24)).all()          # operands of v_mul_u32_u24
    # the oracle's chroma blend (pinned on the reference plugin) on every (alpha, colour) pair: bf = 255 keeps s2, bf = 0 keeps s1
    o = po.oracle()
    w, h = 256, 255
    p2 = np.zeros((h, w * 4), np.uint8)
    p1 = np.zeros((h, w * 4), np.uint8)
    for ch in range(3):
        p2[:, ch::4] = np.arange(256, dtype=np.uint8)[None, :]
        p1[:, ch::4] = np.arange(256, dtype=np.uint8)[None, :]
    p2[:, 3::4] = np.arange(255, dtype=np.uint8)[:, None]
    p1[:, 3::4] = 255
    for bf, g in ((255, g2), (0, g1)):
        out = np.zeros_like(p1)
        o.orc_blend_chroma(P(p1), w * 4, P(p2), w * 4, P(out), w * 4, w, h, 4, 0, bf)
        want = (255 * g[:255]) >> 8          # blend[s2][s1] = (bf * s2 + (255 - bf) * s1) >> 8 with the other weight 0
        assert (out[:, 0::4] == want).all(), "bf=%d" % bf


def test_clamped_luma_premultiply_tables_have_an_exact_integer_form():
    """what k_premult_yuva evaluates instead of gathering from init_unal's 64 KB tables (src/colourspace.c:1141-1160): alcy / unalcy[i][j] ==
    floor((2 j i + 255) / 510) > lim ? cap : floor((2 (j - 16) i + 8415) / 510) for every (alpha, value) pair -- 2 j i + 255 is odd and 510 even, so the float
    arithmetic of the reference never sits on a rounding boundary; and floor(n / 510) == (n * 2155905153) >> 40 over the whole range"""
    import ctypes
    import numpy as np
    from lives_amd import lib
    L = lib.load()
    t = [np.zeros((256, 256), np.uint8) for _ in range(4)]
    L.lgpu_premult_yuv_tables.argtypes = [ctypes.c_void_p] * 4
    assert L.lgpu_premult_yuv_tables(*[x.ctypes.data for x in t]) == 1
    i = np.arange(256, dtype=np.int64)[:, None]
    j = np.arange(256, dtype=np.int64)[None, :]
    n = np.arange(0, 140000, dtype=np.int64)
    assert ((n * 2155905153 >> 40) == n // 510).all()
    for tab, lim, cap in ((t[0], 219, 235), (t[1], 224, 240)):
        cand = np.where((2 * j * i + 255) // 510 > lim, cap, (2 * (j - 16) * i + 8415) // 510)
        assert (cand == tab).all()
