"""CPU: the palette resolution in front of resize_layer_full's body (get_resizable and friends, src/colourspace.c:14500-14669) and the planner's capability
queries (src/colourspace.h:400-407).  Three parties: tests/golden/resizable.npz (the outputs of the reference's own lines, oracle/ref/gen_golden_resizable.py),
the oracle's restatement (oracle/orc_resizable.c) and the product (lives_gpu_get_resizable & co. in liblivesgpu.so, host logic: no device needed)."""
import ctypes
import os

import numpy as np
import pytest

from lives_amd import lib
from oracle import pyoracle as po
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "resizable.npz"))
PALS, HINTS = [int(x) for x in G["pals"]], [int(x) for x in G["hints"]]
ip = ctypes.POINTER(ctypes.c_int)


def cases():
    for i, p in enumerate(PALS):
        for j, h in enumerate(HINTS):
            for up in (0, 1):
                for cl in (0, 1):
                    yield p, h, up, cl, [int(x) for x in G["resizable"][i, j, up, cl]]


def test_oracle_restatement_equals_the_reference_lines():
    O = po.oracle()
    O.orc_get_resizable.argtypes = [ip]
    for p, h, up, cl, want in cases():
        io = (ctypes.c_int * 5)(p, h, cl, up, 0)
        r = O.orc_get_resizable(io)
        assert r == want[0], (p, h, up, cl)
        if r == 1:
            assert list(io) == want[1:], (p, h, up, cl, list(io), want)
        else:
            assert list(io)[:4] == [p, h, cl, up], "a failed resolution leaves its arguments alone"
    for i, a in enumerate(PALS):
        for j, b in enumerate(PALS):
            assert O.orc_get_tgt_gamma(a, b) == G["tgt_gamma"][i, j]
            assert O.orc_can_inline_gamma(a, b) == G["inline_gamma"][i, j], (a, b)
            assert O.orc_pconv_can_inplace(a, b) == G["inplace"][i, j], (a, b)


def test_what_the_fixture_says_about_the_cases_the_review_named():
    """YUV420P with the hint res_substep passes (src/nodemodel.c:1187) is converted and scaled, not refused; the routes the reference itself cannot take end in its LIVES_FATAL"""
    def row(p, h, up, cl=0):
        return [int(x) for x in G["resizable"][PALS.index(p), HINTS.index(h), up, cl]]
    assert row(512, 3, 0) == [1, 3, 3, 0, 3, 3] and row(512, 3, 1) == [1, 3, 3, 0, 3, 3]      # YUV420P, hint RGBA32 -> RGBA32
    assert row(5, 3, 0) == [1, 3, 3, 0, 3, 3]                                                  # ARGB32, hint RGBA32 -> RGBA32 (get_inter_pal)
    assert row(564, 1, 0)[1] == 1 and row(564, 1, 1)[1] == 1                                    # UYVY, hint RGB24
    assert row(512, 512, 0)[0] == -1 and row(512, -1, 0)[0] == -1                                # no resizable route: LIVES_FATAL
    assert row(512, -1, 1) == [1, 1, 1, 0, 1, 1]                                                # ... but an upscale converts to RGB24 first
    assert row(5, -1, 0)[1] == 1 and row(5, -1, 1)[1] == 588                                    # ARGB32 without a hint: RGB24 down, YUV888 up
    assert row(3, 588, 1, 0) == [1, 3, 3, 0, 588, 588]                                          # both sides in the switch: scaled as it is, the conversion is the caller's


def test_product_get_resizable_equals_the_reference_lines_on_the_pixbuf_backend():
    L = lib.load()
    L.lives_gpu_get_resizable.argtypes = [ip, ip, ip, ip, ip, ctypes.c_int]
    assert L.lives_gpu_get_resize_backend() == 1, "the pinned body is the default"
    for p, h, up, cl, want in cases():
        pal, xpal, ocl, opal, xopal = (ctypes.c_int(v) for v in (p, 0, cl, h, 0))
        r = L.lives_gpu_get_resizable(ctypes.byref(pal), ctypes.byref(xpal), ctypes.byref(ocl), ctypes.byref(opal), ctypes.byref(xopal), up)
        assert r == (1 if want[0] == 1 else 0), (p, h, up, cl)            # the reference's LIVES_FATAL is a plain FAIL here
        if r == 1:
            assert [pal.value, xpal.value, ocl.value, opal.value, xopal.value] == want[1:], (p, h, up, cl)
        else:
            assert (pal.value, ocl.value, opal.value) == (p, cl, h)
    # the two optional outputs may be NULL (get_resize_ops, src/nodemodel.c:143)
    pal, ocl, opal = ctypes.c_int(512), ctypes.c_int(0), ctypes.c_int(3)
    assert L.lives_gpu_get_resizable(ctypes.byref(pal), None, ctypes.byref(ocl), ctypes.byref(opal), None, 0) == 1 and pal.value == 3
    for i, a in enumerate(PALS):
        for j, b in enumerate(PALS):
            assert L.lives_gpu_get_tgt_gamma(a, b) == G["tgt_gamma"][i, j]


def test_product_capability_answers_describe_its_own_bodies():
    """can_inline_gamma / pconv_can_inplace answer for the library's conversions (include/lives_gpu_layer.h): never TRUE where the reference's rule says FALSE for
    gamma; in place only where the layer keeps its pixel_data"""
    L = lib.load()
    RGB = [1, 2, 3, 4, 5]
    for i, a in enumerate(PALS):
        for j, b in enumerate(PALS):
            mine = L.lives_gpu_can_inline_gamma(a, b)
            assert mine in (0, 1) and (not mine or G["inline_gamma"][i, j] == 1), (a, b)
            want = (a in RGB and b in RGB) or (a in (512, 513, 522) and b in RGB) or (a in RGB and b in (564, 565))
            assert mine == int(want), (a, b)
            assert L.lives_gpu_pconv_can_inplace(a, b) == int((a, b) in ((564, 565), (565, 564))), (a, b)


def test_polyphase_backend_resolves_with_its_own_switch():
    L = lib.load()
    L.lives_gpu_get_resizable.argtypes = [ip, ip, ip, ip, ip, ctypes.c_int]
    assert L.lives_gpu_set_resize_backend(0) == 0
    try:
        for p, h, want in ((512, 3, 512), (5, 3, 5), (564, 1, 1), (595, 512, 512), (588, 3, 3), (3, 588, 3)):
            pal, ocl, opal = ctypes.c_int(p), ctypes.c_int(0), ctypes.c_int(h)
            assert L.lives_gpu_get_resizable(ctypes.byref(pal), None, ctypes.byref(ocl), ctypes.byref(opal), None, 0) == 1 and pal.value == want, (p, h, pal.value)
        for p, h in ((588, -1), (564, 565), (595, 0)):                       # no masquerades in that body: what it cannot scale and cannot convert to the hint it refuses
            pal, ocl, opal = ctypes.c_int(p), ctypes.c_int(0), ctypes.c_int(h)
            assert L.lives_gpu_get_resizable(ctypes.byref(pal), None, ctypes.byref(ocl), ctypes.byref(opal), None, 0) == 0, (p, h)
    finally:
        assert L.lives_gpu_set_resize_backend(1) == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(po.REFDIR, "libresizableref.so")), reason="oracle/_ref/libresizableref.so not built")
def test_fixture_equals_the_live_reference_build():
    R = ctypes.CDLL(os.path.join(po.REFDIR, "libresizableref.so"))
    for p, h, up, cl, want in cases():
        io = (ctypes.c_int * 5)(p, h, cl, up, 0)
        r = R.rsref_get_resizable(io)
        assert r == want[0] and (r != 1 or list(io) == want[1:]), (p, h, up, cl)
