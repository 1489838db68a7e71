"""The layer-op seam under the reference's own names: a small C host (tests/c/dropin_host.c) declares the prototypes of
src/colourspace.h:377-423, is linked against lives_amd/liblivesgpu_dropin.so and calls them on genuine weed layers."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.util import align, frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "lives_amd")
P = po.P
# SURVEY 8(b)1 + the rest of the header range the seam cites
NAMES = ["convert_layer_palette", "convert_layer_palette_full", "convert_layer_palette_with_sampling", "gamma_convert_layer", "gamma_convert_layer_variant",
         "gamma_convert_sub_layer", "alpha_premult", "resize_layer", "resize_layer_full", "letterbox_layer", "unletterbox_layer", "compact_rowstrides",
         "create_empty_pixel_data", "weed_layer_clear_pixel_data", "calc_rowstrides"]


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dropin") / "libdropin_host.so")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "c", "dropin_host.c"),
                           "-L" + LIBDIR, "-llivesgpu_dropin", "-Wl,-rpath," + LIBDIR, "-Wl,--no-undefined"])
    from lives_amd import lib
    lib.load()
    return ctypes.CDLL(out)


def test_reference_names_resolve(host):
    shim = ctypes.CDLL(os.path.join(LIBDIR, "liblivesgpu_dropin.so"))
    for n in NAMES:
        assert hasattr(shim, n), "liblivesgpu_dropin.so does not export " + n
    assert host.host_name_count() == len(NAMES)
    assert host.host_rowstride(640, 1) == 1920 and host.host_rowstride(1918, 512) == 1920      # calc_rowstrides needs no device


@pytest.mark.gpu
@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref (reference libweed) not built")
def test_c_host_runs_a_convert_chain_by_reference_names(host, orc):
    from lives_amd import lib
    from tests import weedhost as wh
    L = lib.load()
    wh.bind(L)
    rng = np.random.default_rng(61)
    w, h = 256, 144
    ys, cs = align(w), align(w) >> 1
    Y = rng.integers(16, 236, (h, ys), dtype=np.uint8)
    U = rng.integers(16, 241, (h // 2, cs), dtype=np.uint8)
    V = rng.integers(16, 241, (h // 2, cs), dtype=np.uint8)
    lay = wh.new_layer(512, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    host.host_convert_chain.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6
    assert host.host_convert_chain(lay, 3, 1, 128, 72, 128, 96) == 0
    ref = wh.new_layer(512, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    assert L.lives_gpu_convert_layer_palette(ref, 3, 0) == 1 and L.lives_gpu_gamma_convert_layer(1, ref) == 1
    assert L.lives_gpu_resize_layer(ref, 128, 72, 3, 0, 0) == 1 and L.lives_gpu_letterbox_layer(ref, 128, 96, 128, 72, 3, 0, 0) == 1
    got, _, rs = wh.planes_of(lay)
    want, _, rs2 = wh.planes_of(ref)
    assert rs == rs2 and (got[0] == want[0]).all()
    assert (wh.geti(lay, "current_palette"), wh.geti(lay, "width"), wh.geti(lay, "height"), wh.geti(lay, "gamma_type")) == (3, 128, 96, 1)
    # the first stage against the oracle directly
    rgba = np.zeros((h, align(w * 4)), np.uint8)
    strides = (ctypes.c_int * 3)(ys, cs, cs)
    orc.orc_yuv420p_to_rgb(P(Y), P(U), P(V), strides, U.size, V.size, P(rgba), rgba.strides[0], w, h, 4, 0, 0, 0, 2, None, 0)
    one = wh.new_layer(512, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1)
    host.host_convert_chain.restype = ctypes.c_int
    shim = ctypes.CDLL(os.path.join(LIBDIR, "liblivesgpu_dropin.so"))
    shim.convert_layer_palette.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    assert shim.convert_layer_palette(one, 3, 0) == 1
    assert (wh.planes_of(one)[0][0][:, :w * 4] == rgba[:, :w * 4]).all()
