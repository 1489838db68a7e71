"""A real weed host for the layer-seam tests: the REFERENCE's libweed (oracle/_ref/libweedall.so) through ctypes.

Layers are genuine weed plants of type WEED_PLANT_LAYER (128, src/layers.h:14) built with libweed's own
accessors; pixel planes are malloc()ed so the seam can free() them like LiVES would.
"""
import ctypes
import os

import numpy as np

from oracle import pyoracle as po

vp, ci = ctypes.c_void_p, ctypes.c_int
libc = ctypes.CDLL(None)
libc.malloc.restype = vp
libc.malloc.argtypes = [ctypes.c_size_t]
libc.free.argtypes = [vp]

_W = None


def weed():
    global _W
    if _W is None:
        # libweed declares its accessors `extern` (libweed/weed.h:117): the HOST module defines the pointer variables.
        # oracle/ref/refhost.c is such a host, so load it first and globally.
        W = ctypes.CDLL(os.path.join(po.REFDIR, "librefhost.so"), mode=ctypes.RTLD_GLOBAL)
        assert W.refhost_init() == 0
        W.weed_set_int_value.argtypes = [vp, ctypes.c_char_p, ci]
        W.weed_get_int_value.argtypes = [vp, ctypes.c_char_p, vp]
        W.weed_set_int_array.argtypes = [vp, ctypes.c_char_p, ci, vp]
        W.weed_set_voidptr_array.argtypes = [vp, ctypes.c_char_p, ci, vp]
        W.weed_get_voidptr_array_counted.argtypes = [vp, ctypes.c_char_p, vp]
        W.weed_get_voidptr_array_counted.restype = ctypes.POINTER(vp)
        W.weed_get_int_array_counted.argtypes = [vp, ctypes.c_char_p, vp]
        W.weed_get_int_array_counted.restype = ctypes.POINTER(ci)
        W.weed_plant_has_leaf.argtypes = [vp, ctypes.c_char_p]
        W.plant_new = ctypes.CFUNCTYPE(vp, ctypes.c_int32)(vp.in_dll(W, "weed_plant_new").value)
        W.fn = {n: vp.in_dll(W, n).value for n in ("weed_leaf_get", "weed_leaf_set", "weed_leaf_num_elements", "weed_leaf_delete", "weed_leaf_get_flags", "weed_leaf_set_flags")}
        _W = W
    return _W


class WeedApi(ctypes.Structure):
    _fields_ = [("leaf_get", vp), ("leaf_set", vp), ("leaf_num_elements", vp), ("leaf_delete", vp), ("pixel_alloc", vp), ("pixel_free", vp)]


def bind(L):
    W = weed()
    api = WeedApi(W.fn["weed_leaf_get"], W.fn["weed_leaf_set"], W.fn["weed_leaf_num_elements"], W.fn["weed_leaf_delete"], None, None)
    L.lives_gpu_bind_weed.argtypes = [ctypes.POINTER(WeedApi)]
    assert L.lives_gpu_bind_weed(ctypes.byref(api)) == 0
    L.lives_gpu_bind_leaf_get_flags.argtypes = [vp]
    assert L.lives_gpu_bind_leaf_get_flags(W.fn["weed_leaf_get_flags"]) == 0
    for name, args in (("lives_gpu_convert_layer_palette", [vp, ci, ci]), ("lives_gpu_convert_layer_palette_full", [vp, ci, ci, ci, ci, ci]),
                       ("lives_gpu_gamma_convert_layer", [ci, vp]), ("lives_gpu_gamma_convert_sub_layer", [ci, ctypes.c_double, vp, ci, ci, ci, ci, ci]),
                       ("lives_gpu_alpha_premult", [vp, ci]), ("lives_gpu_resize_layer", [vp, ci, ci, ci, ci, ci]),
                       ("lives_gpu_letterbox_layer", [vp, ci, ci, ci, ci, ci, ci, ci]), ("lives_gpu_create_empty_pixel_data", [vp, ci, ci]),
                       ("lives_gpu_layer_pin", [vp]), ("lives_gpu_layer_sync", [vp]), ("lives_gpu_layer_unpin", [vp]), ("lives_gpu_layer_forget", [vp]),
                       ("lives_gpu_transfer_stats", [vp, vp]), ("lives_gpu_convert_layer_palette_with_sampling", [vp, ci, ci]),
                       ("lives_gpu_gamma_convert_layer_variant", [ctypes.c_double, ci, vp]), ("lives_gpu_resize_layer_full", [vp, ci, ci, ci, ci, ci, ci, ci, ci]),
                       ("lives_gpu_unletterbox_layer", [vp, ci, ci, ci, ci, ci, ci]), ("lives_gpu_compact_rowstrides", [vp]),
                       ("lives_gpu_weed_layer_clear_pixel_data", [vp]), ("lives_gpu_layer_copy", [vp, vp]), ("lives_gpu_layer_set_opaque", [vp, ci])):
        getattr(L, name).argtypes = args
    L.lives_gpu_calc_rowstrides.argtypes = [ci, ci, vp, vp]
    L.lives_gpu_calc_rowstrides.restype = ctypes.POINTER(ci)
    return W


def malloc_copy(arr):
    p = libc.malloc(arr.nbytes + 64)
    ctypes.memmove(p, arr.ctypes.data, arr.nbytes)
    return p


def new_layer(pal, width, height, planes, gamma=None, clamping=None, subspace=None, flags=None):
    """planes: list of 2-D uint8 arrays (rows x rowstride); copied into malloc()ed memory owned by the layer"""
    W = weed()
    layer = W.plant_new(128)
    W.weed_set_int_value(layer, b"current_palette", pal)
    W.weed_set_int_value(layer, b"width", width)
    W.weed_set_int_value(layer, b"height", height)
    rs = (ci * len(planes))(*[p.strides[0] for p in planes])
    W.weed_set_int_array(layer, b"rowstrides", len(planes), rs)
    pd = (vp * len(planes))(*[malloc_copy(np.ascontiguousarray(p)) for p in planes])
    W.weed_set_voidptr_array(layer, b"pixel_data", len(planes), pd)
    if gamma is not None:
        W.weed_set_int_value(layer, b"gamma_type", gamma)
    if clamping is not None:
        W.weed_set_int_value(layer, b"YUV_clamping", clamping)
    if subspace is not None:
        W.weed_set_int_value(layer, b"YUV_subspace", subspace)
    if flags is not None:
        W.weed_set_int_value(layer, b"host_flags", flags)
    return layer


def geti(layer, key, default=None):
    W = weed()
    if not W.weed_plant_has_leaf(layer, key.encode()):
        return default
    return W.weed_get_int_value(layer, key.encode(), None)


def planes_of(layer):
    """[(2-D uint8 view copy rows x rowstride)] using the layer's own leaves"""
    W = weed()
    n = ci()
    pd = W.weed_get_voidptr_array_counted(layer, b"pixel_data", ctypes.byref(n))
    rs = W.weed_get_int_array_counted(layer, b"rowstrides", ctypes.byref(ci()))
    pal, h = geti(layer, "current_palette"), geti(layer, "height")
    out = []
    for i in range(n.value):
        ph = h if (i == 0 or pal in (544, 545, 522)) else h >> 1
        buf = (ctypes.c_uint8 * (rs[i] * ph)).from_address(pd[i])
        out.append(np.frombuffer(buf, np.uint8).reshape(ph, rs[i]).copy())
    return out, [pd[i] for i in range(n.value)], [rs[i] for i in range(n.value)]
