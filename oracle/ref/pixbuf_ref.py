"""TEST INFRASTRUCTURE.  ctypes binding of the gdk-pixbuf runtime library (libgdk_pixbuf-2.0.so.0) -- the third-party scaler the
reference's non-swscale resize body calls: resize_layer_full -> lives_pixbuf_scale_simple == gdk_pixbuf_scale_simple
(/root/reference/src/colourspace.c:15262-15322, call at :15295; LIVES_INTERP_BEST / NORMAL / FAST = GDK_INTERP_HYPER / BILINEAR / NEAREST,
src/widget-helper-gtk.h:1136-1138), and the compositor's layer scaler (lives-plugins/weed-plugins/gdk/compositor.c:263-265).

The library is a binary in this image (no headers needed); nothing of it is copied.  Used by gen_golden_pixbuf.py to make the committed
fixtures and by tests/ to compare the restatement (oracle/lives_oracle.c: orc_pixbuf_scale) against the live library where it is present.
"""
import ctypes

import numpy as np

GDK_INTERP_NEAREST, GDK_INTERP_TILES, GDK_INTERP_BILINEAR, GDK_INTERP_HYPER = 0, 1, 2, 3
_lib = None


def lib():
    global _lib
    if _lib is None:
        g = ctypes.CDLL("libgdk_pixbuf-2.0.so.0")
        gobj = ctypes.CDLL("libgobject-2.0.so.0")
        g.gdk_pixbuf_new_from_data.restype = ctypes.c_void_p
        g.gdk_pixbuf_new_from_data.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        g.gdk_pixbuf_scale_simple.restype = ctypes.c_void_p
        g.gdk_pixbuf_scale_simple.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        for n in ("gdk_pixbuf_get_width", "gdk_pixbuf_get_height", "gdk_pixbuf_get_rowstride", "gdk_pixbuf_get_n_channels",
                  "gdk_pixbuf_get_has_alpha"):
            getattr(g, n).restype = ctypes.c_int
            getattr(g, n).argtypes = [ctypes.c_void_p]
        g.gdk_pixbuf_get_pixels.restype = ctypes.c_void_p
        g.gdk_pixbuf_get_pixels.argtypes = [ctypes.c_void_p]
        gobj.g_object_unref.argtypes = [ctypes.c_void_p]
        g._unref = gobj.g_object_unref
        _lib = g
    return _lib


def available():
    try:
        lib()
        return True
    except OSError:
        return False


def version():
    return ctypes.c_char_p.in_dll(lib(), "gdk_pixbuf_version").value.decode()


def scale_simple(src, width, channels, dw, dh, interp):
    """src: uint8 [h, rowstride] (rowstride >= width * channels); channels 3 (no alpha) or 4 (has_alpha, as
    lives_pixbuf_new_from_data_wrapper sets for RGBA32 / BGRA32 / YUVA8888, colourspace.c:14219-14225).
    Returns uint8 [dh, dw * channels] (the pixbuf's own rowstride padding removed)."""
    g = lib()
    src = np.ascontiguousarray(src)
    h, rs = src.shape
    pb = g.gdk_pixbuf_new_from_data(src.ctypes.data, 0, 1 if channels == 4 else 0, 8, width, h, rs, None, None)
    assert pb
    out = g.gdk_pixbuf_scale_simple(pb, dw, dh, interp)
    assert out, "gdk_pixbuf_scale_simple failed"
    ow, oh, ors, och = g.gdk_pixbuf_get_width(out), g.gdk_pixbuf_get_height(out), g.gdk_pixbuf_get_rowstride(out), g.gdk_pixbuf_get_n_channels(out)
    assert (ow, oh, och) == (dw, dh, channels)
    px = g.gdk_pixbuf_get_pixels(out)
    nbytes = ors * (oh - 1) + ow * och        # a pixbuf's last row may be short
    raw = np.ctypeslib.as_array(ctypes.cast(px, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).copy()
    res = np.zeros((oh, ow * och), np.uint8)
    for y in range(oh):
        res[y] = raw[y * ors: y * ors + ow * och]
    g._unref(out)
    g._unref(pb)
    return res
