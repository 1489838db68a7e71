"""ctypes view of liblivesgpu.so (the C ABI declared in include/lives_gpu.h).

The product is the shared library; this module is the thinnest possible Python binding for tests and
bench.py.  It fails loudly when the library is missing or a call returns an error -- there is no
Python / CPU fallback for any pixel work.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LGPU_SO") or os.path.join(HERE, "liblivesgpu.so")      # LGPU_SO: a profiling build of the same library (tools/)

vp = ctypes.c_void_p
ci = ctypes.c_int
cl = ctypes.c_long
cd = ctypes.c_double
u8p = ctypes.POINTER(ctypes.c_uint8)

LGPU_CHAIN_MAX_TRACKS = 64

(SWAP3, SWAP4, SWAP3ADDPOST, SWAP3ADDPRE, SWAP3POSTALPHA, SWAP3PREALPHA, ADDPOST, ADDPRE, SWAP3DELPOST, DELPOST,
 DELPRE, SWAP3DELPRE, SWAPPREPOST) = range(13)
YUV_FIX_EDGES = 1


class LgpuError(RuntimeError):
    pass


class ChainTrack(ctypes.Structure):
    _fields_ = [("src_d", vp), ("layer2_d", vp), ("dst_d", vp)]


class YuvFrame(ctypes.Structure):
    """lgpu_yuv_frame"""
    _fields_ = [("y_d", vp), ("u_d", vp), ("v_d", vp), ("dst_d", vp)]


class CompLayer(ctypes.Structure):
    """lgpu_comp_layer (include/lives_gpu.h)"""
    _fields_ = [("src_d", ctypes.c_void_p), ("irow", ctypes.c_int), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("offs_x", ctypes.c_int), ("offs_y", ctypes.c_int), ("alpha", ctypes.c_double)]


class Canvas(ctypes.Structure):
    """lgpu_canvas"""
    _fields_ = [("nwidth", ci), ("nheight", ci), ("offs_x", ci), ("offs_y", ci)]


class ChainParams(ctypes.Structure):
    _fields_ = [("sw", ci), ("sh", ci), ("irow", ci), ("dw", ci), ("dh", ci), ("irow2", ci), ("orow", ci),
                ("swap_rb", ci), ("interp", ci), ("do_blur", ci), ("bf", ci), ("use_lut", ci),
                ("lut8", ctypes.c_uint8 * 256), ("param_block_d", vp)]


class FxFrame(ctypes.Structure):
    _fields_ = [("in0", vp * 4), ("in1", vp * 4), ("out", vp * 4)]


class FxParams(ctypes.Structure):
    _fields_ = [("op", ci), ("width", ci), ("height", ci), ("palette", ci), ("irow0", ci * 4), ("irow1", ci * 4), ("orow", ci * 4), ("ip", ci * 4), ("dp", ctypes.c_double * 2),
                ("frame_dp0", ctypes.POINTER(ctypes.c_double))]


# name -> argtypes; every entry point include/lives_gpu.h declares must appear here (tests check both ways)
PROTOTYPES = {
    "lgpu_abi_version": [],
    "lgpu_init": [ci],
    "lgpu_device_count": [],
    "lgpu_current_device": [vp],
    "lgpu_set_device": [ci],
    "lgpu_stream_query": [vp],
    "lgpu_malloc": [ctypes.POINTER(vp), ctypes.c_size_t],
    "lgpu_free": [vp],
    "lgpu_malloc_ordered": [ctypes.POINTER(vp), ctypes.c_size_t, vp],
    "lgpu_free_ordered": [vp, vp],
    "lgpu_stream_create": [ctypes.POINTER(vp), ctypes.c_int],
    "lgpu_stream_destroy": [vp],
    "lgpu_event_create": [ctypes.POINTER(vp)],
    "lgpu_event_destroy": [vp],
    "lgpu_event_record": [vp, vp],
    "lgpu_stream_wait_event": [vp, vp],
    "lgpu_debug_fail_alloc": [ci],
    "lgpu_upload": [vp, vp, ctypes.c_size_t, vp],
    "lgpu_download": [vp, vp, ctypes.c_size_t, vp],
    "lgpu_copy": [vp, vp, ctypes.c_size_t, vp],
    "lgpu_fill": [vp, ci, ctypes.c_size_t, vp],
    "lgpu_sync": [vp],
    "lgpu_dist_bind": [ctypes.c_char_p],
    "lgpu_dist_unique_id": [vp],
    "lgpu_dist_comm_create": [vp, ci, ci, ctypes.POINTER(vp)],
    "lgpu_dist_comm_create_timeout": [vp, ci, ci, ci, ctypes.POINTER(vp)],
    "lgpu_dist_comm_count": [vp],
    "lgpu_dist_comm_destroy": [vp],
    "lgpu_params_broadcast": [vp, ci, vp, vp],
    "lgpu_status_allreduce": [vp, vp, vp],
    "lgpu_fan_in": [vp, ci, ci, ci, ci, vp, ctypes.c_size_t, vp, vp],
    "lgpu_params_set": [vp, vp, vp],
    "lgpu_stepper_create": [vp, ci, ci, vp, vp, ctypes.POINTER(vp)],
    "lgpu_chain_step": [vp, vp, vp, vp, ci],
    "lgpu_stepper_feed": [vp, vp, ci],
    "lgpu_stepper_overlap": [vp, vp],
    "lgpu_stepper_failed": [vp],
    "lgpu_stepper_wait": [vp, ci],
    "lgpu_chain_check": [ctypes.POINTER(ChainParams), ctypes.POINTER(ChainTrack), ci],
    "lgpu_params_set_n": [vp, vp, ci, vp],
    "lgpu_params_broadcast_n": [vp, ci, vp, ci, vp],
    "lgpu_stepper_block": [vp, ci],
    "lgpu_stepper_destroy": [vp],
    "lgpu_copy_rows": [vp, ci, vp, ci, ci, ci, vp],
    "lgpu_fill_pattern": [vp, ci, vp, ci, ci, ci, vp],
    "lgpu_conversion_tables": [ci, vp, vp],
    "lgpu_gamma_lut8": [cd, ci, ci, cd, vp],
    "lgpu_calc_rowstrides": [ci, ci, ci, vp],
    "lgpu_swizzle": [ci, ci, vp, ci, vp, ci, ci, ci, vp, vp],
    "lgpu_gamma_apply": [vp, ci, ci, ci, ci, ci, ci, ci, vp, vp],
    "lgpu_alpha_premult": [vp, ci, ci, ci, ci, ci, vp],
    "lgpu_yuv420p_to_rgb": [vp, vp, vp, vp, cl, cl, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp],
    "lgpu_yuv420_tuning": [ci, ci, ci],
    "lgpu_tuning_set": [ctypes.c_char_p, ci],
    "lgpu_tuning_get": [ctypes.c_char_p],
    "lgpu_debug_recip_check": [ctypes.c_uint32, ctypes.c_uint32, vp],
    "lgpu_debug_stream_probe": [vp, vp, ci, ci, vp, vp],
    "lgpu_debug_pixbuf_cache_entries": [],
    "lgpu_debug_premult_yuv_tables_device": [vp],
    "lgpu_yuv420p_to_rgb_lut16": [vp, vp, vp, vp, cl, cl, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp],
    "lgpu_gamma_lut16": [cd, ci, ci, cd, vp],
    "lgpu_alpha_scalers": [vp, vp],
    "lgpu_letterbox": [vp, ci, ci, ci, vp, ci, ci, ci, ci, vp, vp],
    "lgpu_letterbox_at": [vp, ci, ci, ci, vp, ci, ci, ci, ci, vp, ci, ci, vp],
    "lgpu_letterbox_bars": [vp, ci, ci, ci, ci, vp, ci, ci, ci, ci, vp],
    "lgpu_resize": [vp, ci, ci, ci, vp, ci, ci, ci, ci, ci, vp, vp],
    "lgpu_pixbuf_scale": [vp, ci, ci, ci, vp, ci, ci, ci, ci, ci, vp],
    "lgpu_swizzle_batch": [ci, ci, vp, ci, vp, ci, ci, ci, vp, ci, vp],
    "lgpu_gamma_apply_batch": [vp, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp],
    "lgpu_alpha_premult_batch": [vp, ci, ci, ci, ci, ci, ci, vp],
    "lgpu_mirror_batch": [ci, vp, ci, vp, ci, ci, ci, ci, ci, vp],
    "lgpu_letterbox_batch": [vp, ci, ci, ci, vp, ci, ci, ci, ci, vp, ci, vp],
    "lgpu_colorkey_batch": [vp, ci, vp, ci, vp, ci, ci, ci, ci, ctypes.c_double, ctypes.c_double, ci, ci, ci, ci, vp],
    "lgpu_rgb_to_yuv_batch": [vp, ci, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, vp],
    "lgpu_yuv_to_rgb_batch": [vp, vp, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, vp],
    "lgpu_chain_amounts": [vp, vp, vp, ci, vp, vp],
    "lgpu_pixbuf_scale_check": [ci, ci, ci, ci, ci, ci, vp],
    "lgpu_pixbuf_scale_batch": [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "lgpu_fx_batch": [ctypes.POINTER(FxParams), ctypes.POINTER(FxFrame), ci, vp],
    "lgpu_chain_canvas": [vp, vp, vp, ci, vp],
    "lgpu_pixbuf_weights": [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ctypes.c_size_t],
    "lgpu_make_filter": [ci, ci, ci, vp, vp, vp, ci],
    "lgpu_gauss5": [vp, ci, vp, ci, ci, ci, ci, vp],
    "lgpu_blend_chroma": [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp],
    "lgpu_blend_luma": [ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp],
    "lgpu_blend_multi": [ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp],
    "lgpu_colorkey": [vp, ci, vp, ci, vp, ci, ci, ci, ci, cd, cd, ci, ci, ci, vp],
    "lgpu_gauss5_colorkey": [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, cd, cd, ci, ci, ci, vp],
    "lgpu_mirror": [ci, vp, ci, vp, ci, ci, ci, ci, vp],
    "lgpu_transition": [ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, cd, vp],
    "lgpu_yuv_repack": [ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "lgpu_rgbdelay_create": [vp],
    "lgpu_rgbdelay_process": [vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp],
    "lgpu_rgbdelay_destroy": [vp],
    "lgpu_fx_luts": [ci, ci, cd, cd, cd, vp],
    "lgpu_byte_luts": [vp, ci, vp, ci, ci, ci, ci, vp, vp],
    "lgpu_yuv420p_to_rgb_batch": [ci, vp, vp, ctypes.c_long, ctypes.c_long, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci, vp],
    "lgpu_pinned_calloc": [ctypes.c_size_t],
    "lgpu_pinned_free": [vp],
    "lgpu_premult_yuv_tables": [vp, vp, vp, vp],
    "lgpu_alpha_premult_yuva": [vp, vp, ci, ci, ci, ci, ci, vp],
    "lgpu_deinterlace": [vp, ci, vp, ci, ci, ci, ci, vp],
    "lgpu_triple_split": [vp, ci, vp, ci, vp, ci, ci, ci, ci, cd, ci, cd, ci, cd, vp, vp],
    "lgpu_dissolve_mask": [ctypes.c_uint64, ci, ci, vp],
    "lgpu_dissolve": [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, cd, vp],
    "lgpu_slide_over": [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "lgpu_softlight": [vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "lgpu_yuv_switch_clamping": [vp, vp, ci, ci, ci, vp],
    "lgpu_rgb_to_yuv": [vp, ci, ci, ci, ci, ci, vp, vp, ci, ci, ci, vp],
    "lgpu_chroma_average_table": [vp],
    "lgpu_rgb_to_yuv_lut16": [vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, vp, vp],
    "lgpu_yuv_to_rgb": [vp, vp, ci, ci, ci, ci, vp, ci, ci, ci, ci, vp],
    "lgpu_rgb_to_yuv411": [vp, ci, ci, ci, ci, ci, vp, ci, vp],
    "lgpu_yuv411_to_rgb": [vp, ci, ci, vp, ci, ci, ci, ci, vp],
    "lgpu_edge": [vp, ci, vp, ci, ci, ci, ci, ci, vp],
    "lgpu_blurzoom_create": [ci, ci, ci, vp],
    "lgpu_blurzoom_process": [vp, vp, ci, vp, ci, ci, ci, vp],
    "lgpu_blurzoom_destroy": [vp],
    "lgpu_composite": [vp, ci, ci, ci, ci, ci, vp, vp, ci, ci, vp],
    "lgpu_chain": [ctypes.POINTER(ChainParams), ctypes.POINTER(ChainTrack), ci, vp],
    "lgpu_chain_timed": [ctypes.POINTER(ChainParams), ctypes.POINTER(ChainTrack), ci, ci, ctypes.POINTER(ctypes.c_float), vp],
}

_lib = None


def load():
    """dlopen liblivesgpu.so; raise (never fall back) when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise LgpuError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(lives_amd/csrc/build.sh).  There is no CPU fallback." % SO_PATH)
        lib = ctypes.CDLL(SO_PATH)
        for name, args in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = ci
        lib.lgpu_pinned_calloc.restype = vp
        lib.lgpu_stepper_block.restype = vp
        lib.lgpu_pinned_free.restype = None
        lib.lgpu_last_error.restype = ctypes.c_char_p
        lib.lgpu_last_error.argtypes = []
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc < 0:
        msg = load().lgpu_last_error()
        raise LgpuError("%s failed (%d): %s" % (what or "lgpu call", rc, msg.decode() if msg else ""))
    return rc


def call(name, *args):
    """call an entry point and raise LgpuError on a negative return code"""
    return check(getattr(load(), name)(*args), name)
