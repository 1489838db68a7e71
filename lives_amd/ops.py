"""Torch-tensor front end over the C ABI (device memory + streams come from PyTorch-ROCm; every pixel is
computed by liblivesgpu.so).  Frames are 2-D uint8 CUDA tensors [rows, rowstride_bytes].
"""
import ctypes

import numpy as np
import torch

from . import lib

vp = ctypes.c_void_p


def stream_ptr():
    return vp(torch.cuda.current_stream().cuda_stream)


def dptr(t, byte_offset=0):
    assert t.is_cuda and t.dtype == torch.uint8
    return vp(t.data_ptr() + byte_offset)


def lut_ptr(lut):
    """host LUT (numpy uint8[256] or None) -> pointer kept alive by the caller"""
    if lut is None:
        return None
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    assert lut.size == 256
    return lut.ctypes.data_as(vp), lut


def init(device=0):
    torch.cuda.set_device(device)
    lib.call("lgpu_init", device)


def tuning(name, value):
    """a launch-shape / ablation switch by name (lgpu_tuning_set; value < 0 or None clears it); returns the previous value (-1: unset)"""
    L = lib.load()
    old = L.lgpu_tuning_get(name.encode())
    assert L.lgpu_tuning_set(name.encode(), -1 if value is None else int(value)) == 0, lib.last_error()
    return old


def swizzle(op, src, dst, width, height, alpha_first=0, lut=None):
    lp = lut_ptr(lut)
    lib.call("lgpu_swizzle", op, alpha_first, dptr(src), src.stride(0), dptr(dst), dst.stride(0), width, height,
             lp[0] if lp else None, stream_ptr())


def gamma_apply(pix, width, height, psize, lut, alpha_first=0, x=0, y=0):
    lp = lut_ptr(lut)
    lib.call("lgpu_gamma_apply", dptr(pix), pix.stride(0), x, y, width, height, psize, alpha_first, lp[0] if lp else None,
             stream_ptr())


def alpha_premult(pix, width, height, alpha_first=0, un=0):
    lib.call("lgpu_alpha_premult", dptr(pix), pix.stride(0), width, height, alpha_first, un, stream_ptr())


def alpha_premult_yuva(planes, width, height, palette, clamped, un=0):
    """K9b: YUVA8888 (589, one packed plane) / YUVA4444P (545, four planes), in place"""
    pp, ss = _plane_tables(planes)
    lib.call("lgpu_alpha_premult_yuva", ctypes.addressof(pp), ctypes.addressof(ss), width, height, palette, int(bool(clamped)), int(bool(un)), stream_ptr())


def rgb_to_yuv411(src, dst, width, height, in_order=0, in_alpha=0, unclamped=0):
    """K4b: RGB family -> compact YUV411 rows ((width >> 2) * 6 bytes)"""
    lib.call("lgpu_rgb_to_yuv411", dptr(src), src.stride(0), width, height, in_order, in_alpha, dptr(dst), int(bool(unclamped)), stream_ptr())


def yuv411_to_rgb(src, dst, width_mp, height, out_order=0, out_alpha=0, unclamped=0):
    """K3b: compact YUV411 rows (width_mp * 6 bytes) -> RGB / BGR / ARGB"""
    lib.call("lgpu_yuv411_to_rgb", dptr(src), width_mp, height, dptr(dst), dst.stride(0), out_order, out_alpha, int(bool(unclamped)), stream_ptr())


def yuv420p_to_rgb(y, u, v, dst, width, height, opsize=4, out_order=0, is_422=0, which_tables=0, pb_quality=2, lut=None,
                   flags=0, u_size=None, v_size=None):
    strides = (ctypes.c_int * 3)(y.stride(0), u.stride(0), v.stride(0))
    lp = lut_ptr(lut)
    lib.call("lgpu_yuv420p_to_rgb", dptr(y), dptr(u), dptr(v), strides, u_size if u_size is not None else u.numel(),
             v_size if v_size is not None else v.numel(), dptr(dst), dst.stride(0), width, height, opsize, out_order, is_422,
             which_tables, pb_quality, lp[0] if lp else None, flags, stream_ptr())


def yuv420p_to_rgb_batch(frames, width, height, opsize=4, out_order=0, is_422=0, which_tables=0, pb_quality=2, lut=None, flags=0):
    """frames: list of (y, u, v, dst) device tensors sharing one geometry and rowstrides; one launch for all of them"""
    n = len(frames)
    arr = (lib.YuvFrame * n)()
    for i, (y, u, v, d) in enumerate(frames):
        arr[i].y_d, arr[i].u_d, arr[i].v_d, arr[i].dst_d = y.data_ptr(), u.data_ptr(), v.data_ptr(), d.data_ptr()
    y, u, v, d = frames[0]
    st = (ctypes.c_int * 3)(y.stride(0), u.stride(0), v.stride(0))
    lp = lut_ptr(lut)
    lib.call("lgpu_yuv420p_to_rgb_batch", n, arr, st, u.numel(), v.numel(), d.stride(0), width, height, opsize, out_order, is_422, which_tables,
             pb_quality, lp[0] if lp else None, flags, stream_ptr())


def yuv420p_to_rgb_lut16(y, u, v, dst, width, height, lut16, opsize=4, out_order=0, is_422=0, which_tables=0, pb_quality=2, flags=0):
    """lut16: device tensor of 65536 16-bit values (the reference's fused LUT16 variant)"""
    assert lut16.is_cuda and lut16.numel() == 65536 and lut16.element_size() == 2
    strides = (ctypes.c_int * 3)(y.stride(0), u.stride(0), v.stride(0))
    lib.call("lgpu_yuv420p_to_rgb_lut16", dptr(y), dptr(u), dptr(v), strides, u.numel(), v.numel(), dptr(dst), dst.stride(0), width, height,
             opsize, out_order, is_422, which_tables, pb_quality, lut16.data_ptr(), flags, stream_ptr())


def letterbox(src, dst, width, height, nwidth, nheight, psize, black):
    b = (ctypes.c_uint8 * 4)(*black)
    lib.call("lgpu_letterbox", dptr(src), src.stride(0), width, height, dptr(dst), dst.stride(0), nwidth, nheight, psize, b,
             stream_ptr())


def resize(src, dst, sw, sh, dw, dh, psize=4, interp=3, lut=None):
    lp = lut_ptr(lut)
    lib.call("lgpu_resize", dptr(src), src.stride(0), sw, sh, dptr(dst), dst.stride(0), dw, dh, psize, interp,
             lp[0] if lp else None, stream_ptr())


def gauss5_colorkey(src0, src1, dst, width, height, psize, is_bgr, delta, opac, col):
    lib.call("lgpu_gauss5_colorkey", dptr(src0), src0.stride(0), dptr(src1), src1.stride(0), dptr(dst), dst.stride(0), width, height, psize, int(is_bgr),
             float(delta), float(opac), int(col[0]), int(col[1]), int(col[2]), stream_ptr())


def pixbuf_scale(src, dst, sw, sh, dw, dh, channels=4, interp=3):
    lib.call("lgpu_pixbuf_scale", dptr(src), src.stride(0), sw, sh, dptr(dst), dst.stride(0), dw, dh, channels, interp, stream_ptr())


def ptr_array(tensors):
    """host array of device pointers (what the *_batch entry points take)"""
    return (ctypes.c_void_p * len(tensors))(*[dptr(t) for t in tensors])


def pixbuf_scale_batch(srcs, dsts, sw, sh, dw, dh, channels=4, interp=3):
    lib.call("lgpu_pixbuf_scale_batch", ptr_array(srcs), ptr_array(dsts), len(srcs), srcs[0].stride(0), sw, sh, dsts[0].stride(0), dw, dh, channels, interp, stream_ptr())


def gauss5(src, dst, width, height, psize=4):
    lib.call("lgpu_gauss5", dptr(src), src.stride(0), dptr(dst), dst.stride(0), width, height, psize, stream_ptr())


def blend_chroma(src1, src2, dst, width, height, psize, bf, alpha_first=0):
    lib.call("lgpu_blend_chroma", dptr(src1), src1.stride(0), dptr(src2), src2.stride(0), dptr(dst), dst.stride(0), width,
             height, psize, alpha_first, bf, stream_ptr())


def blend_luma(kind, src1, src2, dst, width, height, psize, pal_order, thresh):
    lib.call("lgpu_blend_luma", kind, dptr(src1), src1.stride(0), dptr(src2), src2.stride(0), dptr(dst), dst.stride(0), width,
             height, psize, pal_order, thresh, stream_ptr())


def blend_multi(kind, src1, src2, dst, width, height, is_bgr, bf):
    lib.call("lgpu_blend_multi", kind, dptr(src1), src1.stride(0), dptr(src2), src2.stride(0), dptr(dst), dst.stride(0), width,
             height, is_bgr, bf, stream_ptr())


def colorkey(src0, src1, dst, width, height, is_bgr, delta, opac, col):
    lib.call("lgpu_colorkey", dptr(src0), src0.stride(0), dptr(src1), src1.stride(0), dptr(dst), dst.stride(0), width, height,
             is_bgr, float(delta), float(opac), int(col[0]), int(col[1]), int(col[2]), stream_ptr())


def mirror(mode, src, dst, width, height, psize):
    lib.call("lgpu_mirror", mode, dptr(src), src.stride(0), dptr(dst), dst.stride(0), width, height, psize, stream_ptr())


def transition(kind, src1, src2, dst, width, height, psize, amount):
    lib.call("lgpu_transition", kind, dptr(src1), src1.stride(0), dptr(src2), src2.stride(0), dptr(dst), dst.stride(0), width, height, psize,
             float(amount), stream_ptr())


def fx_luts(kind, palette, p0=0., p1=0., p2=0.):
    """host tables of negate (0) / posterise (1, p0 = levels) / ccorrect (2, p0..p2 = r, g, b factors); returns uint8 [psize][256] or None"""
    out = np.zeros((4, 256), np.uint8)
    ps = lib.load().lgpu_fx_luts(kind, palette, float(p0), float(p1), float(p2), out.ctypes.data)
    return out[:ps].copy() if ps else None


def byte_luts(src, dst, width, height, psize, luts):
    luts = np.ascontiguousarray(luts, dtype=np.uint8)
    assert luts.shape == (psize, 256)
    lib.call("lgpu_byte_luts", dptr(src), src.stride(0), dptr(dst), dst.stride(0), width, height, psize, luts.ctypes.data, stream_ptr())


def deinterlace(src, dst, width, height, palette):
    """deinterlace.c:45-308; src is dst = in place"""
    lib.call("lgpu_deinterlace", dptr(src), src.stride(0), dptr(dst), dst.stride(0), width, height, palette, stream_ptr())


def triple_split(src1, src2, dst, width, height, is_bgr, start, sym, end, vert, bw, rgb):
    """layout_blends.c:24-113; src1 is dst = in place"""
    col = (ctypes.c_int * 3)(*[int(v) for v in rgb])
    lib.call("lgpu_triple_split", dptr(src1), src1.stride(0), dptr(src2), src2.stride(0), dptr(dst), dst.stride(0), width, height, int(bool(is_bgr)),
             float(start), int(bool(sym)), float(end), int(bool(vert)), float(bw), ctypes.addressof(col), stream_ptr())


def dissolve_mask(seed, width, height):
    """host float32 mask of the dissolve transition for an instance seed (multi_transitions.c:41-69)"""
    m = np.zeros(width * height, np.float32)
    assert lib.load().lgpu_dissolve_mask(ctypes.c_uint64(seed), width, height, m.ctypes.data) == 1
    return m


def dissolve(src1, src2, dst, width, height, psize, mask, amount):
    """mask: float32 device tensor of width * height values"""
    lib.call("lgpu_dissolve", dptr(src1), src1.stride(0), dptr(src2), src2.stride(0), dptr(dst), dst.stride(0), width, height, psize,
             mask.data_ptr(), float(amount), stream_ptr())


def slide_over(src1, src2, dst, width, height, psize, amount, direction, slide_lower=True, slide_upper=False):
    """slide_over.c:54-146; direction 1..4 as sover_init stores it"""
    lib.call("lgpu_slide_over", dptr(src1), src1.stride(0), dptr(src2), src2.stride(0), dptr(dst), dst.stride(0), width, height, psize,
             int(amount), int(direction), int(bool(slide_lower)), int(bool(slide_upper)), stream_ptr())


def _plane_tables(planes):
    n = len(planes)
    pp = (ctypes.c_void_p * 4)(*([dptr(t) for t in planes] + [None] * (4 - n)))
    ss = (ctypes.c_int * 4)(*([t.stride(0) for t in planes] + [0] * (4 - n)))
    return pp, ss


def rgb_to_yuv(src, dst_planes, width, height, in_order, in_alpha, out_fmt, out_alpha, which_tables):
    """K4 (see include/lives_gpu.h): src 2-D uint8 device tensor, dst_planes list of 2-D uint8 device tensors"""
    dp, ds = _plane_tables(dst_planes)
    lib.call("lgpu_rgb_to_yuv", dptr(src), src.stride(0), width, height, in_order, int(in_alpha), ctypes.addressof(dp), ctypes.addressof(ds),
             out_fmt, int(out_alpha), which_tables, stream_ptr())


def rgb_to_yuv_lut16(src, dst, width, height, in_order, in_alpha, out_fmt, unclamped, lut16):
    """K4 with the 16-bit gamma LUT inline (UYVY / YUYV only); lut16: device tensor of 65536 16-bit entries"""
    assert lut16.is_cuda and lut16.numel() == 65536 and lut16.element_size() == 2
    lib.call("lgpu_rgb_to_yuv_lut16", dptr(src), src.stride(0), width, height, in_order, int(in_alpha), dptr(dst), dst.stride(0), out_fmt,
             int(bool(unclamped)), lut16.data_ptr(), stream_ptr())


def yuv_to_rgb(src_planes, dst, width, height, in_fmt, in_alpha, out_order, out_alpha, which_tables):
    sp, ss = _plane_tables(src_planes)
    lib.call("lgpu_yuv_to_rgb", ctypes.addressof(sp), ctypes.addressof(ss), width, height, in_fmt, int(in_alpha), dptr(dst), dst.stride(0),
             out_order, int(out_alpha), which_tables, stream_ptr())


def yuv_switch_clamping(planes, palette, height, to_unclamped):
    pp, ss = _plane_tables(planes)
    lib.call("lgpu_yuv_switch_clamping", ctypes.addressof(pp), ctypes.addressof(ss), palette, height, int(to_unclamped), stream_ptr())


def yuv_repack(in_pal, out_pal, src_planes, dst_planes, width, height, unclamped=False, sampling=0):
    """YUV -> YUV repack (colourspace.c K5b); palettes are WEED_PALETTE_* numbers; raises LgpuError (LGPU_E_UNSUPPORTED) for pairs
    the library does not take"""
    sp, ss = _plane_tables(src_planes)
    dp, ds = _plane_tables(dst_planes)
    lib.call("lgpu_yuv_repack", in_pal, out_pal, ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(dp), ctypes.addressof(ds),
             width, height, int(bool(unclamped)), int(sampling), stream_ptr())


def softlight(src_planes, dst_planes, width, height, palette, unclamped):
    """planar YUV softlight (softlight.c): src_planes / dst_planes are lists of 2-D uint8 device tensors, one per plane"""
    n = len(src_planes)
    sp = (ctypes.c_void_p * 4)(*[dptr(t) for t in src_planes])
    dp = (ctypes.c_void_p * 4)(*[dptr(t) for t in dst_planes])
    ss = (ctypes.c_int * 4)(*[t.stride(0) for t in src_planes])
    ds = (ctypes.c_int * 4)(*[t.stride(0) for t in dst_planes])
    assert n in (3, 4)
    lib.call("lgpu_softlight", ctypes.addressof(sp), ctypes.addressof(ss), ctypes.addressof(dp), ctypes.addressof(ds), width, height,
             palette, int(unclamped), stream_ptr())


FX_SOFTLIGHT, FX_TRANSITION, FX_YUV411_TO_RGB, FX_GAUSS5_COLORKEY, FX_BLEND_CHROMA, FX_BLEND_LUMA, FX_BLEND_MULTI = 1, 2, 3, 4, 5, 6, 7


def fx_batch(op, ins0, outs, width, height, ins1=None, palette=0, ip=(0, 0, 0, 0), dp=(0., 0.), frame_dp0=None):
    """lgpu_fx_batch: ins0 / ins1 / outs are lists (one entry per frame) of lists of plane tensors; strides are taken from frame 0;
    frame_dp0: a value per frame in place of dp[0] (transitions: the amount)"""
    n = len(ins0)
    frames = (lib.FxFrame * n)()
    for f in range(n):
        for k, t in enumerate(ins0[f]):
            frames[f].in0[k] = dptr(t)
        for k, t in enumerate(ins1[f] if ins1 else []):
            frames[f].in1[k] = dptr(t)
        for k, t in enumerate(outs[f]):
            frames[f].out[k] = dptr(t)
    prm = lib.FxParams()
    prm.op, prm.width, prm.height, prm.palette = op, width, height, palette
    for k, t in enumerate(ins0[0]):
        prm.irow0[k] = t.stride(0)
    for k, t in enumerate(ins1[0] if ins1 else []):
        prm.irow1[k] = t.stride(0)
    for k, t in enumerate(outs[0]):
        prm.orow[k] = t.stride(0)
    for k in range(4):
        prm.ip[k] = int(ip[k]) if k < len(ip) else 0
    prm.dp[0], prm.dp[1] = float(dp[0]), float(dp[1]) if len(dp) > 1 else 0.
    if frame_dp0 is not None:
        per_frame = (ctypes.c_double * n)(*[float(v) for v in frame_dp0])
        prm.frame_dp0 = ctypes.cast(per_frame, ctypes.POINTER(ctypes.c_double))
    lib.call("lgpu_fx_batch", ctypes.byref(prm), frames, n, stream_ptr())


def edge(src, dst, width, height, palette, mode):
    lib.call("lgpu_edge", dptr(src), src.stride(0), dptr(dst), dst.stride(0), width, height, palette, mode, stream_ptr())


def comp_geometry(owidth, oheight, offs_x, offs_y, scale_x, scale_y):
    """pixel geometry of one compositor layer (compositor.c:197-212): offsets truncate, sizes round to even"""
    return (int(offs_x * float(owidth)), int(offs_y * float(oheight)),
            (int(owidth * scale_x + 1.) >> 1) << 1, (int(oheight * scale_y + 1.) >> 1) << 1)


def composite(dst, owidth, oheight, psize, layers, bgcol=(0, 0, 0), is_bgr=0, revz=0):
    """layers: list of (tensor or None, width, height, offs_x, offs_y, alpha) already scaled to their on-screen size"""
    n = len(layers)
    arr = (lib.CompLayer * max(1, n))()
    for i, (t, w, h, ox, oy, al) in enumerate(layers):
        arr[i].src_d = dptr(t) if t is not None else None
        arr[i].irow = t.stride(0) if t is not None else 0
        arr[i].width, arr[i].height, arr[i].offs_x, arr[i].offs_y, arr[i].alpha = w, h, ox, oy, float(al)
    bg = (ctypes.c_int * 3)(*[int(c) for c in bgcol])
    lib.call("lgpu_composite", dptr(dst), dst.stride(0), owidth, oheight, psize, int(is_bgr), ctypes.addressof(bg), ctypes.addressof(arr), n,
             int(revz), stream_ptr())


class Blurzoom:
    """stateful blurzoom instance (lgpu_blurzoom_*): one per filter instance and frame geometry"""

    def __init__(self, width, height, palette):
        h = ctypes.c_void_p()
        lib.call("lgpu_blurzoom_create", width, height, palette, ctypes.addressof(h))
        self.h = h

    def process(self, src, dst, mode=0, pattern=0):
        lib.call("lgpu_blurzoom_process", self.h, dptr(src), src.stride(0), dptr(dst), dst.stride(0), mode, pattern, stream_ptr())

    def close(self):
        if self.h:
            lib.load().lgpu_blurzoom_destroy(self.h)
            self.h = None


class RgbDelay:
    """stateful RGBdelay / YUVdelay instance (lgpu_rgbdelay_*): the frame ring lives in HBM inside the handle"""

    def __init__(self):
        h = ctypes.c_void_p()
        lib.call("lgpu_rgbdelay_create", ctypes.addressof(h))
        self.h = h

    def process(self, src, dst, width, height, palette, maxcache, on, strength, yuv_clamped=False):
        on = np.ascontiguousarray(on, dtype=np.int32)
        strength = np.ascontiguousarray(strength, dtype=np.float64)
        assert on.size == 153 and strength.size == 51
        lib.call("lgpu_rgbdelay_process", self.h, dptr(src), src.stride(0), dptr(dst), dst.stride(0), width, height, palette, int(bool(yuv_clamped)),
                 int(maxcache), on.ctypes.data, strength.ctypes.data, stream_ptr())

    def close(self):
        if self.h:
            lib.load().lgpu_rgbdelay_destroy(self.h)
            self.h = None


def chain_params(sw, sh, irow, dw, dh, irow2, orow, swap_rb=1, interp=3, do_blur=0, bf=128, lut=None, param_block=None):
    p = lib.ChainParams()
    p.param_block_d = param_block.data_ptr() if param_block is not None else None
    p.sw, p.sh, p.irow, p.dw, p.dh, p.irow2, p.orow = sw, sh, irow, dw, dh, irow2, orow
    p.swap_rb, p.interp, p.do_blur, p.bf = swap_rb, interp, do_blur, bf
    p.use_lut = 1 if lut is not None else 0
    if lut is not None:
        ctypes.memmove(p.lut8, np.ascontiguousarray(lut, dtype=np.uint8).ctypes.data, 256)
    return p


def chain_tracks(srcs, layer2s, dsts):
    n = len(srcs)
    arr = (lib.ChainTrack * n)()
    for i in range(n):
        arr[i].src_d, arr[i].layer2_d, arr[i].dst_d = srcs[i].data_ptr(), (layer2s[i].data_ptr() if layer2s is not None else None), dsts[i].data_ptr()
    return arr


def chain(params, tracks):
    lib.call("lgpu_chain", ctypes.byref(params), tracks, len(tracks), stream_ptr())


def chain_canvas(params, tracks, nwidth, nheight, offs_x, offs_y):
    cv = lib.Canvas(nwidth, nheight, offs_x, offs_y)
    lib.call("lgpu_chain_canvas", ctypes.byref(params), ctypes.byref(cv), tracks, len(tracks), stream_ptr())


def chain_amounts(params, tracks, amounts, canvas=None):
    """lgpu_chain_amounts: the chain with a blend amount per track; canvas = (nwidth, nheight, offs_x, offs_y) or None"""
    am = (ctypes.c_uint8 * len(tracks))(*[int(a) & 0xFF for a in amounts]) if amounts is not None else None      # None: with LGPU_INTERP_NOBLEND (0x400) in params.interp
    cv = lib.Canvas(*canvas) if canvas is not None else None
    lib.call("lgpu_chain_amounts", ctypes.byref(params), ctypes.byref(cv) if cv is not None else None, tracks, len(tracks), am, stream_ptr())


def stream_probe(params, tracks, reps):
    """lgpu_debug_stream_probe: the chain's algorithmic bytes as a bare stream on the same frames, ms for `reps` launches (destinations left dirty)"""
    ms = ctypes.c_float()
    lib.call("lgpu_debug_stream_probe", ctypes.byref(params), tracks, len(tracks), reps, ctypes.byref(ms), stream_ptr())
    return ms.value


def chain_timed(params, tracks, reps):
    ms = ctypes.c_float()
    lib.call("lgpu_chain_timed", ctypes.byref(params), tracks, len(tracks), reps, ctypes.byref(ms), stream_ptr())
    return ms.value
