#!/usr/bin/env python3
"""tools/bench_seam.py -- the weed_layer_t seam on one 1080p frame: YUV420P -> RGBA32 -> gamma -> resize 0.5x -> letterbox, unpinned (every call
crosses PCIe twice) and pinned (lives_gpu_layer_pin: one upload, one download).  Wall-clock per frame on the host, PCIe bytes from the seam's counters.
Needs oracle/_ref/libweedall.so (the reference's libweed, built by oracle/ref/build_ref.sh) for genuine weed plants."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from lives_amd import lib
    from tests import weedhost as wh
    L = lib.load()
    wh.bind(L)
    assert L.lgpu_init(0) == 0
    rng = np.random.default_rng(5)
    w, h = 1920, 1080
    Y = rng.integers(16, 236, (h, w), dtype=np.uint8)
    U = rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8)
    V = rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8)

    def stats():
        a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        L.lives_gpu_transfer_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value + b.value

    per_call = [0.0, 0.0, 0.0, 0.0]

    def chain(lay):
        t = [time.perf_counter()]
        assert L.lives_gpu_convert_layer_palette(lay, 3, 0) == 1
        t.append(time.perf_counter())
        assert L.lives_gpu_gamma_convert_layer(1, lay) == 1
        t.append(time.perf_counter())
        assert L.lives_gpu_resize_layer(lay, 960, 540, 3, 0, 0) == 1
        t.append(time.perf_counter())
        assert L.lives_gpu_letterbox_layer(lay, 960, 600, 960, 540, 3, 0, 0) == 1
        t.append(time.perf_counter())
        for i in range(4):
            per_call[i] += t[i + 1] - t[i]

    for pinned in (0, 1):
        n = 30
        layers = [wh.new_layer(512, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1) for _ in range(n + 3)]
        for lay in layers[:3]:                      # warm-up
            if pinned:
                L.lives_gpu_layer_pin(lay)
            chain(lay)
            if pinned:
                L.lives_gpu_layer_unpin(lay)
        b0 = stats()
        per_call[:] = [0.0, 0.0, 0.0, 0.0]
        t_pin = t_chain = t_sync = 0.0
        t0 = time.perf_counter()
        for lay in layers[3:]:
            a = time.perf_counter()
            if pinned:
                L.lives_gpu_layer_pin(lay)
            b = time.perf_counter()
            chain(lay)
            c = time.perf_counter()
            if pinned:
                L.lives_gpu_layer_unpin(lay)
            d = time.perf_counter()
            t_pin += b - a; t_chain += c - b; t_sync += d - c
        dt = (time.perf_counter() - t0) / n
        print("%s: %.3f ms per frame (4 seam calls), %.1f MB over PCIe per frame" % ("pinned  " if pinned else "unpinned", dt * 1e3, (stats() - b0) / n / 1e6), flush=True)
        if pinned:
            print("          of which pin (upload 3.1 MB + sync) %.3f ms | the four calls (enqueue only, no synchronisation) %.3f ms | unpin (wait for the kernels, download 2.3 MB) %.3f ms"
                  % (t_pin / n * 1e3, t_chain / n * 1e3, t_sync / n * 1e3), flush=True)
            print("          per call, us: convert %.1f | gamma %.1f | resize %.1f | letterbox %.1f" % tuple(x / n * 1e6 for x in per_call), flush=True)
    # the same unpinned chain when the host's frames are page-locked: lives_gpu_pinned_calloc / _free bound as the seam's pixel_alloc / pixel_free, the decoder's
    # planes from it too -- every plane then crosses PCIe by DMA straight from / to its own memory instead of through the 4 MB staging chunks
    L.lives_gpu_pinned_calloc.restype = ctypes.c_void_p
    L.lives_gpu_pinned_calloc.argtypes = [ctypes.c_size_t]
    W = wh.weed()
    api = wh.WeedApi(W.fn["weed_leaf_get"], W.fn["weed_leaf_set"], W.fn["weed_leaf_num_elements"], W.fn["weed_leaf_delete"],
                     ctypes.cast(L.lives_gpu_pinned_calloc, ctypes.c_void_p).value, ctypes.cast(L.lives_gpu_pinned_free, ctypes.c_void_p).value)
    assert L.lives_gpu_bind_weed(ctypes.byref(api)) == 0
    keep_malloc_copy = wh.malloc_copy

    def pinned_copy(arr):
        pm = L.lives_gpu_pinned_calloc(arr.nbytes + 64)
        ctypes.memmove(pm, arr.ctypes.data, arr.nbytes)
        return pm
    wh.malloc_copy = pinned_copy
    n = 12
    layers = [wh.new_layer(512, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1) for _ in range(n + 2)]
    for lay in layers[:2]:
        chain(lay)
    b0 = stats()
    t0 = time.perf_counter()
    for lay in layers[2:]:
        chain(lay)
    dt = (time.perf_counter() - t0) / n
    print("unpinned, page-locked frames (lives_gpu_pinned_calloc as the frame allocator; one hipHostMalloc + memset per new plane -- a host would pool them): %.3f ms per frame (4 seam calls), %.1f MB over PCIe per frame"
          % (dt * 1e3, (stats() - b0) / n / 1e6), flush=True)
    wh.malloc_copy = keep_malloc_copy
    wh.bind(L)
    # the host allocator's share: the convert call frees the decoder's three planes (3.1 MB the host wrote) with the bound pixel_free -- libc's free here,
    # i.e. an munmap of ~760 touched pages; LiVES' own frames come from its bigblock pool (src/memory.c) and cost nothing to release
    libc = ctypes.CDLL("libc.so.6")
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    blocks = []
    for _ in range(20):
        pm = libc.malloc(Y.nbytes + U.nbytes + V.nbytes + 64)
        ctypes.memmove(pm, Y.ctypes.data, Y.nbytes)
        ctypes.memmove(pm + Y.nbytes, U.ctypes.data, U.nbytes)
        ctypes.memmove(pm + Y.nbytes + U.nbytes, V.ctypes.data, V.nbytes)
        blocks.append(pm)
    t0 = time.perf_counter()
    for pm in blocks:
        libc.free(pm)
    print("          (libc free() of one decoder frame's touched 3.1 MB, which the convert call includes: %.1f us)" % ((time.perf_counter() - t0) / 20 * 1e6), flush=True)
    # a resident chain as a render loop runs it: layers pinned once (decoder output uploaded), the chain enqueued for a batch of frames, one wait at the end
    n = 30
    layers = [wh.new_layer(512, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1) for _ in range(n)]
    for lay in layers:
        L.lives_gpu_layer_pin(lay)
    L.lgpu_sync(None)
    per_call[:] = [0.0, 0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for lay in layers:
        chain(lay)
    t1 = time.perf_counter()
    L.lgpu_sync(None)
    t2 = time.perf_counter()
    print("resident : %.3f ms per frame host time for the four calls, %.3f ms per frame until the device has finished all %d frames" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, n), flush=True)
    print("          per call, us: convert %.1f | gamma %.1f | resize %.1f | letterbox %.1f" % tuple(x / n * 1e6 for x in per_call), flush=True)
    for lay in layers:
        L.lives_gpu_layer_unpin(lay)


if __name__ == "__main__":
    main()
