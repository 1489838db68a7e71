#!/usr/bin/env python3
"""tools/bench_seam_mt.py -- the layer seam from N host threads (LiVES' plan steps run on pool threads, src/threading.c): every thread keeps its own
1080p layers resident and runs convert -> gamma -> resize 0.5x -> letterbox on them; frames per second over all threads, for N = 1, 2, 4, 8, 16.
The threads are C threads (tests/c/mt_host.c: Python threads spend the run handing the interpreter lock to each other); each enqueues on its own
stream, so the small kernels of different tracks overlap on the device.  Needs oracle/_ref/libweedall.so for genuine weed plants."""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Calls(ctypes.Structure):
    _fields_ = [("convert", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("resize", ctypes.c_void_p), ("letterbox", ctypes.c_void_p), ("layer_sync", ctypes.c_void_p), ("consume", ctypes.c_void_p),
                ("outpl", ctypes.c_int), ("gamma_type", ctypes.c_int), ("w", ctypes.c_int), ("h", ctypes.c_int), ("nw", ctypes.c_int), ("nh", ctypes.c_int)]


def build_host():
    out = os.path.join(tempfile.mkdtemp(prefix="mt_host"), "libmt_host.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "c", "mt_host.c"), "-lpthread"])
    H = ctypes.CDLL(out)
    H.mt_run_chains.argtypes = [ctypes.POINTER(Calls), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    return H


def fn(L, name):
    return ctypes.cast(getattr(L, name), ctypes.c_void_p).value


def main():
    from lives_amd import lib
    from tests import weedhost as wh
    L = lib.load()
    W = wh.bind(L)
    H = build_host()
    # LiVES' frames come from its bigblock pool (src/memory.c: no system call per frame).  With libc blocks every call would mmap / munmap its planes
    # (~170 us to hand back 3 MB of touched pages, all threads queueing for the process's memory-map lock), which is then what this bench measures.
    # tests/c/mt_host.c carries a pool of that kind; it is bound as the seam's pixel_alloc / pixel_free and the decoder's planes come from it too.
    sizes = (ctypes.c_size_t * 3)(768 << 10, 2560 << 10, 9 << 20)
    counts = (ctypes.c_int * 3)(4000, 4000, 256)
    assert H.mt_pool_init(sizes, counts) == 0
    H.mt_pool_alloc.restype = ctypes.c_void_p
    H.mt_pool_alloc.argtypes = [ctypes.c_size_t]
    api = wh.WeedApi(W.fn["weed_leaf_get"], W.fn["weed_leaf_set"], W.fn["weed_leaf_num_elements"], W.fn["weed_leaf_delete"],
                     ctypes.cast(H.mt_pool_alloc, ctypes.c_void_p).value, ctypes.cast(H.mt_pool_free, ctypes.c_void_p).value)
    assert L.lives_gpu_bind_weed(ctypes.byref(api)) == 0

    def pool_copy(arr):
        p = H.mt_pool_alloc(arr.nbytes + 64)
        ctypes.memmove(p, arr.ctypes.data, arr.nbytes)
        return p
    wh.malloc_copy = pool_copy
    assert L.lgpu_init(0) == 0
    rng = np.random.default_rng(5)
    w, h = 1920, 1080
    Y = rng.integers(16, 236, (h, w), dtype=np.uint8)
    U = rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8)
    V = rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8)
    calls = Calls(fn(L, "lives_gpu_convert_layer_palette"), fn(L, "lives_gpu_gamma_convert_layer"), fn(L, "lives_gpu_resize_layer"), fn(L, "lives_gpu_letterbox_layer"),
                  fn(L, "lives_gpu_layer_sync"), fn(L, "lives_gpu_layer_forget"), 3, 1, 960, 540, 960, 600)
    per = 32
    inplace = len(sys.argv) > 1 and sys.argv[1] == "inplace"      # diagnostic: four in-place gamma calls per RGBA layer instead of the chain
    if inplace:
        calls.outpl = -1
        R = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    for nthreads in (1, 1, 2, 4, 8, 16):
        if inplace:
            layers = [wh.new_layer(3, w, h, [R], gamma=1) for _ in range(nthreads * (1 + per))]
        else:
            layers = [wh.new_layer(512, w, h, [Y, U, V], gamma=-1, clamping=0, subspace=1) for _ in range(nthreads * (1 + per))]
        for lay in layers:
            assert L.lives_gpu_layer_pin(lay) == 0
        arr = (ctypes.c_void_p * len(layers))(*layers)
        secs, enq = ctypes.c_double(), ctypes.c_double()
        rc = H.mt_run_chains(ctypes.byref(calls), arr, nthreads, per, ctypes.byref(secs), ctypes.byref(enq))
        assert rc == 0, rc
        n = nthreads * per
        print("%2d host thread(s): %4d resident 1080p frames through the four calls in %6.2f ms -> %6.0f frames/s (%.1f us per frame; slowest thread spent %.2f ms enqueueing)"
              % (nthreads, n, secs.value * 1e3, n / secs.value, secs.value / n * 1e6, enq.value * 1e3), flush=True)
        gc = (ctypes.c_double * 4).in_dll(H, "g_call")
        if not inplace:
            print("      mean host time per call, us: convert %.1f | gamma %.1f | resize %.1f | letterbox %.1f" % tuple(gc[k] / n * 1e6 for k in range(4)), flush=True)
        for lay in layers:
            assert L.lives_gpu_layer_unpin(lay) == 0


if __name__ == "__main__":
    main()
