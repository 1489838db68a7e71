"""shared helpers for the parity tests"""
import numpy as np


def align(n, a=32):
    return (n + a - 1) // a * a


def frame(rng, w, h, psize, stride=None, extra_rows=0, alpha_mix=False, pad_px=0):
    """random packed frame, LiVES-style 32-byte aligned rowstride unless given"""
    if stride is None:
        stride = align((w + pad_px) * psize)
    a = rng.integers(0, 256, (h + extra_rows, stride), dtype=np.uint8)
    if alpha_mix and psize == 4:
        al = a[:, 3::4]
        al[rng.random(al.shape) < 0.5] = 255
    return a


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy()


def assert_same(got, want, w, h, psize, what="", mask=None):
    g = got[:h, :w * psize].reshape(h, w, psize)
    x = want[:h, :w * psize].reshape(h, w, psize)
    diff = (g != x).any(axis=2)
    if mask is not None:
        diff &= ~mask
    n = int(diff.sum())
    assert n == 0, "%s: %d mismatching pixels, first at %s got %s want %s" % (
        what, n, np.argwhere(diff)[0].tolist(), g[tuple(np.argwhere(diff)[0])].tolist(), x[tuple(np.argwhere(diff)[0])].tolist())


def assert_padding_untouched(got, before, w, h, psize, what=""):
    assert (got[:h, w * psize:] == before[:h, w * psize:]).all(), what + ": row padding was written"
    assert (got[h:] == before[h:]).all(), what + ": rows past the frame were written"
