#!/usr/bin/env python3
"""HBM calibration on this box: device-to-device copy and read-only reduction of a buffer larger than the Infinity Cache
(torch / hipMemcpy kernels), for reading the chain kernel's rate against what the memory system delivers here."""
import json
import torch

def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

def main():
    nbytes = 800 * 1024 * 1024
    a = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print(json.dumps({"op": "d2d copy 800 MiB", "seconds": round(t, 6), "read_plus_write_GBs": round(2 * nbytes / t / 1e9, 1)}))
    # (a read-only figure comes from tools/hbm_stream.hip: torch.sum is a reduction kernel, not a bandwidth probe)
    c = torch.empty((nbytes // 4,), dtype=torch.int32, device="cuda")
    t = timeit(lambda: c.fill_(7))
    print(json.dumps({"op": "fill 800 MiB (write only)", "seconds": round(t, 6), "write_GBs": round(nbytes / t / 1e9, 1)}))

if __name__ == "__main__":
    main()
