#!/usr/bin/env python3
"""tools/bench_resize.py -- lgpu_resize on the ratios the headline kernel does not cover (everything but the exact 2:1 bicubic case), RGBA32,
HIP events around back-to-back launches on rotating buffers; algorithmic bytes = source read + destination written."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
from lives_amd import ops   # noqa: E402

CASES = [("3840x2160 -> 1280x720 bicubic", 3840, 2160, 1280, 720, 3), ("1920x1080 -> 3840x2160 lanczos (BEST, enlarging)", 1920, 1080, 3840, 2160, 3),
         ("3840x2160 -> 1920x1080 bicubic (k_half8s)", 3840, 2160, 1920, 1080, 3), ("3840x2160 -> 1706x960 bicubic (letterbox inner size)", 3840, 2160, 1706, 960, 3),
         ("1920x1080 -> 1280x720 bicubic", 1920, 1080, 1280, 720, 3), ("3840x2160 -> 1920x1080 bilinear", 3840, 2160, 1920, 1080, 2)]


PIXBUF_CASES = [("pixbuf 3840x2160 -> 1920x1080 HYPER rgba", 3840, 2160, 1920, 1080, 3, 4), ("pixbuf 3840x2160 -> 1920x1080 BILINEAR rgba", 3840, 2160, 1920, 1080, 2, 4),
                ("pixbuf 3840x2160 -> 1920x1080 NEAREST rgba", 3840, 2160, 1920, 1080, 0, 4), ("pixbuf 3840x2160 -> 1920x1080 HYPER rgb24", 3840, 2160, 1920, 1080, 3, 3),
                ("pixbuf 3840x2160 -> 1706x960 HYPER rgba", 3840, 2160, 1706, 960, 3, 4), ("pixbuf 3840x2160 -> 1280x720 HYPER rgba", 3840, 2160, 1280, 720, 3, 4),
                ("pixbuf 1920x1080 -> 3840x2160 HYPER rgba", 1920, 1080, 3840, 2160, 3, 4), ("pixbuf 1920x1080 -> 3840x2160 BILINEAR rgba", 1920, 1080, 3840, 2160, 2, 4),
                ("pixbuf 1920x1080 -> 1280x720 HYPER rgba", 1920, 1080, 1280, 720, 3, 4), ("pixbuf 3840x2160 -> 1280x720 BILINEAR rgba", 3840, 2160, 1280, 720, 2, 4),
                ("pixbuf 1280x720 -> 1920x1080 HYPER rgba", 1280, 720, 1920, 1080, 3, 4), ("pixbuf 1280x720 -> 1920x1080 BILINEAR rgba", 1280, 720, 1920, 1080, 2, 4),
                ("pixbuf 1920x1080 -> 2560x1440 HYPER rgba", 1920, 1080, 2560, 1440, 3, 4), ("pixbuf 3840x2160 -> 960x540 HYPER rgba", 3840, 2160, 960, 540, 3, 4),
                ("pixbuf 1280x720 -> 3840x2160 HYPER rgba", 1280, 720, 3840, 2160, 3, 4), ("pixbuf 1280x720 -> 3840x2160 BILINEAR rgba", 1280, 720, 3840, 2160, 2, 4)]


def main():
    ops.init(0)
    pixbuf = "--pixbuf" in sys.argv
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    reps, nb = 200, 6
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for case in (PIXBUF_CASES if pixbuf else CASES):
        name, sw, sh, dw, dh, interp = case[:6]
        if only and not any(o in name for o in only):
            continue
        ch = case[6] if pixbuf else 4
        srcs = [torch.randint(0, 256, (sh, sw * ch), dtype=torch.uint8, device="cuda", generator=g) for _ in range(nb)]
        if pixbuf and ch == 4 and "--opaque" in sys.argv:
            for t_ in srcs:
                t_[:, 3::4] = 255           # the common frame: no translucency anywhere
            name += " (opaque)"
        dsts = [torch.zeros((dh, dw * ch), dtype=torch.uint8, device="cuda") for _ in range(nb)]
        if pixbuf:
            def run(s_, d_):
                ops.pixbuf_scale(s_, d_, sw, sh, dw, dh, channels=ch, interp=interp)
        else:
            def run(s_, d_):
                ops.resize(s_, d_, sw, sh, dw, dh, psize=4, interp=interp)
        t_end = time.perf_counter() + 0.08          # ~80 ms of the same launch first: clocks / power state as in a running pipeline
        i = 0
        while time.perf_counter() < t_end:
            for _ in range(50):
                run(srcs[i % nb], dsts[i % nb])
                i += 1
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            run(srcs[i % nb], dsts[i % nb])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        # the same launches replayed from a HIP graph: a Python / ctypes call costs ~8 us, which a 10 us kernel does not hide completely
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for i in range(4 * nb):
                        run(srcs[i % nb], dsts[i % nb])
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(8):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / (8 * 4 * nb))
        except Exception:           # noqa: BLE001
            torch.cuda.synchronize()
        ab = sw * sh * ch + dw * dh * ch
        print(json.dumps({"op": name, "us": round(us, 2), "algorithmic_bytes": ab, "GBs": round(ab / us / 1e3, 1), "frac_of_8TBs": round(ab / us / 1e3 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
