#!/usr/bin/env python3
"""tools/bench_one.py <op> [<op> ...] -- device time per launch of single entry points (the cases of tools/prof_one.py) by HIP-graph replay over rotating buffers:
no host time between the kernels.  Prints one line per op: us, GB/s of algorithmic bytes, fraction of 8 TB/s.  LGPU_* switches apply (read once per process).
--two-streams: launches alternate between two streams inside the graph.  --cold: as many buffer sets per op as make 1.6 GB -- every launch then reads bytes the 256 MiB memory-side cache has long lost; the default six
sets of a 1080p op (under 100 MB together) stay inside that cache from one replay to the next, which a frame that has just been uploaded does not."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch         # noqa: E402
from lives_amd import ops   # noqa: E402
import prof_one      # noqa: E402


def main():
    ops.init(0)
    names = sys.argv[1:]
    cold = two = False
    nstreams = 2
    while names and names[0].startswith("--"):
        if names[0] == "--cold":
            cold = True
        elif names[0] == "--two-streams":      # launches alternate between two streams inside the graph (independent buffer sets): the drain of one overlaps the ramp-up of the next
            two = True
        elif names[0].startswith("--streams="):      # the same over N streams (buffer sets a multiple of N)
            two = True
            nstreams = int(names[0].split("=")[1])
        names = names[1:]
    for op in names:
        prof_one.NB = 6
        if cold:
            prof_one.NB = 1
            _, nbytes = prof_one.case(op)
            prof_one.NB = max(4, int(1.6e9 / nbytes) + 1)
            prof_one.NB += (-prof_one.NB) % (nstreams if two else 2)
            prof_one.COLD = True
            torch.cuda.empty_cache()
        fn, nbytes = prof_one.case(op)
        for i in range(60):
            fn(i)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        others = [torch.cuda.Stream() for _ in range(nstreams - 1)]
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        per = 4 * prof_one.NB if prof_one.NB <= 8 else prof_one.NB
        per -= per % nstreams if two else 0
        if two and prof_one.NB % nstreams:
            raise SystemExit("N streams need a multiple of N buffer sets (a set stays on one stream)")
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                if two:
                    for o in others:
                        o.wait_stream(side)
                for i in range(per):
                    if two and (i % nstreams):
                        with torch.cuda.stream(others[i % nstreams - 1]):
                            fn(i)
                    else:
                        fn(i)
                if two:
                    for o in others:
                        side.wait_stream(o)
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (20 * per))
        print("%-28s %7.2f us  %7.1f GB/s  %.3f of 8 TB/s%s" % (op, best, nbytes / best / 1e3, nbytes / best / 1e3 / 8000.0, ("   (cold: %d buffer sets)" % prof_one.NB if cold else "") + ("   (%d streams)" % nstreams if two else "")), flush=True)
        del fn, graph
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
