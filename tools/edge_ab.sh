#!/bin/bash
# edge detect 1080p RGBA: the general kernels (EDGE_NO_S=1: 4 launches per pass) against the quad form (3 launches), back-to-back launches, rocprofv3 kernel stats of both
cd $GRAFT_REPO_ROOT
cat > /tmp/edge_time.py <<'PY'
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from lives_amd import ops
import prof_one
ops.init(0)
for mode in (0, 2):
    w, h = 1920, 1080
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    src = [torch.randint(0, 256, (h, w * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(6)]
    dst = [torch.zeros_like(t) for t in src]
    for i in range(50): ops.edge(src[i % 6], dst[i % 6], w, h, 3, mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(300): ops.edge(src[i % 6], dst[i % 6], w, h, 3, mode)
    e1.record(); torch.cuda.synchronize()
    print("mode %d: %.2f us per call" % (mode, e0.elapsed_time(e1) * 1e3 / 300))
PY
for rep in 1 2; do
  echo "general: $(LGPU_EDGE_NO_S=1 python /tmp/edge_time.py 2>/dev/null | tr '\n' ' ')"
  echo "quads 16: $(LGPU_EDGE_TH=16 python /tmp/edge_time.py 2>/dev/null | tr '\n' ' ')"
  echo "quads 32: $(LGPU_EDGE_TH=32 python /tmp/edge_time.py 2>/dev/null | tr '\n' ' ')"
done
for eh in 16 32; do LGPU_EDGE_TH=$eh bash tools/pmc_case.sh gpurun_out/r04/pmc_edge_quads$eh k_edge python tools/prof_one.py edge 2>&1 | head -7; done
