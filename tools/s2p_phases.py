import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lives_amd import ops
ops.init(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
for (sw, sh, dw, dh) in ((3840, 2160, 1280, 720), (1920, 1080, 3840, 2160), (1920, 1080, 1280, 720)):
    src = torch.randint(0, 256, (sh, sw * 4), dtype=torch.uint8, device="cuda", generator=g)
    dst = torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda")
    os.environ.pop("X", None)
    for i in range(3):
        ops.resize(src, dst, sw, sh, dw, dh, psize=4, interp=3)
    torch.cuda.synchronize()
