"""Run under `python -m torch.distributed.run --nproc-per-node 2`: the library's RCCL entry points with a REAL peer (tests/test_rccl_peers.py launches this
when the box has two GPUs).  Checks, through the C entry points only:
  lgpu_params_broadcast : a block written on rank 0 arrives on rank 1
  lgpu_fan_in           : 5 tracks over 2 ranks (rank 0 owns 0, 2, 4; rank 1 owns 1, 3) land on the root in TRACK order (slot r + i * world)
  lgpu_chain_step       : 20 steps of the C stepper; every rank's kernel of step s blends with the amount the root scheduled for step s
Prints RCCL_PEERS_OK on rank 0."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    from lives_amd import dist as ld, ops
    from oracle import pyoracle as po
    ops.init(local)
    comm = ld.RcclComm("cuda")
    # ---- broadcast
    blk = torch.tensor([123, 45, 6, 7] if rank == 0 else [0, 0, 0, 0], dtype=torch.int32, device="cuda")
    comm.broadcast_params(blk, root=0)
    torch.cuda.synchronize()
    assert blk.cpu().tolist() == [123, 45, 6, 7], (rank, blk.cpu().tolist())
    # ---- status word
    st = torch.tensor([3 if rank == 1 else 0], dtype=torch.int32, device="cuda")
    comm.status_max(st)
    torch.cuda.synchronize()
    assert int(st.item()) == 3
    # ---- fan-in: 5 tracks, frame t is filled with bytes t + 1
    ntracks, fb = 5, 4096
    mine = ld.shard_tracks(ntracks, rank, world)
    frames = torch.cat([torch.full((fb,), t + 1, dtype=torch.uint8, device="cuda") for t in mine])
    gathered = torch.zeros(ntracks * fb, dtype=torch.uint8, device="cuda") if rank == 0 else None
    comm.fan_in(frames, ntracks, fb, gathered, root=0)
    torch.cuda.synchronize()
    if rank == 0:
        g = gathered.cpu().numpy().reshape(ntracks, fb)
        for t in range(ntracks):
            assert (g[t] == t + 1).all(), "fan-in slot %d holds %d" % (t, int(g[t, 0]))
    # ---- the C stepper: every rank its own track, the root's schedule
    rng = np.random.default_rng(0x2C1 + rank)
    sw, sh, dw, dh = 256, 144, 128, 72
    src = rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8)
    l2 = rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8)
    schedule = [int(v) for v in np.random.default_rng(99).integers(0, 256, 20)]        # the same list on both ranks; only rank 0 hands it to the stepper
    d_src, d_l2 = torch.from_numpy(src).cuda(), torch.from_numpy(l2).cuda()
    outs = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device="cuda") for _ in schedule]
    prm = ops.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3, do_blur=0, bf=1, lut=None)
    stp = ld.Stepper(comm, [schedule[0]] if rank == 0 else [0])
    for s in range(len(schedule)):
        nxt = None if s + 1 == len(schedule) else ([schedule[s + 1], 0, 0, 0] if rank == 0 else [0, 0, 0, 0])
        stp.step(nxt, prm, ops.chain_tracks([d_src], [d_l2], [outs[s]]))
    torch.cuda.synchronize()
    stp.close()
    o = po.oracle()
    for s, bf in enumerate(schedule):
        want = np.zeros((dh, dw * 4), np.uint8)
        assert o.orc_chain(po.P(src), sw * 4, sw, sh, po.P(l2), dw * 4, po.P(want), dw * 4, dw, dh, 1, 3, 0, bf, None) == 0
        assert (outs[s].cpu().numpy() == want).all(), "rank %d step %d (blend amount %d)" % (rank, s, bf)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_PEERS_OK")


if __name__ == "__main__":
    main()
