"""Multi-GPU layer of the engine: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The path shards by independent units -- frames of live clip tracks -- so there is NO data-path collective:
track t belongs to rank t % world.  The only exchange per frame batch is the shared transition
parameter block (blend amount, key colour, ...), broadcast from the control rank; it stays in device memory
and the chain kernel reads it from there (lgpu_chain_params.param_block_d), so the broadcast is stream
ordered and needs no host round trip.
"""
import torch
import torch.distributed as dist

PARAM_BLOCK_INTS = 4      # int32[0] = blend amount; the rest reserved (key colour, timecode lo / hi)


def shard_tracks(ntracks, rank, world):
    """tracks owned by `rank`: t % world == rank"""
    return [t for t in range(ntracks) if t % world == rank]


def new_param_block(device):
    return torch.zeros(PARAM_BLOCK_INTS, dtype=torch.int32, device=device)


def publish_params(block, values=None, src=0):
    """control rank writes `values` (sequence of ints or a tensor) into the block, everyone receives it"""
    if values is not None and (not dist.is_initialized() or dist.get_rank() == src):
        if torch.is_tensor(values):
            block.copy_(values, non_blocking=True)
        else:
            block.copy_(torch.tensor(list(values) + [0] * (PARAM_BLOCK_INTS - len(values)), dtype=torch.int32))
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(block, src=src)
    return block


class ParamPipeline:
    """Double-buffered parameter block.  prefetch(s + 1) is called BEFORE the kernel of batch s is launched: the broadcast
    is issued behind everything already on the launch stream (the kernel of batch s - 1, the last reader of that buffer)
    and then travels over xGMI on RCCL's own stream while the kernel of batch s runs; acquire(s) makes the launch stream
    wait for it (a stream-level wait, no host sync)."""

    def __init__(self, device, src=0):
        self.blocks = [new_param_block(device), new_param_block(device)]
        self.pending = [None, None]
        self.src = src

    def prefetch(self, s, values):
        b = self.blocks[s & 1]
        multi = dist.is_initialized() and dist.get_world_size() > 1
        if values is not None and (not multi or dist.get_rank() == self.src):
            if torch.is_tensor(values):
                b.copy_(values, non_blocking=True)
            else:
                b.copy_(torch.tensor(list(values) + [0] * (PARAM_BLOCK_INTS - len(values)), dtype=torch.int32))
        if multi:
            self.pending[s & 1] = dist.broadcast(b, src=self.src, async_op=True)

    def acquire(self, s):
        w = self.pending[s & 1]
        if w is not None:
            w.wait()
            self.pending[s & 1] = None
        return self.blocks[s & 1]


def max_over_ranks(seconds, device):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fan_in(local_frames, ntracks, dst=0):
    """Compositing fan-in (SURVEY 8f "next" 1): gather the processed frames of all tracks on rank `dst`.

    local_frames: this rank's frames in the order of shard_tracks(ntracks, rank, world) -- equal-shaped uint8 tensors.
    Returns the ntracks frames in track order on `dst`, None elsewhere.  Over RCCL this is a gather on the xGMI links
    into `dst`; every rank contributes ceil(ntracks / world) slots (short ranks pad with their last frame, dropped on
    arrival) so that one collective moves the whole batch.
    """
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return list(local_frames)
    rank, world = dist.get_rank(), dist.get_world_size()
    slots = (ntracks + world - 1) // world
    mine = list(local_frames)
    assert len(mine) == len(shard_tracks(ntracks, rank, world)) and mine, "every rank must own at least one track"
    while len(mine) < slots:
        mine.append(mine[-1])
    send = torch.stack(mine)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    out = [None] * ntracks
    for r in range(world):
        for i, t in enumerate(shard_tracks(ntracks, r, world)):
            out[t] = recv[r][i]
    return out

