"""Multi-GPU layer of the engine: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The path shards by independent units -- frames of live clip tracks -- so there is NO data-path collective:
track t belongs to rank t % world.  The only exchange per frame batch is the shared transition
parameter block (blend amount, key colour, ...), broadcast from the control rank; it stays in device memory
and the chain kernel reads it from there (lgpu_chain_params.param_block_d), so the broadcast is stream
ordered and needs no host round trip.  On GPUs the exchange runs through the library's C entry points (RcclComm ->
lgpu_params_broadcast / lgpu_fan_in, RCCL called directly); the torch.distributed forms below remain for the CPU (gloo) tests
of the protocol and as the id side channel.
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


class RcclComm:
    """This library's own RCCL communicator (include/lives_gpu.h, "multi-GPU exchange"): the per-batch traffic of the path -- the
    parameter block, the status word, the compositing fan-in -- goes through the C entry points a C render worker would call
    (lgpu_params_broadcast / lgpu_status_allreduce / lgpu_fan_in), not through torch.distributed.  torch.distributed (any backend)
    is only the side channel that carries the 128-byte communicator id to the other ranks once."""

    def __init__(self, device="cuda"):
        import ctypes
        from . import lib
        self.lib = lib
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        idbuf = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            lib.call("lgpu_dist_unique_id", idbuf)
        if self.world > 1:
            t = torch.tensor(list(idbuf), dtype=torch.uint8, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(t, src=0)
            idbuf = (ctypes.c_uint8 * 128)(*t.cpu().tolist())
        c = ctypes.c_void_p()
        lib.call("lgpu_dist_comm_create", idbuf, self.rank, self.world, ctypes.byref(c))
        self.comm = c

    def broadcast_params(self, block, root=0, stream=None):
        self.lib.call("lgpu_params_broadcast", self.comm, root, block.data_ptr(), _sp(stream))

    def status_max(self, status, stream=None):
        self.lib.call("lgpu_status_allreduce", self.comm, status.data_ptr(), _sp(stream))

    def fan_in(self, frames, ntracks, frame_bytes, gathered, root=0, stream=None):
        """frames: this rank's processed frames as ONE contiguous device tensor (nlocal x frame_bytes); gathered: on root, ntracks x frame_bytes"""
        self.lib.call("lgpu_fan_in", self.comm, root, self.rank, self.world, ntracks, frames.data_ptr(), frame_bytes,
                      gathered.data_ptr() if gathered is not None else None, _sp(stream))

    def close(self):
        if self.comm:
            self.lib.call("lgpu_dist_comm_destroy", self.comm)
            self.comm = None


class Stepper:
    """lgpu_stepper (include/lives_gpu.h): the per-step host path in C -- wait for this step's parameter block, exchange the next one on a side
    stream beside the kernel, launch the chain -- as one ctypes call.  comm: RcclComm or None (one GPU, nothing to exchange)."""

    def __init__(self, comm, first_values, root=0, stream=None):
        import ctypes
        from . import lib
        self.lib, self.ct = lib, ctypes
        self.h = ctypes.c_void_p()
        vals = (ctypes.c_int32 * 4)(*(list(first_values) + [0] * 4)[:4])
        lib.call("lgpu_stepper_create", comm.comm if comm is not None else None, root, comm.rank if comm is not None else 0, _sp(stream), vals, ctypes.byref(self.h))
        self._vals = (ctypes.c_int32 * 4)()

    def step(self, next_values, params, tracks):
        """next_values: the block of the following step (used on the root), None after the last one"""
        nv = None
        if next_values is not None:
            for i in range(4):
                self._vals[i] = int(next_values[i]) if i < len(next_values) else 0
            nv = self._vals
        self.lib.call("lgpu_chain_step", self.h, nv, self.ct.byref(params), tracks, len(tracks))

    def block_ptr(self, which):
        return self.lib.load().lgpu_stepper_block(self.h, which)

    def close(self):
        if self.h:
            self.lib.call("lgpu_stepper_destroy", self.h)
            self.h = None


def _sp(stream):
    if stream is not None:
        return stream.cuda_stream
    return torch.cuda.current_stream().cuda_stream


class ParamPipeline:
    """Double-buffered parameter block.  prefetch(s + 1) is called BEFORE the kernel of batch s is launched: the broadcast
    is issued behind everything already on the launch stream (the kernel of batch s - 1, the last reader of that buffer)
    and then travels over xGMI on RCCL's own stream while the kernel of batch s runs; acquire(s) makes the launch stream
    wait for it (a stream-level wait, no host sync)."""

    def __init__(self, device, src=0, comm=None):
        self.blocks = [new_param_block(device), new_param_block(device)]
        self.pending = [None, None]
        self.src = src
        self.comm = comm                  # RcclComm: the broadcast goes through lgpu_params_broadcast on a side stream
        if comm is not None:
            self.side = torch.cuda.Stream()
            self.events = [torch.cuda.Event(), torch.cuda.Event()]

    def prefetch(self, s, values):
        b = self.blocks[s & 1]
        if self.comm is not None:
            # behind everything already on the launch stream (the last reader of this buffer), then on the side stream while the next kernel runs
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                if values is not None and self.comm.rank == self.src:
                    b.copy_(values if torch.is_tensor(values) else torch.tensor(list(values) + [0] * (PARAM_BLOCK_INTS - len(values)), dtype=torch.int32), non_blocking=True)
                if self.comm.world > 1:
                    self.comm.broadcast_params(b, root=self.src, stream=self.side)
                self.events[s & 1].record(self.side)
            self.pending[s & 1] = self.events[s & 1]
            return
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
            if self.comm is not None:
                torch.cuda.current_stream().wait_event(w)     # a stream-level wait
            else:
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

