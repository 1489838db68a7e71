"""Multi-GPU layer of the engine: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The path shards by independent units -- frames of live clip tracks -- so there is NO data-path collective:
track t belongs to rank t % world.  The only exchange per frame batch is the shared transition
parameter block (blend amount, key colour, ...), broadcast from the control rank; it stays in device memory
and the chain kernel reads it from there (lgpu_chain_params.param_block_d), so the broadcast is stream
ordered and needs no host round trip.  On GPUs the exchange runs through the library's C entry points (RcclComm ->
lgpu_params_broadcast / lgpu_fan_in, RCCL called directly); the torch.distributed forms below remain for the CPU (gloo) tests
of the protocol and as the id side channel.
"""
import os

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
        # a rank of the job that never arrives must not hang the others for ever: LGPU_E_TIMEOUT after LGPU_COMM_TIMEOUT_MS (default 120 s) names the rank that waited
        lib.call("lgpu_dist_comm_create_timeout", idbuf, self.rank, self.world, int(os.environ.get("LGPU_COMM_TIMEOUT_MS", "120000")), ctypes.byref(c))
        self.comm = c

    def count(self):
        """ncclCommCount: the ranks this communicator really has"""
        return int(self.lib.load().lgpu_dist_comm_count(self.comm))

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

    def feed(self, rows):
        """lgpu_stepper_feed: the blocks of the next len(rows) steps in one exchange (rows: lists of up to 4 ints; read on the root only)"""
        n = len(rows)
        flat = (self.ct.c_int32 * (4 * n))()
        for i, r in enumerate(rows):
            for j in range(min(4, len(r))):
                flat[4 * i + j] = int(r[j])
        self.lib.call("lgpu_stepper_feed", self.h, flat, n)

    def overlap(self, second_stream):
        """lgpu_stepper_overlap: odd steps on a second launch stream (a torch.cuda.Stream, or None to switch it off); synchronise both before reading results"""
        self.lib.call("lgpu_stepper_overlap", self.h, second_stream.cuda_stream if second_stream is not None else None)
        self._second = second_stream

    def wait(self, timeout_ms=0):
        """lgpu_stepper_wait: everything fed and launched so far has completed, or LgpuError (LGPU_E_TIMEOUT) naming what hangs"""
        self.lib.call("lgpu_stepper_wait", self.h, int(timeout_ms))

    def failed(self):
        return bool(self.lib.load().lgpu_stepper_failed(self.h))

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


class TorchComm:
    """the same three exchanges on torch.distributed (gloo on CPU boxes): what preflight() runs on when there is no GPU, and the shape of the fallback"""

    def __init__(self):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def broadcast_params(self, block, root=0, stream=None):
        if self.world > 1:
            dist.broadcast(block, src=root)

    def status_max(self, status, stream=None):
        if self.world > 1:
            dist.all_reduce(status, op=dist.ReduceOp.MAX)

    def fan_in(self, frames, ntracks, frame_bytes, gathered, root=0, stream=None):
        mine = shard_tracks(ntracks, self.rank, self.world)
        got = fan_in([frames[i * frame_bytes:(i + 1) * frame_bytes] for i in range(len(mine))], ntracks, dst=root)
        if got is not None and gathered is not None:
            for t, f in enumerate(got):
                gathered[t * frame_bytes:(t + 1) * frame_bytes] = f

    def close(self):
        pass


def preflight(comm, device="cuda", ops=None, steps=20):
    """The exchanges of the path with the ranks of THIS job, before anything is timed (bench.py, N > 1): the parameter block from the root arrives on every rank,
    the status word is the maximum over the ranks, lgpu_fan_in puts 2 * world + 1 tracks into track order on the root, and -- with a GPU (`ops`) -- `steps` steps of the
    C stepper use the ROOT's value on every rank (oracle-free: each step's output equals a plain lgpu_chain launch of the same frame with that value).
    The ranks agree on a verdict after EVERY stage (through torch.distributed, not through the communicator under test), so a rank that fails a check never leaves
    the others inside the next collective.  Every rank returns the same string: "ok", or "failed: rank r: <reason>" for the lowest failing rank of the first
    failing stage.  A rank that hangs inside RCCL is not caught."""
    rank, world = comm.rank, comm.world

    def stage_broadcast():
        blk = torch.tensor([123 + world, 45, 6, 7] if rank == 0 else [0, 0, 0, 0], dtype=torch.int32, device=device)
        comm.broadcast_params(blk, root=0)
        if blk.cpu().tolist() != [123 + world, 45, 6, 7]:
            raise RuntimeError("broadcast: rank %d holds %s" % (rank, blk.cpu().tolist()))

    def stage_status():
        st = torch.tensor([rank + 1], dtype=torch.int32, device=device)
        comm.status_max(st)
        if int(st.cpu().item()) != world:
            raise RuntimeError("status word: %d instead of %d" % (int(st.cpu().item()), world))

    def stage_fan_in():
        ntracks, fb = 2 * world + 1, 4096
        mine = shard_tracks(ntracks, rank, world)
        frames = torch.cat([torch.full((fb,), (t + 1) & 255, dtype=torch.uint8, device=device) for t in mine])
        gathered = torch.zeros(ntracks * fb, dtype=torch.uint8, device=device) if rank == 0 else None
        comm.fan_in(frames, ntracks, fb, gathered, root=0)
        if rank == 0:
            g = gathered.cpu().view(ntracks, fb)
            for t in range(ntracks):
                if not bool((g[t] == ((t + 1) & 255)).all()):
                    raise RuntimeError("fan-in: slot %d holds track %d" % (t, int(g[t, 0]) - 1))

    def stage_stepper():
        g_ = torch.Generator(device=device)
        g_.manual_seed(0x2C1 + rank)
        sw, sh, dw, dh = 256, 144, 128, 72
        src = torch.randint(0, 256, (sh, sw * 4), dtype=torch.uint8, device=device, generator=g_)
        l2 = torch.randint(0, 256, (dh, dw * 4), dtype=torch.uint8, device=device, generator=g_)
        schedule = [(37 * s + 11) & 255 for s in range(steps)]                 # the same list on every rank; only the root hands it to the stepper
        outs = [torch.zeros((dh, dw * 4), dtype=torch.uint8, device=device) for _ in schedule]
        prm = ops.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | 0x100, do_blur=0, bf=1, lut=None)
        stp = Stepper(comm, [schedule[0]] if rank == 0 else [0])
        try:
            fed = 1
            for s in range(steps):
                if fed == s + 1 and fed < steps:             # blocks of up to 8 steps per exchange
                    rows = [[v if rank == 0 else 0, 0, 0, 0] for v in schedule[fed:fed + 8]]
                    stp.feed(rows)
                    fed += len(rows)
                stp.step(None, prm, ops.chain_tracks([src], [l2], [outs[s]]))
            torch.cuda.synchronize()
        finally:
            stp.close()
        ref = torch.zeros((dh, dw * 4), dtype=torch.uint8, device=device)
        for s, bf in enumerate(schedule):
            p2 = ops.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | 0x100, do_blur=0, bf=bf, lut=None)
            ops.chain(p2, ops.chain_tracks([src], [l2], [ref]))
            torch.cuda.synchronize()
            if not torch.equal(ref, outs[s]):
                raise RuntimeError("stepper: step %d did not blend with the root's amount %d" % (s, bf))

    def agree(why):          # one verdict for all
        if not (dist.is_initialized() and world > 1):
            return "failed: rank 0: " + why if why else None
        bad = torch.tensor([rank if why else world], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(bad, op=dist.ReduceOp.MIN)
        first = int(bad.item())
        if first >= world:
            return None
        msg = [why if rank == first else None]
        dist.broadcast_object_list(msg, src=first)
        return "failed: rank %d: %s" % (first, msg[0])

    for stage in [stage_broadcast, stage_status, stage_fan_in] + ([stage_stepper] if ops is not None else []):
        why = ""
        try:
            stage()
        except Exception as e:          # noqa: BLE001 -- whatever went wrong, the job goes on without the C exchange and says so
            why = "%s: %s" % (type(e).__name__, e)
        verdict = agree(why)
        if verdict:
            return verdict
    return "ok"


def max_over_ranks(seconds, device):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_over_ranks(seconds, device):
    """every rank's value, in rank order (a list of one without a process group)"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [seconds]
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = seconds
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


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

