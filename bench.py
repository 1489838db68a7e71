#!/usr/bin/env python3
"""bench.py -- effect-chain frames/sec at 3840x2160 RGBA32 (BASELINE.json metric) on N MI355X GPUs.

A "step" is one pass of the headline chain over one batch of synthetic tracks resident in HBM:
   3840x2160 BGRA32 -> [convert to RGBA32] -> resize 0.5x (LIVES_INTERP_BEST) -> chroma blend (bf) with a 1920x1080
   RGBA32 layer -> gamma LUT (linear->sRGB) -> 1920x1080 RGBA32        (north_star: convert->resize->blend->gamma)
The resize stage runs on the reference's gdk-pixbuf arithmetic by default (--resize-backend pixbuf: GDK_INTERP_HYPER, alpha-weighted, bit-exact to
gdk_pixbuf_scale_simple 2.42.8 -- the pinned parity path); --resize-backend polyphase is the repo's own bicubic spec standing in for libswscale.
one fused launch of liblivesgpu.so's lgpu_chain() per step, TRACKS_PER_GPU independent tracks per launch.
Multi-GPU: one process per GPU (torch.distributed / RCCL), tracks sharded track-per-rank with no data-path
collective; the shared transition parameter block (blend amount) is broadcast from rank 0 over RCCL every
step (lgpu_params_broadcast, the library's C entry point, on its own communicator) and read by the kernel from
device memory.  Weak scaling: per-GPU work is fixed.  `--tracks 1` is BASELINE config 5's shape
(one 4K frame per GPU per step); the default keeps 16 tracks per GPU so that one step exceeds the Infinity Cache.

Before the W warm-up steps the device is woken up with ~60 ms of the same launch (clock / power ramp; not part of the
schedule).  Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : algorithmic bytes / launch  divided by  the kernel's average launch duration, measured live with
                 HIP events on the launch stream (lgpu_chain_timed), against the 8 TB/s HBM3E peak
  cpu_baseline : the CPU oracle (a port of the reference path; oracle/) timed on this host's cores on a
                 bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SW, SH, DW, DH = 3840, 2160, 1920, 1080
TRACKS_PER_GPU = 16           # 16 x 33 MB sources = 531 MB per step: larger than the 256 MiB Infinity Cache
# 33,177,600 B source + 8,294,400 B layer 2 read + 8,294,400 B written, per frame per track
ALGO_BYTES_PER_FRAME = SW * SH * 4 + DW * DH * 4 + DW * DH * 4
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--tracks", type=int, default=TRACKS_PER_GPU)
    ap.add_argument("--blur", type=int, default=0, help="1: add the 5x5 gaussian stage (BASELINE config 5 chain)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--config5", action="store_true", help="N = 1 without a launcher: run the one-frame-per-step legs (config5_*) as well; launched jobs always do")
    ap.add_argument("--launch-streams", type=int, default=1, choices=[1, 2],
                    help="2: consecutive steps alternate between two launch streams (independent buffer sets; needs an even --sets): 4-6 %% more frames/s, "
                         "but a profiler then sees pairs of overlapping launches of about twice the duration each")
    ap.add_argument("--sets", type=int, default=2,
                    help="independent buffer sets (sources, layer 2, destinations) the steps rotate through; with 2, consecutive steps share no byte, "
                         "so nothing a step reads can still sit in the 256 MiB Infinity Cache from the step before")
    ap.add_argument("--l2-translucent", type=float, default=0.5,
                    help="fraction of layer-2 pixels with alpha < 255 (they take the reference's float scaling path); 0 = an opaque layer 2")
    ap.add_argument("--resize-backend", choices=["polyphase", "pixbuf"], default="pixbuf",
                    help="arithmetic of the resize stage: the repo's polyphase spec in the swscale body's place (bicubic; parity unpinned, libswscale is not in the "
                         "image) or the reference's gdk-pixbuf body (GDK_INTERP_HYPER, alpha-weighted; bit-exact to gdk-pixbuf 2.42.8)")
    ap.add_argument("--no-seam", action="store_true", help="skip the seam_chain leg (a profiled run then holds the launches behind `value` only under the headline kernel's name)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / reduce / print path only, on the gloo backend with no GPU work (tests/test_dist_cpu.py runs this on a CPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` run plainly: become the launcher -- one rank per GPU under torch.distributed.run on this node, same arguments
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or unset WORLD_SIZE and let bench.py launch itself" % (
        args.gpus, world, args.gpus)
    if args.dry_run:
        return dry_run(args, rank, world)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from lives_amd import dist as ld, lib, ops   # noqa: F401
    ops.init(local_rank)

    class RawStream:          # a HIP stream of the library's own making, with the one attribute lives_amd.dist wants
        def __init__(self):
            h = ctypes.c_void_p()
            lib.call("lgpu_stream_create", ctypes.byref(h), 1)
            self.cuda_stream = h.value
    # the two launch streams of the one-frame-per-GPU leg, made BEFORE anything else creates streams: HIP deals its few hardware queues out in creation order, and two
    # launch streams that share a queue do not overlap (tools/worker.c measured it: 13.4 us per step behind the communicator's streams, 8.8 us created first)
    launch_a, launch_b = RawStream(), RawStream()

    # ---- synthetic, device-resident inputs (seeded per rank) ----
    g = torch.Generator(device="cuda")
    g.manual_seed(0x11FE5 + rank)
    T = args.tracks
    nsets = max(1, args.sets)
    keep, trks = [], []
    for _ in range(nsets):
        srcs = [torch.randint(0, 256, (SH, SW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
        l2s = [torch.randint(0, 256, (DH, DW * 4), dtype=torch.uint8, device="cuda", generator=g) for _ in range(T)]
        for t in l2s:   # default: half of layer 2 opaque (integer blend path), half translucent (the reference's float scaling path), scattered per pixel
            a = t[:, 3::4]
            a[torch.rand(a.shape, device="cuda", generator=g) >= args.l2_translucent] = 255
        dsts = [torch.zeros((DH, DW * 4), dtype=torch.uint8, device="cuda") for _ in range(T)]
        keep.append((srcs, l2s, dsts))
        trks.append(ops.chain_tracks(srcs, l2s, dsts))

    def groups_of(n):
        """The frames of all buffer sets as launches of n tracks each (the 8-track and one-frame legs rotate through ALL of them, like the 16-track steps do)."""
        out = []
        for srcs, l2s, dsts in keep:
            for i in range(0, T - n + 1, n):
                out.append(ops.chain_tracks(srcs[i:i + n], l2s[i:i + n], dsts[i:i + n]))
        return out

    from lives_amd.lib import load
    lut = np.zeros(256, np.uint8)
    assert load().lgpu_gamma_lut8(1.0, -1, 1, 1.4, lut.ctypes.data) == 1     # WEED_GAMMA_LINEAR -> WEED_GAMMA_SRGB

    # shared transition parameter block: int32[4], [0] = blend amount; broadcast from rank 0 each step
    pblock = ld.new_param_block("cuda")
    schedule = torch.tensor([[(96 + 7 * s) % 256, 0, 0, 0] for s in range(args.steps + args.warmup)], dtype=torch.int32, device="cuda")
    prm = ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=1, interp=3 | (0x100 if args.resize_backend == "pixbuf" else 0), do_blur=args.blur, bf=128, lut=lut,
                           param_block=pblock)
    trk = trks[0]

    sched_base = schedule.data_ptr()

    # N > 1: the library's own RCCL communicator (lgpu_dist_comm_create; torch.distributed only carries its 128-byte id) and the C entry
    # point lgpu_params_broadcast for the per-step exchange, on a side stream, double buffered
    comm = None
    multi = world > 1 or bool(os.environ.get("LGPU_BENCH_FORCE_EXCHANGE"))     # FORCE: the N > 1 host path (one-rank RCCL communicator, C stepper, config-5 leg) on one GPU
    if multi:
        # the chain kernel is persistent and fills every CU with two 71.9 KB workgroups; a kernel with a large LDS footprint on another stream -- RCCL's
        # broadcast -- would wait for it to end (tools/corun_probe.py: 164 us instead of 20).  Sixteen free workgroup slots (two per XCD) cost 1 % of the
        # launch and let the broadcast of the next parameter block run beside it (profiles/r02/corun_probe.txt)
        if load().lgpu_tuning_get(b"CHAIN_SPARE_WGS") < 0:
            load().lgpu_tuning_set(b"CHAIN_SPARE_WGS", 16)
    if multi and not os.environ.get("LGPU_BENCH_TORCH_DIST"):
        # every rank first checks that it can bind RCCL at all (dlopen + symbols); the communicator is only created when ALL can, so that no rank
        # waits in ncclCommInitRank for one that gave up -- otherwise the parameter block falls back to torch.distributed's broadcast
        can = torch.tensor([1 if load().lgpu_dist_bind(None) == 0 else 0], dtype=torch.int32, device="cuda")
        if world > 1:
            dist.all_reduce(can, op=dist.ReduceOp.MIN)
        if int(can.item()) == 1:
            comm = ld.RcclComm("cuda")
        elif rank == 0:
            print("bench.py: librccl could not be bound on every rank; the parameter block goes through torch.distributed", file=sys.stderr)
    rccl_preflight = None
    if comm is not None and world > 1:
        # the exchanges of the path with the ranks of this job before anything is timed: broadcast with real peers, fan-in slots for 2 * world + 1 tracks, 20 steps of
        # the C stepper against plain launches.  A failure does not kill the job: the parameter block then goes through torch.distributed and the line says so
        rccl_preflight = ld.preflight(comm, "cuda", ops=ops)
        if rccl_preflight != "ok":
            if rank == 0:
                print("bench.py: RCCL preflight %s; the parameter block goes through torch.distributed" % rccl_preflight, file=sys.stderr)
            comm.close()
            comm = None
    nsched = args.steps + args.warmup
    sched_host = [[(96 + 7 * s) % 256, 0, 0, 0] for s in range(nsched)]
    # N > 1 with RCCL bound: the whole per-step host path is ONE C call, lgpu_chain_step (wait for this step's block, exchange the next on a side stream, launch)
    # Consecutive steps work on different buffer sets and are independent, so they MAY alternate between two launch streams (tools/worker_overlap16.sh,
    # profiles/r04/worker_overlap16.txt: 145 -> 136.5 us per 16-track step, 76 -> 70 per 8-track step).  Step s and step s + 2 share a stream and a buffer set, so
    # every buffer has one stream; with an odd number of sets the steps stay on one stream.
    # OFF by default: two launches that are in flight together share the device from their first workgroup on (the dispatcher serves both queues), so each lasts
    # about twice as long while a pair takes less than two -- rocprofv3's per-kernel average then reads ~250 us beside a 156 us `roofline.launch_us`
    # (profiles/r04/box_c/README).  The default bench keeps one stream, so that its profile and its line say the same thing; `--launch-streams 2` is the faster host.
    two_streams = args.launch_streams == 2 and nsets % 2 == 0
    stepper = ld.Stepper(comm, sched_host[0], stream=launch_a) if comm is not None else None
    if stepper is not None and two_streams:
        stepper.overlap(launch_b)
    pipe = ld.ParamPipeline("cuda", comm=None) if (world > 1 and comm is None) else None
    if pipe is not None:
        pipe.prefetch(0, schedule[0] if rank == 0 else None)

    AHEAD = 16          # parameter blocks per exchange (lgpu_stepper_feed): the schedule of a render is known ahead
    fed = [1]           # blocks handed to the stepper so far (the first one by its constructor)

    def step(s):
        if stepper is not None:
            if fed[0] == s + 1 and fed[0] < nsched:
                stepper.feed(sched_host[fed[0]:fed[0] + AHEAD])
                fed[0] = min(nsched, fed[0] + AHEAD)
            stepper.step(None, prm, trks[s % nsets])
            return
        if world > 1:
            # RCCL broadcast over xGMI, no host sync: the block of step s was sent while step s - 1 ran; send the next one now
            blk = pipe.acquire(s)
            if s + 1 < nsched:
                pipe.prefetch(s + 1, schedule[s + 1] if rank == 0 else None)
            prm.param_block_d = blk.data_ptr()
        else:
            prm.param_block_d = sched_base + 16 * s      # one GPU: nothing to exchange, the kernel reads step s of the resident schedule
            if two_streams:
                lib.call("lgpu_chain", ctypes.byref(prm), trks[s % nsets], T, (launch_b if s & 1 else launch_a).cuda_stream)
                return
        ops.chain(prm, trks[s % nsets])

    STEP_TIMEOUT_MS = int(os.environ.get("LGPU_STEP_TIMEOUT_MS", "120000"))

    def fence():
        if stepper is not None:
            stepper.wait(STEP_TIMEOUT_MS)      # a peer that never arrives ends the job with a message (which rank waited for what), not with a hang in the synchronise below
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # device wake-up (not a step of the schedule): ~60 ms of the same launch so that the power / clock state is the steady one
    # whatever --warmup says; measured: the first ~100 launches after idle run ~15 % slower
    for i in range(300):
        ops.chain(prm, trks[i % nsets])
    torch.cuda.synchronize()
    for s in range(args.warmup):
        step(s)
    fence()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s)
    fence()
    dt_own = time.perf_counter() - t0
    dt = ld.max_over_ranks(dt_own, "cuda")
    dt_all = ld.all_over_ranks(dt_own, "cuda")          # every rank's own clock around the same K steps (value uses the max)

    frames = world * T * args.steps
    fps = frames / dt

    # N > 1: BASELINE config 5's own shape as well -- ONE 4K frame per GPU per step, the exchange on -- through the same C step: once on the chain of the headline
    # (north_star's convert -> resize -> blend -> gamma) and once with config 5's 5x5 gaussian between the scaler and the blend
    prm_blur = prm if args.blur else ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=1, interp=3 | (0x100 if args.resize_backend == "pixbuf" else 0), do_blur=1, bf=128, lut=lut,
                                                      param_block=pblock)
    config5 = None
    # every rank count of a launched job (the driver's scaling run starts N = 1 under torch.distributed.run as well): at N = 1 the same C step without a communicator.
    # NOT in the plain `python bench.py`: its ~2,000 one-frame launches carry the 16-track launch's kernel name and would drag the average rocprofv3 --stats prints for it
    # away from roofline.launch_us (`--config5` asks for the leg there)
    if stepper is not None or (not multi and (args.config5 or "WORLD_SIZE" in os.environ)):
        if stepper is not None:
            stepper.close()
        stepper = None
        one = groups_of(1)                    # every frame of every buffer set in turn: the one-frame launches, too, see buffers the memory-side cache has long lost
        n5 = max(args.steps, 200)

        def one_frame_leg(p5):
            st5 = ld.Stepper(comm, sched_host[0], stream=launch_a)
            st5.overlap(launch_b)             # odd steps on the second launch stream: consecutive frames are independent, the drain of one launch overlaps the ramp-up of the next
            tot5, f5 = 50 + n5, 1

            def step5(i):
                nonlocal f5
                if f5 == i + 1 and f5 < tot5:
                    k = min(AHEAD, tot5 - f5)
                    st5.feed([sched_host[(f5 + j) % nsched] for j in range(k)])
                    f5 += k
                st5.step(None, p5, one[i % len(one)])
            for i in range(50):
                step5(i)
            st5.wait(STEP_TIMEOUT_MS)
            fence()
            t5 = time.perf_counter()
            for i in range(50, tot5):
                step5(i)
            st5.wait(STEP_TIMEOUT_MS)
            fence()
            d5 = ld.max_over_ranks(time.perf_counter() - t5, "cuda")
            st5.close()
            return d5
        d5 = one_frame_leg(prm)
        config5 = {"config5_fps": round(world * n5 / d5, 1), "config5_ms_per_step": round(d5 / n5 * 1e3, 4), "config5_steps": n5,
                   "config5_shape": "one 3840x2160 frame per GPU per step, lgpu_chain_step (C) with %s (%d blocks per lgpu_stepper_feed), "
                                    "steps alternating between two launch streams (lgpu_stepper_overlap)" % ("the RCCL parameter exchange on" if comm is not None else "no communicator (one GPU)", AHEAD)}
        if args.resize_backend == "pixbuf" and not args.blur:
            d5b = one_frame_leg(prm_blur)
            config5.update({"config5_blur_fps": round(world * n5 / d5b, 1), "config5_blur_ms_per_step": round(d5b / n5 * 1e3, 4)})

    # ---- roofline of the dominant kernel: HIP events on the launch stream around K launches ----
    reps = max(10, min(args.steps, 200))
    if nsets == 1:
        ms = ops.chain_timed(prm, trk, reps)       # the library's own event pair (hipEventRecord on the launch stream around `reps` launches)
    else:                                          # the same, around launches that rotate through the buffer sets: ops.chain launches on torch's current stream,
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)    # which is the stream these events are recorded on
        prm.param_block_d = sched_base if world == 1 else pblock.data_ptr()
        e0.record()
        for i in range(reps):
            ops.chain(prm, trks[i % nsets])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    launch_s = ms * 1e-3 / reps
    algo = ALGO_BYTES_PER_FRAME * T
    achieved = algo / launch_s / 1e9
    traffic, traffic_source = None, None
    try:   # HBM bytes per launch from the PMC passes of this round's build on this exact workload (tools/pmc.sh + tools/pmc_traffic.py -> profiles/pmc_traffic*.json, which names the commit)
        tf = "pmc_traffic_pixbuf.json" if args.resize_backend == "pixbuf" else "pmc_traffic.json"
        with open(os.path.join(ROOT, "profiles", tf)) as f:
            pj = json.load(f)
        if pj.get("tracks") == T and pj.get("blur") == args.blur:
            traffic = pj.get("hbm_bytes_per_launch")
            traffic_source = "profiles/%s@%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of that commit's build; not measured in this run)" % (tf, pj.get("commit"))
    except (OSError, ValueError):
        pass

    # the denominator of "aggregate at 8 GPUs on an 8-track batch": the same 8 tracks as ONE launch on ONE GPU (every rank measures its own; rank 0's is printed)
    batch8 = None
    if T >= 8:
        prm.param_block_d = sched_base if world == 1 else pblock.data_ptr()
        t8 = groups_of(8)                     # 16 tracks x 2 sets = four 8-track groups, 1.6 GB: with two groups the 256 MiB memory-side cache still holds part of a group when its turn
        for i in range(10):                   # comes again (tools/sets_ab.sh, profiles/r04/sets_ab.txt: 8 tracks 69 us over two groups, 82 us over four or more; 16 tracks 156 us either way)
            ops.chain(prm, t8[i % len(t8)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n8 = 20          # few launches: they carry the same kernel name as the 16-track launch in a rocprofv3 --stats run (30 of ~1,600: the average moves by < 1 %)
        e0.record()
        for i in range(n8):
            ops.chain(prm, t8[i % len(t8)])
        e1.record()
        torch.cuda.synchronize()
        us8 = e0.elapsed_time(e1) * 1e3 / n8
        batch8 = {"batch8_1gpu_fps": round(8e6 / us8, 1), "batch8_1gpu_us_per_step": round(us8, 2), "batch8_groups_rotated": len(t8)}
        if args.resize_backend == "pixbuf" and not args.blur:      # the same with config 5's gaussian in the chain
            prm_blur.param_block_d = prm.param_block_d
            for i in range(6):
                ops.chain(prm_blur, t8[i % len(t8)])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n8):
                ops.chain(prm_blur, t8[i % len(t8)])
            e1.record()
            torch.cuda.synchronize()
            batch8["batch8_blur_1gpu_us_per_step"] = round(e0.elapsed_time(e1) * 1e3 / n8, 2)
            # ... and on frames whose alpha is 255 everywhere (decoded video), with the caller's word for it (LGPU_INTERP_OPAQUE): the lighter instantiation, same bytes
            opq = []
            for srcs, l2s, dsts in keep:
                for i in range(0, T - 7, 8):
                    o_s = [t.clone() for t in srcs[i:i + 8]]
                    for t in o_s:
                        t[:, 3::4] = 255
                    opq.append((o_s, ops.chain_tracks(o_s, l2s[i:i + 8], dsts[i:i + 8])))
            prm_op = ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=1, interp=3 | 0x100 | 0x200, do_blur=1, bf=128, lut=lut, param_block=pblock)
            prm_op.param_block_d = prm.param_block_d
            for i in range(6):
                ops.chain(prm_op, opq[i % len(opq)][1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n8):
                ops.chain(prm_op, opq[i % len(opq)][1])
            e1.record()
            torch.cuda.synchronize()
            batch8["batch8_blur_opaque_sources_1gpu_us_per_step"] = round(e0.elapsed_time(e1) * 1e3 / n8, 2)
            del opq

    # ---- the same chain through the REFERENCE's API (N = 1, outside the timed region of `value`): the tracks as pinned weed_layer_t's, one host thread per track,
    # convert_layer_palette -> resize_layer -> "chroma blend" process_func -> gamma_convert_layer by the reference names (liblivesgpu_dropin.so + livesgpu_fx.so),
    # lives_gpu_layers_flush once per tick: the calls are recorded on the planes and become ONE lgpu_chain launch (tools/seam_host.c; include/lives_gpu_layer.h)
    seam = None
    if world == 1 and not args.blur and args.resize_backend == "pixbuf" and not args.dry_run and not args.no_seam:
        try:
            seam = seam_chain_leg(keep[0][0], keep[0][1], ops, fps, T)
        except Exception as e:      # noqa: BLE001 -- a second measurement, never fatal for the line
            seam = {"error": str(e)}

    # what this box's memory system gives the launch's own algorithmic bytes as a bare stream (no arithmetic, no re-reads): tells a slow box from a regression
    box = None
    if True:
        try:
            ops.stream_probe(prm, trks[0], 3)
            sp = sum(ops.stream_probe(prm, trks[i % nsets], 5) for i in range(2)) / 10.0 * 1e3      # 10 launches over both buffer sets, us per launch
            box = {"stream_probe_us": round(sp, 2), "stream_probe_GBps": round(algo / sp / 1e3, 1), "kernel_over_stream": round(launch_s * 1e6 / sp, 3),
                   "class": "fast-stream" if algo / sp / 1e3 >= 5600.0 else "slow-stream",
                   "what": "lgpu_debug_stream_probe: the launch's algorithmic bytes (every source / layer-2 byte read once, every result byte written once, 16-byte non-temporal accesses, no arithmetic)"}
        except Exception as e:      # noqa: BLE001 -- a measurement aid, never fatal
            box = {"error": str(e)}
    roof = {"bound": "hbm", "kernel": pixbuf_kernel_name(args) if args.resize_backend == "pixbuf" else "lgpu::k_half8s<0,0>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": algo, "launch_us": round(launch_s * 1e6, 2),
            "timed_as": "HIP events around %d back-to-back launches on ONE stream: the kernel's own duration, what rocprofv3 --kernel-trace --stats reports for it%s" % (
                reps, "; the steps behind `value` alternate between two launch streams, so ms_per_step is below it" if two_streams else ""),
            "box_class": box}

    out = None
    if rank == 0:
        out = {
            "metric": "effect-chain frames/sec at 3840x2160 RGBA32", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "3840x2160 BGRA32 -> convert(RGBA32) -> %s resize 0.5x%s -> chroma blend with 1920x1080 RGBA32 layer -> gamma LUT (linear->sRGB)"
                                   % ("gdk-pixbuf HYPER (alpha-weighted, pinned)" if args.resize_backend == "pixbuf" else "bicubic", " -> 5x5 gaussian" if args.blur else ""),
                       "resize_backend": args.resize_backend,
                       "tracks_per_gpu": T, "frames_per_step": world * T, "inputs": "HBM-resident", "parallelism": "track-per-gpu x%d" % world,
                       "param_exchange": ("none (one GPU: the kernel reads step s of the resident schedule)" if not multi else
                                          "lgpu_stepper_feed + lgpu_chain_step (C): the blocks of 16 steps per lgpu_params_set_n + ncclBroadcast on a side stream, ahead of the kernels that read them" if comm is not None else
                                          "torch.distributed broadcast (fallback)"),
                       "launch_streams": 2 if two_streams else 1,
                       "launches_per_step": 2 if (args.blur and args.resize_backend != "pixbuf") else 1, "layer2_translucent_fraction": args.l2_translucent, "buffer_sets_rotated": nsets},
            "roofline": roof,
        }
        if seam:
            out["seam_chain"] = seam
        if batch8:
            out["config"].update(batch8)
        if config5:
            out["config"].update(config5)
            if batch8:       # what 8 GPUs with one frame each would make of the 8-track batch that one GPU runs as one launch
                out["config"]["projected_batch8_speedup"] = round(batch8["batch8_1gpu_us_per_step"] / (config5["config5_ms_per_step"] * 1e3), 2)
                if "config5_blur_ms_per_step" in config5 and "batch8_blur_1gpu_us_per_step" in batch8:
                    out["config"]["projected_batch8_blur_speedup"] = round(batch8["batch8_blur_1gpu_us_per_step"] / (config5["config5_blur_ms_per_step"] * 1e3), 2)
        out["per_rank_ms_per_step"] = {"min": round(min(dt_all) / args.steps * 1e3, 4), "max": round(max(dt_all) / args.steps * 1e3, 4), "ranks": len(dt_all)}
        if multi:
            # the ranks the communicator really has (ncclCommCount), not what the launcher promised: a line that says N GPUs over a smaller communicator is refused
            nr = comm.count() if comm is not None else 0
            if nr < 0:           # ncclCommCount is bound as optional: a librccl without it (or a failed call) is "unknown", not a reason to lose the measurement
                from lives_amd import lib as _lib
                msg = _lib.load().lgpu_last_error()
                nr = "unavailable (%s)" % (msg.decode() if msg else "error %d" % nr)
            else:
                assert comm is None or nr == world, "the RCCL communicator has %d ranks, the job %d" % (nr, world)
            out["config"]["rccl_ranks"] = nr
            if world > 1:
                out["config"]["rccl_preflight"] = rccl_preflight if rccl_preflight is not None else "not run (librccl could not be bound on every rank)"
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.blur, args.resize_backend == "pixbuf")
    ctypes.CDLL(None).fflush(None)          # every rank: whatever librccl left in C stdio's buffer goes out now, not at exit behind rank 0's line
    if comm is not None:
        if world > 1:
            dist.barrier()
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)      # librccl prints its version banner through C stdio, which a pipe buffers until exit: out with it BEFORE the one JSON line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def seam_chain_leg(srcs, l2s, ops, value_fps, T, ticks=600, warm=150):
    """the headline chain through the two seams by the reference's names, from C host threads (tools/libseam_host.so)"""
    import numpy as np
    import torch
    from lives_amd import lib
    L = lib.load()
    so = os.path.join(ROOT, "tools", "libseam_host.so")
    if not os.path.exists(so):
        raise RuntimeError("tools/libseam_host.so is missing: run __graft_entry__.build()")
    Hs = ctypes.CDLL(so)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    Hs.seam_host_run.argtypes = [ci, ci, ci, ci, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(vp), ctypes.POINTER(ci)]
    if Hs.seam_host_init(os.path.join(ROOT, "lives_amd", "livesgpu_fx.so").encode()) != 0:
        raise RuntimeError("seam_host_init failed")
    L.lives_gpu_deferred_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    L.lives_gpu_deferred_stats.restype = None
    sp = (vp * T)(*[t.data_ptr() for t in srcs])
    lp = (vp * T)(*[t.data_ptr() for t in l2s])
    BT709 = 2
    res = {}
    for threads in (1, 0):
        st0, st1 = (ctypes.c_ulonglong * 4)(), (ctypes.c_ulonglong * 4)()
        ms, outs, orow = ctypes.c_double(), (vp * T)(), ci()
        L.lives_gpu_deferred_stats(st0)
        rc = Hs.seam_host_run(T, SW, SH, DW, DH, sp, lp, 128, BT709, ticks, warm, threads, ctypes.byref(ms), outs, ctypes.byref(orow))
        L.lives_gpu_deferred_stats(st1)
        if rc:
            raise RuntimeError("seam_host_run failed at step %d" % rc)
        n = ticks + warm
        res[threads] = (ms.value, (st1[1] - st0[1], st1[2] - st0[2], st1[3] - st0[3]), n)
        if threads:      # the bytes of the last tick against the same frames through lgpu_chain directly (same table, same amount)
            lut = np.zeros(256, np.uint8)
            assert L.lgpu_gamma_lut8(1.0, 1, BT709, 1.4, lut.ctypes.data) == 1
            ref = torch.zeros((DH, DW * 4), dtype=torch.uint8, device="cuda")
            prm = ops.chain_params(SW, SH, SW * 4, DW, DH, DW * 4, DW * 4, swap_rb=1, interp=3 | 0x100, do_blur=0, bf=128, lut=lut)
            ops.chain(prm, ops.chain_tracks([srcs[T - 1]], [l2s[T - 1]], [ref]))
            torch.cuda.synchronize()
            got = torch.empty_like(ref)
            lib.call("lgpu_copy_rows", got.data_ptr(), DW * 4, outs[T - 1], orow.value, DW * 4, DH, None)
            torch.cuda.synchronize()
            same = bool(torch.equal(got, ref))
        Hs.seam_host_release()
    ms_t, (launches, tracks, staged), n = res[1]
    fps = T * ticks / (ms_t * 1e-3)
    return {"fps": round(fps, 1), "frac_of_value": round(fps / value_fps, 3), "ms_per_tick": round(ms_t / ticks, 4), "tracks": T, "host_threads": T, "ticks": ticks,
            "one_host_thread_fps": round(T * ticks / (res[0][0] * 1e-3), 1),
            "chain_launches_per_tick": round(launches / n, 3), "tracks_per_launch": round(tracks / max(1, launches), 2), "programs_run_stage_by_stage": staged,
            "same_bytes_as_lgpu_chain": same,
            "what": "per tick and track, on one host thread per track: a BGRA32 weed layer whose frame is already in HBM (lives_gpu_layer_pin_device) -> convert_layer_palette(RGBA32) -> "
                    "resize_layer(1920x1080, LIVES_INTERP_BEST) -> process_func of livesgpu_fx.so's \"chroma blend\" (in place, amount 128) -> gamma_convert_layer(WEED_GAMMA_BT709), by the "
                    "reference's names out of liblivesgpu_dropin.so; then lives_gpu_layers_flush(layers, n) once per tick.  Wall clock over the ticks, host work included."}


def self_launch(n):
    """re-exec under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on 127.0.0.1 with a free port; the ranks inherit the arguments,
    rank 0 prints the one JSON line"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """everything around the GPU work: rendezvous (gloo), the track sharding, the max-over-ranks reduction, rank 0's one line"""
    import torch.distributed as dist
    from lives_amd import dist as ld
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    mine = ld.shard_tracks(world * args.tracks, rank, world)
    pf = ld.preflight(ld.TorchComm(), "cpu") if world > 1 else "ok"          # the preflight's plumbing (verdict shared by every rank) on gloo; the C stepper part needs a GPU
    dt = ld.max_over_ranks(1e-3 * (rank + 1), "cpu")
    dts = ld.all_over_ranks(1e-3 * (rank + 1), "cpu")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "effect-chain frames/sec at 3840x2160 RGBA32", "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "tracks_of_rank0": mine, "max_over_ranks_s": dt, "all_over_ranks_s": dts, "rccl_preflight": pf}))


def pixbuf_kernel_name(args):
    """the instantiation pb_half_geometry() (pixbuf.hip) picks for this launch, as rocprofv3 prints it: <CHAIN, HYPER, BLUR, ALIGNED>; ALIGNED (strips of 64 storing lanes)
    is the default (round 4); the last argument is SWAP (BGRA -> RGBA)"""
    from lives_amd.lib import load
    aligned = (not args.blur) and load().lgpu_tuning_get(b"PBH_ALIGNED") != 0        # the default since round 4; LGPU_PBH_ALIGNED=0 keeps the feeder-lane strips
    return "lgpu::k_pb_half<1, 1, %d, %d, 1, 0>" % (args.blur, 1 if aligned else 0)          # <CHAIN, HYPER, BLUR, ALIGNED, SWAP, OPAQUE>


def cpu_baseline(blur, pixbuf=True):
    """the oracle's threaded runner (reference row-slice rule, one thread per core) on a bounded sample"""
    from oracle import pyoracle as po
    so = po.build_oracle(native=True)
    lib_ = ctypes.CDLL(so)
    lib_.orc_bench_chain2.restype = ctypes.c_double
    lib_.orc_bench_chain2.argtypes = [ctypes.c_int] * 8
    interp = 3 | (0x100 if pixbuf else 0)          # the same resize arithmetic as the GPU leg
    cores = os.cpu_count() or 1
    probe = lib_.orc_bench_chain2(SW, SH, DW, DH, cores, 2, blur, interp)
    per = max(probe / 2, 1e-4)
    n = int(max(4, min(400, 12.0 / per)))          # ~12 s of CPU work
    secs = lib_.orc_bench_chain2(SW, SH, DW, DH, cores, n, blur, interp)
    one = lib_.orc_bench_chain2(SW, SH, DW, DH, 1, 2, blur, interp) / 2
    return {"value": round(n / secs, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same 3840x2160 chain (%s resize), %d threads (reference row-slice rule), gcc -O3 -march=native" % (n, "gdk-pixbuf HYPER" if pixbuf else "polyphase bicubic", cores),
            "single_thread_fps": round(1.0 / one, 2)}


if __name__ == "__main__":
    main()
