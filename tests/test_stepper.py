"""lgpu_stepper / lgpu_chain_step: the C render-worker step (north_star "host code stays C").  One GPU here: with comm = NULL (no exchange) and with a
ONE-RANK RCCL communicator, which runs the whole host path -- side stream, events, lgpu_params_set, ncclBroadcast -- that the multi-GPU batch runs."""
import numpy as np
import pytest

from oracle import pyoracle as po
from util import dev, host

P = po.P
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("ahead", [1, 5, 16])
@pytest.mark.parametrize("exchange", [0, 1])
def test_steps_use_the_block_of_their_own_step(gpu, orc, exchange, ahead, overlap):
    import torch
    from lives_amd import dist as ld
    rng = np.random.default_rng(0x57E9 + exchange)
    sw, sh, dw, dh = 256, 144, 128, 72
    srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(2)]
    l2s = [rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8) for _ in range(2)]
    schedule = [17, 200, 0, 255, 96, 131, 64] + [int(v) for v in rng.integers(0, 256, 143)]      # 150 steps: the 64-slot block ring wraps twice, with fences
    # ahead == 1: one block ahead through lgpu_chain_step's next_values; otherwise lgpu_stepper_feed hands over the blocks of the next `ahead` steps in one exchange
    comm = ld.RcclComm("cuda") if exchange else None
    d_srcs, d_l2s = [dev(s) for s in srcs], [dev(s) for s in l2s]
    outs = [[dev(np.zeros((dh, dw * 4), np.uint8)) for _ in range(2)] for _ in range(32)]
    wants = {}
    for bf in set(schedule):
        for i in range(2):
            w_ = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_chain(P(srcs[i]), sw * 4, sw, sh, P(l2s[i]), dw * 4, P(w_), dw * 4, dw, dh, 1, 3, 0, bf, None) == 0
            wants[(bf, i)] = w_
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3, do_blur=0, bf=1, lut=None)      # bf = 1 must never be used
    st = ld.Stepper(comm, [schedule[0]])
    if overlap:                         # odd steps on a second launch stream (lgpu_stepper_overlap)
        st.overlap(torch.cuda.Stream())
    try:
        fed = 1
        for s, bf in enumerate(schedule):
            nxt = [schedule[s + 1], 0, 0, 0] if s + 1 < len(schedule) else None
            if ahead > 1:
                nxt = None
                if fed == s + 1 and fed < len(schedule):
                    rows = [[v, 0, 0, 0] for v in schedule[fed:fed + ahead]]
                    st.feed(rows)
                    fed += len(rows)
            st.step(nxt, prm, gpu.chain_tracks(d_srcs, d_l2s, outs[s % 32]))
            if s % 32 == 31 or s == len(schedule) - 1:         # compare in groups of 32 steps (the output buffers are reused)
                torch.cuda.synchronize()
                for t in range(s - s % 32, s + 1):
                    for i in range(2):
                        assert (host(outs[t % 32][i]) == wants[(schedule[t], i)]).all(), "step %d (blend amount %d) track %d" % (t, schedule[t], i)
        torch.cuda.synchronize()
    finally:
        st.close()
        if comm is not None:
            comm.close()


def test_stepper_refuses_what_would_put_it_out_of_step(gpu):
    """argument errors are found before anything is enqueued (the call can be repeated); a step without a block, a feed beyond the ring and a negative ring index are refused"""
    import ctypes
    from lives_amd import dist as ld, lib
    L = lib.load()
    rng = np.random.default_rng(5)
    sw, sh, dw, dh = 64, 32, 32, 16
    d_src, d_l2, d_out = dev(rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8)), dev(rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8)), dev(np.zeros((dh, dw * 4), np.uint8))
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=0, interp=3, do_blur=0, bf=1, lut=None)
    trk = gpu.chain_tracks([d_src], [d_l2], [d_out])
    st = ld.Stepper(None, [10])
    try:
        assert L.lgpu_stepper_block(st.h, -1) is None
        bad = gpu.chain_params(sw, sh, sw * 4 - 4, dw, dh, dw * 4, dw * 4, swap_rb=0, interp=3, do_blur=0, bf=1, lut=None)      # rowstride smaller than a row
        assert L.lgpu_chain_step(st.h, None, ctypes.byref(bad), trk, 1) == -2          # LGPU_E_BADARG, nothing changed ...
        # every check lgpu_chain makes is made before the next block is fed: a misaligned frame, a chain without a resize stage, an interp value the pixbuf
        # arithmetic does not have -- each with next_values given, none of them may feed it
        mis = gpu.chain_tracks([d_src], [d_l2], [d_out])
        mis[0].src_d = d_src.data_ptr() + 2
        nxt = (ctypes.c_int32 * 4)(77, 0, 0, 0)
        same = gpu.chain_params(sw, sh, sw * 4, sw, sh, sw * 4, sw * 4, swap_rb=0, interp=3, do_blur=0, bf=1, lut=None)
        tiles = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=0, interp=1 | 0x100, do_blur=0, bf=1, lut=None)
        assert L.lgpu_chain_step(st.h, nxt, ctypes.byref(prm), mis, 1) == -2
        assert L.lgpu_chain_step(st.h, nxt, ctypes.byref(same), trk, 1) == -2
        assert L.lgpu_chain_step(st.h, nxt, ctypes.byref(tiles), trk, 1) == -2
        assert L.lgpu_chain_check(ctypes.byref(tiles), trk, 1) == -2 and L.lgpu_chain_check(ctypes.byref(prm), trk, 1) == 0
        assert L.lgpu_stepper_failed(st.h) == 0
        st.step(None, prm, trk)                                                          # ... so the same step goes through afterwards
        assert L.lgpu_chain_step(st.h, None, ctypes.byref(prm), trk, 1) == -2          # no block for step 1
        st.feed([[v] for v in range(64)])                                                # a whole ring
        assert L.lgpu_stepper_feed(st.h, (ctypes.c_int32 * 4)(), 1) == -2               # one more does not fit
        for _ in range(64):
            st.step(None, prm, trk)
    finally:
        st.close()


def test_wait_covers_a_stepper_on_the_null_stream(gpu):
    """lgpu_stepper_create accepts the NULL stream as its launch stream (torch's default stream IS handle 0): lgpu_stepper_wait has to poll it like any other --
    it returns only once the launches are complete (queried right after it, with no other synchronisation), not after a look at the side stream alone"""
    import torch
    from lives_amd import dist as ld, lib
    L = lib.load()
    rng = np.random.default_rng(6)
    sw, sh, dw, dh = 3840, 2160, 1920, 1080
    d_src = dev(rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8))
    d_l2, d_out = dev(rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8)), dev(np.zeros((dh, dw * 4), np.uint8))
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3 | 0x100, do_blur=1, bf=1, lut=None)
    trk = gpu.chain_tracks([d_src] * 8, [d_l2] * 8, [d_out] * 8)
    assert torch.cuda.current_stream().cuda_stream == 0
    torch.cuda.synchronize()
    st = ld.Stepper(None, [10])                      # stream=None -> torch's current stream = the NULL stream
    try:
        st.feed([[v] for v in range(1, 40)])
        for _ in range(40):                           # ~40 x 8 tracks x 20 us: a few milliseconds of queued work
            st.step(None, prm, trk)
        assert L.lgpu_stream_query(None) == 0, "the launches are still running when the host gets here"
        st.wait(20000)
        assert L.lgpu_stream_query(None) == 1, "lgpu_stepper_wait came back while its launch stream was busy"
    finally:
        st.close()


def test_every_exchange_of_the_path_runs_against_self(gpu):
    """a ONE-RANK RCCL communicator on the GPU: the parameter broadcast, the status all-reduce, the compositing fan-in (lgpu_fan_in: its Send / Recv slots inside one
    RCCL group, 2 * world + 1 tracks) and 20 steps of the C stepper -- the whole of lives_amd.dist.preflight, which bench.py runs with real peers before it times
    anything.  Until a multi-GPU node exists this is as close to hardware as the three exchanges get."""
    from lives_amd import dist as ld
    comm = ld.RcclComm("cuda")
    try:
        assert comm.count() in (1,) or comm.count() < 0          # 1, or a negative LGPU_E_* where the bound librccl has no ncclCommCount
        assert ld.preflight(comm, "cuda", ops=gpu) == "ok"
    finally:
        comm.close()
