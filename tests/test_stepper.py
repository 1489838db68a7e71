"""lgpu_stepper / lgpu_chain_step: the C render-worker step (north_star "host code stays C").  One GPU here: with comm = NULL (no exchange) and with a
ONE-RANK RCCL communicator, which runs the whole host path -- side stream, events, lgpu_params_set, ncclBroadcast -- that the multi-GPU batch runs."""
import numpy as np
import pytest

from oracle import pyoracle as po
from util import dev, host

P = po.P
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("exchange", [0, 1])
def test_steps_use_the_block_of_their_own_step(gpu, orc, exchange):
    import torch
    from lives_amd import dist as ld
    rng = np.random.default_rng(0x57E9 + exchange)
    sw, sh, dw, dh = 256, 144, 128, 72
    srcs = [rng.integers(0, 256, (sh, sw * 4), dtype=np.uint8) for _ in range(2)]
    l2s = [rng.integers(0, 256, (dh, dw * 4), dtype=np.uint8) for _ in range(2)]
    schedule = [17, 200, 0, 255, 96, 131, 64] + [int(v) for v in rng.integers(0, 256, 38)]      # 45 steps: the 16-slot block ring wraps twice, 5 fences
    comm = ld.RcclComm("cuda") if exchange else None
    d_srcs, d_l2s = [dev(s) for s in srcs], [dev(s) for s in l2s]
    outs = [[dev(np.zeros((dh, dw * 4), np.uint8)) for _ in range(2)] for _ in schedule]
    prm = gpu.chain_params(sw, sh, sw * 4, dw, dh, dw * 4, dw * 4, swap_rb=1, interp=3, do_blur=0, bf=1, lut=None)      # bf = 1 must never be used
    st = ld.Stepper(comm, [schedule[0]])
    try:
        for s, bf in enumerate(schedule):
            nxt = [schedule[s + 1], 0, 0, 0] if s + 1 < len(schedule) else None
            st.step(nxt, prm, gpu.chain_tracks(d_srcs, d_l2s, outs[s]))
        torch.cuda.synchronize()
    finally:
        st.close()
        if comm is not None:
            comm.close()
    for s, bf in enumerate(schedule):
        for i in range(2):
            want = np.zeros((dh, dw * 4), np.uint8)
            assert orc.orc_chain(P(srcs[i]), sw * 4, sw, sh, P(l2s[i]), dw * 4, P(want), dw * 4, dw, dh, 1, 3, 0, bf, None) == 0
            assert (host(outs[s][i]) == want).all(), "step %d (blend amount %d) track %d" % (s, bf, i)
