"""bench.py itself on the GPU, short: the line's contract (one JSON object with roofline and the 8-GPU denominators), the two-stream option, and the N > 1 host path
forced onto one GPU (library RCCL communicator of one rank, C stepper, config-5 legs with and without the gaussian)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [1, 2])
def test_bench_line(streams):
    j = run_bench(["--launch-streams", str(streams)])
    assert j["metric"] == "effect-chain frames/sec at 3840x2160 RGBA32" and j["unit"] == "frames/s" and j["n_gpus"] == 1 and j["steps"] == 8 and j["warmup"] == 2
    assert j["value"] > 2000 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["dtype"] == "u8" and j["vs_baseline"] is None
    c, r = j["config"], j["roofline"]
    assert c["launch_streams"] == streams and c["tracks_per_gpu"] == 16 and c["buffer_sets_rotated"] == 2
    assert c["batch8_groups_rotated"] == 4 and 40 < c["batch8_1gpu_us_per_step"] < 400 and c["batch8_blur_1gpu_us_per_step"] > c["batch8_1gpu_us_per_step"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.2 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 16 * (3840 * 2160 * 4 + 2 * 1920 * 1080 * 4) and "ONE stream" in r["timed_as"]
    assert r["box_class"]["stream_probe_us"] > 0


@pytest.mark.gpu
def test_bench_multi_gpu_host_path_on_one_gpu():
    j = run_bench([], env={"LGPU_BENCH_FORCE_EXCHANGE": "1"})
    c = j["config"]
    assert "lgpu_chain_step" in c["param_exchange"]
    for k in ("config5_ms_per_step", "config5_blur_ms_per_step", "projected_batch8_speedup", "projected_batch8_blur_speedup"):
        assert c[k] > 0, k
    assert c["config5_blur_ms_per_step"] > c["config5_ms_per_step"]
    assert c["projected_batch8_speedup"] > 3.0
    assert c["rccl_ranks"] == 1 and j["per_rank_ms_per_step"]["ranks"] == 1


@pytest.mark.gpu
def test_bench_under_the_drivers_launcher_agrees_with_the_plain_run():
    """The driver's N = 1 leg of the scaling run is `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`: the same workload, a value within 5 %
    of the plain run's, the per-rank clocks and the config-5 figures present at every rank count."""
    import socket
    plain = run_bench(["--steps", "200", "--warmup", "50"])
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu", "--steps", "200", "--warmup", "50"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["config"]["workload"] == plain["config"]["workload"] and j["config"]["tracks_per_gpu"] == plain["config"]["tracks_per_gpu"] and j["n_gpus"] == 1
    assert abs(j["value"] - plain["value"]) <= 0.05 * plain["value"], (j["value"], plain["value"])
    for line in (j, plain):
        assert line["per_rank_ms_per_step"]["ranks"] == 1 and abs(line["per_rank_ms_per_step"]["max"] - line["ms_per_step"]) < 1e-3
    assert j["config"]["config5_ms_per_step"] > 0 and j["config"]["config5_blur_ms_per_step"] > j["config"]["config5_ms_per_step"]
    assert "config5_ms_per_step" not in plain["config"], "the plain run keeps its profile clean of one-frame launches (--config5 asks for them)"


@pytest.mark.gpu
def test_a_missing_peer_ends_with_a_message_not_a_hang(gpu):
    """lgpu_dist_comm_create_timeout: rank 0 of a two-rank job whose rank 1 never starts gets LGPU_E_TIMEOUT and a message naming the rank that waited;
    lgpu_stepper_wait: a launch stream that does not drain within the limit fails the stepper with what it was waiting for"""
    import ctypes
    import time
    import numpy as np
    import torch
    from lives_amd import dist as ld, lib
    L = lib.load()
    idbuf = (ctypes.c_uint8 * 128)()
    if L.lgpu_dist_bind(None) == 0:
        assert L.lgpu_dist_unique_id(idbuf) == 0
        c = ctypes.c_void_p()
        t0 = time.time()
        rc = L.lgpu_dist_comm_create_timeout(idbuf, 0, 2, 1500, ctypes.byref(c))
        assert rc == -7 and 1.0 < time.time() - t0 < 30.0
        assert b"rank 0 of 2" in L.lgpu_last_error()
    # a stepper whose launch stream is held up by a host callback that sleeps past the limit
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
    nap = CB(lambda _: time.sleep(1.5))
    own = torch.cuda.Stream()
    st = ld.Stepper(None, [10], stream=own)
    try:
        st.wait(2000)                                       # nothing outstanding: returns at once
        assert hip.hipLaunchHostFunc(ctypes.c_void_p(own.cuda_stream), nap, None) == 0
        with pytest.raises(lib.LgpuError) as ei:
            st.wait(200)
        assert "waited" in str(ei.value) and "launch" in str(ei.value) and st.failed()
        torch.cuda.synchronize()
    finally:
        st.close()
