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
