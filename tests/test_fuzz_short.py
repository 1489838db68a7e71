"""A short run of the randomized differential tool (tools/fuzz_ops.py: 33 kinds of entry-point calls on random sizes, strides, alignments and launch-shape switches,
GPU against the oracle bit for bit) inside the GPU suite, in a process of its own (the tool flips the library's process-wide switches)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [20260929, 7])
def test_fuzz_ops_finds_no_mismatch(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ops.py"), "6000", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    m = re.search(r"fuzz: (\d+) iterations, (\d+) mismatching cases", text)
    assert out.returncode == 0 and m, text[-3000:]
    assert int(m.group(2)) == 0, text[-3000:]
    assert "MISMATCH" not in text


@pytest.mark.gpu
def test_chain_forms_on_random_geometry(gpu):
    """tools/fuzz_chain.py: lgpu_chain_amounts at random sizes, ratios, canvases, with and without a layer 2 / a resize, against the oracle's stages (600 cases)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_chain
    ran, tracks, bad = fuzz_chain.run(600, 77)
    assert ran > 400 and tracks > 800 and bad == 0
