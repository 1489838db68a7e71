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
