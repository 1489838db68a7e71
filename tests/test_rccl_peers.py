"""The RCCL branches that a one-GPU box cannot reach: lgpu_params_broadcast with a real peer, lgpu_fan_in's ncclSend / ncclRecv slots (5 tracks over 2 ranks),
lgpu_chain_step across ranks.  Needs two GPUs on the box: skipped otherwise (the pool's test boxes have one; the 8-GPU node of the scaling run has them)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_entry_points_with_a_real_peer():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "mp", "rccl_peer_check.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0 and "RCCL_PEERS_OK" in text, text[-4000:]
