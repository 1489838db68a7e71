"""CPU, world_size 2 over gloo: the multi-GPU host logic (track sharding + parameter-block broadcast)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from lives_amd import dist as ld
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = ld.shard_tracks(8, rank, world)
    # every track owned exactly once
    owned = torch.zeros(8, dtype=torch.int32)
    owned[mine] = 1
    dist.all_reduce(owned)
    assert owned.tolist() == [1] * 8, owned
    blk = ld.new_param_block("cpu")
    for step in range(5):
        ld.publish_params(blk, [96 + 7 * step, step] if rank == 0 else None)
        assert blk.tolist() == [96 + 7 * step, step, 0, 0], (rank, blk)
    # the pipelined form bench.py uses: block s + 1 is sent while step s would run; acquire(s) hands out step s' values
    pipe = ld.ParamPipeline("cpu")
    pipe.prefetch(0, [96, 0] if rank == 0 else None)
    for step in range(6):
        b = pipe.acquire(step)
        if step + 1 < 6:
            pipe.prefetch(step + 1, [96 + 7 * (step + 1), step + 1] if rank == 0 else None)
        assert b.tolist() == [96 + 7 * step, step, 0, 0], (rank, step, b)
    t = ld.max_over_ranks(1.0 + rank, "cpu")
    assert t == float(world), t
    # compositing fan-in: 5 tracks over 2 ranks (rank 1 pads one slot), frame t is filled with t + 1
    mine5 = ld.shard_tracks(5, rank, world)
    got = ld.fan_in([torch.full((4, 16), t + 1, dtype=torch.uint8) for t in mine5], 5, dst=0)
    if rank == 0:
        assert [int(g[0, 0]) for g in got] == [1, 2, 3, 4, 5] and all(g.shape == (4, 16) for g in got)
    else:
        assert got is None
    dist.barrier()
    os.write(1, ("rank " + str(rank) + " ok " + str(mine) + chr(10)).encode())   # one write: the two ranks share a pipe
''') % ROOT


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 ok [0, 2, 4, 6]" in r.stdout and "rank 1 ok [1, 3, 5, 7]" in r.stdout, r.stdout


def test_bench_as_the_driver_launches_it_for_8_gpus():
    """the driver's own command line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...` -- with
    --dry-run (gloo, no GPU work): eight ranks rendezvous, shard 8 x 2 tracks `t % world`, reduce the maximum and rank 0 alone prints the line"""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1", "--tracks", "2"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["dry_run"] is True and j["tracks_of_rank0"] == [0, 8], j
    assert j["rccl_preflight"] == "ok", j            # the preflight's own plumbing (broadcast, status word, fan-in of 17 tracks over 8 ranks, one verdict for all) on gloo


def test_bench_launches_itself_for_n_gpus():
    """`python bench.py --gpus 2` run plainly (no WORLD_SIZE): bench.py re-executes itself under torch.distributed.run with one rank per GPU;
    --dry-run keeps the GPU work out so that the spawn, the rendezvous on 127.0.0.1, the sharding, the max-over-ranks reduction and rank 0's
    single JSON line run on a CPU box"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--tracks", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["tracks_of_rank0"] == [0] and abs(j["max_over_ranks_s"] - 2e-3) < 1e-9


def _preflight_worker(rank, world, port, q, break_it):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from lives_amd import dist as ld
    comm = ld.TorchComm()
    if break_it and rank == 1:
        real = comm.status_max
        comm.status_max = lambda st, stream=None: (real(st), st.zero_())          # rank 1 takes part in the exchange and then "loses" its result
    q.put((rank, ld.preflight(comm, "cpu")))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("break_it", [False, True])
def test_preflight_gives_every_rank_the_same_verdict(break_it):
    """lives_amd.dist.preflight (what bench.py runs before timing when N > 1) on 2 gloo ranks: "ok" on both, and when one rank's exchange is broken every
    rank learns which rank failed and why -- the bench then falls back to torch.distributed instead of dying"""
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_preflight_worker, args=(r, 2, port, q, break_it)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    if break_it:
        assert got[0] == got[1] and got[0].startswith("failed: rank 1: RuntimeError: status word"), got
    else:
        assert got == {0: "ok", 1: "ok"}, got
