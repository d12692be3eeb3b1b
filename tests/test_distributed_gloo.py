"""world_size-2 (and 3) run of the sharded driver over torch.distributed/gloo on CPU.

Each process is one rank of ``driver.TorchWorld`` (launched exactly as the bench is, with
``python -m torch.distributed.run``); the kernels are the NumPy test double
(tests/cpu_ops.py), the exchange is a real ``all_to_all_single`` and the convergence count
a real ``all_reduce``.  Every rank must end with the full matrix the reference produced."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["SimRank_er64", "SimRankPP_er64_weighted", "AprioriSimRank_er64",
         "BipartiteSimRank_b5030", "BipartiteSimRankPP_b40", "SimRank_toy5",
         "AprioriSimRank_er64_asym", "BipartitleAprioriSimRank_b40_asym"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_staged_exchange_with_empty_blocks_does_not_hang():
    """N = 5 over 4 ranks in 2 stages: rank 3 owns an empty block and still has to join every
    stage's all_to_all_single (round-1 advisor finding: it skipped it and the peers hung)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "2", "SimRank_toy5", "SimRankPP_toy5"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(4):
        assert f"RANK {r} ok" in p.stdout


@pytest.mark.parametrize("world,stages", [(2, 1), (3, 1), (2, 3), (3, 2)])
def test_sharded_driver_over_gloo(world, stages):
    """stages > 1: leg 1 cut into column slices, one asynchronous all_to_all_single per slice
    (the transfer of a slice overlaps the kernels of the next ones on the GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"),
           str(stages), *NAMES]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        assert f"RANK {r} ok" in p.stdout


@pytest.mark.parametrize("world,stages", [(2, 1), (3, 2)])
def test_half_form_leg2_over_gloo(world, stages):
    """N = 64 x world: every rank runs leg 2 in its half form and the mirrored tiles travel in a
    second (real) all_to_all_single; the result equals the one-rank result."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"),
           str(stages), "half:SimRank", "half:SimRankPP"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        assert f"RANK {r} ok" in p.stdout


def test_shard_form_is_measured_and_agreed_over_gloo():
    """TorchWorld(symmetric_shards="auto"), two ranks: one update is timed in each form of leg 2, the MAX over
    the ranks decides, every rank ends with the same choice and the one-rank result."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "1", "auto:SimRank"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(2):
        assert f"RANK {r} ok" in p.stdout


@pytest.mark.parametrize("world,stages", [(2, 1), (3, 2)])
def test_fp16_wire_over_gloo(world, stages):
    """TorchWorld(exchange_precision="fp16"): the exchange buffers travel as fp16 (half the link bytes), the kernels
    stay f32 — full and half form, staged and not; the result is the f32 wire's to a few fp16 roundings and equals
    LocalWorld's emulation of the same wire."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"),
           str(stages), "wire:SimRank", "wire:SimRankPP"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        assert f"RANK {r} ok" in p.stdout
