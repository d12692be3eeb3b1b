"""SimRank().fit on several MI355X of one node: one process per GPU, S column-sharded, one RCCL
all-to-all per update.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/fit_multi_gpu.py [N] [average degree]

Every rank passes the same edge list.  `top_k=10` hands the ten most similar nodes of every node to every
rank without moving N x N values over PCIe; without it the dense similarity DataFrame goes to rank 0 only
(the other ranks' fit() returns None; TorchWorld(handback="all") gives it to every rank).  What runs is the
sharded loop behind the C ABI (simrank_shardplan_*, csrc/shardplan.hip) over the library's own RCCL communicator; the
library picks the pipeline depth of the exchange (stages=0) and runs leg 2 in its half form from eight ranks on
(TorchWorld(symmetric_shards=True / False) fixes the form; `bench.py --gpus N` times both on the node's links)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import simrank_amd.SimRank as SimRank            # same import path shape as the reference's
from simrank_amd import synth
from simrank_amd.driver import TorchWorld

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
deg = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
edges = synth.powerlaw_directed(n, deg, seed=n)          # columns 'from', 'to', 'weight'

t0 = time.perf_counter()
est = SimRank.SimRank()
top = est.fit(edges, C=0.8, iterations=100, eps=1e-4, verbose=dist.get_rank() == 0,
              device=local_rank, world=TorchWorld(stages=0), top_k=10)
if dist.get_rank() == 0:
    print(f"\nN={n}: converged at iteration {est.converged_at} in {time.perf_counter() - t0:.2f} s "
          f"on {dist.get_world_size()} GPU(s)")
    print(top.head())
dist.destroy_process_group()
