"""Worker of tests/test_gpu_parity.py::test_rccl_world_values: one rank of an RCCL ("nccl") world
on the available GPU, launched by ``python -m torch.distributed.run``.  The TorchWorld path —
torch-owned exchange buffers, all_to_all_single issued on the engine's stream stage by stage
(no host synchronisation), all_reduce of the convergence count — must give the values of the
in-process world and of the golden vectors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    import simrank_amd.SimRank as SRA
    from simrank_amd import synth
    from tests.pydriver import LocalWorld, TorchWorld
    from tests.conftest import Golden
    from tests.helpers import check_against_golden, run_estimator
    rank = dist.get_rank()
    # golden vectors (labels, values, convergence iteration, stdout) through the RCCL world
    for name, stages in (("SimRank_er128", 1), ("SimRankPP_er128", 3), ("BipartiteSimRankPP_b40", 2),
                         ("AprioriSimRank_er64_asym", 2)):
        g = Golden(name)
        est, res, text = run_estimator(g, world=TorchWorld(stages=stages, stage_single_rank=True),
                                       mode="sparse")
        if rank == 0:
            check_against_golden(g, est, res, text)
        else:                                   # root-only hand-back (TorchWorld default)
            assert res is None and text == ""
    # the same vectors through the sharded loop behind the C ABI over the library's own RCCL communicator (what a fit on
    # several RCCL ranks runs since round 5: cshard.CShardSolver -> simrank_shardplan_* / simrank_shardbiplan_*)
    import simrank_amd.cshard as cshard
    made_c = []
    orig_c = cshard.CShardSolver.__init__

    def spy_c(self, *a, **k):
        orig_c(self, *a, **k)
        made_c.append(self)
    cshard.CShardSolver.__init__ = spy_c
    for name in ("SimRank_er128", "SimRankPP_er128", "AprioriSimRank_er64", "AprioriSimRank_er64_asym", "BipartiteSimRank_b5030",
                 "BipartiteSimRankPP_b40", "BipartitleAprioriSimRank_b40", "BipartitleAprioriSimRank_b40_asym"):
        g = Golden(name)
        n_before = len(made_c)
        est, res, text = run_estimator(g, world=TorchWorld(stages=2, loop="c"), mode="sparse")
        assert len(made_c) == n_before + 1, name
        if rank == 0:
            check_against_golden(g, est, res, text)
    cshard.CShardSolver.__init__ = orig_c
    # a graph with dense sets and several panels per stage, against the in-process world
    df = synth.powerlaw_directed(3000, 24, seed=5)
    want = SRA.SimRank().fit(df, iterations=5, eps=0, verbose=False, mode="sparse", world=LocalWorld(1))
    for stages in (1, 4):
        got = SRA.SimRank().fit(df, iterations=5, eps=0, verbose=False, mode="sparse",
                                world=TorchWorld(stages=stages, stage_single_rank=True, handback="all"))
        np.testing.assert_allclose(got.values, want.values, rtol=2e-6, atol=1e-30)
    # leg 2 in its half form (simrank_spmm_shard, the second all_to_all_single on the engine's stream,
    # simrank_shard_unpack); N must be a multiple of 32 x ranks.  "force" takes it in a one-rank world too
    import tests.pydriver as drv
    made = []
    orig = drv.Solver.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    drv.Solver.__init__ = spy
    try:
        df = synth.powerlaw_directed(2048 * dist.get_world_size(), 16, seed=9)
        want = SRA.SimRankPP().fit(df, iterations=4, eps=0, verbose=False, mode="sparse", world=LocalWorld(1))
        for stages in (1, 2):
            got = SRA.SimRankPP().fit(df, iterations=4, eps=0, verbose=False, mode="sparse",
                                      world=TorchWorld(stages=stages, stage_single_rank=True, handback="all",
                                                       symmetric_shards="force"))
            assert all(sd.shard_sym for sd in made[-1].sides[0].values())
            np.testing.assert_allclose(got.values, want.values, rtol=2e-6, atol=1e-30)
        # the fp16 wire (exchange_precision="fp16"): simrank_narrow_h16 -> all_to_all_single of the fp16 bytes ->
        # simrank_widen_h16, all on the engine's stream; same roundings as the in-process emulation of that wire
        P = dist.get_world_size()
        for stages, form in ((1, "force"), (2, "force"), (2, False)):
            emu = SRA.SimRankPP().fit(df, iterations=4, eps=0, verbose=False, mode="sparse",
                                      world=LocalWorld(P, symmetric_shards=bool(form), exchange_precision="fp16"))
            got = SRA.SimRankPP().fit(df, iterations=4, eps=0, verbose=False, mode="sparse",
                                      world=TorchWorld(stages=stages, stage_single_rank=True, handback="all",
                                                       symmetric_shards=form, exchange_precision="fp16"))
            assert made[-1].sides[0][rank].x1.wire is not None
            big = want.values > 1e-6
            rel = np.abs(got.values - want.values)[big] / want.values[big]
            assert rel.max() < 4e-3, rel.max()
            if P > 1 or stages > 1:             # (one rank, one stage: the rank's buffers alias, nothing travels)
                assert rel.max() > 1e-7, rel.max()
            if P > 1:                           # (the emulation of one rank has no wire either)
                np.testing.assert_allclose(got.values, emu.values, rtol=1e-5, atol=1e-9)
    finally:
        drv.Solver.__init__ = orig
    # the measured choice between the two forms of leg 2 (driver.resolve_shard_form): both solvers, the barriers
    # and the MAX all-reduce of the times over RCCL; whichever form wins, the values are the one-rank values
    drv.MEASURE_FORM_FROM_N = 0
    world = TorchWorld(stages=2, stage_single_rank=True, handback="all", measure_single_rank=dist.get_world_size() == 1)
    got = SRA.SimRankPP().fit(df, iterations=4, eps=0, verbose=False, mode="sparse", world=world)
    m = world.form_measured
    assert m and m["half_ms"] > 0 and m["full_ms"] > 0 and m["chosen"] in ("half", "full"), m
    np.testing.assert_allclose(got.values, want.values, rtol=2e-6, atol=1e-30)
    drv.MEASURE_FORM_FROM_N = 4096
    # fp16-held matrices on the ranks of an RCCL world through the class surface (simrank_amd/cshard.py: the sharded loop
    # behind the C ABI on the library's own communicator, made from an id broadcast through torch.distributed)
    from oracle import simrank_oracle as O
    df = synth.powerlaw_directed(1024 * dist.get_world_size(), 16, seed=9)
    want = O.fit_simrank_pp(df, verbose=False)
    est = SRA.SimRankPP()
    got = est.fit(df, verbose=False, storage_precision="fp16", world=TorchWorld(handback="all"))
    assert est.converged_at is not None and est.converged_at >= want["k"] - 1
    assert np.abs(got.values - want["S"]).max() < 1e-4 * 0.8 / 0.2 + 6e-4
    top = SRA.SimRankPP().fit(df, verbose=False, storage_precision="fp16", world=TorchWorld(), top_k=3)
    assert len(top) == 3 * len(got)                   # every rank gets the top-k frame
    dist.barrier()
    print("RCCL WORLD ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
