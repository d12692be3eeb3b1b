"""Worker of tests/test_distributed_gloo.py: one rank of a gloo world, launched by
``python -m torch.distributed.run``.  Kernels = NumPy test double, collectives = real."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402


def main(argv):
    stages, names = int(argv[0]), argv[1:]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    import tests.pydriver as drv
    drv.STAGE_ALIGN = 4              # small fixtures: let the stages really split the columns
    from tests.pydriver import TorchWorld
    from tests.conftest import Golden
    from tests.cpu_ops import NumpyOps
    from tests.helpers import check_against_golden, run_estimator
    for name in [n for n in names if n.startswith("half:")]:
        # half-form leg 2 (driver.Side.shard_sym): N divisible by 32 x world, second all-to-all real
        import simrank_amd.SimRank as SRA
        from simrank_amd import synth
        from tests.pydriver import LocalWorld
        cls = name.split(":")[1]
        # (256 nodes per rank = 8 column tiles when the exchange is staged: leg 2 and its second all-to-all
        # are then cut into stages too, Side.sh_stages)
        frame = synth.powerlaw_directed((256 if stages > 1 else 64) * dist.get_world_size(), 5, 3)
        kw = dict(weighted=True) if cls.endswith("PP") else {}
        one = NumpyOps()
        want = getattr(SRA, cls)().fit(frame, verbose=False, world=LocalWorld(1), mode="sparse",
                                       _ops_factory=lambda r: one, **kw)
        ops = NumpyOps()
        got = getattr(SRA, cls)().fit(frame, verbose=False, mode="sparse", _ops_factory=lambda r: ops,
                                      world=TorchWorld(stages=stages, handback="all", symmetric_shards=True),
                                      **kw)
        assert any(c[0] == "spmm_shard" for c in ops.calls) and any(c[0] == "shard_unpack" for c in ops.calls)
        assert list(got.index) == list(want.index)
        np.testing.assert_allclose(got.values, want.values, rtol=2e-5, atol=1e-30)
    for name in [n for n in names if n.startswith("wire:")]:
        # exchange_precision="fp16": both all-to-alls (exchange 1 staged or not, the mirrors of the half form) move
        # fp16 values (as bytes); the result is the f32 wire's to a few fp16 roundings, and it is what LocalWorld's emulation
        # of the same wire gives (same roundings of the same values)
        import simrank_amd.SimRank as SRA
        from simrank_amd import synth
        from tests.pydriver import LocalWorld
        cls = name.split(":")[1]
        P = dist.get_world_size()
        frame = synth.powerlaw_directed((256 if stages > 1 else 64) * P, 5, 3)
        kw = dict(weighted=True) if cls.endswith("PP") else {}
        for half in (True, False):
            one = NumpyOps()
            exact = getattr(SRA, cls)().fit(frame, verbose=False, world=LocalWorld(1), mode="sparse", iterations=6, eps=0,
                                            _ops_factory=lambda r: one, **kw)
            emu_ops = NumpyOps()
            emu = getattr(SRA, cls)().fit(frame, verbose=False, mode="sparse", iterations=6, eps=0,
                                          world=LocalWorld(P, symmetric_shards=half, exchange_precision="fp16"),
                                          _ops_factory=lambda r: emu_ops, **kw)
            ops = NumpyOps()
            got = getattr(SRA, cls)().fit(frame, verbose=False, mode="sparse", iterations=6, eps=0,
                                          _ops_factory=lambda r: ops,
                                          world=TorchWorld(stages=stages, handback="all", symmetric_shards=half,
                                                           exchange_precision="fp16"), **kw)
            assert list(got.index) == list(exact.index)
            big = exact.values > 1e-6
            rel = np.abs(got.values - exact.values)[big] / exact.values[big]
            assert 1e-7 < rel.max() < 4e-3, rel.max()          # really rounded, and only by a few fp16 roundings
            if stages == 1:
                np.testing.assert_allclose(got.values, emu.values, rtol=1e-6, atol=1e-12)
    for name in [n for n in names if n.startswith("auto:")]:
        # TorchWorld(symmetric_shards="auto") with the measurement forced on a small graph: both forms are
        # timed, every rank adopts the same one, the result is the one-rank result either way
        import simrank_amd.SimRank as SRA
        from simrank_amd import synth
        from tests.pydriver import LocalWorld
        drv.MEASURE_FORM_FROM_N = 0
        frame = synth.powerlaw_directed(64 * dist.get_world_size(), 5, 3)
        one = NumpyOps()
        want = SRA.SimRank().fit(frame, verbose=False, world=LocalWorld(1), mode="sparse", _ops_factory=lambda r: one)
        ops = NumpyOps()
        world = TorchWorld(stages=stages, handback="all")
        assert world.symmetric_shards == "auto"
        got = SRA.SimRank().fit(frame, verbose=False, mode="sparse", _ops_factory=lambda r: ops, world=world)
        m = world.form_measured
        assert isinstance(world.symmetric_shards, bool) and m and m["chosen"] == ("half" if world.symmetric_shards else "full")
        votes = [None] * dist.get_world_size()
        dist.all_gather_object(votes, (m["chosen"], m["half_ms"], m["full_ms"]))
        assert len(set(votes)) == 1, votes                   # same numbers, same decision on every rank
        np.testing.assert_allclose(got.values, want.values, rtol=2e-5, atol=1e-30)
        drv.MEASURE_FORM_FROM_N = 4096
    names = [n for n in names if not n.startswith("half:") and not n.startswith("auto:") and not n.startswith("wire:")]
    for name in names:
        g = Golden(name)
        ops = NumpyOps()
        # hand-back to the root only (the default): the other ranks get None
        est, res, text = run_estimator(g, lambda r: ops, world=TorchWorld(stages=stages), mode="sparse")
        if rank == 0:
            check_against_golden(g, est, res, text)
            n = len(res[0] if isinstance(res, tuple) else res)
        else:
            assert text == "" and res is None
            n = None
        # hand-back to every rank
        ops = NumpyOps()
        est, res, text = run_estimator(g, lambda r: ops, world=TorchWorld(stages=stages, handback="all"),
                                       mode="sparse")
        if rank == 0:
            check_against_golden(g, est, res, text)
        else:                        # other ranks are silent but hold the same result
            assert text == ""
            first = res[0] if isinstance(res, tuple) else res
            want = g.out["S1"] if isinstance(res, tuple) else g.out["S"]
            np.testing.assert_allclose(first.values, want, rtol=1e-5, atol=1e-30)
        # (a rank whose column block is empty at this N launches nothing but joins every collective)
        from simrank_amd.ingest import partition
        first = res[0] if isinstance(res, tuple) else res
        lo, hi = partition(len(first), dist.get_world_size(), rank)
        assert hi == lo or any(c[0] == "spmm" and c[1] for c in ops.calls)
    dist.barrier()
    print(f"RANK {rank} ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
