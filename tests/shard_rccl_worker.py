"""Worker of tests/test_gpu_shardplan.py: one rank of an RCCL world driving `simrank_shardplan_*` through ctypes — no
torch in this process (the library loads RCCL itself).  Rank 0 writes the communicator id to a file the other ranks
wait for (the host program's job in a real deployment: MPI, a socket, a file)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["SHARD_RANK"]), int(os.environ["SHARD_WORLD"])
    idfile = os.environ["SHARD_ID_FILE"]
    from simrank_amd import _lib, ingest, synth
    from simrank_amd.engine import HipOps, Plan, ShardPlans
    ops = HipOps(rank % max(1, _lib.device_count()))
    if rank == 0:
        uid = ShardPlans.rccl_unique_id(ops.lib)
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise SystemExit("no communicator id")
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    comm = ShardPlans.rccl_comm(ops.lib, uid, rank, world)
    n = 1024 * world
    df = synth.powerlaw_directed(n, 10, seed=6)
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    scale = ingest.spread(csr) * csr.rowscale
    one = Plan(ops, csr, rowscale=scale, evidence=True)
    want_done, want_conv = one.run(30, 1e-4)
    want = one.result()
    one.free()
    for form, stages, wire in ((0, 1, False), (0, 3, False), (1, 2, False), (1, 4, False), (0, 2, True), (1, 3, True)):
        sp = ShardPlans(ops, csr, rowscale=scale, world=world, comm=comm, evidence=True, leg2_form=form, stages=stages,
                        wire_fp16=wire)
        done, conv = sp.run(30, 1e-4)
        got = sp.result(root=0, i_am_root=rank == 0)
        if wire:
            if rank == 0:
                big = want > 1e-6
                assert (np.abs(got - want)[big] / want[big]).max() < 5e-3
        else:
            assert (done, conv) == (want_done, want_conv), (done, conv, want_done, want_conv)
            if rank == 0:
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-30)
        blk, ids = sp.block(0)
        if rank == 0:
            assert np.array_equal(got[:, ids], blk)
        c = sp.step(0.0, exact_count=True)
        assert c > 0
        sp.free()
    # a prior that is not symmetric: leg 2's product goes round a second exchange on RCCL's stream, epilogue as its own pass
    prior = np.random.default_rng(4).random((n, n)).astype(np.float32)
    one = Plan(ops, csr, rowscale=scale, evidence=True, apriori=prior, lbd=0.3)
    want_a = one.run(30, 1e-4)
    want_s = one.result()
    one.free()
    for stages in (1, 3):
        sp = ShardPlans(ops, csr, rowscale=scale, world=world, comm=comm, evidence=True, apriori=prior, lbd=0.3, stages=stages)
        assert sp.run(30, 1e-4) == want_a
        got = sp.result(root=0, i_am_root=rank == 0)
        if rank == 0:
            assert not np.array_equal(got, got.T)
            np.testing.assert_allclose(got, want_s, rtol=1e-5, atol=1e-30)
        sp.free()
    # top-k hand-back: every rank's candidates meet on rank 0
    sp = ShardPlans(ops, csr, rowscale=scale, world=world, comm=comm, evidence=True)
    sp.run(5, 0.0)
    full = sp.result(root=0, i_am_root=rank == 0)
    idx, val = sp.topk(6, root=0, i_am_root=rank == 0)
    sp.free()
    if rank == 0:
        for a in range(0, n, 37):
            cand = np.array([c for c in range(n) if c != a])
            order = cand[np.lexsort((cand, -full[a, cand]))][:6]
            assert list(idx[a]) == list(order), a
    # fp16-held matrices on the ranks (config 5's reduced precision on shards): against the f32 result
    sp = ShardPlans(ops, csr, rowscale=scale, world=world, comm=comm, evidence=True, stages=2, storage="fp16")
    done, conv = sp.run(30, 1e-4)
    got = sp.result(root=0, i_am_root=rank == 0)
    assert conv is not None and conv >= want_conv - 1
    if rank == 0:
        assert np.abs(got - want).max() < 1e-4 * 0.8 / 0.2 + 6e-4
    sp.free()
    ops.lib.simrank_comm_destroy(comm)
    if world == 1:
        # simrank_comm_adopt: a communicator the host program made itself (here through ctypes on the same RCCL build)
        import ctypes as C
        rccl = C.CDLL(os.environ.get("SIMRANK_RCCL_LIB", "librccl.so.1"))

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid2 = UniqueId()
        rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        assert rccl.ncclGetUniqueId(C.byref(uid2)) == 0
        theirs = C.c_void_p()
        assert rccl.ncclCommInitRank(C.byref(theirs), 1, uid2, 0) == 0
        adopted = C.c_void_p()
        from simrank_amd._lib import check
        check(ops.lib.simrank_comm_adopt(theirs, 0, 1, C.byref(adopted)), "simrank_comm_adopt")
        sp = ShardPlans(ops, csr, rowscale=scale, world=1, comm=adopted, evidence=True, stages=2)
        done, conv = sp.run(30, 1e-4)
        assert (done, conv) == (want_done, want_conv)
        np.testing.assert_allclose(sp.result(), want, rtol=1e-5, atol=1e-30)
        sp.free()
        ops.lib.simrank_comm_destroy(adopted)            # (does not destroy what it adopted)
        assert rccl.ncclCommDestroy(theirs) == 0
    print("SHARDPLAN RCCL ok", flush=True)


if __name__ == "__main__":
    main()
