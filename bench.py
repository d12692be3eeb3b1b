#!/usr/bin/env python3
"""bench.py — SimRank iterations/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pl32768] [--mode auto]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one loop body of the reference's ``fit`` (SimRank.py:138-140): both legs of
S <- C.W.S.W^T with the fused diagonal/convergence epilogue, plus the read-back of the
convergence count (the reference tests convergence every iteration, SimRank.py:130).
Inputs (the CSR graph and S) are resident in HBM when the timed region starts.  Default
workload = BASELINE.json configs[3] as stated: synthetic power-law directed graph, N = 32768, average
degree 32 AFTER de-duplication (pl32768d32: 1 048 576 distinct edges), fp32 — the configuration the
metric is quoted on; it fits one GPU (3 x 4 GiB).  The lighter graph of rounds 1-2 (pl32768: the
SURVEY.md recipe keeps 783 100 of its 1 048 576 draws) is reported beside it for continuity.
With N > 1 ranks S is column-sharded and each update does one RCCL all-to-all (strong
scaling: total work fixed).

One JSON line is printed by rank 0.  ``roofline`` is for the dominant launch (the slower
of the two legs; on one rank leg 1 is ONE kernel, fused_trans_kernel: the shared columns of
every 128-row block on the matrix cores, the rest gathered, by the same workgroup), achieved =
algorithmic bytes per launch / mean launch duration measured with HIP events on the engine's
stream inside the timed region; ``roofline_mfma`` prices the matrix-core work of that launch
against the bf16 MFMA peak (a lower bound: the whole launch time is charged to it).
``cpu_baseline`` (rank 0, N = 1 only) times the oracle's dense float64 update on a bounded
row slab of the same workload and scales it to a full iteration.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TF = 2500.0     # v_mfma_f32_32x32x16_bf16 dense peak


def leg_bytes(n_rows, n_cols_in, n_cols_x, nnz, leg2, has_evidence=False, triangle=False, reads_previous=True, elt=4):
    """Algorithmic HBM bytes of one gather-leg launch:
    read X once, write Y once, CSR (col + rowptr) once per launch; leg 2 also reads the
    previous iterate for the convergence count (and 1 B/elt of evidence counts).
    ``triangle``: the single-rank form of leg 2 computes the tiles on/above the diagonal and
    stores their mirror image — it still writes all of Y and gathers from all of X, but reads
    only half of the previous iterate and of the evidence counts.
    ``reads_previous=False``: the short-circuit convergence test (epilogue count_any) reads the
    previous iterate only until a difference has been found (~1 % of it): not charged.
    ``elt``: bytes per stored value (2 for the fp16-held matrices of half.hip)."""
    b = elt * n_cols_in * n_cols_x + elt * n_rows * n_cols_x + 8 * nnz + 4 * (n_rows + 1)
    if leg2:
        half = 2 if triangle else 1
        if reads_previous:
            b += elt * n_rows * n_cols_x // half
        if has_evidence:
            b += n_rows * n_cols_x // half
    return b


def config_roofline(key, legs, note=None):
    """``roofline`` object of a secondary configuration: ``legs`` = [(kernel, "leg1" | "leg2", mean ms per launch,
    algorithmic bytes per launch)]; the slower launch first, the other under ``other``.  ``traffic`` = HBM-side bytes
    per launch of the per-config rocprofv3 --pmc passes (profiles/pmc_traffic.json[key], tools/gpu_profile_configs.sh),
    not measured in this run."""
    rec = {}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key, {})
    except Exception:
        pass
    objs = []
    for kernel, leg, ms, b in legs:
        gbs = b / (ms * 1e-3) / 1e9
        o = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": gbs / HBM_PEAK_GBS, "traffic": rec.get(leg), "ms": ms, "algorithmic_bytes": b}
        if rec.get(leg):
            o["traffic_over_algorithmic"] = rec[leg] / b
            o["traffic_source"] = (f"profiles/pmc_traffic.json[{key!r}]: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc "
                                   f"passes at commit {rec.get('commit', '?')}, NOT measured in this run")
            if rec.get(leg + "_ms_rocprof"):
                o["ms_rocprof"] = rec[leg + "_ms_rocprof"]
        objs.append(o)
    objs.sort(key=lambda o: -o["ms"])
    top = objs[0]
    top["other"] = objs[1:]
    if note:
        top["note"] = note
    return top



# ---- the printed line (round 6): numbers first, <= 8 KB.  The driver keeps the TAIL of what bench.py prints; a 30 KB line
# of prose left it with the middle of `config5`.  So: the contract keys, then `summary` (every BASELINE config's it/s, ms,
# roofline fraction and traffic ratio), `fit_wall_s`, `converge`, then the per-variant detail with every prose string
# removed (what each key means is in profiles/bench_notes.md) and floats cut to five significant digits; sections are
# dropped from the end of a priority list while the line is still longer than LINE_LIMIT (named under "dropped").
# The complete record — prose included — goes to --full-json (default gpurun_out/bench_full.json when that can be written).
LINE_LIMIT = 8000
KEEP_STRINGS = {"metric", "unit", "scaling", "dtype", "data", "bound", "kind", "leg2_form", "form", "error", "headline_loop", "name",
                "python_world_error", "skipped", "workload", "sample", "kernel"}
DROP_ORDER = ["mfma_dense_leg", "roofline_other", "continuity_pl32768", "shards_emulated_p8", "convergence_test",
              "sharded_c_loop", "config5", "bipartite_pp", "secondary", "roofline_mfma", "fit_wall"]


def _num(x):
    return float(f"{x:.5g}") if isinstance(x, float) else x


def _strip(o, key=None):
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if isinstance(v, str) and (k not in KEEP_STRINGS or len(v) > 160):
                continue
            out[k] = _strip(v, k)
        return out
    if isinstance(o, (list, tuple)):
        return [_strip(x, key) for x in o]
    return _num(o)


def _roof(r):
    """(leg-1 frac, leg-2 frac, leg-1 and leg-2 traffic / algorithmic) of a `roofline` object whose slower launch comes first."""
    if not isinstance(r, dict):
        return None, None, None, None
    legs = [r] + list(r.get("other") or [])
    f1 = f2 = t1 = t2 = None
    for o in legs:
        name = str(o.get("kernel", ""))
        if "leg 2" in name or "leg2" in name:
            f2, t2 = o.get("frac"), o.get("traffic_over_algorithmic")
        else:
            f1, t1 = o.get("frac"), o.get("traffic_over_algorithmic")
    return f1, f2, t1, t2


def summary_of(out):
    """Every BASELINE config on one screen: it/s, ms per update, HBM roofline fraction of both legs, leg-1 traffic ratio."""
    def row(value, ms, roof, **extra):
        f1, f2, t1, t2 = _roof(roof)
        r = {"it_s": value, "ms": ms, "leg1_frac": f1, "leg2_frac": f2, "leg1_traffic_x": t1, "leg2_traffic_x": t2}
        r.update(extra)
        return {k: v for k, v in r.items() if v is not None}
    s = {}
    sec, bp, c5 = out.get("secondary") or {}, out.get("bipartite_pp") or {}, out.get("config5") or {}
    if sec:
        fl = sec.get("fit_loop") or {}
        s["cfg2_er8192"] = row(fl.get("value", sec.get("value")), fl.get("ms_per_step", sec.get("ms_per_step")), sec.get("roofline"),
                               it_s_step_by_step=sec.get("value"))
    if bp:
        fl = bp.get("fit_loop") or {}
        s["cfg3_ml1m_bipartite_pp"] = row(fl.get("value", bp.get("value")), fl.get("ms_per_step", bp.get("ms_per_step")),
                                          bp.get("roofline"), it_s_step_by_step=bp.get("value"),
                                          leg1_mfma_frac=(bp.get("roofline_mfma") or {}).get("frac"))
    hr = dict(out.get("roofline") or {})
    if hr:
        hr = dict(hr, kernel="leg 1", other=[dict(out.get("roofline_other") or {}, kernel="leg 2")] if out.get("roofline_other") else [])
        for o in [hr] + hr["other"]:
            if o.get("traffic") and o.get("algorithmic_bytes"):
                o["traffic_over_algorithmic"] = o["traffic"] / o["algorithmic_bytes"]
    s["cfg4_" + str((out.get("config") or {}).get("name", "headline"))] = row(out.get("value"), out.get("ms_per_step"), hr or None)
    for key, name in (("f32_exact_dense_blocks", "cfg5_pl65536_pp_f32"), ("fp16_storage", "cfg5_pl65536_pp_fp16_held")):
        v = c5.get(key) or {}
        if v:
            s[name] = row(v.get("value"), v.get("ms_per_step"), v.get("roofline"))
    return s


def compact_line(out, limit=LINE_LIMIT):
    head_keys = ["metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "gpu_over_cpu"]
    line = {k: _strip(out[k], k) for k in head_keys if k in out}
    for k, v in out.items():
        if k not in line and k != "converge":
            line[k] = _strip(v, k)
    tail = {"summary": _strip(summary_of(out))}
    fw = out.get("fit_wall") or {}
    if fw:
        tail["fit_wall_s"] = {k: _num(v.get("fit_wall_s")) for k, v in fw.items() if isinstance(v, dict) and "fit_wall_s" in v}
    if "converge" in out:
        tail["converge"] = _strip(out["converge"])
    dropped = []

    def text_of():
        d = dict(line)
        if dropped:
            d["dropped"] = dropped
        d.update(tail)                       # (last: the driver's record keeps the TAIL of what is printed)
        return json.dumps(d, default=str, separators=(",", ":"))
    text = text_of()
    for k in DROP_ORDER:
        if len(text) <= limit:
            break
        if k in line:
            del line[k]
            dropped.append(k)
            text = text_of()
    return text


def write_full(out, path):
    if not path:
        return
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, default=str)
    except OSError:
        pass

def cpu_share():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup's quota (the GPU boxes show 256 hardware
    threads and grant 16 CPUs of time: cpu.max = "1600000 100000")."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(csr, S_host, coef, budget_s=45.0):
    """Oracle (dense float64 NumPy, the reference's arithmetic) on a row slab of one
    iteration; value = iterations/s of a full iteration extrapolated from the slab."""
    from oracle import simrank_oracle as O
    share = cpu_share()
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        if threads > share:
            # BLAS sized its pool by the hardware threads it sees; more runnable threads than granted CPUs only get
            # throttled by the cgroup — the baseline runs on the CPUs it really has
            threadpool_limits(limits=share)
            threads = share
    except Exception:
        threads = min(os.cpu_count() or 1, share)
    n = csr.n_rows
    G = csr.dense()                                        # what the reference keeps as `Graph`
    probe = slice(0, 128)
    O.update_rows(G, S_host, coef, slice(0, 8))            # warm the BLAS threads
    t0 = time.perf_counter()
    O.update_rows(G, S_host, coef, probe)
    t_probe = time.perf_counter() - t0
    rows = int(min(n, max(128, 128 * budget_s / max(t_probe, 1e-3))))
    rows = min(rows, 8192)
    slab = slice(0, rows)
    t0 = time.perf_counter()
    new = O.update_rows(G, S_host, coef, slab)
    t_update = time.perf_counter() - t0
    t0 = time.perf_counter()
    old = S_host[slab].copy()                              # copy.deepcopy, SimRank.py:138
    O.converged(old, new, 1e-4)                            # SimRank.py:130
    t_rest = time.perf_counter() - t0
    per_iter = (t_update + t_rest) * n / rows
    return {"value": 1.0 / per_iter, "unit": "iterations/s", "cores": int(threads),
            "kind": "port",
            "sample": f"oracle dense f64 update, rows 0..{rows - 1} of N={n}: {t_update + t_rest:.2f} s x{n / rows:.1f} = "
                      f"{per_iter:.1f} s/iteration; {share} CPUs granted of {os.cpu_count()}, BLAS threads {threads}"}


def link_probe(world, dist, torch, rank, mib=64, reps=5):
    """What the links of THIS node move: every rank sends `mib` MiB of f32 to every peer and receives as much from each, all
    peers at once (ncclSend / ncclRecv pairs in one group: the traffic shape of the update's all-to-all), `reps` times behind
    a warm-up; -> GB/s per rank and direction (sum over its peers), and per peer link.  DESIGN.md §5's projection of the
    8-GPU update (448 MiB per rank over 7 links in ~0.9 ms = 75 GB/s per link) rests on this number; no multi-GPU node has
    been available to take it (one rank: nothing to time, the keys say so)."""
    P = world.size
    rec = {"ranks": P, "MiB_per_peer": mib, "peers": P - 1}
    if P < 2:
        rec["skipped"] = "one rank: no peer to exchange with"
        return rec
    try:
        n = mib * (1 << 20) // 4
        send = [torch.full((n,), float(rank), dtype=torch.float32, device="cuda") for _ in range(P)]
        recv = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(P)]

        def once():
            ops_ = []
            for h in range(P):
                if h != rank:
                    ops_.append(dist.P2POp(dist.isend, send[h], h, group=world.group))
                    ops_.append(dist.P2POp(dist.irecv, recv[h], h, group=world.group))
            for w in dist.batch_isend_irecv(ops_):
                w.wait()
        once()
        torch.cuda.synchronize()
        dist.barrier(group=world.group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            once()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=world.group)
        ms = float(t.item())
        ok = all(float(recv[h][0].item()) == float(h) for h in range(P) if h != rank)
        sent = (P - 1) * mib * (1 << 20)
        rec.update({"ms": ms, "GBps_per_rank_each_way": sent / (ms * 1e-3) / 1e9,
                    "GBps_per_peer_link_each_way": sent / (P - 1) / (ms * 1e-3) / 1e9, "payload_ok": ok})
    except Exception as e:                           # (a probe must never sink the line)
        rec["error"] = f"{type(e).__name__}: {e}"
    return rec


def c_loop_section(args, ops, world, dist, torch, synth, ingest, csr, coef, gpu, barrier, rank, out=None):
    """The sharded loop behind the C ABI on this world's ranks: every variant `steps` updates with the exact count, the
    MAX over the ranks of the wall time, per-piece HIP-event times of rank 0.  Over gloo (the CPU rehearsal of the launch
    path) nothing can run — the loop moves data with RCCL or inside one process — and every key says so."""
    P = world.size
    n = csr.n_rows
    variants = [("f32_full_form", dict(leg2_form=0), True),
                ("f32_half_form", dict(leg2_form=1), True),
                ("f32_half_form_fp16_wire", dict(leg2_form=1, wire_fp16=True), False),
                ("fp16_held_full_form", dict(storage="fp16", leg2_form=0), False)]
    out = {} if out is None else out             # (filled in place: the caller's watchdog prints what is there if a rank hangs)
    out.update({"ranks": P, "variants": {}, "stages": args.stages})
    if not gpu:
        why = "not run: the C loop exchanges over RCCL (or inside one process); this is the gloo rehearsal of the launch path"
        for name, _, grade in variants:
            out["variants"][name] = {"skipped": why, "parity_grade": grade}
        out["config5"] = {"skipped": why}
        out["config3"] = {"skipped": why}
        out["form_measured"] = {"skipped": why}
        out["link_probe"] = {"skipped": "gloo rehearsal: no xGMI links to time"}
        return out
    from simrank_amd import cshard
    from simrank_amd.engine import ShardPlans
    comm = cshard._rccl_comm(world, ops)

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(ok):
        t = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) == 0.0

    def time_variant(c, rowscale, evidence, kw, steps, warmup):
        applies = None
        if kw.get("leg2_form") == 1 and c.n_rows % (32 * P):
            applies = f"needs n % (32 x {P}) == 0"
        if kw.get("storage") == "fp16" and c.n_rows % (64 * P):
            applies = f"needs n % (64 x {P}) == 0"
        if applies:
            return {"skipped": applies}
        sp, err = None, None
        try:
            sp = ShardPlans(ops, c, rowscale=rowscale, world=P, comm=comm, coef=coef, evidence=evidence,
                            stages=args.stages, **kw)
        except Exception as e:                       # (every rank must learn of a rank's failure before a collective)
            err = f"{type(e).__name__}: {e}"
        if not all_ok(err is None):
            if sp is not None:
                sp.free()
            return {"error": err or "another rank failed to create its plan"}
        for _ in range(warmup):
            sp.step(0.0, exact_count=True)
        sp.set_timing(steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            sp.step(0.0, exact_count=True)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0) / steps
        rec = {"value": 1.0 / dt, "unit": "iterations/s", "ms_per_step": dt * 1e3, "rank0_events_ms": sp.timings(),
               "info": sp.info(0)}
        sp.set_timing(0)
        sp.free()
        return rec

    spec = (csr.rowscale if not args.pp else ingest.spread(csr) * csr.rowscale)
    for name, kw, grade in variants:
        rec = time_variant(csr, spec, args.pp, kw, args.steps, max(1, args.warmup))
        rec["parity_grade"] = grade
        out["variants"][name] = rec
    full, half = out["variants"]["f32_full_form"], out["variants"]["f32_half_form"]
    if "value" in full and "value" in half:
        out["form_measured"] = {"full_ms": full["ms_per_step"], "half_ms": half["ms_per_step"],
                                "chosen": "half" if half["ms_per_step"] < full["ms_per_step"] else "full"}
    if not args.no_extras and args.workload == "pl32768d32" and not args.pp:
        try:
            df5 = synth.WORKLOADS["pl65536"][0]()
            _, csr5 = ingest.directed(df5, False, "from", "to", "weight")
            scale5 = ingest.spread(csr5) * csr5.rowscale
            c5 = {"workload": f"pl65536: N={csr5.n_rows} nnz={csr5.nnz} SimRank++ (evidence + spread) on {P} ranks, C loop"}
            for name, kw in (("f32_full_form", dict(leg2_form=0)), ("f32_half_form", dict(leg2_form=1)),
                             ("fp16_held_full_form", dict(storage="fp16", leg2_form=0))):
                c5[name] = time_variant(csr5, scale5, True, kw, 6, 2)
            out["config5"] = c5
        except Exception as e:
            out["config5"] = {"error": f"{type(e).__name__}: {e}"}
        # BASELINE config 3 (MovieLens-1M-shaped BipartiteSimRankPP) through the two-matrix sharded loop (simrank_shardbiplan_*)
        try:
            from simrank_amd.engine import ShardBiPlans
            df3 = synth.WORKLOADS["ml1m"][0]()
            _, _, _, _, g12, g21 = ingest.bipartite(df3, False, "user", "item", "weight")
            bp, err = None, None
            try:
                bp = ShardBiPlans(ops, g12, g12.rowscale, g21.rowscale, world=P, comm=comm, c1=coef, c2=coef, evidence=True,
                                  leg2_form=0, stages=args.stages)
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
            if not all_ok(err is None):
                if bp is not None:
                    bp.free()
                out["config3"] = {"error": err or "another rank failed to create its plans"}
            else:
                for _ in range(2):
                    bp.step(0.0, True)
                barrier()
                t0 = time.perf_counter()
                for _ in range(10):
                    bp.step(0.0, True)
                barrier()
                dt = max_over_ranks(time.perf_counter() - t0) / 10
                out["config3"] = {"workload": f"ml1m: {g12.n_rows} x {g12.n_cols}, nnz={g12.nnz}, BipartiteSimRankPP on {P} ranks, C loop "
                                              "(two exchanges per loop body, full form)",
                                  "value": 1.0 / dt, "unit": "iterations/s", "ms_per_step": dt * 1e3}
                bp.free()
        except Exception as e:
            out["config3"] = {"error": f"{type(e).__name__}: {e}"}
    # LAST (a probe that hangs must not cost the measurements above: the watchdog prints what is there): what the links move
    barrier()
    out["link_probe"] = link_probe(world, dist, torch, rank)
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """``python bench.py --gpus N`` outside torchrun: start the N ranks as a CHILD process
    (python -m torch.distributed.run ... bench.py <same arguments>), relay rank 0's JSON line
    and return the child's exit code.  Runs before torch or HIP is touched in this process:
    nothing that has initialised the GPU is ever exec'ed or re-exec'ed."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(p.stdout)
    return p.returncode


def workload_frame(synth, name):
    """A named BASELINE.json workload, or an ad-hoc ``er:N:p`` / ``pl:N:deg`` graph."""
    if name.startswith("er:"):
        _, n, p = name.split(":")
        return synth.er_directed(int(n), float(p), int(n)), "directed"
    if name.startswith("pl:"):
        _, n, d = name.split(":")
        return synth.powerlaw_directed(int(n), float(d), int(n)), "directed"
    factory, kind = synth.WORKLOADS[name]
    return factory(), kind


STATE = {}        # what main() has built so far: the fallback of a multi-rank run picks it up (c_loop_only)


def c_loop_only(err):
    """A run on several RCCL ranks: time the sharded loop behind the C ABI — the loop fit() runs there — and report THAT as the
    line, through the library's own communicator (``err``: why the tests' Python world, asked for with --python-world, did not
    finish; None by default).  If the ranks do not all get through, the watchdog prints what there is and exits non-zero."""
    import threading
    st = STATE
    args, rank = st["args"], st["rank"]
    out = {"metric": "simrank_iterations_per_sec", "value": None, "unit": "iterations/s", "n_gpus": st["world_size"],
           "rccl_ranks": st["world_size"], "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.workload}: synthetic directed graph N={st['csr'].n_rows} nnz={st['csr'].nnz} SimRank C=0.8 "
                                  "fp32, eps test every iteration (every element compared, exact count)",
                      "name": args.workload, "N": st["csr"].n_rows, "nnz": st["csr"].nnz, "mode": "sparse",
                      "sharding": f"S column-sharded over {st['world_size']} rank(s), all-to-all per update"},
           "python_world_error": err, "sharded_c_loop": {}}
    deadline = float(os.environ.get("SIMRANK_BENCH_CLOOP_DEADLINE", "300"))

    def give_up():
        out["sharded_c_loop"]["error"] = f"watchdog: not finished after {deadline:.0f} s"
        if rank == 0:
            st["emit"](compact_line(out))
        os._exit(3)
    dog = threading.Timer(deadline, give_up)
    dog.daemon = True
    dog.start()
    def barrier():
        st["ops"].synchronize()
        st["torch"].cuda.synchronize()
        st["dist"].barrier()
    try:
        c_loop_section(args, st["ops"], st["world"], st["dist"], st["torch"], st["synth"], st["ingest"], st["csr"], 0.8, True,
                       st.get("barrier", barrier), rank, out=out["sharded_c_loop"])
    except Exception as e:
        out["sharded_c_loop"]["error"] = f"{type(e).__name__}: {e}"
    dog.cancel()
    best = None
    for name, rec in out["sharded_c_loop"].get("variants", {}).items():
        if rec.get("parity_grade") and "value" in rec and (best is None or rec["value"] > best[1]["value"]):
            best = (name, rec)
    if best is not None:
        out["value"], out["ms_per_step"] = best[1]["value"], best[1]["ms_per_step"]
        out["headline_loop"] = f"simrank_shardplan_step behind the C ABI, {best[0]}" + (
            " (the tests' Python world failed: python_world_error)" if err else "")
        # dominant launch of a rank: leg 1 on its column block (X block read once, the transposed product written once, the CSR)
        ev = best[1].get("rank0_events_ms") or {}
        if ev.get("leg1_ms"):
            n_, P_, z_ = st["csr"].n_rows, st["world_size"], st["csr"].nnz
            b1 = leg_bytes(n_, n_, -(-n_ // P_), z_, leg2=False)
            gbs = b1 / (ev["leg1_ms"] * 1e-3) / 1e9
            out["roofline"] = {"kernel": "leg 1 of rank 0 (fused_trans_kernel on its column block, chunked transposed store)",
                               "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                               "traffic": None, "ms": ev["leg1_ms"], "algorithmic_bytes": b1,
                               "other_ms": {k: v for k, v in ev.items() if k != "leg1_ms"}}
    if not err:
        out.pop("python_world_error", None)
    if rank == 0:
        write_full(out, getattr(args, "full_json", None))
        st["emit"](compact_line(out))
    try:
        st["dist"].destroy_process_group()
    except Exception:
        pass
    return 0 if best is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pl32768d32")
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--panel", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where the complete record (prose notes included) is written; the printed line is its compact form")
    ap.add_argument("--exact-only", action="store_true",
                    help="skip the extra steps with the short-circuit convergence test (profiling runs: every leg-2 "
                         "dispatch is then of the timed, exact-count form)")
    ap.add_argument("--stages", type=int, default=0,
                    help="pipeline depth of the sharded exchange (all_to_all_single calls per update; "
                         "0 = by the width of a rank's column block)")
    ap.add_argument("--shard-form", choices=["auto", "half", "full"], default="auto",
                    help="leg 2 of a sharded run: half = tiles i <= j per rank + a second all-to-all of the "
                         "mirrored tiles, full = every tile; auto = half from 8 ranks on (DESIGN.md §5)")
    ap.add_argument("--pp", action="store_true",
                    help="SimRank++ (evidence-gated update, spread weights: SimRank.py:351-362) instead of SimRank")
    ap.add_argument("--dense-precision", default="f32", choices=["f32", "fp16"],
                    help="operand precision of the matrix-core part: f32 = exact (three bf16 terms), fp16 = one "
                         "fp16 term (BASELINE.json config 5's reduced-precision dense leg; outside the parity bar)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL on the GPUs (the measurement); gloo = CPU rehearsal of the launch "
                         "path with the NumPy test double (tests only)")
    ap.add_argument("--exchange-precision", default="f32", choices=["f32", "fp16"],
                    help="wire format of the all-to-alls of a sharded run: f32 = exact (default, the parity path); "
                         "fp16 = value x 2^14 in fp16 on the links, f32 kernels (half the link bytes, one fp16 "
                         "rounding per update: outside the 1e-5 bar, reported as such)")
    ap.add_argument("--python-world", action="store_true",
                    help="several ranks: also time the tests' Python choreography (tests/pydriver.py over torch.distributed) "
                         "before the C loop; default: the C loop alone")
    ap.add_argument("--c-loop-only", action="store_true",
                    help="with --force-dist: take the several-rank path (the C loop alone) in a one-rank RCCL world")
    ap.add_argument("--force-dist", action="store_true",
                    help="use the torch.distributed world even with one rank (exercises RCCL)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))           # before anything touches torch or HIP

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world_size

    # Rank 0 owes its caller ONE JSON line on stdout.  Libraries print there too (RCCL: a five-line version banner whenever a
    # communicator is made — torch's eager init below, the library's own in cshard._rccl_comm): from here on file descriptor 1
    # is stderr's, and the line goes out through a duplicate of the real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (line + "\n").encode())
    STATE.update(args=args, rank=rank, world_size=world_size, emit=emit)

    import torch
    import torch.distributed as dist
    from simrank_amd import ingest, synth
    from tests.pydriver import LocalWorld, SideSpec, Solver, TorchWorld

    gpu = args.backend == "nccl"
    use_dist = world_size > 1 or args.force_dist
    if gpu:
        torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world_size,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        form = {"auto": "auto", "half": True, "full": False}[args.shard_form]
        world = TorchWorld(stages=args.stages, stage_single_rank=args.force_dist, symmetric_shards=form,
                           exchange_precision=args.exchange_precision)
    else:
        world = LocalWorld(1)
    one_rank = LocalWorld(1)          # (the single-GPU sections below run on this process alone whatever the headline's world)

    if gpu:
        from simrank_amd.engine import HipOps
        ops = HipOps(local_rank)
        if args.panel is not None:
            ops.set_tuning(panel=args.panel)
    else:
        # rehearsal of the multi-rank launch path on CPU (tests/test_bench_launch.py): real
        # collectives over gloo, the kernels replaced by the NumPy test double; no number
        # from this path is a measurement
        from tests.cpu_ops import NumpyOps
        ops = NumpyOps()
    df, kind = workload_frame(synth, args.workload)
    assert kind == "directed", "bench workloads are the directed SimRank configurations"
    _, csr = ingest.directed(df, False, "from", "to", "weight")
    n, nnz = csr.n_rows, csr.nnz
    coef = 0.8
    rehearse_failure = os.environ.get("SIMRANK_BENCH_FAIL_PYTHON_WORLD") == "1"      # (test hook of the fallback below)
    STATE.update(ops=ops, world=world, dist=dist, torch=torch, synth=synth, ingest=ingest, csr=csr,
                 c_fallback=bool(gpu and use_dist and (world_size > 1 or rehearse_failure) and args.mode in ("auto", "sparse")))
    if rehearse_failure and use_dist:
        raise RuntimeError("SIMRANK_BENCH_FAIL_PYTHON_WORLD=1: rehearsal of a failing Python world")
    if gpu and use_dist and (world_size > 1 or args.c_loop_only) and not args.python_world and args.mode in ("auto", "sparse"):
        # Several RCCL ranks: the sharded loop behind the C ABI is the ONLY loop fit() runs there (round 6), so it is the
        # only thing timed by default — under a watchdog that turns a rank stuck in a collective into a partial line and a
        # non-zero exit.  The tests' Python choreography over torch's own collectives (never run on more than one GPU either)
        # is a comparison on request: --python-world.
        sys.exit(c_loop_only(None))

    def make_spec(c, pp, terms=3, storage="f32"):
        if not pp:
            return SideSpec(c, c.rowscale, coef, dense_terms=terms, storage=storage)
        return SideSpec(c, ingest.spread(c) * c.rowscale, coef, evidence_from=c,      # SimRank.py:322-337, :311-320
                        dense_terms=terms, storage=storage)

    solver = Solver(lambda r: ops, world,
                    [make_spec(csr, args.pp, 1 if args.dense_precision == "fp16" else 3)], args.mode)
    solver.reset()

    def barrier():
        ops.synchronize()
        if gpu:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
    STATE["barrier"] = barrier

    # The headline is timed with the EXACT count of moved elements in every step: every element of S' is
    # compared with the previous iterate, as `_converged` (SimRank.py:74-77) literally does.  What fit()
    # runs by default — the short-circuit form of the same test — is timed afterwards and reported beside it.
    # One GPU, gather legs: the loop behind the C ABI (simrank_plan_step: what fit() runs since round 5, cplan.PlanSolver),
    # its legs stamped with HIP events on the plan's own stream.  Several ranks / GEMM modes: driver.Solver.
    hplan = None
    terms_now = 1 if args.dense_precision == "fp16" else 3
    if gpu and not use_dist and solver.mode == "sparse" and getattr(solver, "blocked", False):
        from simrank_amd.engine import Plan
        spec0 = make_spec(csr, args.pp, terms_now)
        hplan = Plan(ops, csr, spec0.rowscale, coef=coef, evidence=args.pp, dense_terms=terms_now)
    solver.exact_count = True
    short_ms = None
    if hplan is not None:
        for _ in range(args.warmup):
            hplan.step(0.0, exact_count=True)
        hplan.set_timing(args.steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hplan.step(0.0, exact_count=True)
        barrier()
        elapsed = time.perf_counter() - t0
        l1p, l2p, n_stamped = hplan.leg_times()
        hplan.set_timing(0)
        legs = {"leg1.0": (l1p, n_stamped), "leg2.0": (l2p, n_stamped)}
        if not args.exact_only:
            hplan.step(0.0, exact_count=False)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                hplan.step(0.0, exact_count=False)
            barrier()
            short_ms = (time.perf_counter() - t0) / args.steps * 1e3
    else:
        for _ in range(args.warmup):
            solver.step(0.0)
        solver.enable_timing(args.steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            solver.step(0.0)
        barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        legs = solver.leg_times()                 # mean ms per launch, measured by HIP events
        solver.events = None
    side = solver.sides[0][rank if use_dist else 0]
    side_stages = side.n_stages
    out = {
        "metric": "simrank_iterations_per_sec", "value": args.steps / elapsed,
        "unit": "iterations/s", "n_gpus": world_size,
        "rccl_ranks": (dist.get_world_size() if use_dist and gpu else 0), "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: synthetic directed graph N={n} nnz={nnz} "
                               f"{'SimRank++ (evidence + spread)' if args.pp else 'SimRank'} C=0.8 fp32"
                               f"{' with fp16 dense blocks' if args.dense_precision == 'fp16' else ''}, "
                               f"eps test every iteration (every element compared, exact count)",
                   "name": args.workload, "N": n, "nnz": nnz, "mode": solver.mode,
                   "loop": ("simrank_plan_step behind the C ABI (csrc/plan.hip): what fit() runs on one GPU" if hplan is not None
                            else "tests/pydriver.Solver.step (the tests' Python choreography over the C ABI's kernels)"),
                   "layout": "panel-blocked (32-column panels)" if getattr(solver, "blocked", False) else "row-major",
                   "sharding": f"S column-sharded over {world_size} rank(s), all-to-all per update"
                               + (f" in {side_stages} overlapped stage(s)" if use_dist else "")
                               + ("; leg 2 in its half form (tiles i <= j per rank, mirrored tiles in a second "
                                  "half-size all-to-all)" if getattr(side, "shard_sym", False) else "")
                               + ("; fp16 WIRE format (reduced precision on the links, outside the parity bar)"
                                  if use_dist and args.exchange_precision == "fp16" else "")},
    }
    if use_dist and getattr(world, "form_measured", None):
        out["shard_form_measured"] = world.form_measured      # both forms of leg 2 timed on this node's links
    if use_dist:
        out["headline_loop"] = "tests/pydriver.Solver over torch.distributed (the tests' Python choreography; replaced below by the C loop)"
        out["python_driver"] = {"value": out["value"], "unit": "iterations/s", "ms_per_step": out["ms_per_step"]}
        # The one shot at a multi-GPU node: the SAME ranks also time the sharded loop behind the C ABI
        # (simrank_shardplan_*: ncclSend / ncclRecv groups on a stream of their own) in its forms — f32 full / half, the fp16
        # wire, fp16-held matrices — with HIP events on the kernels' and the exchanges' streams, and BASELINE config 5
        # (N = 65536 SimRank++) through the same loop.  The headline is the faster PARITY-GRADE (f32) loop, named.
        # No P > 1 RCCL world has ever run this loop (one GPU per box until now), and a rank stuck in a collective would take the
        # whole line with it: a watchdog prints what has been measured (the Python world's headline + the variants finished) and
        # ends the rank if the section is not done after SIMRANK_BENCH_CLOOP_DEADLINE seconds (default 300; the section takes well under a minute).
        import threading
        section = {}
        out["sharded_c_loop"] = section
        deadline = float(os.environ.get("SIMRANK_BENCH_CLOOP_DEADLINE", "300"))

        def give_up():
            section["error"] = (f"watchdog: the C-loop section had not finished after {deadline:.0f} s (a rank stuck in a collective?); "
                                "the variants above it are what had been measured; value = the Python world's loop")
            if rank == 0:
                try:
                    line = compact_line(out)
                except Exception:
                    line = compact_line({k: v for k, v in out.items() if k != "sharded_c_loop"})
                emit(line)
            # (a hung run must not read as a pass: torchrun, launch_ranks() and tools/gpu_final.sh look at the exit code)
            os._exit(3)
        dog = threading.Timer(deadline, give_up)
        dog.daemon = True
        dog.start()
        try:
            c_loop_section(args, ops, world, dist, torch, synth, ingest, csr, coef, gpu, barrier, rank, out=section)
        except Exception as e:
            section["error"] = f"{type(e).__name__}: {e}"
        finally:
            dog.cancel()
        best = None
        for name, rec in out["sharded_c_loop"].get("variants", {}).items():
            if rec.get("parity_grade") and "value" in rec and (best is None or rec["value"] > best[1]["value"]):
                best = (name, rec)
        if best is not None:
            # (round 6: the sharded loop behind the C ABI is the ONLY loop fit() runs on several ranks — it is the headline
            # whatever the Python choreography of tests/pydriver.py, timed above for comparison, reached)
            out["value"], out["ms_per_step"] = best[1]["value"], best[1]["ms_per_step"]
            out["headline_loop"] = f"simrank_shardplan_step behind the C ABI, {best[0]}"
    if short_ms is not None:
        out["convergence_test"] = {
            "timed_form": "exact: every element of S' is compared with the previous iterate and the moved ones "
                          "are counted (SimRank.py:74-77)",
            "short_circuit_form": "what fit() runs: `_converged` uses its sum as a truth value, so comparing stops "
                                  "once an element that moved by more than eps has been found (epilogue count_any); "
                                  "same results, same convergence iteration",
            "ms_per_step_short_circuit": short_ms, "iterations_per_sec_short_circuit": 1e3 / short_ms}
    if not gpu:
        out["data"] = "synthetic; gloo rehearsal with the NumPy test double: NOT a measurement"
        out["ranks"] = dist.get_world_size() if use_dist else 1
        if use_dist:
            out["exchange_ms"] = {k: v[0] for k, v in legs.items() if k.startswith("exchange")}
        args.no_extras = args.no_cpu_baseline = True
    elif solver.mode == "sparse":
        l1 = legs["leg1.0"][0]
        l2 = legs["leg2.0"][0]
        b1 = leg_bytes(side.M, side.K, side.Lk, nnz, leg2=False)
        tri = world_size == 1 and not use_dist          # upper-triangle + mirror form of leg 2
        half = bool(getattr(side, "shard_sym", False))   # sharded: tiles i <= j + exchanged mirror images
        short = not solver.exact_count                 # the timed steps used the short-circuit test
        b2 = leg_bytes(side.M, side.K, side.Lm, nnz, leg2=True, has_evidence=args.pp, triangle=tri or half,
                       reads_previous=not short)
        b2_full = leg_bytes(side.M, side.K, side.Lm, nnz, leg2=True, has_evidence=args.pp, reads_previous=not short)
        # the matrix-core part of leg 1
        nt, dk, cov = ops.dense_stats(side.graph)
        dense_ms = None
        fsteps, fcov, frem = ops.fused_stats(side.graph) if hasattr(ops, "fused_stats") else (0, 0, nnz)
        fused = bool(getattr(solver, "blocked", False)) and fsteps > 0 and side.K <= ops.get_tuning("fuse_max_rows")
        if fused:
            # one launch: 16-column steps x 128 rows x 32 columns x 3 bf16 terms per panel
            flop = 2.0 * 3 * 128 * 16 * fsteps * side.Lk
            tf = flop / (l1 * 1e-3) / 1e12
            out["roofline_mfma"] = {
                "kernel": "matrix-core phase of fused_trans_kernel (leg 1: 0/1 pattern bits x operand rows, bf16 hi+mid+lo)",
                "bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / MFMA_BF16_PEAK_TF, "ms": l1, "flop": flop, "steps_per_panel": fsteps,
                "entries_covered": fcov, "entries_covered_frac": fcov / max(1, nnz),
                "entries_gathered": frem,
                "note": "lower bound: the launch also gathers the remainder and stores the tiles, and its whole "
                        "duration is charged here; exact f32 products (three bf16 MFMAs per operand term)"}
            if gpu and world_size == 1 and not args.no_extras:
                # the matrix-core phase by itself: the same launch on a graph whose gather phase is switched
                # off (diagnostic knob probe_flags = 1; wrong results, timing only) minus the launch with both
                # phases off (5): what the phase costs when nothing else runs beside it
                try:
                    os.environ["SIMRANK_ENABLE_PROBES"] = "1"
                    S_in = solver.cur[0][0]
                    scratch = ops.matrix(side.K, side.M, blocked=True)
                    t_probe = {}
                    for flags in (1, 4, 5):
                        gp = ops.graph(side.spec.csr, side.spec.rowscale, knobs={"probe_flags": flags})
                        for _ in range(2):
                            ops.spmm(gp, S_in, scratch, n_cols=side.Lk, transpose_out=True)
                        e0, e1 = ops.event(), ops.event()
                        ops.record(e0)
                        for _ in range(5):
                            ops.spmm(gp, S_in, scratch, n_cols=side.Lk, transpose_out=True)
                        ops.record(e1)
                        ops.synchronize()
                        t_probe[flags] = ops.elapsed_ms(e0, e1) / 5
                        gp.free()
                    scratch.free()
                    ph = t_probe[1] - t_probe[5]
                    out["roofline_mfma"]["phase_alone"] = {
                        "ms": ph, "achieved": flop / (ph * 1e-3) / 1e12, "unit": "TFLOP/s",
                        "frac": flop / (ph * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                        "ms_launch_without_gather_phase": t_probe[1], "ms_launch_without_mfma_phase": t_probe[4],
                        "ms_launch_with_neither_phase": t_probe[5], "ms_launch": l1,
                        "ms_mfma_phase_adds_to_the_launch": l1 - t_probe[4],
                        "note": "diagnostic launches (probe_flags 1 / 4 / 5: wrong results, timing only) of the "
                                "same plan on the same operand; operand segments come from the real "
                                "(L2-missing) ids (DESIGN.md §4.10)"}
                except Exception as e:
                    out["roofline_mfma"]["phase_alone"] = {"error": f"{type(e).__name__}: {e}"}
                finally:
                    os.environ.pop("SIMRANK_ENABLE_PROBES", None)
        elif dk and side.Lk:
            S_in = solver.cur[0][rank if use_dist else 0]
            for _ in range(2):
                ops.dense_part(side.graph, S_in, side.Lk)
            e0, e1 = ops.event(), ops.event()
            ops.record(e0)
            for _ in range(5):
                ops.dense_part(side.graph, S_in, side.Lk)
            ops.record(e1)
            dense_ms = ops.elapsed_ms(e0, e1) / 5
            flop = 2.0 * 3 * 128 * dk * side.Lk
            tf = flop / (dense_ms * 1e-3) / 1e12
            out["roofline_mfma"] = {
                "kernel": "dense_tiles (leg 1: 0/1 pattern blocks x operand rows, bf16 hi+mid+lo)",
                "bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / MFMA_BF16_PEAK_TF, "ms": dense_ms, "flop": flop,
                "row_blocks_with_dense_set": nt, "dense_columns": dk,
                "entries_covered": cov, "entries_covered_frac": cov / max(1, nnz),
                "note": "exact f32 products: three bf16 MFMAs per f32 operand term; the f32-equivalent "
                        "rate is a third of this"}
        rl = []
        leg1_name = ("leg 1 = fused_trans_kernel (dense sets on MFMA + gathered remainder, transposed store)" if fused
                     else "leg 1 = dense_tiles + spmm_gather (transposed store)")
        for name, ms, b in ((leg1_name, l1, b1),
                            ("leg 2 = spmm_gather (" + ("upper triangle" if tri else "half form" if half else
                                                        "full form") + ", fused epilogue)", l2, b2)):
            gbs = b / (ms * 1e-3) / 1e9
            rl.append({"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                       "ms": ms, "algorithmic_bytes": b,
                       "gathered_bytes": 4 * nnz * (side.Lk if "leg 1" in name else side.Lm)})
        if dense_ms is not None:
            rl[0]["parts_ms"] = {"dense_tiles": dense_ms, "spmm_gather": l1 - dense_ms}
            rl[0]["gathered_bytes"] = 4 * (nnz - cov) * side.Lk
            rl[0]["gather_kernel_alone"] = {"achieved": b1 / ((l1 - dense_ms) * 1e-3) / 1e9, "unit": "GB/s",
                                            "frac": b1 / ((l1 - dense_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if fused:
            # segments pulled through the vector-memory path: one per gathered entry and one per (block, dense
            # column) pair, per 32-column panel
            rl[0]["gathered_bytes"] = 4 * (frem + 16 * fsteps) * side.Lk
        # The second roofline of a gather leg: what the vector-memory path returns.  Every (entry, panel) pair
        # is one 128-byte segment pulled into registers; a loop of nothing but such loads (tools/micro/
        # gather_ceiling.hip, profiles/r02_gather_ceiling.log) reads 32-33 TB/s chip-wide from an L2-resident
        # slice and 10.4 TB/s from beyond the L2.
        for r, ms_k in ((rl[0], l1 - (dense_ms or 0.0)), (rl[1], l2)):
            if (tri or half) and r is rl[1]:
                continue                      # (the triangle / half form gathers a row-dependent share: not charged)
            g_tbs = r["gathered_bytes"] / (ms_k * 1e-3) / 1e12
            r["gather_path"] = {"gathered_TBps": g_tbs, "ceiling_TBps_L2_resident": 32.5,
                                "ceiling_TBps_beyond_L2": 10.4, "frac_of_L2_resident_ceiling": g_tbs / 32.5,
                                "note": "gathered segments only (4 x entries x columns bytes; id streams, partial "
                                        "sums and the store ride on the same path on top: +15 % by the L1 counters); "
                                        "ceilings measured with tools/micro/gather_ceiling.hip"}
        if half and "unpack.0" in legs:
            rl[1]["unpack_ms"] = legs["unpack.0"][0]
        if use_dist:
            # what the engine's stream spent waiting for / running the collectives (rank 0's events)
            out["exchange_ms"] = {k: v[0] for k, v in legs.items() if k.startswith("exchange")}
        if tri:
            rl[1]["algorithmic_bytes_full_form"] = b2_full
            rl[1]["note"] = ("single rank: leg 2 computes the tiles on/above the diagonal and stores their "
                             "mirror image (S' is symmetric); `algorithmic_bytes` are those of this form (all of "
                             "Tt read, all of S' written, half of the previous iterate read), not of the full form")
        # what the runtime's own device-to-device copy of S moves per second on this GPU (read +
        # write): the practical ceiling next to the 8 TB/s spec figure used for `frac`
        try:
            src, dst = solver.cur[0][rank if use_dist else 0], solver.nxt[0][rank if use_dist else 0]
            ops.copy(dst, src)
            e0, e1 = ops.event(), ops.event()
            ops.record(e0)
            for _ in range(3):
                ops.copy(dst, src)
            ops.record(e1)
            copy_gbs = 2 * src.nbytes / (ops.elapsed_ms(e0, e1) / 3 * 1e-3) / 1e9
            for r in rl:
                r["box_memcpy_GBps"] = copy_gbs
                r["frac_of_box_memcpy"] = r["achieved"] / copy_gbs
        except Exception:
            pass
        rl.sort(key=lambda r: -r["ms"])
        out["roofline"], out["roofline_other"] = rl[0], rl[1]
    else:
        flops = 2.0 * side.M * side.M * side.K
        ms = legs["leg2.0"][0]
        tf = flops / (ms * 1e-3) / 1e12
        out["roofline"] = {"kernel": "gemm_nt_mfma (leg 2, fused epilogue)", "bound": "mfma",
                           "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                           "frac": tf / MFMA_F32_PEAK_TF, "traffic": None, "ms": ms}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and "roofline" in out:
        try:
            # (the recorded passes are of the single-rank solver's kernels: a sharded run — also --force-dist's world of one — launches
            # other instantiations on other layouts and carries no traffic figure)
            key = f"{args.workload}:{world_size}" + (":pp" if args.pp else "") + \
                  (":fp16" if args.dense_precision == "fp16" else "") + (":sharded" if use_dist else "")
            rec = json.load(open(pmc)).get(key, {})
            for r in (out["roofline"], out.get("roofline_other", {})):
                key = "leg1" if "leg 1" in r.get("kernel", "") else "leg2"
                if key in rec:
                    r["traffic"] = rec[key]
                    r["traffic_source"] = ("profiles/pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc "
                                           f"passes at commit {rec.get('commit', '?')}, NOT measured in this run")
        except Exception:
            pass

    if not args.no_extras:
        # wall-clock to converge with the reference's defaults (eps = 1e-4)
        barrier()
        t0 = time.perf_counter()
        k = hplan.run(100, 1e-4)[1] if hplan is not None else solver.run(100, 1e-4)
        barrier()
        out["converge"] = {"eps": 1e-4, "iterations": k, "seconds": time.perf_counter() - t0}

    if not args.no_extras and world_size == 1 and args.workload == "pl32768d32" and not args.pp:
        # continuity with rounds 1-2: the lighter graph of the SURVEY.md recipe (783 100 of its 1 048 576
        # draws survive de-duplication: mean degree 23.9)
        try:
            dfd = synth.WORKLOADS["pl32768"][0]()
            _, csrd = ingest.directed(dfd, False, "from", "to", "weight")
            sd = Solver(lambda r: ops, one_rank, [make_spec(csrd, False)], args.mode)
            sd.exact_count = True
            sd.reset()
            for _ in range(3):
                sd.step(0.0)
            sd.enable_timing(10)
            ops.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                sd.step(0.0)
            ops.synchronize()
            dt = (time.perf_counter() - t0) / 10
            lt = sd.leg_times()
            ntd, dkd, covd = ops.dense_stats(next(iter(sd.sides[0].values())).graph)
            out["continuity_pl32768"] = {
                "workload": f"pl32768: synthetic directed graph N={csrd.n_rows} nnz={csrd.nnz} (mean degree "
                            f"{csrd.nnz / csrd.n_rows:.1f} after de-duplication; the headline of rounds 1-2) "
                            f"SimRank C=0.8 fp32",
                "value": 1.0 / dt, "unit": "iterations/s", "ms_per_step": dt * 1e3,
                "leg1_ms": lt["leg1.0"][0], "leg2_ms": lt["leg2.0"][0],
                "leg1_algorithmic_GBps": leg_bytes(csrd.n_rows, csrd.n_rows, csrd.n_rows, csrd.nnz, False)
                / (lt["leg1.0"][0] * 1e-3) / 1e9,
                "entries_in_dense_blocks_frac": covd / max(1, csrd.nnz)}
            sd.release()
            del sd
        except Exception as e:
            out["continuity_pl32768"] = {"error": f"{type(e).__name__}: {e}"}

    if not args.no_extras and world_size == 1 and args.workload != "er8192":
        # BASELINE.json configs[1] (ER N=8192, p=0.001) next to the headline configuration
        try:
            df2 = synth.WORKLOADS["er8192"][0]()
            _, csr2 = ingest.directed(df2, False, "from", "to", "weight")
            s2 = Solver(lambda r: ops, one_rank, [SideSpec(csr2, csr2.rowscale, coef)], args.mode)
            s2.exact_count = True
            s2.reset()
            for _ in range(3):
                s2.step(0.0)
            s2.enable_timing(50)
            ops.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                s2.step(0.0)
            ops.synchronize()
            dt = (time.perf_counter() - t0) / 50
            lt = s2.leg_times()
            n2, z2 = csr2.n_rows, csr2.nnz
            out["secondary"] = {
                "workload": f"er8192: synthetic ER directed graph N={n2} nnz={z2} SimRank C=0.8 fp32",
                "value": 1.0 / dt, "unit": "iterations/s", "ms_per_step": dt * 1e3,
                "leg1_ms": lt["leg1.0"][0], "leg2_ms": lt["leg2.0"][0],
                "leg1_algorithmic_GBps": leg_bytes(n2, n2, n2, z2, False) / (lt["leg1.0"][0] * 1e-3) / 1e9,
                "leg2_algorithmic_GBps": leg_bytes(n2, n2, n2, z2, True, triangle=True)
                                         / (lt["leg2.0"][0] * 1e-3) / 1e9}
            try:
                # the Python driver's loop (Solver.run: update k + 1 queued before the count of update k is read)
                s2.events = None
                s2.run(5, 0.0)
                ops.synchronize()
                t0 = time.perf_counter()
                s2.run(100, 0.0)
                ops.synchronize()
                dtf = (time.perf_counter() - t0) / 100
                out["secondary"]["python_driver_loop"] = {"value": 1.0 / dtf, "unit": "iterations/s", "ms_per_step": dtf * 1e3,
                                                          "note": "driver.Solver.run, 100 updates, eps = 0 (what fit() ran until round 4)"}
            except Exception as e:
                out["secondary"]["python_driver_loop"] = {"error": f"{type(e).__name__}: {e}"}
            if gpu:
                try:
                    # what fit() runs: the loop behind the C ABI with the reference's console hooks attached
                    # (cplan.PlanSolver -> simrank_plan_run_cb: update k + 1 queued before the count of update k is read)
                    from simrank_amd.cplan import PlanSolver
                    ps = PlanSolver(ops, LocalWorld(1), [SideSpec(csr2, csr2.rowscale, coef)])
                    ps.run(5, 0.0)
                    ops.synchronize()
                    ticks = []
                    t0 = time.perf_counter()
                    ps.run(100, 0.0, on_iteration=ticks.append)
                    dtp = (time.perf_counter() - t0) / 100
                    out["secondary"]["fit_loop"] = {"value": 1.0 / dtp, "unit": "iterations/s", "ms_per_step": dtp * 1e3,
                                                    "note": "cplan.PlanSolver.run = simrank_plan_run_cb, 100 updates, eps = 0, a "
                                                            "progress hook called every loop index: the loop of fit()"}
                    # the same plan step by step with the exact count, legs stamped on its stream
                    pl2 = ps.plan
                    pl2.reset()
                    for _ in range(3):
                        pl2.step(0.0, exact_count=True)
                    pl2.set_timing(50)
                    ops.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        pl2.step(0.0, exact_count=True)
                    ops.synchronize()
                    dts = (time.perf_counter() - t0) / 50
                    a1, a2, _n = pl2.leg_times()
                    out["secondary"]["c_plan_steps"] = {"value": 1.0 / dts, "unit": "iterations/s", "ms_per_step": dts * 1e3,
                                                        "leg1_ms": a1, "leg2_ms": a2,
                                                        "note": "simrank_plan_step x 50, exact count read every update"}
                    out["secondary"]["roofline"] = config_roofline("er8192:1", [
                        ("leg 1 (transposed store)", "leg1", a1, leg_bytes(n2, n2, n2, z2, False)),
                        ("leg 2 = gather3_kernel (upper triangle, fused epilogue, exact count)", "leg2", a2,
                         leg_bytes(n2, n2, n2, z2, True, triangle=True))],
                        note="N = 8192: S, the transposed product and S' are 268 MB each - most of what the legs read is served "
                             "by the 256 MB Infinity Cache and the L2s, so the HBM figure is not what binds here: the vector-"
                             "memory path of the gathers is (the launches are 0.1 ms long; see `gather_path` of the headline)")
                    ps.plan.free()
                except Exception as e:
                    out["secondary"]["fit_loop"] = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0 and not args.no_cpu_baseline:
                # the oracle on the whole N=8192 workload (no sampling needed at this size)
                from oracle import simrank_oracle as O
                G2 = csr2.dense()
                S2 = s2.result(0)
                O.update_rows(G2, S2, coef, slice(0, 8))
                t0 = time.perf_counter()
                new = O.update(G2, S2, coef)
                O.converged(S2.copy(), new, 1e-4)
                t_cpu = time.perf_counter() - t0
                out["secondary"]["cpu_baseline"] = {
                    "value": 1.0 / t_cpu, "unit": "iterations/s", "kind": "port",
                    "sample": f"one full oracle iteration (dense f64, N={n2}): {t_cpu:.2f} s"}
            s2.release()
            del s2
        except Exception as e:
            out["secondary"] = {"error": f"{type(e).__name__}: {e}"}

    if not args.no_extras and world_size == 1:
        # BASELINE.json configs[2]: MovieLens-1M-shaped bipartite SimRank++ (6040 x 3706, 1.0 M
        # edges; corrected Evidence_N2 — the reference cannot run n1 != n2)
        try:
            df3 = synth.WORKLOADS["ml1m"][0]()
            _, _, _, _, g12, g21 = ingest.bipartite(df3, False, "user", "item", "weight")
            s3 = Solver(lambda r: ops, one_rank,
                        [SideSpec(g12, g12.rowscale, coef, evidence_from=g12),
                         SideSpec(g21, g21.rowscale, coef, evidence_from=g21)], args.mode)
            s3.exact_count = True
            s3.reset()
            for _ in range(2):
                s3.step(0.0)
            s3.enable_timing(10)
            ops.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                s3.step(0.0)
            ops.synchronize()
            dt = (time.perf_counter() - t0) / 10
            st3 = [ops.fused_stats(next(iter(sd.values())).graph) for sd in s3.sides]     # (steps, covered, gathered) per panel
            out["bipartite_pp"] = {
                "workload": f"ml1m: synthetic MovieLens-1M-shaped bipartite graph {g12.n_rows} x "
                            f"{g12.n_cols}, nnz={g12.nnz}, BipartiteSimRankPP C1=C2=0.8 fp32",
                "value": 1.0 / dt, "unit": "iterations/s", "ms_per_step": dt * 1e3,
                "legs_ms": {k: v[0] for k, v in s3.leg_times().items()},
                "legs_ms_note": "HIP events of the Python driver's launches of the same kernels (the C loop queues them "
                                "back to back)",
                "entries_in_dense_sets": [c / max(1, g12.nnz) for _, c, _ in st3]}
            try:
                # the mean launch of each leg over both groups (what the per-kernel rocprofv3 figures average too):
                # group 1: S1' = C1 W12 S2 W12^T (leg 1 reads S2, n2 x n2, writes n1 x n2; leg 2 reads that, writes S1,
                # n1 x n1, upper triangle + mirror); group 2 likewise with the sizes swapped
                lm = out["bipartite_pp"]["legs_ms"]
                m1, m2, z3 = g12.n_rows, g12.n_cols, g12.nnz
                out["bipartite_pp"]["roofline"] = config_roofline("ml1m:1:pp", [
                    ("leg 1 (fused_trans_kernel / gather3_kernel<1>), mean of both groups", "leg1",
                     0.5 * (lm["leg1.0"] + lm["leg1.1"]),
                     0.5 * (leg_bytes(m1, m2, m2, z3, False) + leg_bytes(m2, m1, m1, z3, False))),
                    ("leg 2 = fused_trans_kernel<SYM> (one launch: matrix cores + gathers + evidence epilogue + exact count, upper "
                     "triangle + mirror), mean of both groups", "leg2",
                     0.5 * (lm["leg2.0"] + lm["leg2.1"]),
                     0.5 * (leg_bytes(m1, m2, m1, z3, True, has_evidence=True, triangle=True)
                            + leg_bytes(m2, m1, m2, z3, True, has_evidence=True, triangle=True)))],
                    note="6040 x 3706: the matrices are 146 MB, 55 MB and 90 MB - Infinity-Cache resident; the HBM figure is "
                         "the contract's, the gathers' vector-memory path is what binds")
                # the launches of this configuration are paced by their matrix-core phase (90 % of the entries in dense sets;
                # DESIGN 4.10b): leg 1 priced against the dense bf16 MFMA peak — 12 v_mfma_f32_32x32x16_bf16 (three bf16 terms
                # x four 32-row tiles) per 16-column step and 32-column panel
                fl = [st3[i][0] * 12 * 32768.0 * ((k + 31) // 32) for i, k in ((0, m2), (1, m1))]
                ms1 = [lm["leg1.0"], lm["leg1.1"]]
                tf = sum(fl) / (sum(ms1) * 1e-3) / 1e12
                out["bipartite_pp"]["roofline_mfma"] = {
                    "kernel": "matrix-core phase of leg 1 (both groups), exact f32 products as three bf16 terms", "bound": "mfma",
                    "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TF,
                    "flop_per_launch": fl, "ms": ms1, "steps_per_panel": [st3[0][0], st3[1][0]],
                    "f32_equivalent_TFLOPs": tf / 3.0}
            except Exception as e:
                out["bipartite_pp"]["roofline"] = {"error": f"{type(e).__name__}: {e}"}
            s3.release()
            del s3
            if gpu:
                # what BipartiteSimRankPP.fit() runs: the loop behind the C ABI (cplan.PlanSolver -> simrank_biplan_run_cb:
                # iteration k + 1 queued before the counts of iteration k are read); 50 loop bodies with eps = 0, a
                # progress hook called every loop index; then 10 loop bodies step by step with the exact counts
                from simrank_amd.cplan import PlanSolver
                t0 = time.perf_counter()
                ps3 = PlanSolver(ops, LocalWorld(1), [SideSpec(g12, g12.rowscale, coef, evidence_from=g12),
                                                      SideSpec(g21, g21.rowscale, coef, evidence_from=g21)])
                ops.synchronize()
                setup_s = time.perf_counter() - t0
                ps3.run(5, 0.0)
                ticks = []
                t0 = time.perf_counter()
                ps3.run(50, 0.0, on_iteration=ticks.append)
                dt_p = (time.perf_counter() - t0) / 50
                bp = ps3.plan
                bp.reset()
                bp.step(0.0, True)
                ops.synchronize()
                t0 = time.perf_counter()
                for _ in range(40):
                    bp.step(0.0, True)
                dt_s = (time.perf_counter() - t0) / 40
                bp.free()
                out["bipartite_pp"]["python_driver"] = {k: out["bipartite_pp"][k] for k in ("value", "unit", "ms_per_step")}
                out["bipartite_pp"]["python_driver"]["note"] = "driver.Solver.step x 10 (what fit() ran until round 4)"
                out["bipartite_pp"].update({"value": 1.0 / dt_s, "ms_per_step": dt_s * 1e3,
                                            "loop": "simrank_biplan_step x 40, exact counts read every loop body (the C loop fit() runs)"})
                out["bipartite_pp"]["fit_loop"] = {"value": 1.0 / dt_p, "unit": "iterations/s", "ms_per_step": dt_p * 1e3,
                                                   "setup_s": setup_s,
                                                   "note": "cplan.PlanSolver.run = simrank_biplan_run_cb, 50 loop bodies, eps = 0, "
                                                           "progress hook every loop index: the loop of BipartiteSimRankPP.fit()"}
        except Exception as e:
            out["bipartite_pp"] = {"error": f"{type(e).__name__}: {e}"}

    if not args.no_extras and world_size == 1 and args.workload == "pl32768d32" and not args.pp:
        # BASELINE.json configs[4]: N = 65536 SimRank++ with evidence weights: exact (f32 storage, three bf16
        # terms on the matrix cores), its "fp16 MFMA dense leg" taken literally (one fp16 operand term for the
        # blocks on the matrix cores; buys nothing) and the mode that pays: S and the intermediate product
        # HELD in fp16 (half.hip).  Reduced-precision runs are priced by their error: against the exact run
        # on sampled rows here, and against the float64 oracle on a whole N = 4096 fit below.
        try:
            df5 = synth.WORKLOADS["pl65536"][0]()
            _, csr5 = ingest.directed(df5, False, "from", "to", "weight")
            rows5 = [0, 11, csr5.n_rows // 2, csr5.n_rows - 300, csr5.n_rows - 2, csr5.n_rows - 1]
            res5, sample = {}, {}
            for prec, terms, storage in (("f32", 3, "f32"), ("fp16", 1, "f32"), ("fp16_storage", 3, "fp16")):
                first_s = None
                for attempt in range(2 if prec == "f32" else 1):
                    # (the first creation of matrices of this size in the process pays the allocator — hipMalloc maps a
                    # 17 GiB block in 0.0 .. 0.5 s depending on the box —; the second takes them from the block pool)
                    t0 = time.perf_counter()
                    s5 = Solver(lambda r: ops, one_rank, [make_spec(csr5, True, terms, storage)], args.mode)
                    ops.synchronize()
                    setup_s = time.perf_counter() - t0
                    if prec == "f32" and attempt == 0:
                        first_s = setup_s
                        s5.release()
                        del s5
                s5.exact_count = True
                s5.reset()
                for _ in range(2):
                    s5.step(0.0)
                s5.enable_timing(6)
                ops.synchronize()
                t0 = time.perf_counter()
                for _ in range(6):
                    s5.step(0.0)
                ops.synchronize()
                dt = (time.perf_counter() - t0) / 6
                lt = s5.leg_times()
                sample[prec] = ops.download_rows(s5.cur[0][0], rows5).astype(np.float64)
                nt5, dk5, cov5 = ops.dense_stats(next(iter(s5.sides[0].values())).graph)
                res5[prec] = {"value": 1.0 / dt, "unit": "iterations/s", "ms_per_step": dt * 1e3,
                              "leg1_ms": lt["leg1.0"][0], "leg2_ms": lt["leg2.0"][0],
                              "setup_s_graph_and_evidence": setup_s}
                if first_s is not None:
                    res5[prec]["setup_s_first_call_in_process"] = first_s
                s5.release()
                del s5

            def err_stats(low, ref):
                pos = ref > 0
                rel = np.abs(low[pos] - ref[pos]) / ref[pos]
                return {"elements": int(pos.sum()), "max_rel": float(rel.max()), "median_rel": float(np.median(rel)),
                        "p99_rel": float(np.quantile(rel, 0.99)), "max_abs": float(np.abs(low - ref).max())}

            res5["fp16_storage"]["vs_f32_ms_per_step"] = res5["fp16_storage"]["ms_per_step"] / res5["f32"]["ms_per_step"]
            n5, z5 = csr5.n_rows, csr5.nnz
            for prec, key5, elt, k1, k2 in (("f32", "pl65536:1:pp", 4, "fused_trans_kernel", "gather3_kernel"),
                                            ("fp16_storage", "pl65536:1:pp:fp16storage", 2, "half_leg_kernel (leg 1)",
                                             "half_leg_kernel (leg 2)")):
                res5[prec]["roofline"] = config_roofline(key5, [
                    (f"leg 1 = {k1} (dense sets on MFMA + gathered remainder, transposed store)", "leg1",
                     res5[prec]["leg1_ms"], leg_bytes(n5, n5, n5, z5, False, elt=elt)),
                    (f"leg 2 = {k2} (upper triangle, evidence epilogue, exact count)", "leg2",
                     res5[prec]["leg2_ms"], leg_bytes(n5, n5, n5, z5, True, has_evidence=True, triangle=True, elt=elt))])
            out["config5"] = {
                "workload": f"pl65536: synthetic directed graph N={csr5.n_rows} nnz={csr5.nnz} SimRank++ "
                            f"(evidence counts in the epilogue, spread weights) C=0.8, one GPU, 8 iterations",
                "entries_in_dense_blocks_frac": cov5 / max(1, csr5.nnz),
                "f32_exact_dense_blocks": res5["f32"], "fp16_dense_blocks": res5["fp16"],
                "fp16_storage": res5["fp16_storage"],
                "fp16_vs_f32_error": dict(rows_sampled=len(rows5), **err_stats(sample["fp16"], sample["f32"])),
                "fp16_storage_vs_f32_error": dict(rows_sampled=len(rows5),
                                                  **err_stats(sample["fp16_storage"], sample["f32"])),
                "note": "fp16_dense_blocks: fp16 for the operand of the matrix-core part only, the gathered remainder "
                        "stays f32 (buys nothing: the bytes are in the gathered operand).  fp16_storage: S, the "
                        "transposed product and the previous iterate held in fp16 on 64-column panels, f32 sums and "
                        "epilogue, one rounding per stored value, matrix-core part on one exact fp16 term — half "
                        "the gathered lines and half the streamed bytes per update.  Never the default.  A full "
                        "dense f16 GEMM of W (2N^3 = 5.6e14 flop) would take >= 225 ms per leg at the 2.5 PFLOP/s "
                        "peak against these timings."}
        except Exception as e:
            out["config5"] = {"error": f"{type(e).__name__}: {e}"}
        # the reduced-precision mode against the float64 oracle: a whole SimRank++ fit to eps = 1e-4 at N = 4096
        if gpu and not args.no_cpu_baseline:
            try:
                import simrank_amd.SimRank as SRA
                from oracle import simrank_oracle as O        # the checker (never the thing measured)
                df4 = synth.powerlaw_directed(4096, 24, 4096)
                checks = {}
                for label, okw in (("ten_updates", dict(iterations=10, eps=1e-30)), ("to_eps_1e-4", {})):
                    want = O.fit_simrank_pp(df4, verbose=False, **okw)
                    fits = {}
                    for name, kw in (("f32", {}), ("fp16_storage", dict(storage_precision="fp16"))):
                        est = SRA.SimRankPP()
                        t0 = time.perf_counter()
                        got = est.fit(df4, verbose=False, **okw, **kw).values
                        fits[name] = dict(fit_wall_s=time.perf_counter() - t0, converged_at=est.converged_at,
                                          **err_stats(got, want["S"]))
                    checks[label] = {"oracle_converged_at": want["k"], **fits}
                out["config5"]["oracle_check_n4096"] = {
                    "workload": "power-law N=4096, mean degree 24 before de-duplication, SimRank++ (the float64 "
                                "oracle is the checker): exactly ten updates on both sides = the arithmetic error; "
                                "to eps = 1e-4 the fp16-held loop may stop later than the reference's (values above "
                                "1/8 are stored with a spacing above eps), its result is then nearer the fixed point",
                    **checks}
            except Exception as e:
                out["config5"]["oracle_check_n4096"] = {"error": f"{type(e).__name__}: {e}"}

    if gpu and not args.no_extras and world_size == 1 and args.workload == "pl32768d32" and not args.pp:
        # BASELINE.json's metric names wall-clock to converge: whole fits through the reference's class surface,
        # ingest + set-up (graph objects, evidence counts) + updates to eps = 1e-4 + hand-back, second call of
        # each (the first pays the allocator), and the evidence kernel by itself
        try:
            import simrank_amd.SimRank as SRA
            walls = {}
            type(ops).trim_pool()            # (the block pool is full of this process's earlier configurations)

            def timed_fit(name, make, *a, **kw):
                # (one warm call — allocator, host frames, plans' first-touch —, then the fastest of three: the hand-back's
                # host threads share the box's CPUs with other tenants, and one busy moment moved these lines by 30 %)
                best, calls = None, []
                for rep in range(4):
                    est = make()
                    t0 = time.perf_counter()
                    res = est.fit(*a, verbose=False, **kw)
                    for frame in (res if isinstance(res, tuple) else (res,)):
                        if hasattr(frame, "values"):
                            frame.values                         # (the similarity frames are the hand-back)
                    dt = time.perf_counter() - t0
                    if rep:
                        calls.append(dt)
                        if best is None or dt < best["fit_wall_s"]:
                            best = dict(fit_wall_s=dt, converged_at=est.converged_at)
                    # (every reference to the N x N host frame goes BEFORE the next call is timed: unmapping 8.6 GB takes
                    # 0.4 s, and the loop variable kept the first call's frame alive into the second call's timed region —
                    # the 0.68 s this line showed for config 4 until round 4; profiles/r04_fit_breakdown_cfg4.log)
                    frame = None
                    del res, est, frame
                best["calls_s"] = calls
                walls[name] = best

            df4 = synth.WORKLOADS["pl32768d32"][0]()
            timed_fit("cfg4_SimRank_pl32768d32_full_handback", SRA.SimRank, df4)
            try:
                # A/B of the hand-back: the symmetric form (upper triangle over PCIe, mirrored by the host threads; same bits)
                os.environ["SIMRANK_SYM_HANDBACK"] = "1"
                timed_fit("cfg4_SimRank_pl32768d32_upper_triangle_over_pcie", SRA.SimRank, df4)
            finally:
                os.environ.pop("SIMRANK_SYM_HANDBACK", None)
            try:
                # the same fit through the C-level plan: create (graph + plans + matrices), run to eps, f64 hand-back
                from simrank_amd.engine import Plan
                # (round 6: the split now holds everything fit() does — edge list -> CSR before, the labelled DataFrame after —
                # so its parts add up to the wall clock of the class surface, the line above)
                import pandas as pd
                best, calls = None, []
                for rep in range(4):                 # (a warm call, then the fastest of three, as timed_fit)
                    ti = time.perf_counter()
                    labels4, csr4 = ingest.directed(df4, False, "from", "to", "weight")
                    t0 = time.perf_counter()
                    plan = Plan(ops, csr4, coef=0.8)
                    t1 = time.perf_counter()
                    done, conv = plan.run(100, 1e-4)
                    t2 = time.perf_counter()
                    res4 = plan.result()
                    t3 = time.perf_counter()
                    frame4 = pd.DataFrame(res4, index=list(labels4), columns=list(labels4), copy=False)
                    t4 = time.perf_counter()
                    plan.free()
                    if rep:
                        calls.append(t4 - ti)
                        if best is None or t4 - ti < best["fit_wall_s"]:
                            best = dict(fit_wall_s=t4 - ti, ingest_s=t0 - ti, create_s=t1 - t0, run_s=t2 - t1,
                                        result_f64_s=t3 - t2, frame_s=t4 - t3, converged_at=conv)
                    frame4 = None
                    del res4, frame4
                best["calls_s"] = calls
                walls["cfg4_SimRank_pl32768d32_c_plan_split"] = best
                csr5p = ingest.directed(synth.WORKLOADS["pl65536"][0](), False, "from", "to", "weight")[1]
                tc = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    plan = Plan(ops, csr5p, ingest.spread(csr5p) * csr5p.rowscale, coef=0.8, evidence=True)
                    tc.append(time.perf_counter() - t0)
                    plan.free()
                walls["cfg5_simrank_plan_create_s"] = {"first": tc[0], "second": tc[1],
                                                       "note": "N = 65536 SimRank++: graph + plans + evidence counts + three 17 GiB "
                                                               "matrices; the second call takes them from the library's block pool"}
            except Exception as e:
                walls["cfg4_SimRank_pl32768d32_c_plan_split"] = {"error": f"{type(e).__name__}: {e}"}
            df3 = synth.WORKLOADS["ml1m"][0]()
            timed_fit("cfg3_BipartiteSimRankPP_ml1m_full_handback", SRA.BipartiteSimRankPP, df3, strict_reference=False)
            df5 = synth.WORKLOADS["pl65536"][0]()
            timed_fit("cfg5_SimRankPP_pl65536_top10_handback", SRA.SimRankPP, df5, top_k=10)
            timed_fit("cfg5_SimRankPP_pl65536_top10_handback_fp16_storage", SRA.SimRankPP, df5, top_k=10,
                      storage_precision="fp16")
            _, csr5 = ingest.directed(df5, False, "from", "to", "weight")
            g5 = ops.graph(csr5)
            cnt = ops.matrix(csr5.n_rows, csr5.n_rows, np.uint8, blocked=True)
            ops.evidence_counts(g5, 0, cnt)
            ev0, ev1 = ops.event(), ops.event()
            ops.record(ev0)
            for _ in range(3):
                ops.evidence_counts(g5, 0, cnt)
            ops.record(ev1)
            ops.synchronize()
            ev_ms = ops.elapsed_ms(ev0, ev1) / 3
            ev_bytes = csr5.n_rows ** 2 + 8 * csr5.nnz
            walls["evidence_counts_kernel_cfg5"] = {
                "ms": ev_ms, "algorithmic_bytes": ev_bytes, "GBps": ev_bytes / ev_ms / 1e6,
                "frac_of_hbm_peak": ev_bytes / ev_ms / 1e6 / HBM_PEAK_GBS,
                "note": "N^2 one-byte counts written + the pattern read twice; the kernel is bound by its LDS "
                        "counter updates (one per 2-hop path), not by these bytes (DESIGN.md §4.2)"}
            cnt.free()
            g5.free()
            out["fit_wall"] = walls
        except Exception as e:
            out["fit_wall"] = {"error": f"{type(e).__name__}: {e}"}

    if not args.no_extras and world_size == 1 and gpu and solver.mode == "sparse" and n % 512 == 0:
        # what ONE rank of an 8-rank world spends in kernels on this graph (north_star's configs 4 and 5 are 8-GPU
        # configurations; no multi-GPU node is needed to time a rank's launches): LocalWorld(8) runs the eight shards one
        # after another on this GPU with the exchanges done by device copies — nothing here is a multi-GPU measurement
        try:
            type(ops).trim_pool()
            emu = {}
            sh = Solver(lambda r: ops, LocalWorld(8), [make_spec(csr, args.pp)], "sparse")
            sh.reset()
            sh.step(0.0)
            sh.enable_timing()
            for _ in range(3):
                sh.step(0.0)
            lt = sh.leg_times()
            side8 = sh.sides[0][0]
            emu["python_driver_f32"] = {
                "leg2_form": "half" if side8.shard_sym else "full",
                "leg1_ms": lt["leg1.0"][0], "leg2_ms": lt["leg2.0"][0], "unpack_ms": lt.get("unpack.0", (0.0, 0))[0],
                "kernels_per_rank_ms": lt["leg1.0"][0] + lt["leg2.0"][0] + lt.get("unpack.0", (0.0, 0))[0],
                "exchange_payload_per_rank_MiB": (4.0 * n * n / 8 * 7 / 8 + (4.0 * side8.sh_chunk * 7 if side8.shard_sym else 0)) / 2**20}
            sh.release()
            del sh
            from simrank_amd.engine import ShardPlans
            spec8 = make_spec(csr, args.pp)
            for label, kw in (("c_loop_f32_half_form", dict(leg2_form=1)), ("c_loop_fp16_held_full_form", dict(storage="fp16"))):
                sp = ShardPlans(ops, csr, rowscale=spec8.rowscale, world=8, coef=coef, evidence=args.pp, stages=1, **kw)
                sp.step(0.0)
                sp.set_timing(3)
                ops.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    sp.step(0.0, exact_count=False)
                ops.synchronize()
                emu[label] = {"per_rank_ms_device_copies_included": (time.perf_counter() - t0) / 3 / 8 * 1e3,
                              "events_ms_all_eight_ranks_in_turn": sp.timings()}
                sp.free()
            emu["note"] = ("eight virtual ranks on ONE GPU, run one after another (driver.LocalWorld / simrank_comm_local_group): a "
                           "rank's kernel time, not a multi-GPU measurement; single rank on this box: "
                           f"{out['ms_per_step']:.2f} ms per step")
            out["shards_emulated_p8"] = emu
            type(ops).trim_pool()
        except Exception as e:
            out["shards_emulated_p8"] = {"error": f"{type(e).__name__}: {e}"}

    if not args.no_extras and world_size == 1 and solver.mode == "sparse" and n <= 32768:
        # BASELINE.json's literal recipe — sparse leg 1, dense f32 MFMA GEMM for leg 2 —
        # measured on the same workload so the dispatch decision is a number, not a claim
        try:
            hy = Solver(lambda r: ops, one_rank, [SideSpec(csr, csr.rowscale, coef)], "hybrid")
            hy.reset()
            hy.step(0.0)
            hy.enable_timing()
            ops.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                hy.step(0.0)
            ops.synchronize()
            dt = (time.perf_counter() - t0) / 2
            ms2 = hy.leg_times()["leg2.0"][0]
            tf = 2.0 * n * n * n / (ms2 * 1e-3) / 1e12
            out["mfma_dense_leg"] = {
                "mode": "hybrid (gather leg 1 + gemm_nt_mfma leg 2 on densified W)",
                "ms_per_step": dt * 1e3, "iterations_per_sec": 1.0 / dt,
                "roofline": {"kernel": "gemm_nt_mfma (leg 2, fused epilogue)", "bound": "mfma",
                             "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                             "frac": tf / MFMA_F32_PEAK_TF, "ms": ms2},
                "note": f"W has density {csr.density:.2e}: the GEMM multiplies "
                        f"{100 * (1 - csr.density):.2f} % zeros; mode=auto sends only the dense blocks "
                        f"of W to the matrix cores (roofline_mfma)"}
            hy.release()
            del hy
        except Exception as e:                       # an extra must never sink the headline
            out["mfma_dense_leg"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        try:
            import psutil
            avail = psutil.virtual_memory().available
        except Exception:
            avail = 0
        need = 3 * 8 * n * n
        if avail and avail < need * 1.2:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = f"skipped: needs {need / 2**30:.0f} GiB host RAM"
        else:
            if hplan is not None:
                hplan.run(3, 0.0)
                S_host = hplan.result()
                hplan.free()
            else:
                S_host = solver.result(0)
            solver.release()
            out["cpu_baseline"] = cpu_baseline(csr, S_host, coef)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        write_full(out, args.full_json)
        emit(compact_line(out))
    STATE["emitted"] = True
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception as exc:
        if not STATE.get("c_fallback") or STATE.get("emitted"):
            raise
        import traceback
        traceback.print_exc()
        sys.exit(c_loop_only(f"{type(exc).__name__}: {exc}"))
