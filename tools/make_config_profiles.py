#!/usr/bin/env python3
"""The rocprofv3 passes of tools/gpu_profile_configs.sh -> profiles/<TAG>_rocprof_summary.md (one table per BASELINE config:
kernel, calls, mean duration, HBM-side bytes per launch), profiles/<TAG>_kernel_stats_<config>.csv and the per-config keys
of profiles/pmc_traffic.json (what bench.py attaches as roofline.traffic).

bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half of a 16 B/lane stream,
MI355X_MICROARCH.md §HBM; Infinity-Cache hits are included in these counters).

    python3 tools/make_config_profiles.py gpurun_out/prof_TAG TAG"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"cfg2_er8192": "er8192:1", "cfg3_ml1m": "ml1m:1:pp", "cfg4_pl32768d32": "pl32768d32:1",
        "cfg5_pl65536_pp": "pl65536:1:pp", "cfg5_pl65536_pp_fp16": "pl65536:1:pp:fp16storage"}
LEG1 = ("fused_trans_kernel<true, false", "fused_trans_kernel<false, false", "gather3_kernel<1", "half_leg_kernel<true, false", "half_leg_kernel<false, false")
LEG2 = ("fused_trans_kernel<true, true", "fused_trans_kernel<false, true", "gather3_kernel<2", "gather3_kernel<0", "half_leg_kernel<true, true", "half_leg_kernel<false, true")
try:
    commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = "?"
commit = os.environ.get("PMC_COMMIT", commit or "?")


def short(name):
    name = name.replace("simrank::", "").replace("void ", "")
    return name[:name.index("(")] if "(" in name else name


def counter_sums(cfg, which, counter):
    """kernel name -> [sum over launches, launches]"""
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out_dir}/{cfg}/{which}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]][0] += float(r["Counter_Value"])
                acc[r["Kernel_Name"]][1] += 1
    return acc


def counters(cfg, which, counter):
    return {k: v[0] / v[1] for k, v in counter_sums(cfg, which, counter).items() if v[1]}


def leg_mean(sums, pats):
    """mean per launch over every instantiation of a leg's kernel (config 3 launches one per group)"""
    tot = sum(v[0] for k, v in sums.items() if any(p in k for p in pats))
    cnt = sum(v[1] for k, v in sums.items() if any(p in k for p in pats))
    return tot / cnt if cnt else 0.0


path = os.path.join(root, "profiles", "pmc_traffic.json")
rec = json.load(open(path)) if os.path.exists(path) else {}
lines = [f"# rocprofv3 at commit {commit} ({tag}): kernel stats and HBM-side bytes per launch, per BASELINE config", "",
         "Commands: `tools/gpu_profile_configs.sh` (`--kernel-trace --stats`; `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in runs of "
         "their own).  bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch (gfx950 correction of the microarchitecture guide; "
         "Infinity-Cache hits included).  Config 3 runs two launches of each leg per loop body (one per group): the means "
         "are over both.", ""]
for cfg, key in KEYS.items():
    stats = glob.glob(f"{out_dir}/{cfg}/stats/**/*kernel_stats.csv", recursive=True)
    if not stats:
        lines += [f"## {cfg}", "", "(no kernel stats: the pass did not complete)", ""]
        continue
    shutil.copy(stats[0], os.path.join(root, "profiles", f"{tag}_kernel_stats_{cfg}.csv"))
    fetch, write = counters(cfg, "fetch", "FETCH_SIZE"), counters(cfg, "write", "WRITE_SIZE")
    lines += [f"## {cfg}  (`pmc_traffic.json` key `{key}`)", "", "| kernel | calls | mean ms | % of GPU time | HBM-side MB per launch |", "|---|---|---|---|---|"]
    fs, ws = counter_sums(cfg, "fetch", "FETCH_SIZE"), counter_sums(cfg, "write", "WRITE_SIZE")
    legs = {leg: (2 * leg_mean(fs, pats) + leg_mean(ws, pats)) * 1024 for leg, pats in (("leg1", LEG1), ("leg2", LEG2))}
    # a leg 2 whose dense column sets run on the matrix cores first (config 3): dense_tiles_kernel + the gather kernel
    dense_n = sum(v[1] for k, v in fs.items() if "dense_tiles_kernel" in k)
    leg2_n = sum(v[1] for k, v in fs.items() if any(p in k for p in LEG2))
    dense_b = (2 * leg_mean(fs, ("dense_tiles_kernel",)) + leg_mean(ws, ("dense_tiles_kernel",))) * 1024
    with_dense = dense_n > 0 and dense_n == leg2_n
    if with_dense:
        legs["leg2"] += dense_b
    leg_ns = {"leg1": [0.0, 0], "leg2": [0.0, 0]}
    for r in csv.DictReader(open(stats[0])):
        name = r["Name"]
        b = (2 * fetch.get(name, 0.0) + write.get(name, 0.0)) * 1024
        ms = float(r["AverageNs"]) / 1e6
        if float(r["Percentage"]) >= 0.3:
            lines.append(f"| `{short(name)}` | {r['Calls']} | {ms:.3f} | {float(r['Percentage']):.1f} | {b / 1e6:.0f} |")
        for leg, pats in (("leg1", LEG1), ("leg2", LEG2)):
            if any(p in name for p in pats):
                leg_ns[leg][0] += float(r["TotalDurationNs"])
                leg_ns[leg][1] += int(r["Calls"])
    lines.append("")
    leg_ms = {leg: (v[0] / v[1] / 1e6 if v[1] else None) for leg, v in leg_ns.items()}
    if with_dense:
        for r in csv.DictReader(open(stats[0])):
            if "dense_tiles_kernel" in r["Name"] and leg_ms["leg2"] is not None:
                leg_ms["leg2"] += float(r["AverageNs"]) / 1e6
        lines += [f"(leg 2 here = `dense_tiles_kernel` + the gather kernel, one of each per launch of the leg: {dense_b / 1e6:.0f} MB of "
                  "the leg's bytes are the former's)", ""]
    lines += [f"mean per launch over a leg's instantiations: leg 1 {leg_ms['leg1'] or 0:.3f} ms, {legs['leg1'] / 1e6:.0f} MB; "
              f"leg 2 {leg_ms['leg2'] or 0:.3f} ms, {legs['leg2'] / 1e6:.0f} MB", ""]
    rec[key] = {"leg1": legs["leg1"], "leg2": legs["leg2"], "leg1_ms_rocprof": leg_ms["leg1"], "leg2_ms_rocprof": leg_ms["leg2"],
                "source": f"prof_{tag}/{cfg}", "commit": commit}
# the headline configuration's cache counters
for which, names in (("l2", ("TCC_HIT_sum", "TCC_MISS_sum")), ("tcp", ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum")),
                     ("sq", ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"))):
    rows = []
    for c in names:
        for k, v in counters("cfg4_pl32768d32", which, c).items():
            if any(p in k for p in LEG1 + LEG2):
                rows.append(f"| `{short(k)}` | {c} | {v:.4g} |")
    if rows:
        lines += [f"## cfg4_pl32768d32: {which} counters per launch", "", "| kernel | counter | mean per launch |", "|---|---|---|"] + rows + [""]
rec["_note"] = ("HBM-side bytes per launch from rocprofv3 PMC (tools/gpu_profile_configs.sh + tools/make_config_profiles.py): "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024 - FETCH_SIZE reads half of a 16 B/lane stream on gfx950 "
                "(MI355X_MICROARCH.md §HBM). Infinity-Cache hits are included. leg1 = fused_trans_kernel (fp16-held: "
                "half_leg_kernel<.., false>), leg2 = gather3_kernel<2> (half_leg_kernel<.., true>), exact-count form; each figure "
                "is the mean per launch over every instantiation of the leg's kernel (config 3 launches one per group: the mean is over both).")
json.dump(rec, open(path, "w"), indent=1)
open(os.path.join(root, "profiles", f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:6000])
