#!/usr/bin/env python3
"""Short view of a bench.py JSON line.   python3 tools/bench_summary.py gpurun_out/bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['value']:.1f} it/s, {d['ms_per_step']:.3f} ms/step —", d["config"]["workload"][:70])
for k in ("roofline", "roofline_other"):
    r = d.get(k, {})
    print(f"  {k}: {r.get('kernel','')[:60]} {r.get('ms',0):.3f} ms, {r.get('achieved',0):.0f} {r.get('unit','')}, frac {r.get('frac',0):.3f}, traffic {r.get('traffic')}")
m = d.get("roofline_mfma", {})
print("  mfma:", m.get("achieved"), "TF; covered", m.get("entries_covered_frac"))
print("  short-circuit:", d.get("convergence_test", {}).get("iterations_per_sec_short_circuit"))
for k in ("continuity_pl32768", "secondary", "bipartite_pp", "converge", "cpu_baseline", "gpu_over_cpu", "mfma_dense_leg"):
    print(f"  {k}:", json.dumps(d.get(k))[:420])
c5 = d.get("config5", {})
print("  config5:", json.dumps({k: c5.get(k) for k in ("f32_exact_dense_blocks", "fp16_dense_blocks", "fp16_vs_f32_error", "error")})[:700])
print("  fit_wall:", json.dumps(d.get("fit_wall"))[:1800])
