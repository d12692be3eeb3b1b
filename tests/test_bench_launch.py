"""``python bench.py --gpus N`` must launch its own ranks (the driver calls it exactly like that
when it is not already under torchrun).  Rehearsed here on CPU: same launch path
(child ``python -m torch.distributed.run``), real collectives over gloo, NumPy test double for the
kernels; the JSON line of rank 0 is relayed and the exit code is the child's."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, full_json):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    # (the complete record goes to the test's own directory: the default path is where a real run's record lives)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra, "--full-json", str(full_json)],
                          capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_bench_launches_its_own_ranks_over_gloo(tmp_path):
    p = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
             "--workload", "er:96:0.06", full_json=tmp_path / "full.json")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1                       # rank 0's line, relayed once
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2
    assert out["metric"] == "simrank_iterations_per_sec" and out["value"] > 0
    assert out["scaling"] == "strong" and "NOT a measurement" in out["data"]
    # the one command for an 8-GPU node carries every key of the sharded runs (here: named, and marked as not run)
    assert out["headline_loop"].startswith("tests/pydriver.Solver") and out["python_driver"]["value"] == out["value"]
    c = out["sharded_c_loop"]
    assert c["ranks"] == 2 and set(c["variants"]) == {"f32_full_form", "f32_half_form", "f32_half_form_fp16_wire",
                                                      "fp16_held_full_form"}
    assert all("skipped" in v and "value" not in v for v in c["variants"].values())
    assert [v["parity_grade"] for v in c["variants"].values()] == [True, True, False, False]
    assert "skipped" in c["config5"] and "skipped" in c["config3"] and "skipped" in c["form_measured"]
    assert "skipped" in c["link_probe"]          # (the xGMI link-rate probe of a real multi-GPU run)
    assert len(lines[0]) <= 8000                 # the driver's record keeps the tail of the line: it must be all of it
    assert "exchange_ms" in out
    assert abs(json.load(open(tmp_path / "full.json"))["value"] - out["value"]) < 1e-3 * out["value"]   # the complete record


def test_bench_child_failure_is_reported(tmp_path):
    p = _run("--gpus", "2", "--backend", "gloo", "--workload", "no-such-workload", full_json=tmp_path / "full.json")
    assert p.returncode != 0
