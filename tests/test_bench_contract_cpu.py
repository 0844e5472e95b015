"""CPU: the bench line committed under profiles/ carries every field of the driver's contract (the same code path prints it on
the GPU box), and bench.py's argument defaults stay within it."""
import ast
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_bench_line():
    files = sorted(f for f in os.listdir(os.path.join(REPO, "profiles")) if f.endswith("_bench_default.json"))
    assert files, "no committed bench line under profiles/"
    return json.loads(open(os.path.join(REPO, "profiles", files[-1])).read().strip().splitlines()[-1])


def test_bench_line_fields():
    d = _latest_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "bf16"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["peak"] == 2500.0 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - d["config"]["global_pairs_per_step"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert "pairs" in base["metric"] and "pairs" in d["metric"]
    # round-4 sub-records (never part of `value`): the other BASELINE.json configurations and the directly timed CPU baseline
    for k, unit in (("thirteen_b", "pairs/s"), ("recipe", "samples/s")):
        if k in d:
            assert "error" not in d[k], d[k]
            assert d[k]["unit"] == unit and d[k]["value"] > 0 and d[k]["ms_per_step"] > 0
    if "in_run_extrapolation" in c:      # round 6: `value` is the directly measured full-depth figure, the bounded in-run sample rides beside it
        assert c["extrapolated"] is False and c["source"].endswith("_cpu_baseline_full.json") and c["seconds_per_pair"] > 0
        assert c["in_run_extrapolation"]["extrapolated"] is True and c["in_run_extrapolation"]["value"] > 0
    par = d.get("parity")
    if par is not None and str(par.get("source", "")).startswith("this run"):      # round 6: produced by the run itself (child process), not read from a committed file
        assert par["assertions_passed"] is True and par["layers"] == 32 and 0 < par["mean"] < 5e-3 and par["oracle_bf16_vs_fp32"]["mean"] > 0
    full = c.get("full_depth_measured")
    if full is not None:
        assert full["extrapolated"] is False and full["value"] > 0 and full["cores"] >= 1 and full["source"].endswith("_cpu_baseline_full.json")
    ev = r.get("event_profiling")
    if ev is not None:
        assert ev["ms_per_step_same_pool_without_events"] > 0


def test_bench_defaults():
    src = open(os.path.join(REPO, "bench.py")).read()
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            name = node.args[0].value if isinstance(node.args[0], ast.Constant) else None
            for kw in node.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[name] = kw.value.value
    assert defaults["--gpus"] == 1 and 1 <= defaults["--steps"] <= 10 and 0 <= defaults["--warmup"] <= 3
    # nothing at run time may read the reference tree
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(REPO, f)).read()
