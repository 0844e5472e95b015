"""CPU: `python bench.py --gpus N` starts its N ranks ITSELF when no launcher environment is present (the driver calls it exactly like
that; /root/reference's run/train_opa_dpo.sh:96-100 uses torchrun for the same purpose).  `--dry-run` runs everything of the N > 1
path that is not a kernel - the re-exec through torch.distributed.run, the rendezvous on 127.0.0.1, the max-over-ranks timing, the
`dist` record and the bucketed ZeRO-1 exchange of optim.FlatAdamW - on gloo, so the spawn logic is exercised without a GPU node."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _run(*argv, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=timeout, env=env or _env(), cwd=REPO)


@pytest.mark.parametrize("mode", ["zero1", "allreduce"])
def test_self_launch_two_ranks_dry_run(mode):
    r = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--model", "tiny", "--steps", "2", "--optimizer-mode", mode)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "launching 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert r.stdout.strip().splitlines()[-1] == lines[0]                # and it is the last thing on stdout
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "dist"):
        assert k in d, k
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 2 and d["steps"] == 2
    ds = d["dist"]
    assert ds["backend"] == "gloo" and ds["world_size"] == 2 and ds["allreduce_of_ones"] == 2.0
    assert [x["rank"] for x in ds["ranks"]] == [0, 1] and [x["local_rank"] for x in ds["ranks"]] == [0, 1]
    assert ds["replicas_identical_after_step"] is True and ds["self_check"] is True
    assert ds["max_abs_diff_vs_1_rank_step_on_averaged_gradient"] <= 2.0 ** -9
    assert 0 < ds["ms_per_step_min_over_ranks"] <= ds["ms_per_step_max_over_ranks"] == d["ms_per_step"]


def test_one_rank_dry_run_needs_no_launcher():
    r = _run("--gpus", "1", "--dry-run", "--backend", "gloo", "--model", "tiny", "--steps", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "launching" not in r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["dist"]["world_size"] == 1 and d["dist"]["self_check"] is True


def test_gpus_without_devices_fails_loudly_not_on_an_assert():
    """No GPU here: a real (non-dry) N = 2 run has nothing to launch - it must say so and return 2, not die on an assertion."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box could really launch it")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", timeout=120)
    assert r.returncode == 2 and "nothing to launch" in r.stderr and "AssertionError" not in r.stderr


def test_world_size_mismatch_is_reported():
    env = _env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = _run("--gpus", "2", "--dry-run", env=env, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_a_dying_rank_ends_the_launch_quickly_with_a_reason():
    """The first node run will be unattended: when one rank raises (here: right after the rendezvous, while rank 0 already sits in the
    pre-flight all-reduce) the launcher must come back non-zero with the reason on stderr - within a minute, not after a collective timeout."""
    import time
    env = _env()
    env["OPADPO_BENCH_FAIL_RANK"] = "1"
    env["OPADPO_DIST_TIMEOUT_S"] = "40"
    t0 = time.time()
    r = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--model", "tiny", "--steps", "1", env=env, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 60, time.time() - t0
    assert "FAILED on rank 1 of 2" in r.stderr and "injected failure" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no bench line from a failed run


def test_dry_run_records_what_each_rank_is_bound_to():
    r = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--model", "tiny", "--steps", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    ds = json.loads(r.stdout.strip().splitlines()[-1])["dist"]
    assert len({x["pid"] for x in ds["ranks"]}) == 2 and all("device" in x for x in ds["ranks"])
    assert "collective_library" in ds


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_dist_record_schema_is_the_same_for_dry_and_real_runs():
    """Round 6: an N > 1 line cannot come back without `allreduce_of_ones` / what each rank was bound to.  bench.check_dist_record is
    called on rank 0 before EITHER line is printed; here: the dry run's record passes it, and records a SCALE line must never carry are refused."""
    import copy
    B = _load_bench()
    r = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--model", "tiny", "--steps", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    ds = json.loads(r.stdout.strip().splitlines()[-1])["dist"]
    for k in B.DIST_FIELDS:
        assert k in ds, k
    assert B.check_dist_record(ds, 2, False)
    gpu = copy.deepcopy(ds)                     # the same record as a GPU run would print it: per-rank device UUIDs are mandatory there
    with pytest.raises(ValueError, match="uuid"):
        B.check_dist_record(gpu, 2, True)
    for i, x in enumerate(gpu["ranks"]):
        x.update(uuid=f"GPU-{i}", arch="gfx950", compute_units=256, hbm_GB=288.0)
    assert B.check_dist_record(gpu, 2, True)
    same = copy.deepcopy(gpu)
    same["ranks"][1]["uuid"] = same["ranks"][0]["uuid"]
    with pytest.raises(ValueError, match="distinct device UUID"):
        B.check_dist_record(same, 2, True)
    for drop in ("allreduce_of_ones", "ranks", "collective_library"):
        bad = {k: v for k, v in gpu.items() if k != drop}
        with pytest.raises(ValueError):
            B.check_dist_record(bad, 2, True)
    short = copy.deepcopy(gpu)
    short["allreduce_of_ones"] = 1.0
    with pytest.raises(ValueError):
        B.check_dist_record(short, 2, True)
