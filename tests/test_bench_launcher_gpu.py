"""GPU: the REAL N > 1 code path of bench.py (self-launch through torch.distributed.run, one process per rank, the product's HIP
kernels, bucketed exchange launched from the backward's layer hook, sharded AdamW, max-over-ranks timing, `dist` record with the
exposed-exchange A/B) on the one MI355X a test box has: both ranks on cuda:0 (OPADPO_BENCH_SHARE_DEVICE=1) and gloo on the wire,
because RCCL refuses two ranks per device.  On an 8-GPU node the same command without the two switches runs on RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, share=False):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if share:
        env["OPADPO_BENCH_SHARE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]), r


@pytest.mark.parametrize("mode", ["zero1", "allreduce"])
def test_bench_two_ranks_self_launched(mode):
    d, r = _run("--gpus", "2", "--model", "tiny", "--pairs", "4", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                "--optimizer-mode", mode, share=True)
    assert "launching 2 ranks" in r.stderr
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_pairs_per_step"] == 8
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    ds = d["dist"]
    assert ds["world_size"] == 2 and ds["allreduce_of_ones"] == 2.0 and [x["rank"] for x in ds["ranks"]] == [0, 1]
    assert ds["ms_per_step_min_over_ranks"] <= ds["ms_per_step_max_over_ranks"] == pytest.approx(d["ms_per_step"])
    ex = ds["exposed_exchange"]
    assert "error" not in ex, ex
    assert ex["ms_per_step_with_collectives"] > 0 and ex["ms_per_step_without"] > 0
    assert d["roofline"] is not None and d["roofline"]["achieved"] > 0


def test_bench_one_rank_line_has_no_dist_record():
    d, r = _run("--gpus", "1", "--model", "tiny", "--pairs", "4", "--steps", "2", "--warmup", "1", "--no-side-legs")
    assert "launching" not in r.stderr and "dist" not in d and d["n_gpus"] == 1 and d["value"] > 0
