#!/usr/bin/env python3
"""End-to-end rate of the rollout launcher (cli_generate.run_rollout) on a synthetic 7B model: dataset -> queries -> prefill +
graph-replayed decode -> decoded columns -> step JSON files.  Generated tokens per second per rollout step INCLUDING every
host-side piece (image preprocessing, graph capture per batch, decode to text, JSON); the first step (model build) is dropped."""
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import cli_generate as cg  # noqa: E402


def main():
    per_step = int(os.environ.get("RC_STEP_ROWS", 8))
    T = int(os.environ.get("RC_RESPONSE_LEN", 256))
    B = int(os.environ.get("RC_BATCH", 4))
    steps = 3
    ns, _ = cg.rollout_parser().parse_known_args(
        ["--synthetic", "7b", "--synthetic_rows", str(per_step * steps), "--output_dir", tempfile.mkdtemp(), "--rollout_batch_size", str(per_step),
         "--rollout_per_device_batch_size", str(B), "--response_len", str(T), "--query_len", "128"])
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    stamps = []

    def log(msg):
        if str(msg).startswith("step "):
            stamps.append(time.perf_counter())
    cg.run_rollout(ns, log=log)
    dts = [b - a for a, b in zip(stamps, stamps[1:])]
    res = {"batch": B, "response_len": T, "rows_per_step": per_step, "step_s": dts, "tokens_per_s": per_step * T / min(dts)}
    print(json.dumps(res))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "gpurun_out", "rollout_cli_bench.json"), "w"))


if __name__ == "__main__":
    main()
