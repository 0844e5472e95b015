#!/usr/bin/env python3
"""The reference's NATIVE unit: OPA-DPO samples/s on one MI355X — per sample 8 no-grad reference sequence
forwards (3 responses on the clean image + 2 on the CoPO-masked image... rollout()) and 5 policy sequence forwards with
grad (3 clean + 2 masked), token-level DPO + CoPO + AncPO loss, LoRA backward, clipped AdamW.  LLaVA-1.5-7B, query 128 +
response 384 (seq512), synthetic batches (synth.synth_rollout_batches), random-init weights.  Runs DPOTrainer.step()."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L  # noqa: E402
from opadpo_amd.dims import LlavaDims  # noqa: E402
from opadpo_amd.ctx import CtxEngine
from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter  # noqa: E402
from opadpo_amd.policy import AutoregressivePolicy  # noqa: E402
from opadpo_amd.synth import init_lora, init_weights, synth_rollout_batches  # noqa: E402
from opadpo_amd.trainer import DPOTrainer  # noqa: E402


def main():
    L.load()
    dev = torch.device("cuda:0")
    B = int(os.environ.get("SB_BATCH", 8))            # samples per micro-batch (the reference runs 2 on 80-GB parts)
    steps = int(os.environ.get("SB_STEPS", 3))
    pack = os.environ.get("SB_PACK", "1") == "1"
    Q, T = 128, int(os.environ.get("SB_T", 384))      # SB_T=896: the shipped recipe's response_len (run/train_opa_dpo.sh)
    d = LlavaDims.llava15_7b()
    base = BaseWeights(d, init_weights(d, seed=0, device=dev), dev, need_backward=True)
    eng = LlavaEngine(base) if os.environ.get("OPADPO_OP_LEVEL") == "1" else CtxEngine(base)      # default: the product path (opadpo_ctx, ragged rows)
    pol = LoraAdapter(d, init_lora(d, seed=1, device=dev), dev, trainable=True)
    ref = LoraAdapter(d, init_lora(d, seed=2, device=dev), dev, trainable=False)
    if os.environ.get("SB_MERGE_REF", "1") == "1":
        ref.merge_into_base(eng.base)
    args = SimpleNamespace(rollout_accumulation_steps=1, gradient_accumulation_steps=1, step_per_device_batch_size=B,
                           rollout_per_device_batch_size=B, rollout_batch_size=B, noptepochs=1, max_grad_norm=1.0,
                           learning_rate=1e-6, warmup_steps=0, total_epochs=1, max_step=1000, save_steps=10 ** 9,
                           output_dir="/tmp/none", seed=0, weight_decay=0.0, CoPO=True, AncPO=True, temperature=1.0,
                           query_len=Q, response_len=T)
    tr = DPOTrainer(args, AutoregressivePolicy(eng, pol, T, pack_responses=pack), AutoregressivePolicy(eng, ref, T, pack_responses=pack))
    tr.total_sched_steps = 1000
    it = iter(synth_rollout_batches(d, args, seed=0))
    tr.step(it, 0)                                       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        stats = tr.step(it, i + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"metric": f"OPA-DPO samples/s (3 responses + CoPO masked image + AncPO), LLaVA-1.5-7B LoRA r256, query {Q} + response {T}, 1x MI355X",
           "value": B / dt, "unit": "samples/s", "samples_per_step": B, "ms_per_step": dt * 1e3, "steps": steps, "query_len": Q, "response_len": T,
           "response_layout": "packed on the shared prefix" if pack else "stacked (reference layout)",
           "sequence_forwards_per_sample": {"reference_no_grad": 5, "policy_with_grad": 5,
                                            "note": "the reference additionally runs a discarded 3-sequence policy forward inside rollout() (Quirk Q2)"},
           "loss": next((float(v) for k, v in stats.items() if k.startswith("dpo/loss") and "grad" not in k), None),
           "grad_norm_post_clip": stats.get("dpo/loss-grad_norm"),
           "hbm_peak_allocated_GB": torch.cuda.max_memory_allocated() / 1e9}
    ms = torch.cuda.memory_stats()
    out["allocator"] = {k: int(ms.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")}
    print(json.dumps(out))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", f"sample_bench_{'packed' if pack else 'stacked'}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
