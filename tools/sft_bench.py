#!/usr/bin/env python3
"""OPA LoRA-SFT step on one MI355X: LLaVA-1.5-7B, LoRA r=256 on the LLM, the CLIP tower and the
projector, query 128 + response 384 (L = 1087), synthetic batches, random-init weights.  samples/s of SFTTrainer.step()."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L  # noqa: E402
from opadpo_amd.dims import LlavaDims  # noqa: E402
from opadpo_amd.ctx import CtxEngine
from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter  # noqa: E402
from opadpo_amd.sft import SFTTrainer  # noqa: E402
from opadpo_amd.synth import init_lora, init_weights, synth_pairs  # noqa: E402
from opadpo_amd.vision_train import VisionLoraAdapter  # noqa: E402


def main():
    L.load()
    dev = torch.device("cuda:0")
    B = int(os.environ.get("SB_BATCH", 24))
    steps = int(os.environ.get("SB_STEPS", 3))
    d = LlavaDims.llava15_7b()
    base = BaseWeights(d, init_weights(d, seed=0, device=dev), dev, need_backward=True)
    eng = LlavaEngine(base) if os.environ.get("OPADPO_OP_LEVEL") == "1" else CtxEngine(base)      # default: the product path (opadpo_ctx, ragged rows)
    lora = init_lora(d, seed=1, device=dev, with_vision=True)
    tr = SFTTrainer(eng, LoraAdapter(d, lora, dev, trainable=True), VisionLoraAdapter(d, lora, dev), response_len=384, lr=1e-6)
    p = synth_pairs(d, B, 128, 384, seed=3, device=dev)
    batch = dict(images=p["images"], queries=p["queries"], queries_attn_masks=p["queries_attn_masks"], responses=p["chosen"])
    tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st = tr.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"metric": "OPA LoRA-SFT samples/s, LLaVA-1.5-7B, LoRA r256 on LLM + CLIP + projector, seq512, 1x MI355X", "value": B / dt,
           "unit": "samples/s", "samples_per_step": B, "ms_per_step": dt * 1e3, "loss": st["loss"], "grad_norm": st["grad_norm"],
           "hbm_peak_allocated_GB": torch.cuda.max_memory_allocated() / 1e9}
    print(json.dumps(out))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "sft_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
