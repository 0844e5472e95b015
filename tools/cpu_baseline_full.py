#!/usr/bin/env python3
"""The CPU baseline at FULL depth, timed directly (no extrapolation): ONE preference pair of BASELINE.json's configuration - LLaVA-1.5-7B, 32 decoder
layers, CLIP-L/14-336, query 128 + response 384 (L = 1087), LoRA r = 256 - through the oracle (the parity-checked CPU restatement of the reference's
forward: vision tower + projector once per image, frozen-reference forward on chosen + rejected without grad, policy forward on both with autograd,
token-level DPO loss, backward into the LoRA tensors) on the host cores of the GPU box, fp32.  ~3-5 minutes; bench.py's default `cpu_baseline`
extrapolates from one layer instead and quotes the record this script writes (gpurun_out/cpu_baseline_full.json -> profiles/r*_cpu_baseline_full.json)."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import dpo_ref as DR  # noqa: E402
from oracle import llava_ref as LR  # noqa: E402


def main():
    cores = min(os.cpu_count() or 8, 64)
    torch.set_num_threads(cores)
    d = LR.LlavaDims()
    t0 = time.time()
    W = LR.init_weights(d, seed=0)
    lora_p = {k: v.requires_grad_(True) for k, v in LR.init_lora(d, seed=1, with_vision=False).items()}
    lora_r = LR.init_lora(d, seed=2, with_vision=False)
    t_init = time.time() - t0
    g = torch.Generator().manual_seed(0)
    B, Q, T = 1, 128, 384
    images = torch.randn(B, 3, d.image_size, d.image_size, generator=g)
    queries = torch.randint(3, d.vocab, (B, Q), generator=g)
    qmask = torch.ones(B, Q, dtype=torch.bool)
    n_pad = 17
    queries[0, :n_pad] = 0
    qmask[0, :n_pad] = False
    queries[0, 40] = -200
    resp = {}
    for k, ln in (("chosen_response", 301), ("rejected_response", 187)):
        ids = torch.randint(3, d.vocab, (B, T), generator=g)
        ids[0, ln] = 2
        ids[0, ln + 1:] = 0
        resp[k] = ids
    from oracle.dpo_ref import policy_head, stack_policy_inputs

    def fwd(lora, feats):
        ids, mask = stack_policy_inputs(queries, qmask, resp)
        lp, _ = policy_head(LR.llava_logits(ids, mask, None, W, lora, d, feats=feats.repeat(2, 1, 1)), ids, Q, T, 1.0)
        return {"chosen_response_logprobs": lp[:B], "rejected_response_logprobs": lp[B:]}
    stamps = {}
    t0 = time.time()
    with torch.no_grad():
        feats = LR.image_features(images, W, None, d)
        stamps["vision_s"] = time.time() - t0
        r = fwd(lora_r, feats)
    stamps["reference_forward_s"] = time.time() - t0 - stamps["vision_s"]
    t1 = time.time()
    o = fwd(lora_p, feats)
    stamps["policy_forward_s"] = time.time() - t1
    t2 = time.time()
    loss, _, _ = DR.plain_pair_loss(DR.DPOConfig(), o["chosen_response_logprobs"], o["rejected_response_logprobs"],
                                    r["chosen_response_logprobs"], r["rejected_response_logprobs"])
    loss.backward()
    stamps["loss_and_backward_s"] = time.time() - t2
    dt = time.time() - t0
    out = {"value": 1.0 / dt, "unit": "pairs/s", "seconds_per_pair": dt, "cores": cores, "kind": "port", "extrapolated": False,
           "sample": f"ONE full pair, all {d.n_layers} decoder layers + CLIP tower + head + DPO loss + LoRA backward, L = {Q + T + d.n_patches - 1}, fp32 torch CPU oracle",
           "loss": float(loss.detach()), "weight_init_s": t_init, **stamps}
    print(json.dumps(out))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "cpu_baseline_full.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
