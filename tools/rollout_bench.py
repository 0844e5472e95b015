#!/usr/bin/env python3
"""Rollout (config 5 of BASELINE.json) micro-benchmark on one MI355X: LLaVA-1.5-7B prefill (Q=128 -> L=703) + KV-cache
sampling decode (top-k 30, top-p 0.95), random-init weights, no LoRA (the shipped rollout config; RB_LORA=1 / 2 adds a frozen
adapter unmerged / merged, RB_FUSE=0 switches the fused SwiGLU gate|up projection off, RB_BATCH = sequences per device, shipped: 4).
Reports prefill ms, decode ms/step and tokens/s; the decode roofline is HBM: >= 13.2 GB of bf16 weights per step."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L  # noqa: E402
from opadpo_amd.dims import LlavaDims  # noqa: E402
from opadpo_amd.generate import Generator  # noqa: E402
from opadpo_amd.ctx import CtxEngine  # noqa: E402
from opadpo_amd.model import BaseWeights, LlavaEngine  # noqa: E402
from opadpo_amd.synth import init_weights, synth_pairs  # noqa: E402


def main():
    L.load()
    dev = torch.device("cuda:0")
    model = os.environ.get("RB_MODEL", "7b")
    d = LlavaDims.llava15_7b() if model == "7b" else LlavaDims.tiny()
    B = int(os.environ.get("RB_BATCH", 8))
    steps = int(os.environ.get("RB_STEPS", 48))
    Q = 128 if model == "7b" else 16
    base = BaseWeights(d, init_weights(d, seed=0, device=dev), dev, need_backward=False)
    eng = LlavaEngine(base) if os.environ.get("RB_OP_LEVEL") == "1" else CtxEngine(base)      # default: the context path (opadpo_decode_*)
    p = synth_pairs(d, B, Q, 8, seed=0, device=dev)
    lora = int(os.environ.get("RB_LORA", 0))       # 0: no adapter (the shipped rollout config), 1: frozen adapter, 2: frozen adapter merged
    ad = None
    if lora:
        from opadpo_amd.model import LoraAdapter
        from opadpo_amd.synth import init_lora
        ad = LoraAdapter(d, init_lora(d, seed=2, device=dev), dev, trainable=False)
    gen = Generator(eng, ad, merge_adapter=lora == 2, fuse_swiglu=os.environ.get("RB_FUSE", "1") == "1", use_graph={"0": False, "1": True}.get(os.environ.get("RB_GRAPH", ""), None))
    feats = eng.encode_images(p["images"])
    res = {}
    for n in (1, steps):
        gen.generate(p["queries"], p["queries_attn_masks"], image_feats=feats, max_new_tokens=n, top_k=30, top_p=0.95, seed=1)   # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen.generate(p["queries"], p["queries_attn_masks"], image_feats=feats, max_new_tokens=n, top_k=30, top_p=0.95, seed=2,
                           suppress_eos=True)
        torch.cuda.synchronize()
        res[n] = time.perf_counter() - t0
    prefill = res[1]
    per_step = (res[steps] - res[1]) / (steps - 1)
    wbytes = 2 * (d.n_layers * (4 * d.hidden ** 2 + 3 * d.hidden * d.ffn) + d.vocab * d.hidden)
    out = {"model": model, "batch": B, "adapter": ["none", "lora", "lora merged"][lora], "fused_swiglu": gen.adapter is not None and getattr(gen.adapter, "merged", None) is not None, "prefill_ms": prefill * 1e3, "decode_ms_per_step": per_step * 1e3,
           "decode_tokens_per_s": B / per_step, "weight_bytes_per_step_GB": wbytes / 1e9,
           "decode_hbm_frac": wbytes / per_step / 8e12}
    print(json.dumps(out))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", f"rollout_bench_b{B}_l{lora}.json"), "w"))


if __name__ == "__main__":
    main()
