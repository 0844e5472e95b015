#!/usr/bin/env python3
"""CPU only (oracle/llava_ref.py): how far is a bf16 pipeline from fp32 arithmetic on the benchmarked model, and which rounding
points carry that distance?

    python tools/bf16_floor.py --layers 32 --pairs 1 --out profiles/r03_bf16_floor.json
    python tools/bf16_floor.py --layers 8 --pairs 1 --ablate --out profiles/r03_bf16_ablation.json

Evaluates the oracle on seeded random-init LLaVA-1.5-7B weights (bf16-rounded, like every test) for synthetic seq512 pairs:
  fp32      : plain fp32 arithmetic (the yardstick of north_star's 1e-3);
  hip_emu   : bf16 rounding at the HBM write points of the HIP pipeline (fp32 residual stream, fp32 logits);
  ref_hf    : the REFERENCE's own arithmetic - transformers 4.34.1 Llama in torch.bfloat16 + flash-attn + peft 0.5.0, restated from
              its published behaviour (oracle.llava_ref.HF_BF16): bf16 residual stream, bf16 rotary tables, bf16 logits;
  --ablate  : hip_emu with ONE rounding point left in fp32 at a time.
Reports per-token relative log-prob error (mean / p99 / max over valid response tokens) of each against fp32."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "opa-dpo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

from oracle import llava_ref as LR  # noqa: E402


def stats(got, want, valid):
    r = ((got - want).abs()[valid] / want.abs()[valid].clamp_min(1e-3)).double()
    return {"mean": float(r.mean()), "p99": float(torch.quantile(r, 0.99)), "max": float(r.max()),
            "mean_abs": float((got - want).abs()[valid].mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--tiny-vision", action="store_true", help="2-layer 128-wide tower on 336 px (the vision share of the drift is then ~0)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    from opadpo_amd.dims import LlavaDims
    from opadpo_amd.synth import synth_pairs
    kw = dict(n_layers=a.layers)
    if a.tiny_vision:
        kw.update(v_hidden=128, v_layers=2, v_heads=2, v_ffn=256)
    od = LR.LlavaDims(**kw)
    t0 = time.time()
    W = {k: v.to(torch.bfloat16).float() for k, v in LR.init_weights(od, seed=0).items()}
    lora = {k: v.to(torch.bfloat16).float() for k, v in LR.init_lora(od, seed=1, with_vision=False).items()}
    d = LlavaDims(**{f: getattr(od, f) for f in ("hidden", "n_layers", "n_heads", "head_dim", "ffn", "vocab", "v_hidden", "v_layers", "v_heads", "v_ffn",
                                                  "image_size", "patch", "lora_r", "lora_alpha")})
    p = synth_pairs(d, a.pairs, 128, 384, seed=21)
    images, queries, qmask = p["images"].float(), p["queries"], p["queries_attn_masks"]
    resp = {"chosen_response": p["chosen"], "rejected_response": p["rejected"]}
    print(f"weights {time.time() - t0:.0f}s", flush=True)

    def run(emu, hf=False, skip=()):
        LR.HF_BF16, LR.ROUND_SKIP = hf, frozenset(skip)
        try:
            t = time.time()
            with torch.no_grad():
                o = LR.policy_forward(images, queries, qmask, resp, W, lora, od, 1.0, emulate_bf16=emu)
            print(f"  pass emu={emu} hf={hf} skip={sorted(skip)}: {time.time() - t:.0f}s", flush=True)
            return o
        finally:
            LR.HF_BF16, LR.ROUND_SKIP = False, frozenset()

    rep = {"model": "LLaVA-1.5-7B width", "layers": a.layers, "pairs": a.pairs, "seq": "query 128 + response 384", "tiny_vision": a.tiny_vision}
    f32 = run(False)
    variants = {"hip_emu": run(True), "ref_hf_bf16": run(True, hf=True)}
    if a.ablate:
        for pt in ("vision", "n", "qkv", "rope", "p", "attn", "t", "gu", "act"):
            variants["hip_emu_without_" + pt] = run(True, skip=(pt,))
    for name, o in variants.items():
        worst = None
        for k in resp:
            st = stats(o[k + "_logprobs"], f32[k + "_logprobs"], resp[k] != 0)
            worst = st if worst is None else {f: max(worst[f], st[f]) for f in st}
        rep[name + "_vs_fp32"] = worst
        print(name, json.dumps(worst), flush=True)
    rep["seconds"] = time.time() - t0
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
