#!/usr/bin/env python3
"""Independent pins for the three pieces of oracle/llava_ref.py that round 3 still checked only against themselves (run in the BUILD
container; nothing on the GPU box or in the product path depends on this script):

* hf_llama_lora.npz  - the installed transformers LlamaForCausalLM (fp32, eager attention) loaded with the PEFT-MERGED weights
  W + (alpha / r) * B @ A, computed HERE with plain torch (not with oracle.merge_llm_lora).  PEFT's lora.Linear computes
  x W^T + (alpha / r) (x A^T) B^T; in fp32 that is the merged model's function, so the oracle's UNMERGED LoRA path
  (_Ctx.linear with a lora dict) must reproduce these logits.
* hf_clip_lora.npz   - the same for the CLIP tower (every q/k/v/out_proj/fc1/fc2 Linear carries a LoRA pair in the OPA adapter).
* nn_projector.npz   - mlp2x_gelu as upstream LLaVA builds it: nn.Sequential(Linear(v_hidden, H), GELU(), Linear(H, H)) - plain, and with
  the LoRA pairs merged into its two Linears.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle import llava_ref as LR  # noqa: E402  (dims, key names and the seeded initialisers only)


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote", name)


def tiny_dims():
    return LR.LlavaDims.tiny(hidden=64, n_layers=2, n_heads=2, head_dim=32, ffn=96, vocab=80, v_hidden=32, v_layers=3, v_heads=2, v_ffn=64,
                             image_size=28, lora_r=8, lora_alpha=16.0)


def merged(W, lora, key, scale):
    a = lora[LR.PEFT_PREFIX + key + ".lora_A.weight"].double()
    b = lora[LR.PEFT_PREFIX + key + ".lora_B.weight"].double()
    return (W[key + ".weight"].double() + scale * (b @ a)).float()


def main():
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM
    d = tiny_dims()
    seed, std, lseed, bstd = 3, 0.2, 5, 0.3            # B ~ N(0, 0.3): the adapter moves the logits by O(1), far above the tolerance
    W = LR.init_weights(d, seed=seed, std=std)
    lora = LR.init_lora(d, seed=lseed, b_std=bstd, with_vision=True)
    s = d.lora_alpha / d.lora_r
    g = torch.Generator().manual_seed(123)

    # ---- Llama with merged LLM LoRA
    Wm = dict(W)
    for i in range(d.n_layers):
        for lin in LR.LLM_LINEARS:
            key = f"{LR.LLM_PREFIX}layers.{i}.{lin}"
            Wm[key + ".weight"] = merged(W, lora, key, s)
    cfg = LlamaConfig(vocab_size=d.vocab, hidden_size=d.hidden, intermediate_size=d.ffn, num_hidden_layers=d.n_layers,
                      num_attention_heads=d.n_heads, num_key_value_heads=d.n_heads, head_dim=d.head_dim, rms_norm_eps=d.rms_eps,
                      rope_theta=d.rope_theta, max_position_embeddings=128, attn_implementation="eager", tie_word_embeddings=False)
    lm = LlamaForCausalLM(cfg).eval()
    missing = lm.load_state_dict({k: v for k, v in Wm.items() if k.startswith("model.layers") or k in
                                  ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")}, strict=False)
    assert not [m for m in missing.missing_keys if "rotary" not in m], missing
    S, Ltxt = 3, 14
    ids = torch.randint(3, d.vocab, (S, Ltxt), generator=g)
    mask = torch.ones(S, Ltxt, dtype=torch.bool)
    ids[0, :4] = 0
    mask[0, :4] = False
    ids[2, -3:] = 0
    mask[2, -3:] = False
    with torch.no_grad():
        logits = lm(input_ids=ids, attention_mask=mask.long()).logits
        lm.load_state_dict({k: v for k, v in W.items() if k.startswith("model.layers")}, strict=False)
        logits_base = lm(input_ids=ids, attention_mask=mask.long()).logits
    save("hf_llama_lora.npz", ids=ids, mask=mask.to(torch.uint8), logits=logits, logits_without_adapter=logits_base,
         seed=seed, std=std, lora_seed=lseed, lora_b_std=bstd)

    # ---- CLIP with merged vision LoRA
    vcfg = CLIPVisionConfig(hidden_size=d.v_hidden, intermediate_size=d.v_ffn, num_hidden_layers=d.v_layers, num_attention_heads=d.v_heads,
                            image_size=d.image_size, patch_size=d.patch, hidden_act="quick_gelu", layer_norm_eps=d.v_eps,
                            attn_implementation="eager")
    vm = CLIPVisionModel(vcfg).eval()
    Wv = dict(W)
    for j in range(d.v_layers):
        for lin in LR.VIS_LINEARS:
            key = f"{LR.VIS_PREFIX}encoder.layers.{j}.{lin}"
            Wv[key + ".weight"] = merged(W, lora, key, s)
    vsd = {k[len(LR.VIS_PREFIX):]: v for k, v in Wv.items() if k.startswith(LR.VIS_PREFIX)}
    missing = vm.load_state_dict(vsd, strict=False)
    assert not [m for m in missing.missing_keys if "post_layernorm" not in m and "position_ids" not in m], missing
    pix = torch.randn(2, 3, d.image_size, d.image_size, generator=g)
    with torch.no_grad():
        hs = vm(pixel_values=pix, output_hidden_states=True).hidden_states
    save("hf_clip_lora.npz", pixels=pix, feats=hs[-2][:, 1:], seed=seed, std=std, lora_seed=lseed, lora_b_std=bstd)

    # ---- mlp2x_gelu
    proj = torch.nn.Sequential(torch.nn.Linear(d.v_hidden, d.hidden), torch.nn.GELU(), torch.nn.Linear(d.hidden, d.hidden)).eval()
    x = torch.randn(2, d.n_patches, d.v_hidden, generator=g)
    outs = {}
    for tag, src in (("plain", W), ("lora", None)):
        sd = {}
        for idx in (0, 2):
            key = f"{LR.LLM_PREFIX}mm_projector.{idx}"
            sd[f"{idx}.weight"] = W[key + ".weight"] if src is not None else merged(W, lora, key, s)
            sd[f"{idx}.bias"] = W[key + ".bias"]
        proj.load_state_dict(sd)
        with torch.no_grad():
            outs[tag] = proj(x)
    save("nn_projector.npz", x=x, plain=outs["plain"], lora=outs["lora"], seed=seed, std=std, lora_seed=lseed, lora_b_std=bstd)


if __name__ == "__main__":
    main()
