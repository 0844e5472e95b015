"""Seeded synthetic weights and batches: there is no network for checkpoints or
datasets, so benchmarks and smoke tests run LLaVA-1.5-shaped random-init models on synthetic
(image, query, chosen, rejected) batches."""
from __future__ import annotations

import math
from typing import Dict

import torch

from .dims import (IMAGE_TOKEN_INDEX, LLM_LINEARS, LLM_PREFIX, PEFT_PREFIX, VIS_LINEARS, VIS_PREFIX, LlavaDims,
                   llm_linear_shape, vis_linear_shape)


def init_weights(d: LlavaDims, seed: int = 0, std: float = 0.02, device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Random base weights ~ N(0, std) with the LLaVA-1.5 HF key names (norm gains ~ 1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}

    def n(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    def gain(k):
        return (1.0 + torch.randn(k, generator=g, device=device, dtype=torch.float32) * 0.1).to(dtype)

    W[LLM_PREFIX + "embed_tokens.weight"] = n(d.vocab, d.hidden)
    for i in range(d.n_layers):
        p = f"{LLM_PREFIX}layers.{i}."
        for lin in LLM_LINEARS:
            W[p + lin + ".weight"] = n(*llm_linear_shape(d, lin))
        W[p + "input_layernorm.weight"] = gain(d.hidden)
        W[p + "post_attention_layernorm.weight"] = gain(d.hidden)
    W[LLM_PREFIX + "norm.weight"] = gain(d.hidden)
    W["lm_head.weight"] = n(d.vocab, d.hidden)
    W[LLM_PREFIX + "mm_projector.0.weight"] = n(d.hidden, d.v_hidden)
    W[LLM_PREFIX + "mm_projector.0.bias"] = n(d.hidden)
    W[LLM_PREFIX + "mm_projector.2.weight"] = n(d.hidden, d.hidden)
    W[LLM_PREFIX + "mm_projector.2.bias"] = n(d.hidden)
    v = VIS_PREFIX
    W[v + "embeddings.patch_embedding.weight"] = n(d.v_hidden, 3, d.patch, d.patch)
    W[v + "embeddings.class_embedding"] = n(d.v_hidden)
    W[v + "embeddings.position_embedding.weight"] = n(d.n_patches + 1, d.v_hidden)
    W[v + "pre_layrnorm.weight"] = gain(d.v_hidden)
    W[v + "pre_layrnorm.bias"] = n(d.v_hidden)
    for j in range(d.v_layers):
        p = f"{v}encoder.layers.{j}."
        for ln in ("layer_norm1", "layer_norm2"):
            W[p + ln + ".weight"] = gain(d.v_hidden)
            W[p + ln + ".bias"] = n(d.v_hidden)
        for lin in VIS_LINEARS:
            o, i_ = vis_linear_shape(d, lin)
            W[p + lin + ".weight"] = n(o, i_)
            W[p + lin + ".bias"] = n(o)
    return W


def init_lora(d: LlavaDims, seed: int = 1, b_std: float = 0.01, device="cpu", dtype=torch.bfloat16,
              with_vision: bool = False) -> Dict[str, torch.Tensor]:
    """LoRA adapter in the PEFT key layout: A ~ kaiming_uniform(a=sqrt 5), B ~ N(0, b_std) (non-zero so
    that the LoRA branch is exercised)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    r = d.lora_r

    def pair(key, out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        out[key + ".lora_A.weight"] = ((torch.rand(r, in_f, generator=g, device=device) * 2 - 1) * bound).to(dtype)
        out[key + ".lora_B.weight"] = (torch.randn(out_f, r, generator=g, device=device) * b_std).to(dtype)

    for i in range(d.n_layers):
        for lin in LLM_LINEARS:
            pair(f"{PEFT_PREFIX}{LLM_PREFIX}layers.{i}.{lin}", *llm_linear_shape(d, lin))
    if with_vision:
        for j in range(d.v_layers):
            for lin in VIS_LINEARS:
                pair(f"{PEFT_PREFIX}{VIS_PREFIX}encoder.layers.{j}.{lin}", *vis_linear_shape(d, lin))
        pair(f"{PEFT_PREFIX}{LLM_PREFIX}mm_projector.0", d.hidden, d.v_hidden)
        pair(f"{PEFT_PREFIX}{LLM_PREFIX}mm_projector.2", d.hidden, d.hidden)
    return out


def synth_pairs(d: LlavaDims, n_pairs: int, q_len: int, t_len: int, seed: int = 0, device="cpu", dense: bool = False) -> Dict[str, torch.Tensor]:
    """(image, query, chosen, rejected) batch per SURVEY.md §8d: pixels ~ N(0,1); queries left-padded with
    0 for n_pad ~ U{0..q_len/2}, ids ~ U{3..V-1}, one IMAGE_TOKEN_INDEX at a random non-pad slot; responses:
    ids for len ~ U{t_len/6 .. t_len-1}, then EOS 2, then pad 0.  dense: no padding anywhere (full-length queries, responses of
    t_len - 1 ids + EOS) - the seq512 shape with nothing for the ragged layout to drop.  The returned dict also carries the
    batch's ragged-row plan as HOST tensors (`row_lead`, `row_lens`: policy.host_row_plan)."""
    g = torch.Generator().manual_seed(seed)
    pixels = torch.randn(n_pairs, 3, d.image_size, d.image_size, generator=g).to(torch.bfloat16)
    queries = torch.randint(3, d.vocab, (n_pairs, q_len), generator=g)
    qmask = torch.ones(n_pairs, q_len, dtype=torch.bool)
    for b in range(n_pairs):
        n_pad = int(torch.randint(0, q_len // 2 + 1, (1,), generator=g))
        if dense:
            n_pad = 0
        queries[b, :n_pad] = 0
        qmask[b, :n_pad] = False
        slot = int(torch.randint(n_pad, q_len, (1,), generator=g))
        queries[b, slot] = IMAGE_TOKEN_INDEX

    def resp():
        ids = torch.randint(3, d.vocab, (n_pairs, t_len), generator=g)
        for b in range(n_pairs):
            ln = int(torch.randint(max(1, t_len // 6), t_len, (1,), generator=g))
            if dense:
                ln = t_len - 1
            ids[b, ln] = 2
            ids[b, ln + 1:] = 0
        return ids

    out = dict(images=pixels, queries=queries, queries_attn_masks=qmask, chosen=resp(), rejected=resp())
    # the ragged-row plan of the batch, from the host tensors (what a collator hands over with the batch): HOST int32
    from .policy import host_row_plan
    lead, lens = host_row_plan(queries, qmask, {"chosen_response": out["chosen"], "rejected_response": out["rejected"]})
    dev_out = {k: v.to(device) for k, v in out.items()}
    dev_out["row_lead"], dev_out["row_lens"] = lead, lens
    return dev_out


def synth_rollout_batches(d: LlavaDims, args, seed: int = 0):
    """Endless stream of collated batches with the keys DataCollatorForCausalLM produces (detailed_report=True)."""
    step = 0
    B, Q, T = args.rollout_per_device_batch_size, args.query_len, args.response_len
    while True:
        p = synth_pairs(d, B, Q, T, seed=seed * 100003 + step)
        g = torch.Generator().manual_seed(seed * 7 + step)
        third = synth_pairs(d, B, Q, T, seed=seed * 100003 + step + 50021)["chosen"]
        batch = {"images": p["images"].float(), "queries": p["queries"], "queries_attention_mask": p["queries_attn_masks"],
                 "standard_response": p["chosen"], "original_generate_response": p["rejected"], "AI_pseudo_response": third}
        choices = torch.tensor([1.0, 1.5, 2.0, 2.5])
        for k in ("original_generate_response", "AI_pseudo_response"):
            m = batch[k] != 0
            batch[k + "_scores"] = choices[torch.randint(0, 4, (B, T), generator=g)] * m
            batch[k + "_image_relations"] = torch.tensor([1.0, 3.0])[torch.randint(0, 2, (B, T), generator=g)] * m
        yield batch
        step += 1


class SyntheticTokenizer:
    """Whitespace word-hash tokenizer for the --synthetic launch modes (no sentencepiece model offline): BOS 1, EOS 2, pad 0, words
    hashed into [3, vocab).  `batch_decode` renders ids as 'w<id>' words; special tokens are dropped on request."""
    pad_token_id, eos_token_id, bos_token_id = 0, 2, 1

    def __init__(self, vocab: int):
        self.vocab = int(vocab)

    def _ids(self, text: str):
        return [self.bos_token_id] + [3 + sum(ord(c) * (i + 1) for i, c in enumerate(w)) % (self.vocab - 3) for w in text.split()]

    def __call__(self, text, **_):
        if isinstance(text, str):
            return {"input_ids": self._ids(text)}
        return {"input_ids": [self._ids(t) for t in text]}

    def batch_decode(self, ids, skip_special_tokens: bool = True, clean_up_tokenization_spaces: bool = True):
        keep = (lambda t: t > 2) if skip_special_tokens else (lambda t: True)
        return [" ".join(f"w{int(t)}" for t in row if keep(int(t))) for row in ids]


def synth_question_rows(n: int, seed: int = 0):
    """n rows in the RLAIF-V layout the rollout stage reads ('question', 'chosen', 'image' = {'bytes', 'path'}) with flat-colour PNGs."""
    import io
    from PIL import Image
    g = torch.Generator().manual_seed(seed)
    rows = []
    for i in range(n):
        col = tuple(int(c) for c in torch.randint(0, 256, (3,), generator=g))
        buf = io.BytesIO()
        Image.new("RGB", (int(torch.randint(8, 40, (1,), generator=g)), int(torch.randint(8, 40, (1,), generator=g))), col).save(buf, format="PNG")
        nq = int(torch.randint(3, 9, (1,), generator=g))
        rows.append({"question": " ".join(f"q{i}w{j}" for j in range(nq)) + " ?", "chosen": " ".join(f"a{i}w{j}" for j in range(4)),
                     "image": {"bytes": buf.getvalue(), "path": f"synthetic_{i}.png"}})
    return rows
