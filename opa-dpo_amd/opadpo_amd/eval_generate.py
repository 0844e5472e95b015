"""Evaluation-time generation from a trained checkpoint.

Tensor-level counterpart of eval_llava_rlhf_coco/model_vqa.py:77-120,213-226 (`model.generate(input_ids, images=...,
do_sample = temperature > 0, temperature, max_new_tokens, use_cache=True)` on base + `PeftModel.from_pretrained(model,
<ckpt>/adapter_model/lora_policy)`): the prompt templating / tokenizer / keyword stopping of that script are host-side text
processing outside the kernel path; this module consumes token ids and returns token ids, re-using the rollout kernels
(prefill + graph-replayed KV-cache decode, LoRA tail fused into the decode GEMMs).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .checkpoint_io import load_adapter
from .dims import EOS_ID, PAD_ID
from .generate import Generator, truncate_after_eos_with_padding
from .model import LlavaEngine, LoraAdapter


def adapter_dir_of(checkpoint: str) -> str:
    """`checkpoint-N/adapter_model/lora_policy` (this stage, dpo_trainer.py:1047-1095) or a bare PEFT directory (OPA stage)."""
    cand = os.path.join(checkpoint, "adapter_model", "lora_policy")
    return cand if os.path.isdir(cand) else checkpoint


@torch.no_grad()
def generate_from_checkpoint(engine: LlavaEngine, checkpoint: Optional[str], queries: torch.Tensor, query_attn_masks: torch.Tensor,
                             images: torch.Tensor, *, max_new_tokens: int = 64, temperature: float = 0.0, top_p: float = 1.0,
                             top_k: int = 0, seed: int = 0, adapter: Optional[LoraAdapter] = None) -> torch.Tensor:
    """-> [B, max_new_tokens] int64, pad after EOS.  temperature == 0 -> greedy (the reference passes do_sample=False then)."""
    if adapter is None and checkpoint is not None:
        adapter = LoraAdapter(engine.d, load_adapter(adapter_dir_of(checkpoint)), engine.dev, trainable=False)
    gen = Generator(engine, adapter)
    if temperature and temperature > 0:
        out = gen.generate(queries, query_attn_masks, images, max_new_tokens=max_new_tokens, temperature=temperature, top_k=top_k,
                           top_p=top_p, seed=seed)
    else:
        out = gen.generate(queries, query_attn_masks, images, max_new_tokens=max_new_tokens, temperature=1.0, top_k=1, top_p=1.0, seed=seed)
    return truncate_after_eos_with_padding(out, EOS_ID, PAD_ID)
