"""Evaluation-time generation from a trained checkpoint.

Tensor-level counterpart of eval_llava_rlhf_coco/model_vqa.py:77-120,213-226 (`model.generate(input_ids, images=...,
do_sample = temperature > 0, temperature, max_new_tokens, use_cache=True)` on base + `PeftModel.from_pretrained(model,
<ckpt>/adapter_model/lora_policy)`): the prompt templating / tokenizer / keyword stopping of that script are host-side text
processing outside the kernel path; this module consumes token ids and returns token ids, re-using the rollout kernels
(prefill + KV-cache decode launched from the context's own loop, LoRA tail fused into the decode GEMMs).
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
                             top_k: int = 0, seed: int = 0, adapter: Optional[LoraAdapter] = None,
                             merge_adapter: bool = False) -> torch.Tensor:
    """-> [B, max_new_tokens] int64, pad after EOS.  temperature == 0 -> greedy (the reference passes do_sample=False then).
    merge_adapter: fold the (frozen) adapter into its own copy of the projections first (generate.Generator)."""
    if adapter is None and checkpoint is not None:
        adapter = LoraAdapter(engine.d, load_adapter(adapter_dir_of(checkpoint)), engine.dev, trainable=False)
    gen = Generator(engine, adapter, merge_adapter=merge_adapter)
    if temperature and temperature > 0:
        out = gen.generate(queries, query_attn_masks, images, max_new_tokens=max_new_tokens, temperature=temperature, top_k=top_k,
                           top_p=top_p, seed=seed)
    else:
        out = gen.generate(queries, query_attn_masks, images, max_new_tokens=max_new_tokens, temperature=1.0, top_k=1, top_p=1.0, seed=seed)
    return truncate_after_eos_with_padding(out, EOS_ID, PAD_ID)


# ---- question file -> answers file (the text side of eval_llava_rlhf_coco/model_vqa.py:33-44,143-262) -----------------------
DEFAULT_TEST_PROMPT = "\nAnswer the question using a single word or phrase."


def question_chunk(questions: list, num_chunks: int, chunk_idx: int) -> list:
    """Chunk k of n, chunks of ceil(len / n) questions (model_vqa.py:33-42; an index past the last chunk raises IndexError)."""
    size = -(-len(questions) // num_chunks)
    return [questions[i:i + size] for i in range(0, len(questions), size)][chunk_idx]


def eval_prompt(question: str, test_prompt: Optional[str] = DEFAULT_TEST_PROMPT) -> str:
    """'<image>\\n' + question (+ test prompt) as the USER turn, ASSISTANT turn left open (model_vqa.py:153-170)."""
    from .rollout_data import IMAGE_PLACEHOLDER, render_prompt
    text = IMAGE_PLACEHOLDER + "\n" + question + (test_prompt or "")
    return render_prompt([{"from": "human", "value": text}, {"from": "gpt", "value": None}])


def answer_questions(engine: LlavaEngine, tokenizer, questions: list, image_folder: str, answers_file: str, *,
                     checkpoint: Optional[str] = None, adapter: Optional[LoraAdapter] = None, model_id: str = "opadpo-hip",
                     temperature: float = 0.0, top_p: Optional[float] = None, short_eval: bool = False, max_new_tokens: Optional[int] = None,
                     test_prompt: Optional[str] = DEFAULT_TEST_PROMPT, image_size: int = 336, pad_to_square: bool = True, seed: int = 0,
                     merge_adapter: bool = False, batch_size: int = 1) -> int:
    """One JSON line per question: question_id, prompt (the bare question), text (stripped answer, a trailing '</s>' removed),
    answer_id, model_id, metadata (model_vqa.py:143-262).  Refuses to overwrite an existing answers file like the script's
    __main__.  max_new_tokens defaults to 64 (`short_eval`) or 1024.  merge_adapter: merge the frozen checkpoint adapter once (PEFT
    `merge_and_unload` equivalent; bf16 rounding of the merged weights) - every question then decodes without LoRA launches.
    batch_size > 1: that many questions share one generation (prompts left-padded to a common length, pads masked; rotary
    attention only sees position differences, so a row's answer does not depend on its padding) - the decode step streams the
    weights once per BATCH: 16 questions cost about 1.5x the time of one.  The reference script runs one question at a time."""
    import json
    import uuid
    from PIL import Image
    from .data import preprocess_image
    from .rollout_data import SEP2, tokenize_with_image
    if os.path.exists(answers_file):
        raise FileExistsError(f"{answers_file} already exists. Please delete it first.")
    if adapter is None and checkpoint is not None:
        adapter = LoraAdapter(engine.d, load_adapter(adapter_dir_of(checkpoint)), engine.dev, trainable=False)
    n_new = max_new_tokens or (64 if short_eval else 1024)
    parent = os.path.dirname(answers_file)
    if parent:
        os.makedirs(parent, exist_ok=True)
    n = 0
    pad = tokenizer.pad_token_id
    with open(answers_file, "w") as f:
        for c0 in range(0, len(questions), max(1, batch_size)):
            chunk = questions[c0:c0 + max(1, batch_size)]
            rows = [tokenize_with_image(eval_prompt(line["text"], test_prompt), tokenizer) for line in chunk]
            width = max(len(r) for r in rows)
            ids = torch.tensor([[pad] * (width - len(r)) + r for r in rows], dtype=torch.long)
            mask = torch.tensor([[0] * (width - len(r)) + [1] * len(r) for r in rows], dtype=torch.long)
            images = torch.stack([preprocess_image(Image.open(os.path.join(image_folder, line["image"])).convert("RGB"), image_size, pad_to_square)
                                  for line in chunk]).to(engine.dev)
            out = generate_from_checkpoint(engine, None, ids.to(engine.dev), mask.to(engine.dev), images, max_new_tokens=n_new,
                                           temperature=temperature, top_p=1.0 if top_p is None else top_p, seed=seed + n, adapter=adapter,
                                           merge_adapter=merge_adapter)
            for line, text in zip(chunk, tokenizer.batch_decode(out.cpu(), skip_special_tokens=True)):
                text = text.strip()
                if text.endswith(SEP2):
                    text = text[:-len(SEP2)]
                f.write(json.dumps({"question_id": line["question_id"], "prompt": line["text"], "text": text.strip(),
                                    "answer_id": uuid.uuid4().hex[:22], "model_id": model_id, "metadata": {}}) + "\n")
                n += 1
            f.flush()
    return n
