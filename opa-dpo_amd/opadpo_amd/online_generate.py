"""One rollout step of the generator stage (host side around the decode kernels): batches of
`rollout_data.QueryResponseDataset` items -> sampled responses -> the column dict that `dataset_build.write_rollout_json`
stores as `step{N}_rank{R}.json`.

Reference behaviour restated (no code shared), opadpo/generator_models/online_generator.py:262-377: per batch sample
`response_len` new tokens (temperature -> top-k -> top-p), cut after EOS / the two question-mark ids, decode responses and
standard responses with special tokens skipped, decode the query with the image placeholder turned into BOS and keep the text
between 'USER:  \\n' and ' ASSISTANT:', ask the feedback model for (pseudo response, generated response, JSON report), and
collect eight equal-length columns.  The GPT-4V feedback client needs the network and is out of scope: `feedback`
is a callable; the default returns empty answers, which the dataset builder's first filter then drops.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch

from .rollout_data import IMAGE_TOKEN_INDEX

QUESTION_MARK_IDS = (1577, 29973)          # '?' alone and after a newline in the Llama vocabulary (online_generator.py:311-315)
QUERY_HEAD, QUERY_TAIL = "USER:  \n", " ASSISTANT:"


def no_feedback(images_url: Sequence[str], queries: Sequence[str], responses: Sequence[str], standard: Sequence[str]) -> Dict[str, list]:
    n = len(responses)
    return {"Pseudo_response": [""] * n, "Generated_response": list(responses), "report_json": [""] * n}


def query_text(decoded: str) -> str:
    """Question part of a decoded prompt; `str.find` semantics kept (a missing marker yields -1 like in the reference)."""
    return decoded[decoded.find(QUERY_HEAD) + len(QUERY_HEAD):decoded.find(QUERY_TAIL)]


def decode_batch(tokenizer, ids: torch.Tensor) -> List[str]:
    return tokenizer.batch_decode(ids, skip_special_tokens=True, clean_up_tokenization_spaces=True)


def rollout_step(batches: Iterable[Dict], tokenizer, sample: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                 feedback: Optional[Callable] = None) -> Dict[str, list]:
    """`sample(queries, query_attn_masks, images) -> responses [B, response_len]` already cut after EOS / '?' (on the GPU:
    `generator_sampler(...)` below).  Returns the response dict of online_generator.py:352-362."""
    feedback = feedback or no_feedback
    out: Dict[str, list] = {k: [] for k in ("query", "image_id", "standard_response", "original_generate_response",
                                            "AI_generate_response", "AI_pseudo_response", "AI_json_report", "image_bytes")}
    for batch in batches:
        queries = batch["queries"]
        responses = sample(queries, batch["query_attn_masks"], batch["images"])
        text_rsp = decode_batch(tokenizer, responses)
        text_std = decode_batch(tokenizer, batch["standard_responses"])
        shown = queries.clone()
        shown[shown == IMAGE_TOKEN_INDEX] = tokenizer.bos_token_id
        text_q = [query_text(q) for q in decode_batch(tokenizer, shown)]
        fb = feedback(batch["images_url"], text_q, text_rsp, text_std)
        if not (len(fb["Pseudo_response"]) == len(fb["Generated_response"]) == len(fb["report_json"]) == len(text_rsp)):
            raise ValueError("feedback must return one entry per response")
        out["query"] += ["<image>\n" + q for q in text_q]
        out["image_id"] += list(batch["images_path"])
        out["standard_response"] += text_std
        out["original_generate_response"] += text_rsp
        out["AI_generate_response"] += list(fb["Generated_response"])
        out["AI_pseudo_response"] += list(fb["Pseudo_response"])
        out["AI_json_report"] += list(fb["report_json"])
        out["image_bytes"] += list(batch["images_bytes"])
    return out


def generator_sampler(generator, *, response_len: int, temperature: float = 1.0, top_k: int = 30, top_p: float = 0.95,
                      seed: int = 0) -> Callable:
    """Sampler over the HIP decode path (`generate.Generator.rollout`); a new seed per batch.  Build the generator with
    `merge_adapter=True` (frozen rollout adapter) or `fuse_swiglu=True` (adapter-free rollout) for the fastest decode step."""
    state = {"n": 0}

    def sample(queries, masks, images):
        state["n"] += 1
        dev = generator.engine.dev
        return generator.rollout(queries.to(dev), masks.to(dev), images.to(dev), response_len=response_len, temperature=temperature,
                                 top_k=top_k, top_p=top_p, seed=seed + state["n"], additional_stop_ids=QUESTION_MARK_IDS).cpu()
    return sample
