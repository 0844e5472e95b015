#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): time the REFERENCE's own `AutoregressivePolicy.forward`
(opadpo/dpo_models/rl_models.py:75-144), stub-imported as in tests/golden/make_golden.py, around the installed HuggingFace CPU Llama at
LLaVA-1.5-7B width - the "reference CPU forward" figure SURVEY.md section 8(d) asks BASELINE.md to record.

Config P of section 8(d): 8 pairs, query 32 + response 96 text ids, L = 703 positions (the one image token of every query stands for 576
patch embeddings: the shim widens every sequence by 575 positions so that the decoder sees the real L), fp32, world 1, decoder
truncated to --layers of the 32 (stated).  The CLIP tower is not part of this timing (the LLaVA wrapper is absent).

    python tools/ref_cpu_forward.py --layers 1
"""
import argparse
import importlib.util
import json
import os
import sys
import time
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=8)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("mg", os.path.join(REPO, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    cu, rl_models, dpo_trainer, generator, lora_utils = mg.import_reference()
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=a.layers, num_attention_heads=32,
                      num_key_value_heads=32, rms_norm_eps=1e-5, max_position_embeddings=2048, attn_implementation="eager",
                      tie_word_embeddings=False)
    lm = LlamaForCausalLM(cfg).eval().float()
    P = 576

    class Shim(torch.nn.Module):
        """What `self.base_model` is to the reference: prepare_inputs_for_generation + forward -> .logits (the multimodal splice of the
        absent LLaVA wrapper is emulated by widening every sequence by P - 1 positions in front of the response)."""

        def __init__(self, lm):
            super().__init__()
            self.lm, self.config = lm, lm.config

        def set_adapter(self, name):
            pass

        def prepare_inputs_for_generation(self, input_ids=None, attention_mask=None, images=None, use_cache=None):
            return dict(input_ids=input_ids, attention_mask=attention_mask)

        def forward(self, input_ids=None, attention_mask=None, output_hidden_states=False):
            S = input_ids.shape[0]
            ids = torch.cat([torch.full((S, P - 1), 5, dtype=input_ids.dtype), input_ids.clamp_min(0)], 1)
            am = torch.cat([torch.ones(S, P - 1, dtype=attention_mask.dtype), attention_mask], 1)
            return self.lm(input_ids=ids, attention_mask=am.long(), output_hidden_states=output_hidden_states, use_cache=False)

    B, Q, T = a.pairs, 32, 96
    g = torch.Generator().manual_seed(0)
    queries = torch.randint(3, 32000, (B, Q), generator=g)
    qmask = torch.ones(B, Q, dtype=torch.bool)
    resp = {}
    for k in ("chosen_response", "rejected_response"):
        ids = torch.randint(3, 32000, (B, T), generator=g)
        for b in range(B):
            ln = int(torch.randint(T // 6, T, (1,), generator=g))
            ids[b, ln] = 2
            ids[b, ln + 1:] = 0
        resp[k] = ids
    pol = rl_models.AutoregressivePolicy(types.SimpleNamespace(temperature=1.0, response_len=T), Shim(lm), types.SimpleNamespace(pad_token_id=0),
                                         adapter_name="lora_policy")
    times = []
    for _ in range(2):
        t0 = time.time()
        with torch.no_grad():
            out = pol(images=torch.zeros(B, 1), queries=queries, queries_attn_masks=qmask, temperature=1.0, **resp)
        times.append(time.time() - t0)
    dt = min(times)
    L = Q + T + P - 1
    rec = {"what": "reference AutoregressivePolicy.forward (stub-imported) around HF LlamaForCausalLM fp32 CPU, no grad",
           "pairs": B, "sequences": 2 * B, "L": L, "layers_run": a.layers, "layers_model": 32, "cores": os.cpu_count(), "seconds": dt,
           "sequence_forwards_per_s_truncated": 2 * B / dt,
           "extrapolated_full_depth_seconds_per_sequence_forward": dt / (2 * B) * 32 / a.layers,
           "keys": sorted(out.keys())}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
