"""Weight I/O at the reference's file-format boundary.

* base LLaVA: HF sharded `pytorch_model-0000x-of-0000y.bin` + `pytorch_model.bin.index.json` or
  `model-*.safetensors` + `model.safetensors.index.json` (keys `model.layers.*`, `lm_head.weight`,
  `model.mm_projector.{0,2}.*`); the CLIP tower comes either inside the LLaVA shards
  (`model.vision_tower.vision_tower.vision_model.*`) or from an `openai/clip-vit-large-patch14-336` directory
  (`vision_model.*`), like `loading_vision_tower_parameter` (opadpo/opadpo_train.py:539-557);
* adapters: PEFT `adapter_model.bin` / `adapter_model.safetensors` + `adapter_config.json`
  (`checkpoint-final/` of the OPA stage, `checkpoint-N/adapter_model/lora_policy/` of this stage).
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Optional

import torch

from .dims import VIS_PREFIX, LlavaDims


def _load_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def load_sharded_state(model_dir: str) -> Dict[str, torch.Tensor]:
    for index in ("pytorch_model.bin.index.json", "model.safetensors.index.json"):
        p = os.path.join(model_dir, index)
        if os.path.exists(p):
            files = sorted(set(json.load(open(p))["weight_map"].values()))
            break
    else:
        files = [os.path.basename(f) for pat in ("pytorch_model*.bin", "model*.safetensors")
                 for f in sorted(glob.glob(os.path.join(model_dir, pat)))]
    if not files:
        raise FileNotFoundError(f"no weight files under {model_dir}")
    state: Dict[str, torch.Tensor] = {}
    for f in files:
        state.update(_load_file(os.path.join(model_dir, f)))
    return state


def load_llava_state(base_model_dir: str, vision_tower_dir: Optional[str] = None) -> Dict[str, torch.Tensor]:
    state = load_sharded_state(base_model_dir)
    if not any(k.startswith(VIS_PREFIX) for k in state):
        if vision_tower_dir is None:
            cfg = json.load(open(os.path.join(base_model_dir, "config.json")))
            vision_tower_dir = cfg.get("image_checkpoint") or cfg.get("mm_vision_tower")
        for k, v in load_sharded_state(vision_tower_dir).items():
            if k.startswith("vision_model."):
                state[VIS_PREFIX + k[len("vision_model."):]] = v
    return state


def dims_from_config(base_model_dir: str, lora_r: int = 256, lora_alpha: float = 512.0) -> LlavaDims:
    c = json.load(open(os.path.join(base_model_dir, "config.json")))
    return LlavaDims(hidden=c["hidden_size"], n_layers=c["num_hidden_layers"], n_heads=c["num_attention_heads"],
                     head_dim=c["hidden_size"] // c["num_attention_heads"], ffn=c["intermediate_size"], vocab=c["vocab_size"],
                     rms_eps=c.get("rms_norm_eps", 1e-5), rope_theta=c.get("rope_theta", 10000.0), lora_r=lora_r,
                     lora_alpha=lora_alpha)


def load_adapter(adapter_dir: str) -> Dict[str, torch.Tensor]:
    for name in ("adapter_model.bin", "adapter_model.safetensors"):
        p = os.path.join(adapter_dir, name)
        if os.path.exists(p):
            sd = _load_file(p)
            # PEFT >= 0.6 writes "...lora_A.weight"; some exports keep the adapter name in the key
            return {k.replace(".lora_policy.", ".").replace(".default.", "."): v for k, v in sd.items()}
    raise FileNotFoundError(f"no adapter_model.* under {adapter_dir}")
