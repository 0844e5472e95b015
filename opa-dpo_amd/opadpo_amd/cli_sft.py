"""Command-line surface of the OPA LoRA-SFT stage — the flags `run/train_opa.sh` passes to `opadpo/opa_train_custom.py` of the
reference (LoRA tuning of LLM + CLIP tower + projector: LORATUNE=True, TUNE_MM_PROJECT / TUNE_BASE_MODEL / TUNE_VISION_TOWER
= True, rank 256 / alpha 512, per-device batch 4 x 8 accumulation steps, 2 epochs, entropy regulariser off).  DeepSpeed /
tf32 / flash-attention switches are accepted and ignored (one process per GPU over RCCL, flat-gradient exchange per
optimizer step like the DPO stage); full fine-tuning (`--full_tune True`) is not built.  `--cfg` is read with PyYAML and only
fills values the command line did not set.  Output: `<output_dir>/checkpoint-final/adapter_model.bin` (+ adapter_config.json),
the adapter the DPO stage starts from.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
from typing import List, Optional

FLAGS = [
    ("cfg", str, "configs/llava/llava_opa.yaml"), ("local-rank", int, 0), ("base_model", str, "./base_models/llava-v1.5-7b"),
    ("output_dir", str, "./output/llava7b_opa_model"), ("image_folder", str, "none"), ("data_dir", str, "./base_datasets/opa_training_data-7B"),
    ("per_device_train_batch_size", int, 4), ("per_device_eval_batch_size", int, 4), ("gradient_accumulation_steps", int, 8),
    ("deepspeed", str, None), ("tf32", "st", None), ("bf16", "st", None), ("use_flash_attention", "st", None),
    ("save_steps", int, 40), ("eval_steps", int, 10), ("mm_vision_select_layer", int, -2), ("mm_projector_type", str, "mlp2x_gelu"),
    ("full_tune", "sbool", "False"), ("tune_mm_mlp_adapter", "sbool", "True"), ("tune_base_model", "sbool", "True"),
    ("tune_vision_tower", "sbool", "True"), ("lora_tune", "sbool", "True"), ("lora_rank", int, 256), ("lora_alpha", int, 512),
    ("lora_drop", float, 0.0), ("num_train_epochs", float, 2.0), ("entropy_loss", "sbool", "False"), ("entropy_mask_ratio", float, 0.8),
    ("entropy_mask_method", str, "random"), ("entropy_loss_coef", float, 0.01), ("entropy_decay_coef", float, 1.0),
    ("learning_rate", float, 2e-5), ("weight_decay", float, 0.0), ("warmup_ratio", float, 0.03), ("lr_scheduler_type", str, "cosine"),
    ("max_grad_norm", float, 1.0), ("seed", int, 42), ("model_max_length", int, 2048), ("query_len", int, 128), ("response_len", int, 896),
    ("image_aspect_ratio", str, "pad"), ("max_steps", int, -1),
    # additions of this build
    ("optimizer_mode", str, "zero1"), ("synthetic", str, None), ("synthetic_samples", int, 64),
]


def make_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="OPA LoRA-SFT training (MI355X-native)")
    for name, typ, default in FLAGS:
        if typ == "st":
            p.add_argument("--" + name, action="store_true")
        elif typ == "sbool":
            p.add_argument("--" + name, type=str, default=default)
        else:
            p.add_argument("--" + name, type=typ, default=default)
    return p


def main(argv: Optional[List[str]] = None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    ns = make_parser().parse_args(argv)
    from .cli import _as_bool, load_yaml_defaults
    load_yaml_defaults(ns, argv)
    for name, typ, _ in FLAGS:
        if typ == "sbool":
            setattr(ns, name, _as_bool(getattr(ns, name)))
    if ns.full_tune or not ns.lora_tune:
        raise SystemExit("only the LoRA recipe of run/train_opa.sh is built (lora_tune True, full_tune False)")
    if not (ns.tune_mm_mlp_adapter and ns.tune_base_model and ns.tune_vision_tower):
        raise SystemExit("the LoRA recipe tunes projector, LLM and vision tower together (TUNE_* = True in run/train_opa.sh)")
    import torch
    import torch.distributed as dist
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from . import checkpoint_io as CK
    from .dims import LlavaDims
    from .ctx import CtxEngine
    from .model import BaseWeights, LoraAdapter
    from .optim import cosine_lr
    from .sft import SFTTrainer, sft_batches_from_dpo_batch
    from .vision_train import VisionLoraAdapter
    torch.manual_seed(ns.seed)
    B = ns.per_device_train_batch_size
    if ns.synthetic:
        from .synth import init_lora, init_weights, synth_pairs
        d = {"tiny": LlavaDims.tiny, "7b": LlavaDims.llava15_7b, "13b": LlavaDims.llava15_13b}[ns.synthetic]()
        state = init_weights(d, seed=ns.seed, device=dev)
        lora = init_lora(d, seed=ns.seed + 1, device=dev, with_vision=True)
        q_len, t_len = (ns.query_len, ns.response_len) if ns.synthetic != "tiny" else (16, 16)
        n_samples = ns.synthetic_samples

        def batches(epoch):
            for i in range(rank, n_samples // B, world):
                p = synth_pairs(d, B, q_len, t_len, seed=ns.seed + 1000 * epoch + i, device=dev)
                yield dict(images=p["images"], queries=p["queries"], queries_attn_masks=p["queries_attn_masks"], responses=p["chosen"])
        steps_per_epoch = max(1, (n_samples // B) // world // ns.gradient_accumulation_steps)
    else:
        d = CK.dims_from_config(ns.base_model, ns.lora_rank, float(ns.lora_alpha))
        state = CK.load_llava_state(ns.base_model)
        from .synth import init_lora
        lora = init_lora(d, seed=ns.seed + 1, device=dev, with_vision=True, b_std=0.0)     # PEFT init: A kaiming-uniform, B = 0
        t_len = ns.response_len
        from datasets import load_from_disk
        from transformers import AutoTokenizer
        from .data import DataCollatorForCausalLM, DPODataset
        tok = AutoTokenizer.from_pretrained(ns.base_model, model_max_length=ns.model_max_length, padding_side="left", use_fast=False)
        tok.pad_token = tok.unk_token
        ds = load_from_disk(ns.data_dir)
        dataset = DPODataset(ds, image_dir=os.environ.get("IMAGE_DIR", ns.image_folder or ""), pad_to_square=ns.image_aspect_ratio == "pad")
        coll = DataCollatorForCausalLM(tok, ns.query_len, ns.response_len, False)

        def batches(epoch):      # every row gives two samples (standard / AI-pseudo response); same permutation on every rank
            g = torch.Generator().manual_seed(ns.seed + epoch)
            perm = torch.randperm(len(dataset), generator=g).tolist()[rank::world]
            for i in range(0, len(perm) - B + 1, B):
                for sb in sft_batches_from_dpo_batch(coll([dataset[j] for j in perm[i:i + B]])):
                    yield sb
        steps_per_epoch = max(1, (2 * (len(dataset) // world) // B) // ns.gradient_accumulation_steps)
    engine = CtxEngine(BaseWeights(d, state, dev, need_backward=True))
    del state
    tr = SFTTrainer(engine, LoraAdapter(d, lora, dev, trainable=True), VisionLoraAdapter(d, lora, dev), response_len=t_len,
                    lr=ns.learning_rate, max_grad_norm=ns.max_grad_norm, weight_decay=ns.weight_decay, optimizer_mode=ns.optimizer_mode,
                    entropy_loss=ns.entropy_loss, entropy_mask_ratio=ns.entropy_mask_ratio, entropy_mask_method=ns.entropy_mask_method,
                    entropy_loss_coef=ns.entropy_loss_coef, entropy_decay_coef=ns.entropy_decay_coef)
    total = int(math.ceil(ns.num_train_epochs * steps_per_epoch)) if ns.max_steps <= 0 else ns.max_steps
    warm = int(math.ceil(ns.warmup_ratio * total))
    step, micro, log = 0, 0, []
    epoch = 0
    while step < total:
        for b in batches(epoch):
            loss = tr.loss_and_backward(b["images"], b["queries"], b["queries_attn_masks"], b["responses"])
            micro += 1
            if micro % ns.gradient_accumulation_steps == 0:
                lr = cosine_lr(step + 1, ns.learning_rate, warm, total) if ns.lr_scheduler_type == "cosine" else ns.learning_rate
                tr.opt_llm.lr = tr.opt_vis.lr = lr
                norm = tr.optimizer_step(grad_accum_div=ns.gradient_accumulation_steps)
                step += 1
                log.append(dict(step=step, loss=loss, grad_norm=norm, lr=lr, **tr.last))
                if rank == 0:
                    print(log[-1], flush=True)
                if ns.save_steps > 0 and step % ns.save_steps == 0 and rank == 0:
                    tr.save(os.path.join(ns.output_dir, f"checkpoint-{step}"), ns.base_model)
                if step >= total:
                    break
        epoch += 1
    if rank == 0:
        tr.save(os.path.join(ns.output_dir, "checkpoint-final"), ns.base_model)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
