"""Command-line surface of the DPO stage — same flags as `opadpo/opadpo_train_custom.py` of the reference
(`run/train_opa_dpo.sh` passes them), including its quirks: Q8 `--bf16 --tf32
--use_flash_attention --resume_from_training --do_train --clean_tokens_after_eos` are store_false (default True);
Q5 `AncPO` is taken from `--CoPO`; Q9 PPO-era / unused flags are accepted and ignored.  Hydra/omegaconf are not
required: `--cfg` is read with PyYAML and only fills values the command line did not set.
"""
from __future__ import annotations

import argparse
import os
import sys
from types import SimpleNamespace
from typing import List, Optional

# (flag, type, default).  type "sbool" = string "True"/"False" like the reference; "sf" = store_false; "st" = store_true
FLAGS = [
    ("cfg", str, "configs/llava/llava_dpo.yaml"), ("local-rank", int, 0), ("lora_rank", int, 256), ("lora_alpha", int, 512),
    ("lora_drop", float, 0.0), ("detailed_report", "sbool", "True"), ("response_score", "sbool", "True"),
    ("response_image_relation", "sbool", "True"), ("standard_pair_coef", float, 1.0), ("AI_pair_coef", float, 1.0),
    ("CoPO", "sbool", "True"), ("CoPO_mask_ratio", float, 0.3), ("CoPO_method", str, "random"), ("CoPO_coef", float, 0.2),
    ("AncPO", "sbool", "True"), ("Anchor_value", float, 0.0), ("mDPO_anchor", "sbool", "True"), ("Anchor_coef", float, 1.0),
    ("reference_free", "sbool", "False"), ("f_divergence_type", str, "reverse_kl"), ("loss_type", str, "sigmoid"),
    ("beta", float, 0.1), ("label_smoothing", float, 0.0), ("advantage_whiten_all", "sbool", "True"),
    ("train_from_sft", "sbool", "True"), ("norm_maintain_32", "sbool", "False"), ("lora_with_projector", "sbool", "False"),
    ("value_head_mode", str, "mlp2x_gelu"), ("ddp_backend", str, "None"), ("ddp_find_unused_parameters", str, "None"),
    ("base_model", str, "./base_models/llava-v1.5-7b"), ("output_dir", str, "./output/llava7b_opadpo_model"),
    ("image_folder", str, None), ("policy_model_name_or_path", str, "./output/llava7b_opa_model/checkpoint-final"),
    ("do_train", "sf", None), ("seed", int, 42), ("rollout_batch_size", int, 128), ("step_batch_size", int, 32),
    ("rollout_per_device_batch_size", int, 16), ("reward_model_per_device_batch_size", int, 16),
    ("step_per_device_batch_size", int, 16), ("learning_rate", float, 3e-5), ("init_value_with_reward", "st", None),
    ("warmup_steps", int, 5), ("total_epochs", int, 1), ("group_by_length", "st", None), ("evaluation_strategy", str, "no"),
    ("save_strategy", str, "steps"), ("weight_decay", float, 0.0), ("lr_scheduler_type", str, "cosine"),
    ("logging_steps", int, 1), ("report_to", str, "wandb"), ("bf16", "sf", None), ("tf32", "sf", None), ("fp16", "st", None),
    ("penalty_reward_value", float, -8.0), ("length_bonus_score", float, -10.0), ("correct_bonus_score", float, 2.0),
    ("relative_stop_token_penalty", "st", None), ("penalize_no_stop_token", "st", None), ("resume_from_training", "sf", None),
    ("kl_coef", float, 0.1), ("max_grad_norm", float, 1.0), ("whitening_async_stats", str, "full_batch"),
    ("clean_tokens_after_eos", "sf", None), ("temperature", float, 1.0), ("model_max_length", int, 2048), ("query_len", int, 256),
    ("response_len", int, 256), ("noptepochs", int, 2), ("use_flash_attention", "sf", None), ("eval_steps", int, 100),
    ("save_steps", int, 10), ("save_total_limit", int, 3), ("gradient_accumulation_steps", int, 16), ("reward_scale", float, 1.0),
    ("max_step", int, 300), ("reward_clip_min", float, -10.0), ("reward_clip_max", float, 10.0), ("cliprange", float, 0.2),
    ("cliprange_value", float, 0.2), ("gamma", float, 1.0), ("lam", float, 1.0),
    ("data_path", str, "./base_datasets/opadpo_training_data-7B"), ("image_aspect_ratio", str, "pad"), ("train_splits", str, "train"),
    ("base_model_name", str, "./base_models/llava-v1.5-7b"), ("vision_tower", str, "different"), ("mm_vision_select_layer", int, -2),
    ("mm_use_im_start_end", "st", None), ("mm_use_im_patch_token", "st", None), ("freeze_mm_mlp_adapter", "st", None),
    # additions of this build
    ("optimizer_mode", str, "zero1"), ("synthetic", str, None), ("merge_ref_adapter", int, 0),
]


def make_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="OPA-DPO training (MI355X-native)")
    for name, typ, default in FLAGS:
        flag = "--" + name
        if typ == "sf":
            p.add_argument(flag, action="store_false")
        elif typ == "st":
            p.add_argument(flag, action="store_true")
        elif typ == "sbool":
            p.add_argument(flag, type=str, default=default)
        else:
            p.add_argument(flag, type=typ, default=default)
    return p


def _as_bool(v) -> bool:
    return v if isinstance(v, bool) else str(v) == "True"


def build_args(ns: argparse.Namespace, world_size: Optional[int] = None) -> SimpleNamespace:
    """argparse namespace -> trainer arguments, with the accumulation arithmetic of
    TrainingArguments.__post_init__ (opadpo/opadpo_train.py:383-433)."""
    a = SimpleNamespace(**vars(ns))
    for name, typ, _ in FLAGS:
        if typ == "sbool":
            setattr(a, name.replace("-", "_"), _as_bool(getattr(a, name.replace("-", "_"))))
    a.AncPO = a.CoPO          # Quirk Q5 (opadpo_train_custom.py:202)
    world = world_size if world_size is not None else int(os.environ.get("WORLD_SIZE", 1))
    a.world_size = world
    for total, per, out in (("rollout_batch_size", "rollout_per_device_batch_size", "rollout_accumulation_steps"),
                            ("step_batch_size", "step_per_device_batch_size", "gradient_accumulation_steps")):
        t, pd = getattr(a, total), getattr(a, per)
        if t % (pd * world) != 0:
            raise ValueError(f"{total} ({t}) must be divisible by {per} ({pd}) x world size ({world})")
        setattr(a, out, t // pd // world)
    a.save_steps_extra_list = []
    return a


def load_yaml_defaults(ns: argparse.Namespace, argv: List[str]) -> None:
    """`--cfg` YAML (flat or nested) fills every known flag that was NOT given on the command line."""
    if not ns.cfg or not os.path.exists(ns.cfg):
        return
    import yaml
    given = {tok[2:].split("=")[0].replace("-", "_") for tok in argv if tok.startswith("--")}

    def walk(node):
        for k, v in (node or {}).items():
            if isinstance(v, dict):
                walk(v)
            elif hasattr(ns, k) and k not in given and v is not None:
                setattr(ns, k, v)
    walk(yaml.safe_load(open(ns.cfg)))


def main(argv: Optional[List[str]] = None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    ns = make_parser().parse_args(argv)
    load_yaml_defaults(ns, argv)
    args = build_args(ns)
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    # CoPO image masks draw from the global CPU generator (losses.mask_single_image: torch.randperm); the reference seeds every
    # library from the configured seed before building anything (opadpo_train.py:564 set_reproducibility(seed))
    torch.manual_seed(args.seed)
    from . import checkpoint_io as CK
    from .data import DataCollatorForCausalLM, DPODataset
    from .dims import LlavaDims
    from .ctx import CtxEngine
    from .model import BaseWeights, LoraAdapter
    from .policy import AutoregressivePolicy
    from .trainer import DPOTrainer, get_last_checkpoint

    if args.synthetic:      # no network in the build environment: random-init model + synthetic rollouts
        from .synth import init_lora, init_weights
        d = {"tiny": LlavaDims.tiny, "7b": LlavaDims.llava15_7b, "13b": LlavaDims.llava15_13b}[args.synthetic]()
        # the OPA-stage adapter carries CLIP-tower + projector LoRA too (frozen here, merged into the vision weights at load)
        state, adapter_sd = init_weights(d, seed=args.seed, device=dev), init_lora(d, seed=args.seed + 1, device=dev, with_vision=True)
        ref_sd, src_cfg = adapter_sd, None
        vision_lora = {k: v for k, v in ref_sd.items() if "vision_tower" in k or "mm_projector" in k}
        last, done = get_last_checkpoint(args.output_dir) if args.resume_from_training else (None, False)
        if done:
            print("training already completed")
            return
        resume_dir = last
        if last:
            adapter_sd = CK.load_adapter(os.path.join(last, "adapter_model", "lora_policy"))
    else:
        d = CK.dims_from_config(args.base_model_name, args.lora_rank, float(args.lora_alpha))
        state = CK.load_llava_state(args.base_model_name)
        ckpt = args.policy_model_name_or_path
        last, done = get_last_checkpoint(args.output_dir) if args.resume_from_training else (None, False)
        if done:
            print("training already completed")
            return
        resume_dir = last
        adapter_sd = CK.load_adapter(os.path.join(last, "adapter_model", "lora_policy") if last else ckpt)
        ref_sd = CK.load_adapter(ckpt)
        vision_lora = {k: v for k, v in ref_sd.items() if "vision_tower" in k or "mm_projector" in k}
        import json
        cfg_path = os.path.join(ckpt, "adapter_config.json")
        src_cfg = json.load(open(cfg_path)) if os.path.exists(cfg_path) else None
    base = BaseWeights(d, state, dev, need_backward=True, vision_lora=vision_lora)
    engine = CtxEngine(base)          # sequence-level C entry points (opadpo_ctx): layer loop, workspace, activations below the ABI
    policy = AutoregressivePolicy(engine, LoraAdapter(d, adapter_sd, dev, True), args.response_len, args.temperature, "lora_policy")
    ref_adapter = LoraAdapter(d, ref_sd, dev, False)
    # Default 0 (round 5): the frozen reference adapter runs through the SAME K-concatenated kernels as the policy, so the log-ratio of two
    # equal adapters is exactly 0 on every token - the reference's semantics (dpo_trainer.py:444-449, 997-1016), and every *_gap_mean / accuracy
    # statistic of the first steps starts from 0 instead of from rounding noise.  1 folds it into its own bf16 weight copy
    # (model.LoraAdapter.merge_into_base: -4 % step time, |log-ratio| 0.03 nat per token at equal adapters); bench.py measures both.
    if args.merge_ref_adapter:
        ref_adapter.merge_into_base(base)
    ref_policy = AutoregressivePolicy(engine, ref_adapter, args.response_len, args.temperature, "lora_ref_policy")
    trainer = DPOTrainer(args, policy, ref_policy, optimizer_mode=args.optimizer_mode, frozen_adapter_state=vision_lora,
                         source_adapter_config=src_cfg)
    if args.synthetic:
        from .synth import synth_rollout_batches
        n = args.rollout_batch_size * 2
        factory = lambda: synth_rollout_batches(d, args, seed=args.seed + trainer.rank)
        trainer.train(factory, n, resume_training_ckpt=resume_dir)
    else:
        from datasets import load_from_disk
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(args.base_model_name, model_max_length=args.model_max_length, padding_side="left", use_fast=False)
        tok.pad_token = tok.unk_token
        ds = load_from_disk(args.data_path)
        ds = ds[args.train_splits] if hasattr(ds, "keys") and args.train_splits in ds else ds
        dataset = DPODataset(ds, image_dir=os.environ.get("IMAGE_DIR", args.image_folder or ""), pad_to_square=args.image_aspect_ratio == "pad")
        coll = DataCollatorForCausalLM(tok, args.query_len, args.response_len, args.detailed_report)

        def factory():
            g = torch.Generator().manual_seed(args.seed)
            while True:          # same permutation on every rank, strided by rank (rl_trainer.py:314-321)
                perm = torch.randperm(len(dataset), generator=g).tolist()[trainer.rank::trainer.world]
                for i in range(0, len(perm) - args.rollout_per_device_batch_size + 1, args.rollout_per_device_batch_size):
                    yield coll([dataset[j] for j in perm[i:i + args.rollout_per_device_batch_size]])
        trainer.train(factory, len(dataset), resume_training_ckpt=resume_dir)
    trainer.save_model(os.path.join(args.output_dir, "checkpoint-final"))
    if trainer.is_main:
        open(os.path.join(args.output_dir, "completed"), "w").close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
