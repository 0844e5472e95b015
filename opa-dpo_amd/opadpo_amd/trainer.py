"""OPA-DPO trainer on the HIP engine: rollout (frozen-reference pass) -> grad-accumulated policy
loss -> exchange -> clip -> AdamW, with the reference's loop structure, step / checkpoint
numbering, statistics and checkpoint layout.

Mirrors opadpo/dpo_models/rl_trainer.py:138-279 (RLTrainer.step_with_rollouts / step / train)
and opadpo/dpo_models/dpo_trainer.py:214-427,475-931 (DPOTrainer.rollout / compute_policy_loss /
record_step_stats / save_model / resume_training).  Deliberate deviations:
Q1 gradients ARE exchanged across ranks, Q2 the discarded policy forward of rollout() is not run,
Q12 no hidden-state output / lm_head only on response rows, Q13 rollouts stay resident in HBM.
"""
from __future__ import annotations

import warnings

import json
import os
import re
from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .dims import PAD_ID
from .losses import DPOArgs, mask_percentage_per_row, mask_single_image, policy_loss
from .optim import FlatAdamW, cosine_lr, layer_buckets
from .policy import AutoregressivePolicy, host_row_plan, response_keys

RESPONSE_KEYS = ("standard_response", "original_generate_response", "AI_pseudo_response")

ADAPTER_MODEL_DIR = "adapter_model"
OPTIMIZER_NAME = "optimizer.pt"
SCHEDULER_NAME = "scheduler.pt"
WEIGHTS_NAME = "adapter_model.bin"
FIRST_STEP_IDX = 1

LLM_TARGET_MODULES = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]


def get_last_checkpoint(checkpoint_dir: str):
    """utils/lora_utils.py:31-47: (path | None, is_completed)."""
    if os.path.isdir(checkpoint_dir):
        is_completed = os.path.exists(os.path.join(checkpoint_dir, "completed"))
        if is_completed:
            return None, True
        max_step = 0
        for fn in os.listdir(checkpoint_dir):
            if os.path.isdir(os.path.join(checkpoint_dir, fn)) and fn.startswith("checkpoint"):
                try:
                    max_step = max(max_step, int(fn.replace("checkpoint-", "")))
                except ValueError:
                    continue   # e.g. checkpoint-final
        if max_step == 0:
            return None, is_completed
        return os.path.join(checkpoint_dir, f"checkpoint-{max_step}"), is_completed
    return None, False


def save_adapter(adapter, directory: str, dims, base_model_name_or_path: str = "", extra_state: Optional[Dict[str, torch.Tensor]] = None,
                 source_config: Optional[dict] = None) -> None:
    """PEFT-0.5 layout (dpo_trainer.py:1047-1095): adapter_model.bin (torch pickle, keys without the adapter name) +
    adapter_config.json with inference_mode forced True.

    `extra_state`: the FROZEN tensors of the same PEFT adapter that this stage does not train - the CLIP-tower and mm_projector
    LoRA of the OPA-stage adapter the policy starts from (qlora_model.py:133-143 attaches LoRA to them too;
    dpo_trainer.py:1022-1030 freezes them).  The reference's `get_peft_model_state_dict(model, adapter_name='lora_policy')` keeps
    them in every checkpoint, so evaluation / PEFT loading of checkpoint-N sees the vision LoRA the model was trained with; they
    are written back unchanged.  `source_config`: adapter_config.json of that source adapter (its target_modules are kept)."""
    os.makedirs(directory, exist_ok=True)
    sd = adapter.to_peft_state()
    for k, v in (extra_state or {}).items():
        if k not in sd:
            sd[k] = v.detach().to(torch.bfloat16).cpu().clone()
    torch.save(sd, os.path.join(directory, WEIGHTS_NAME))
    targets = list(LLM_TARGET_MODULES)
    if source_config and source_config.get("target_modules"):
        targets = list(source_config["target_modules"])
    elif extra_state:
        for k in extra_state:                     # derive the module names from the kept keys (e.g. out_proj, fc1, fc2, "0", "2")
            mod = k.split(".lora_")[0].rsplit(".", 1)[-1]
            if mod not in targets:
                targets.append(mod)
    cfg = {"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": dims.lora_r, "lora_alpha": dims.lora_alpha,
           "lora_dropout": 0.0, "bias": "none", "target_modules": targets,
           "base_model_name_or_path": base_model_name_or_path, "inference_mode": True, "fan_in_fan_out": False,
           "init_lora_weights": True, "modules_to_save": None, "layers_to_transform": None,
           "layers_pattern": None, "revision": None}
    with open(os.path.join(directory, "adapter_config.json"), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)


class DPOTrainer:
    def __init__(self, args, policy: AutoregressivePolicy, ref_policy: AutoregressivePolicy,
                 train_dataset=None, data_collator: Optional[Callable] = None, *, optimizer_mode: str = "allreduce",
                 frozen_adapter_state: Optional[Dict[str, torch.Tensor]] = None, source_adapter_config: Optional[dict] = None,
                 layers_per_bucket: int = 4, exchange_dtype: Optional[torch.dtype] = None, optimizer_kwargs: Optional[dict] = None):
        """`args` carries the reference's TrainingArguments fields that are used on this path:
        DPOArgs fields + rollout_accumulation_steps, gradient_accumulation_steps, step_per_device_batch_size,
        rollout_per_device_batch_size, rollout_batch_size, noptepochs, max_grad_norm, learning_rate, warmup_steps,
        total_epochs, max_step, save_steps, save_steps_extra_list, output_dir, seed, weight_decay.
        frozen_adapter_state / source_adapter_config: vision-tower + projector LoRA tensors and adapter_config.json of the adapter
        the policy starts from; written unchanged into every checkpoint (save_adapter)."""
        self.args = args
        self.policy, self.ref_policy = policy, ref_policy
        self.engine = policy.engine
        self.train_dataset, self.data_collator = train_dataset, data_collator
        self.frozen_adapter_state, self.source_adapter_config = frozen_adapter_state, source_adapter_config
        self.loss_args = DPOArgs(**{k: getattr(args, k) for k in DPOArgs.__dataclass_fields__ if hasattr(args, k)})
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        ad = policy.adapter
        self.optimizer = FlatAdamW(ad.master, ad.grad, ad.work, lr=getattr(args, "learning_rate", 1e-6),
                                   weight_decay=getattr(args, "weight_decay", 0.0),
                                   max_grad_norm=getattr(args, "max_grad_norm", 1.0), mode=optimizer_mode,
                                   bucket_bounds=layer_buckets(ad.layer_numel, self.engine.d.n_layers, layers_per_bucket),
                                   exchange_dtype=exchange_dtype, **(optimizer_kwargs or {}))
        self.optimizer.release_full_master(ad)       # ZeRO-1 across ranks: the fp32 master lives in this rank's slices only
        self.sched_step = 0
        self.total_sched_steps = 1
        self.log_history: List[dict] = []

    # ---------------------------------------------------------------------------------------------
    @property
    def is_main(self) -> bool:
        return self.rank == 0

    def _set_lr(self) -> None:
        a = self.args
        self.optimizer.lr = cosine_lr(self.sched_step, getattr(a, "learning_rate", 1e-6), getattr(a, "warmup_steps", 5),
                                      self.total_sched_steps)

    # ---- rollout: frozen-reference log-probs (dpo_trainer.py:214-427) -----------------------------------
    @torch.no_grad()
    def rollout(self, queries_data: Iterable[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        a = self.loss_args
        dev = self.engine.dev
        outs: List[Dict[str, torch.Tensor]] = []
        for batch in queries_data:
            # ragged-row plan from the collator's HOST tensors (valid lengths are known here: no device->host read in any pass)
            plan = None
            if getattr(self.engine, "ragged", False) and not batch["queries"].is_cuda:
                plan = host_row_plan(batch["queries"], batch["queries_attention_mask"], {k: batch[k] for k in RESPONSE_KEYS})
            b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
            images = b["images"].to(torch.bfloat16)
            rb = {"images": images, "queries": b["queries"], "queries_attn_masks": b["queries_attention_mask"]}
            for k in ("standard_response", "original_generate_response", "AI_pseudo_response"):
                rb[k] = b[k]
                rb[k + "_attention_mask"] = b.get(k + "_attention_mask", b[k] != PAD_ID)
            feats = self.engine.encode_images(images)
            rb["image_feats"] = feats
            pk = {}
            if plan is not None:                   # host tensors; they travel with the rollout rows (cat / index like everything else)
                rb["row_lead"] = plan[0]
                for k in RESPONSE_KEYS:
                    rb["rowlen_" + k[:-len("_response")]] = plan[1][k]
                pk = dict(row_lead=plan[0], row_lens=plan[1])
            new_rb = None
            if a.CoPO:
                if a.CoPO_method in ("random", "blockwise"):
                    masked = torch.stack([mask_single_image(images[i].unsqueeze(0), a.CoPO_mask_ratio, a.CoPO_method)
                                          for i in range(images.size(0))]).squeeze(1)
                    rb["masked_images"] = masked
                    rb["masked_image_feats"] = self.engine.encode_images(masked)
                    new_rb = dict(images=masked, image_feats=rb["masked_image_feats"], queries=rb["queries"],
                                  queries_attn_masks=rb["queries_attn_masks"])
                elif a.CoPO_method == "attention":
                    qm = rb["queries_attn_masks"].clone().bool()
                    im = torch.ones(qm.size(0), self.engine.d.n_patches, dtype=torch.bool, device=dev)
                    im = mask_percentage_per_row(im, a.CoPO_mask_ratio)
                    rb["masked_query_attn_masks"] = torch.cat([im, qm], dim=1)
                    new_rb = dict(images=images, image_feats=feats, queries=rb["queries"],
                                  queries_attn_masks=rb["masked_query_attn_masks"])
                else:
                    raise NotImplementedError(a.CoPO_method)
                new_rb["standard_response"] = rb["standard_response"]
                new_rb["AI_pseudo_response"] = rb["AI_pseudo_response"]
            ref_out = self.ref_policy(**{k: v for k, v in rb.items() if k not in ("masked_images", "masked_image_feats",
                                                                                  "masked_query_attn_masks", "row_lead")},
                                      temperature=a.temperature, **pk)
            for k, v in ref_out.items():
                rb["ref_base_" + k] = v
            if new_rb is not None:
                ref_new = self.ref_policy(**new_rb, temperature=a.temperature, **pk)
                for k, v in ref_new.items():
                    rb["ref_mask_" + k] = v
            if a.detailed_report and (a.response_score or a.response_image_relation):
                for k in ("original_generate_response_scores", "AI_pseudo_response_scores",
                          "original_generate_response_image_relations", "AI_pseudo_response_image_relations"):
                    rb[k] = b[k]
            outs.append(rb)
        return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}

    # ---- policy loss with grad (dpo_trainer.py:475-802) ---------------------------------------------------
    def compute_policy_loss(self, rollouts: Dict[str, torch.Tensor]):
        a = self.loss_args
        common = dict(queries=rollouts["queries"], queries_attn_masks=rollouts["queries_attn_masks"],
                      temperature=a.temperature)
        pk, pk_m = {}, {}
        if "row_lead" in rollouts:
            lens = {k: rollouts["rowlen_" + k[:-len("_response")]] for k in RESPONSE_KEYS}
            pk = dict(row_lead=rollouts["row_lead"], row_lens=lens)
            pk_m = dict(row_lead=rollouts["row_lead"], row_lens={"mask_standard_response": lens["standard_response"],
                                                                 "mask_AI_pseudo_response": lens["AI_pseudo_response"]})
        out = self.policy(images=rollouts["images"], image_feats=rollouts.get("image_feats"), **common, **pk,
                          standard_response=rollouts["standard_response"],
                          original_generate_response=rollouts["original_generate_response"],
                          AI_pseudo_response=rollouts["AI_pseudo_response"])
        out_m = None
        if a.CoPO:
            resp = dict(mask_standard_response=rollouts["standard_response"],
                        mask_AI_pseudo_response=rollouts["AI_pseudo_response"])
            if a.CoPO_method in ("random", "blockwise"):
                out_m = self.policy(images=rollouts["masked_images"], image_feats=rollouts.get("masked_image_feats"),
                                    **common, **resp, **pk_m)
            else:
                out_m = self.policy(images=rollouts["images"], image_feats=rollouts.get("image_feats"),
                                    queries=rollouts["queries"], queries_attn_masks=rollouts["masked_query_attn_masks"],
                                    temperature=a.temperature, **resp, **pk_m)
        return policy_loss(a, rollouts, out, out_m)

    # ---- grad-accumulate -> exchange -> clip -> step (rl_trainer.py:138-179) ---------------------------------
    def step_with_rollouts(self, rollouts: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        a = self.args
        n = rollouts["queries"].size(0)
        bs = a.step_per_device_batch_size
        accum = a.gradient_accumulation_steps
        g = torch.Generator().manual_seed(getattr(a, "seed", 42) + self.sched_step)
        stats_list = []
        for _ in range(getattr(a, "noptepochs", 1)):
            perm = torch.randperm(n, generator=g).tolist()             # rollouts dataloader shuffle=True (:325-345)
            micro = [perm[i:i + bs] for i in range(0, n - bs + 1, bs)]  # drop_last
            for bi, idx in enumerate(micro, 1):
                mb = {k: v[idx] for k, v in rollouts.items()}
                loss, st = self.compute_policy_loss(mb)
                # last micro-batch of the accumulation window: the gradient of a bucket of layers is final as soon as the LAST
                # backward pass of this micro-batch has left its lowest layer -> its exchange starts there, next to the backward
                # of the earlier layers (the reference's DDP buckets, rl_trainer.py:155-175)
                self.policy.layer_done_hook = self._bucket_hook if bi % accum == 0 else None
                loss.backward()
                self.policy.layer_done_hook = None
                if bi % accum == 0:
                    self.optimizer.step(grad_accum_div=accum)
                    st["loss/grad_norm"] = torch.tensor(self.optimizer.grad_norm_post_clip())
                    self.optimizer.zero_grad()
                    self.policy.adapter.refresh_transposed()
                    if self.policy._pending_bwd:          # a forward whose backward never ran would silence the bucket hook for good
                        warnings.warn(f"{self.policy._pending_bwd} policy forward(s) of this step never reached backward; exchange overlap counter reset")
                        self.policy._pending_bwd = 0
                    stats_list.append({k: v.detach().float().cpu() for k, v in st.items()})
        return {k: torch.stack([s[k] for s in stats_list]) for k in stats_list[0]} if stats_list else {}

    def _bucket_hook(self, layer: int) -> None:
        ad, opt = self.policy.adapter, self.optimizer
        pos = layer * ad.layer_numel
        for bi, b in enumerate(opt.buckets):
            if b.lo == pos:                       # `layer` is the lowest layer of bucket bi: every layer above it is done too
                opt.launch_bucket(bi)

    def step(self, train_iter, step_idx: int) -> dict:
        batches = [next(train_iter) for _ in range(self.args.rollout_accumulation_steps)]
        rollouts = self.rollout(batches)
        train_stats = self.step_with_rollouts(rollouts)
        self.sched_step += 1                       # scheduler steps once per OUTER step (Quirk Q14)
        self._set_lr()
        return self.record_step_stats(train_stats, rollouts, step_idx)

    def record_step_stats(self, train_stats, rollouts, step_idx: int) -> dict:
        """dpo_trainer.py:804-835 key naming: 'objective/lr', 'objective/<..entropies>', 'dpo/loss-*',
        'dpo/policy-*', 'logprobs/*', 'dpo/loss-grad_norm'."""
        stats = {"objective/lr": self.optimizer.lr}
        for k, v in rollouts.items():
            if "entropies" in k:
                m = (v != 0.0)
                stats[f"objective/{k}"] = ((v * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)).mean().item()
        for k, v in train_stats.items():
            stats[(k if "logprobs/" in k else f"dpo/{k}")] = v.mean(dim=0).item()
        stats = {(k[:k.find('/') + 1] + k[k.find('/') + 1:].replace('/', '-') if '/' in k else k): v
                 for k, v in stats.items()}
        stats["step"] = step_idx
        return stats

    # ---- outer loop (rl_trainer.py:215-279) ------------------------------------------------------------------
    def train(self, train_iter_factory: Callable[[], Iterable], num_samples: int,
              resume_training_ckpt: Optional[str] = None) -> List[dict]:
        a = self.args
        total_steps = num_samples * a.total_epochs // a.rollout_batch_size
        self.total_sched_steps = min(total_steps, a.max_step)
        self._set_lr()
        skipping = 0
        if resume_training_ckpt is not None:
            skipping = self.resume_training(resume_training_ckpt)
        it = iter(train_iter_factory())
        for step_idx in range(FIRST_STEP_IDX, total_steps + FIRST_STEP_IDX):
            if step_idx < skipping:
                for _ in range(a.rollout_accumulation_steps):
                    next(it)
                continue
            if step_idx >= a.max_step:
                break
            if step_idx % a.save_steps == 0 or step_idx in getattr(a, "save_steps_extra_list", []):
                if step_idx > skipping:
                    self.save_model(os.path.join(a.output_dir, f"checkpoint-{step_idx}"))   # saved BEFORE stepping (Q18)
            stats = self.step(it, step_idx)
            self.log_history.append(stats)
        return self.log_history

    # ---- checkpoints (dpo_trainer.py:837-931) ----------------------------------------------------------------
    def save_model(self, output_dir: str) -> None:
        """dpo_trainer.py:837-931: adapter (PEFT layout) + full optimizer state + scheduler.  Under ZeRO-1 the optimizer state is
        assembled from every rank's slices (collective: ALL ranks call this), so optimizer.pt is independent of the world size."""
        opt_sd = self.optimizer.state_dict()
        if self.is_main:
            os.makedirs(output_dir, exist_ok=True)
            save_adapter(self.policy.adapter, os.path.join(output_dir, ADAPTER_MODEL_DIR, "lora_policy"), self.engine.d,
                         getattr(self.args, "base_model_name", ""), self.frozen_adapter_state, self.source_adapter_config)
            torch.save({"optimizer": opt_sd, "sched_step": self.sched_step}, os.path.join(output_dir, OPTIMIZER_NAME))
            torch.save({"last_epoch": self.sched_step, "total": self.total_sched_steps},
                       os.path.join(output_dir, SCHEDULER_NAME))
            parent = os.path.dirname(output_dir.rstrip("/"))
            for fn in os.listdir(parent) if os.path.isdir(parent) else []:       # keep only the newest optimizer.pt (:885-896)
                p = os.path.join(parent, fn, OPTIMIZER_NAME)
                if fn.startswith("checkpoint-") and os.path.join(parent, fn) != output_dir.rstrip("/") and os.path.exists(p):
                    os.remove(p)
        if self.world > 1:
            dist.barrier()

    def resume_training(self, checkpoint_dir: str) -> int:
        """dpo_trainer.py:1098-1137: optimizer (m, v, step AND the fp32 master weights - a resume continues the trajectory instead
        of restarting from bf16-rounded weights), scheduler position; every rank keeps its share of the current ZeRO-1 layout."""
        opt = os.path.join(checkpoint_dir, OPTIMIZER_NAME)
        if os.path.exists(opt):
            sd = torch.load(opt, map_location="cpu")
            self.optimizer.load_state_dict(sd["optimizer"])
            self.sched_step = int(sd["sched_step"])
            self._set_lr()
            if self.policy.adapter.trainable:
                self.policy.adapter.refresh_transposed()
        m = re.search(r"checkpoint-(\d+)", os.path.basename(checkpoint_dir.rstrip("/")))
        return int(m.group(1)) if m else 0
