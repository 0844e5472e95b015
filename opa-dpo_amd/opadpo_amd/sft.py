"""OPA LoRA-SFT step at tensor level.

Reference: opadpo/opa_train.py + opadpo/opa_models/opa_trainer.py:58-125 — HF Trainer over LLaVA with LoRA on every
nn.Linear except lm_head (LLM projections AND the CLIP tower's q/k/v/out_proj/fc1/fc2 AND mm_projector.0/.2), loss = the
causal-LM cross-entropy over `labels != -100` (mean over the labelled tokens of the micro-batch), AdamW, global-norm clipping
over ALL trainable tensors; optional entropy regulariser (`entropy_loss`, default False in configs/llava/llava_opa.yaml): a
second forward on the randomly masked image, loss += coef * mean_b( -sum_t (H_masked - H_clean) m / sum_t m ), gradient through
BOTH forwards' entropies (opa_trainer.py:64-90; mask methods 'random' / 'blockwise' via losses.mask_single_image, 'attention' =
image keys dropped from the attention mask).

Tensor contract of this build (the DPO collator's layout): `queries [B,Q]` left-padded with one image token, `responses
[B,T]` right-padded with pad 0 — the labelled tokens are the non-pad response tokens (prompt tokens carry -100 in the
reference's collator, utils/data_utils_sft.py).  Everything on the GPU path runs on the HIP kernels: vision_train.VisionTrainPath
(trainable CLIP + projector LoRA), LlavaEngine.seq_logprobs_fwd/bwd (LLM LoRA, d_feats), optim.FlatAdamW x2 with ONE shared
clipping norm; data parallel = the same flat-gradient exchange as the DPO stage, once per buffer.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .dims import PAD_ID
from .model import LlavaEngine, LoraAdapter
from .optim import FlatAdamW
from .policy import AutoregressivePolicy
from .vision_train import VisionLoraAdapter, VisionTrainPath


def sft_batches_from_dpo_batch(batch: Dict[str, torch.Tensor]):
    """utils/data_utils_sft.py:187-246: every OPA dataset row (query, standard_response, AI_pseudo_response, image) yields TWO
    single-turn SFT samples, human = query, gpt = the GPT-4V corrected response resp. the AI pseudo response (the two copies of
    the dataset are concatenated and shuffled there).  Given a collated DPO batch (data.DataCollatorForCausalLM: left-padded
    queries, right-padded responses) this returns the two SFT batches in this build's tensor contract."""
    common = dict(images=batch["images"], queries=batch["queries"],
                  queries_attn_masks=batch.get("queries_attention_mask", batch.get("queries_attn_masks")))
    return [dict(common, responses=batch["standard_response"]), dict(common, responses=batch["AI_pseudo_response"])]


class SFTTrainer:
    def __init__(self, engine: LlavaEngine, llm_adapter: LoraAdapter, vis_adapter: VisionLoraAdapter, *, response_len: int,
                 lr: float = 2e-5, max_grad_norm: Optional[float] = 1.0, weight_decay: float = 0.0, optimizer_mode: str = "allreduce",
                 entropy_loss: bool = False, entropy_mask_ratio: float = 0.2, entropy_mask_method: str = "random",
                 entropy_loss_coef: float = 1.0, entropy_decay_coef: float = 1.0):
        assert llm_adapter.trainable
        self.entropy_loss, self.entropy_mask_ratio, self.entropy_mask_method = entropy_loss, entropy_mask_ratio, entropy_mask_method
        self.entropy_loss_coef, self.entropy_decay_coef = entropy_loss_coef, entropy_decay_coef
        self.last = {}
        self.engine, self.llm, self.vis = engine, llm_adapter, vis_adapter
        self.vision = VisionTrainPath(engine.base, vis_adapter)
        self._policy = AutoregressivePolicy(engine, llm_adapter, response_len)      # batch building only
        kw = dict(lr=lr, max_grad_norm=max_grad_norm, weight_decay=weight_decay, mode=optimizer_mode)
        self.opt_llm = FlatAdamW(llm_adapter.master, llm_adapter.grad, llm_adapter.work, **kw)
        self.opt_vis = FlatAdamW(vis_adapter.master, vis_adapter.grad, vis_adapter.work, **kw)

    def loss_and_backward(self, images: torch.Tensor, queries: torch.Tensor, queries_attn_masks: torch.Tensor,
                          responses: torch.Tensor, loss_scale: float = 1.0, masked_images: Optional[torch.Tensor] = None,
                          image_key_mask: Optional[torch.Tensor] = None) -> float:
        """CE over the non-pad response tokens (+ the entropy regulariser); accumulates into both flat gradients.  Returns the
        (unscaled) loss; self.last holds base_sft_loss / mask_sft_loss / entropy_loss like the reference's log (opa_trainer.py:92-94)."""
        eng, d, dev = self.engine, self.engine.d, self.engine.dev
        B = queries.shape[0]
        _, batch = self._policy.build_batch(queries, queries_attn_masks, {"response": responses})
        mask = (responses.to(dev) != PAD_ID)
        n = mask.sum().clamp_min(1).float()
        feats, vsv = self.vision.forward(images)
        logp, ent, sv = eng.seq_logprobs_fwd(self.llm, batch, feats.view(B, d.n_patches, d.hidden), 1.0, train=True)
        loss = -(logp * mask).sum() / n
        self.last = {"base_sft_loss": float(loss), "mask_sft_loss": 0.0, "entropy_loss": 0.0}
        d_ent = None
        d_feats = torch.zeros(B, d.n_patches, d.hidden, dtype=torch.float32, device=dev)
        if self.entropy_loss:
            if self.entropy_mask_method == "attention" and masked_images is None:
                # opa_trainer.py:75-81: same pixels, 'entropy_mask_ratio' of the image KEYS dropped from the attention mask
                # (the reference hard-codes 1369 keys = a 518-px tower; here the tower's own patch count)
                from .losses import mask_percentage_per_row
                im = image_key_mask if image_key_mask is not None else mask_percentage_per_row(
                    torch.ones(B, d.n_patches, dtype=torch.bool), self.entropy_mask_ratio)
                _, batch2 = self._policy.build_batch(queries, torch.cat([im.to(queries_attn_masks.device), queries_attn_masks.bool()], 1),
                                                     {"response": responses})
                feats2, vsv2, d_feats2 = feats, None, d_feats          # one vision pass: both LLM backwards add into d_feats
            else:
                if masked_images is None:      # global CPU RNG, like the reference (mask_single_image draws torch.randperm)
                    from .losses import mask_single_image
                    masked_images = torch.stack([mask_single_image(images[i].unsqueeze(0).cpu(), self.entropy_mask_ratio,
                                                                   self.entropy_mask_method) for i in range(B)]).squeeze(1)
                batch2 = batch
                feats2, vsv2 = self.vision.forward(masked_images)
                d_feats2 = torch.zeros(B, d.n_patches, d.hidden, dtype=torch.float32, device=dev)
            logp2, ent2, sv2 = eng.seq_logprobs_fwd(self.llm, batch2, feats2.view(B, d.n_patches, d.hidden), 1.0, train=True)
            per_row = mask.sum(1).clamp_min(1).float()
            e_loss = (-((ent2 - ent) * mask).sum(1) / per_row).mean()
            self.entropy_loss_coef *= self.entropy_decay_coef                      # opa_trainer.py:119 (decay applied before use)
            coef = self.entropy_loss_coef
            self.last.update(mask_sft_loss=float(-(logp2 * mask).sum() / n), entropy_loss=float(e_loss))
            loss = loss + coef * e_loss
            w = (coef / B) * mask.float() / per_row[:, None]                       # d e_loss / d ent2 = -w, / d ent = +w
            d_ent = w * loss_scale
            eng.seq_logprobs_bwd(self.llm, sv2, torch.zeros_like(logp2), d_feats=d_feats2, d_ent=-d_ent)
            if vsv2 is not None:
                self.vision.backward(vsv2, d_feats2.view(B * d.n_patches, d.hidden))
        eng.seq_logprobs_bwd(self.llm, sv, -(mask.float() / n) * loss_scale, d_feats=d_feats, d_ent=d_ent)
        self.vision.backward(vsv, d_feats.view(B * d.n_patches, d.hidden))
        return float(loss)

    def optimizer_step(self, grad_accum_div: float = 1.0) -> float:
        """Clip by the global norm over BOTH buffers, AdamW, refresh the K-major copies.  Returns the pre-clip global norm."""
        self.opt_llm.prepare(grad_accum_div)
        self.opt_vis.prepare(grad_accum_div)
        FlatAdamW.share_sumsq(self.opt_llm, self.opt_vis)
        norm = float(self.opt_llm.sumsq.sqrt()) * self.opt_llm._grad_div
        self.opt_llm.apply()
        self.opt_vis.apply()
        self.opt_llm.zero_grad()
        self.opt_vis.zero_grad()
        self.llm.refresh_transposed()
        self.vis.refresh_transposed()
        return norm

    def save(self, directory: str, base_model_name_or_path: str = "") -> None:
        """`checkpoint-final/`-style PEFT adapter (adapter_model.bin + adapter_config.json) holding the LLM, CLIP and projector
        LoRA tensors: the file the DPO stage starts from (policy_model_name_or_path; its loader merges the vision part)."""
        import json
        import os
        os.makedirs(directory, exist_ok=True)
        state = self.llm.to_peft_state()
        state.update(self.vis.to_peft_state())
        torch.save(state, os.path.join(directory, "adapter_model.bin"))
        d = self.engine.d
        cfg = {"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": d.lora_r, "lora_alpha": d.lora_alpha, "lora_dropout": 0.0, "bias": "none",
               "target_modules": ["q_proj", "k_proj", "v_proj", "o_proj", "out_proj", "gate_proj", "up_proj", "down_proj", "fc1", "fc2",
                                  "mm_projector.0", "mm_projector.2"],
               "base_model_name_or_path": base_model_name_or_path, "inference_mode": True, "fan_in_fan_out": False}
        with open(os.path.join(directory, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
        loss = self.loss_and_backward(batch["images"], batch["queries"], batch["queries_attn_masks"], batch["responses"],
                                      masked_images=batch.get("masked_images"))
        return {"loss": loss, "grad_norm": self.optimizer_step(), **self.last}
