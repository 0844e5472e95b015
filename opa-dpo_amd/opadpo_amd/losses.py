"""Token-level DPO / CoPO / AncPO objective of OPA-DPO on [B,T] grids (device tensors).

Mirrors DPOTrainer.dpo_loss (opadpo/dpo_models/dpo_trainer.py:429-473) and
DPOTrainer.compute_policy_loss (:475-802) including their quirks (SURVEY.md Appendix A:
Q3 token-level pairing with .mean() over all B*T cells, Q4 value-based masks, Q6 anchor signs).
These grids are a few KB; the arithmetic runs as torch ops on the GPU tensors the HIP path
produced and autograd hands d(loss)/d(logp) to LlavaEngine.seq_logprobs_bwd.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .dims import PAD_ID


@dataclass
class DPOArgs:
    """The loss-relevant subset of TrainingArguments (opadpo/opadpo_train.py:148-458); defaults =
    run/train_opa_dpo.sh."""
    beta: float = 0.1
    label_smoothing: float = 0.0
    reference_free: bool = False
    f_divergence_type: str = "reverse_kl"
    alpha_divergence_coef: float = 1.0
    loss_type: str = "sigmoid"
    standard_pair_coef: float = 1.0
    AI_pair_coef: float = 1.0
    CoPO: bool = True
    CoPO_method: str = "random"
    CoPO_coef: float = 0.2
    CoPO_mask_ratio: float = 0.3
    AncPO: bool = True
    mDPO_anchor: bool = True
    Anchor_value: float = 0.0
    Anchor_coef: float = 1.0
    detailed_report: bool = True
    response_score: bool = True
    response_image_relation: bool = True
    temperature: float = 1.0


def _cap_exp(v: torch.Tensor) -> torch.Tensor:
    cap = torch.floor(torch.log(torch.tensor(torch.finfo(v.dtype).max, dtype=v.dtype)) * 10 ** 4) / 10 ** 4
    return torch.exp(torch.clamp(v, max=float(cap)))


def dpo_loss(a: DPOArgs, pol_c, pol_r, ref_c, ref_r, chosen_scores=None, rejected_scores=None):
    """dpo_trainer.py:429-473 -> (losses [B,T], beta*chosen_logratio, beta*rejected_logratio)."""
    if chosen_scores is None:
        chosen_scores = torch.ones_like(pol_c)
    if rejected_scores is None:
        rejected_scores = torch.ones_like(pol_r)
    use_ref = 0.0 if a.reference_free else 1.0
    cl = pol_c - use_ref * ref_c
    rl = pol_r - use_ref * ref_r
    if a.f_divergence_type == "alpha_divergence":
        c = a.alpha_divergence_coef
        logits = (_cap_exp(rl * -c) - _cap_exp(cl * -c)) / c
    else:
        logits = chosen_scores * cl - rejected_scores * rl
        if a.f_divergence_type == "js_divergence":
            logits = logits - (F.softplus(cl) - F.softplus(rl))
    if a.loss_type != "sigmoid":
        raise ValueError(f"Unknown loss type: {a.loss_type}.")
    losses = (-F.logsigmoid(a.beta * logits) * (1 - a.label_smoothing)
              - F.logsigmoid(-a.beta * logits) * a.label_smoothing)
    return losses, a.beta * cl, a.beta * rl


def masked_mean(values, mask, axis=None):
    if axis is not None:
        return (values * mask).sum(axis=axis, keepdim=True) / mask.sum(axis=axis, keepdim=True)
    return (values * mask).sum() / mask.sum()


def policy_loss(a: DPOArgs, rollouts: Dict[str, torch.Tensor], out: Dict[str, torch.Tensor],
                out_masked: Optional[Dict[str, torch.Tensor]] = None):
    """compute_policy_loss (:475-802) given the policy forwards' log-probs.  Returns (loss, stats) with the
    reference's 32 stat keys ('loss/..', 'policy/..', 'logprobs/..')."""
    pad = PAD_ID
    ref_std = rollouts["ref_base_standard_response_logprobs"]
    ref_org = rollouts["ref_base_original_generate_response_logprobs"]
    ref_ai = rollouts["ref_base_AI_pseudo_response_logprobs"]
    if a.detailed_report and (a.response_score or a.response_image_relation):
        org_scores = rollouts["original_generate_response_scores"]
        ai_scores = rollouts["AI_pseudo_response_scores"]
        ai_rel = rollouts["AI_pseudo_response_image_relations"]
    else:
        org_scores = (ref_org != pad).to(ref_org.dtype)
        ai_scores = (ref_ai != pad).to(ref_ai.dtype)
        ai_rel = ai_scores
    p_std = out["standard_response_logprobs"]
    p_org = out["original_generate_response_logprobs"]
    p_ai = out["AI_pseudo_response_logprobs"]
    use_scores = a.detailed_report and a.response_score
    use_rel = a.detailed_report and a.response_image_relation

    l1, c1, r1 = dpo_loss(a, p_std, p_org, ref_std, ref_org)
    c1m, r1m = ref_std != pad, ref_org != pad
    l2, c2, r2 = dpo_loss(a, p_ai, p_org, ref_ai, ref_org,
                          ai_scores if use_scores else None, org_scores if use_scores else None)
    c2m, r2m = ref_ai != pad, r1m
    loss = l1.mean() * a.standard_pair_coef + l2.mean() * a.AI_pair_coef
    c3m = r3m = c1m
    c4m = r4m = c2m
    std_lp, org_lp, ai_lp = p_std.detach(), p_org.detach(), p_ai.detach()
    if a.CoPO:
        pm_std = out_masked["mask_standard_response_logprobs"]
        pm_ai = out_masked["mask_AI_pseudo_response_logprobs"]
        l3, c3, r3 = dpo_loss(a, p_std, pm_std, ref_std, rollouts["ref_mask_standard_response_logprobs"])
        l4, c4, r4 = dpo_loss(a, p_ai, pm_ai, ref_ai, rollouts["ref_mask_AI_pseudo_response_logprobs"],
                              ai_rel if use_rel else None, ai_rel if use_rel else None)
        std_mask_lp, ai_mask_lp = pm_std.detach(), pm_ai.detach()
        loss = loss + (l3.mean() * a.standard_pair_coef * a.CoPO_coef + l4.mean() * a.AI_pair_coef * a.CoPO_coef)
    else:
        std_mask_lp = ai_mask_lp = torch.zeros_like(std_lp)
        l3 = c3 = r3 = l4 = c4 = r4 = torch.zeros_like(loss)
    if a.AncPO:
        v = a.Anchor_value
        if a.mDPO_anchor:
            anc = (-F.logsigmoid(c1 - v) - F.logsigmoid(-c2 + v) - F.logsigmoid(c3 - v) - F.logsigmoid(-c4 + v))
        else:
            anc = (c1 - v) ** 2 + (c2 - v) ** 2 + (c3 - v) ** 2 + (c4 - v) ** 2
        anc = anc.mean()
        loss = loss + anc * a.Anchor_coef
    else:
        anc = torch.zeros_like(loss)

    def mmean(v_, m):
        return masked_mean(v_, m).mean()

    def mmin(v_, m):
        return (v_ * m + ~m * 1e9).min(dim=1).values.mean()

    def mmax(v_, m):
        return (v_ * m + ~m * -1e9).max(dim=1).values.mean()

    org_m, ai_m, std_m = org_lp != 0.0, ai_lp != 0.0, std_lp != 0.0
    stats = {
        "loss/stand_gen": l1.mean(), "loss/AI_gen": l2.mean(), "loss/stand_mask": l3.mean(),
        "loss/AI_mask": l4.mean(), "loss/AncPO": anc,
        "policy/stand_gen_chosen_mean": mmean(c1, c1m), "policy/stand_gen_reject_mean": mmean(r1, r1m),
        "policy/stand_gen_gap_mean": mmean(c1, c1m) - mmean(r1, r1m),
        "policy/AI_gen_chosen_mean": mmean(c2, c2m), "policy/AI_gen_reject_mean": mmean(r2, r2m),
        "policy/AI_gen_gap_mean": mmean(c2, c2m) - mmean(r2, r2m),
        "policy/stand_mask_chosen_mean": mmean(c3, c3m), "policy/stand_mask_reject_mean": mmean(r3, r3m),
        "policy/stand_mask_gap_mean": mmean(c3, c3m) - mmean(r3, r3m),
        "policy/AI_mask_chosen_mean": mmean(c4, c4m), "policy/AI_mask_reject_mean": mmean(r4, r4m),
        "policy/AI_mask_gap_mean": mmean(c4, c4m) - mmean(r4, r4m),
    }
    for suffix, fn in (("", mmean), ("_min", mmin), ("_max", mmax)):
        stats["logprobs/original_logprobs" + suffix] = fn(org_lp, org_m)
        stats["logprobs/standard_logprobs" + suffix] = fn(std_lp, std_m)
        stats["logprobs/AI_logprobs" + suffix] = fn(ai_lp, ai_m)
        stats["logprobs/standard_mask_logprobs" + suffix] = fn(std_mask_lp, std_m)
        stats["logprobs/AI_mask_logprobs" + suffix] = fn(ai_mask_lp, ai_m)
    return loss, {k: v_.detach() for k, v_ in stats.items()}


def pair_loss(a: DPOArgs, pol_c, pol_r, ref_c, ref_r):
    """One (chosen, rejected) pair per image = pair 1 of policy_loss with CoPO/AncPO off — the
    benchmark's unit of work."""
    losses, c, r = dpo_loss(a, pol_c, pol_r, ref_c, ref_r)
    return losses.mean(), c, r


def mask_single_image(base_image: torch.Tensor, mask_percentage: float, mask_method: str = "random") -> torch.Tensor:
    """CoPO negative image (dpo_trainer.py:83-109): fill int(H*W*ratio) random pixel positions (all
    channels) with the image mean; draws torch.randperm from the global CPU RNG like the reference."""
    image = base_image.clone()
    mean_value = image.mean()
    _, Cn, H, W = image.shape
    if mask_method == "random":
        idx = torch.randperm(H * W)[: int(H * W * mask_percentage)].to(image.device)
        flat = image.view(Cn, -1)
        flat[:, idx] = mean_value
        return flat.view(1, Cn, H, W)
    if mask_method == "blockwise":
        bs = 14
        hb, wb = H // bs, W // bs
        idx = torch.randperm(hb * wb)[: int(hb * wb * mask_percentage)]
        v = image.view(Cn, hb, bs, wb, bs)
        for i in idx.tolist():
            v[:, i // wb, :, i % wb, :] = mean_value
        return v.view(1, Cn, H, W)
    raise NotImplementedError(mask_method)


def mask_percentage_per_row(matrix: torch.Tensor, percentage: float) -> torch.Tensor:
    """dpo_trainer.py:119-125 (CoPO 'attention')."""
    n = int(matrix.size(1) * percentage)
    for i in range(matrix.size(0)):
        matrix[i, torch.randperm(matrix.size(1))[:n].to(matrix.device)] = False
    return matrix
