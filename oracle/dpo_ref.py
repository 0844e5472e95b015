"""Oracle for the arithmetic the reference owns on the DPO hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain torch on CPU; every
function cites the reference lines it restates.  Golden vectors produced by the
reference's own code live in tests/golden/ref_*.npz.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

PAD_ID = 0  # tokenizer.pad_token_id == unk == 0 (opadpo/opadpo_train.py:680-698)
EOS_ID = 2
IMAGE_TOKEN_INDEX = -200  # utils/constants.py:28


# ---------------------------------------------------------------------------
# head ops
# ---------------------------------------------------------------------------
def compute_logprobs(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = PAD_ID) -> torch.Tensor:
    """utils/common_utils.py:112-118 — logp[b,t] = z[b,t,label] - logsumexp(z[b,t,:]);
    cells whose label == ignore_index give -0.0 (cross_entropy returns +0.0 there, negated)."""
    z = logits.float()
    lse = torch.logsumexp(z, dim=-1)
    picked = z.gather(-1, labels.clamp_min(0).unsqueeze(-1)).squeeze(-1)
    lp = picked - lse
    neg_zero = torch.full_like(lp, -0.0)
    return torch.where(labels == ignore_index, neg_zero, lp).to(logits.dtype)


def entropy_from_logits(logits: torch.Tensor) -> torch.Tensor:
    """opadpo/dpo_models/rl_models.py:128 — H = -sum softmax(z) * log_softmax(z)."""
    z = logits.float()
    logp = z - torch.logsumexp(z, dim=-1, keepdim=True)
    return (-(logp.exp() * logp).sum(-1)).to(logits.dtype)


def policy_head(logits_all: torch.Tensor, input_ids: torch.Tensor, query_len: int, response_len: int,
                temperature: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """rl_models.py:112-132 — slice logits[:, -T-1:-1] (next-token shift), divide by
    temperature, gather labels = input_ids[:, -T:], mask with response != pad."""
    T = response_len
    response_mask = input_ids[:, query_len:] != PAD_ID
    logits = logits_all[:, -T - 1:-1] / temperature
    labels = input_ids[:, -T:]
    lp = compute_logprobs(logits, labels, PAD_ID) * response_mask
    ent = entropy_from_logits(logits) * response_mask
    return lp, ent


def response_keys(kwargs: Dict[str, torch.Tensor]) -> List[str]:
    """rl_models.py:91-92 — which kwargs of Policy.forward are response id tensors."""
    return [k for k in kwargs
            if "response" in k and "_mask" not in k and "scores" not in k and "image_relations" not in k]


def stack_policy_inputs(queries: torch.Tensor, queries_attn_masks: torch.Tensor,
                        responses: Dict[str, torch.Tensor]):
    """rl_models.py:95-112 — per response key: ids = cat[query, response]; attention mask =
    (ids != pad) with the query part replaced by queries_attn_masks (or, when the query
    mask is 576+Q wide — CoPO 'attention' — cat[query_mask, response != pad]); keys are
    stacked on the batch dimension in dict order."""
    ids, masks = [], []
    Q = queries.size(1)
    for k in response_keys(responses):
        i = torch.cat([queries, responses[k]], dim=1)
        if Q == queries_attn_masks.size(1):
            m = i != PAD_ID
            m[:, :Q] = queries_attn_masks.bool()
        else:
            m = torch.cat([queries_attn_masks.bool(), responses[k] != PAD_ID], dim=1)
        ids.append(i)
        masks.append(m)
    return torch.cat(ids, 0), torch.cat(masks, 0)


# ---------------------------------------------------------------------------
# loss
# ---------------------------------------------------------------------------
@dataclass
class DPOConfig:
    """Loss hyper-parameters (defaults = run/train_opa_dpo.sh + configs/llava/llava_dpo.yaml)."""
    beta: float = 0.1
    label_smoothing: float = 0.0
    reference_free: bool = False
    f_divergence_type: str = "reverse_kl"  # | js_divergence | alpha_divergence
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
    """dpo_trainer.py:73-81 — exp clamped at floor(log(finfo.max)*1e4)/1e4."""
    cap = torch.floor(torch.log(torch.tensor(torch.finfo(v.dtype).max, dtype=v.dtype)) * 10 ** 4) / 10 ** 4
    return torch.exp(torch.clamp(v, max=cap.item()))


def dpo_loss(cfg: DPOConfig, pol_c, pol_r, ref_c, ref_r, chosen_scores=None, rejected_scores=None):
    """dpo_trainer.py:429-473 — TOKEN-level DPO (Quirk Q3): everything is [B,T] elementwise.
    returns (losses, beta*chosen_logratio, beta*rejected_logratio)."""
    if chosen_scores is None:
        chosen_scores = torch.ones_like(pol_c)
    if rejected_scores is None:
        rejected_scores = torch.ones_like(pol_r)
    use_ref = 0.0 if cfg.reference_free else 1.0
    cl = pol_c - use_ref * ref_c
    rl = pol_r - use_ref * ref_r
    if cfg.f_divergence_type == "alpha_divergence":
        a = cfg.alpha_divergence_coef
        logits = (_cap_exp(rl * -a) - _cap_exp(cl * -a)) / a
    else:
        logits = chosen_scores * cl - rejected_scores * rl
        if cfg.f_divergence_type == "js_divergence":
            logits = logits - (F.softplus(cl) - F.softplus(rl))
    if cfg.loss_type != "sigmoid":
        raise ValueError(f"Unknown loss type: {cfg.loss_type}.")
    losses = (-F.logsigmoid(cfg.beta * logits) * (1 - cfg.label_smoothing)
              - F.logsigmoid(-cfg.beta * logits) * cfg.label_smoothing)
    return losses, cfg.beta * cl, cfg.beta * rl


def masked_mean(values, mask, axis=None):
    """dpo_trainer.py:1040-1045."""
    if axis is not None:
        return (values * mask).sum(axis=axis, keepdim=True) / mask.sum(axis=axis, keepdim=True)
    return (values * mask).sum() / mask.sum()


def compute_policy_loss(cfg: DPOConfig, rollouts: Dict[str, torch.Tensor],
                        policy_out: Dict[str, torch.Tensor],
                        policy_out_masked: Optional[Dict[str, torch.Tensor]] = None):
    """dpo_trainer.py:475-802 given the policy forwards' outputs.

    ``policy_out`` holds ``{standard,original_generate,AI_pseudo}_response_logprobs`` (clean
    image, :573-580); ``policy_out_masked`` holds ``mask_{standard,AI_pseudo}_response_logprobs``
    (CoPO negative, :644-651).  Returns (loss, flat stats dict with the reference's keys).
    """
    pad = PAD_ID
    ref_std = rollouts["ref_base_standard_response_logprobs"]
    ref_org = rollouts["ref_base_original_generate_response_logprobs"]
    ref_ai = rollouts["ref_base_AI_pseudo_response_logprobs"]
    if cfg.detailed_report and (cfg.response_score or cfg.response_image_relation):
        org_scores = rollouts["original_generate_response_scores"]
        ai_scores = rollouts["AI_pseudo_response_scores"]
        ai_rel = rollouts["AI_pseudo_response_image_relations"]
    else:  # :524-528
        org_scores = (ref_org != pad).to(ref_org.dtype)
        ai_scores = (ref_ai != pad).to(ref_ai.dtype)
        ai_rel = ai_scores

    p_std = policy_out["standard_response_logprobs"]
    p_org = policy_out["original_generate_response_logprobs"]
    p_ai = policy_out["AI_pseudo_response_logprobs"]

    use_scores = cfg.detailed_report and cfg.response_score
    use_rel = cfg.detailed_report and cfg.response_image_relation

    l1, c1, r1 = dpo_loss(cfg, p_std, p_org, ref_std, ref_org)                                  # :583-588
    c1m = ref_std != pad
    r1m = ref_org != pad
    l2, c2, r2 = dpo_loss(cfg, p_ai, p_org, ref_ai, ref_org,                                    # :595-602
                          ai_scores if use_scores else None, org_scores if use_scores else None)
    c2m = ref_ai != pad
    r2m = r1m
    loss = l1.mean() * cfg.standard_pair_coef + l2.mean() * cfg.AI_pair_coef                     # :631
    c3m = r3m = c1m
    c4m = r4m = c2m

    std_lp, org_lp, ai_lp = p_std.detach(), p_org.detach(), p_ai.detach()
    if cfg.CoPO:
        assert policy_out_masked is not None
        pm_std = policy_out_masked["mask_standard_response_logprobs"]
        pm_ai = policy_out_masked["mask_AI_pseudo_response_logprobs"]
        rm_std = rollouts["ref_mask_standard_response_logprobs"]
        rm_ai = rollouts["ref_mask_AI_pseudo_response_logprobs"]
        l3, c3, r3 = dpo_loss(cfg, p_std, pm_std, ref_std, rm_std)                               # :665-670
        l4, c4, r4 = dpo_loss(cfg, p_ai, pm_ai, ref_ai, rm_ai,                                   # :673-680
                              ai_rel if use_rel else None, ai_rel if use_rel else None)
        std_mask_lp, ai_mask_lp = pm_std.detach(), pm_ai.detach()
        loss = loss + (l3.mean() * cfg.standard_pair_coef * cfg.CoPO_coef
                       + l4.mean() * cfg.AI_pair_coef * cfg.CoPO_coef)                           # :693
    else:
        std_mask_lp = ai_mask_lp = torch.zeros_like(std_lp)
        l3 = c3 = r3 = l4 = c4 = r4 = torch.zeros_like(loss)                                     # :700

    if cfg.AncPO:
        a = cfg.Anchor_value
        if cfg.mDPO_anchor:   # Quirk Q6: + on pairs 1,3 ; - on pairs 2,4   (:704-705)
            anc = (-F.logsigmoid(c1 - a) - F.logsigmoid(-c2 + a)
                   - F.logsigmoid(c3 - a) - F.logsigmoid(-c4 + a))
        else:
            anc = (c1 - a) ** 2 + (c2 - a) ** 2 + (c3 - a) ** 2 + (c4 - a) ** 2
        anc = anc.mean()
        loss = loss + anc * cfg.Anchor_coef
    else:
        anc = torch.zeros_like(loss)

    def mmean(v, m):
        return masked_mean(v, m).mean()

    def mmin(v, m):
        return (v * m + ~m * 1e9).min(dim=1).values.mean()

    def mmax(v, m):
        return (v * m + ~m * -1e9).max(dim=1).values.mean()

    org_m, ai_m, std_m = org_lp != 0.0, ai_lp != 0.0, std_lp != 0.0                             # Quirk Q4 (:730-732)
    logprobs = {}
    for suffix, fn in (("", mmean), ("_min", mmin), ("_max", mmax)):
        logprobs["original_logprobs" + suffix] = fn(org_lp, org_m)
        logprobs["standard_logprobs" + suffix] = fn(std_lp, std_m)
        logprobs["AI_logprobs" + suffix] = fn(ai_lp, ai_m)
        logprobs["standard_mask_logprobs" + suffix] = fn(std_mask_lp, std_m)
        logprobs["AI_mask_logprobs" + suffix] = fn(ai_mask_lp, ai_m)

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
    stats.update({"logprobs/" + k: v for k, v in logprobs.items()})
    return loss, {k: v.detach() for k, v in stats.items()}


def plain_pair_loss(cfg: DPOConfig, pol_c, pol_r, ref_c, ref_r):
    """The benchmark's unit of work: ONE (chosen, rejected) pair per image =
    pair-1 of compute_policy_loss with CoPO/AncPO off: loss = dpo_loss(...).mean() over all
    B*T cells including padding (Quirk Q3, dpo_trainer.py:631)."""
    losses, c, r = dpo_loss(cfg, pol_c, pol_r, ref_c, ref_r)
    return losses.mean(), c, r


# ---------------------------------------------------------------------------
# CoPO image masking, rollout truncation, sampling processors
# ---------------------------------------------------------------------------
def mask_single_image(base_image: torch.Tensor, mask_percentage: float, mask_method: str = "random") -> torch.Tensor:
    """dpo_trainer.py:83-109 — consumes torch's global CPU RNG exactly like the reference
    (one randperm(H*W) for 'random'; one randperm(H/14*W/14) for 'blockwise')."""
    image = base_image.clone()
    mean_value = image.mean()
    _, C, H, W = image.shape
    if mask_method == "random":
        n = int(H * W * mask_percentage)
        idx = torch.randperm(H * W)[:n]
        flat = image.view(C, -1)
        flat[:, idx] = mean_value
        return flat.view(1, C, H, W)
    if mask_method == "blockwise":
        bs = 14
        hb, wb = H // bs, W // bs
        n = int(hb * wb * mask_percentage)
        idx = torch.randperm(hb * wb)[:n]
        v = image.view(C, hb, bs, wb, bs)
        for i in idx:
            v[:, i // wb, :, i % wb, :] = mean_value
        return v.view(1, C, H, W)
    raise NotImplementedError(mask_method)


def mask_percentage_per_row(matrix: torch.Tensor, percentage: float) -> torch.Tensor:
    """dpo_trainer.py:119-125 (CoPO 'attention': drop image keys)."""
    n = int(matrix.size(1) * percentage)
    for i in range(matrix.size(0)):
        matrix[i, torch.randperm(matrix.size(1))[:n]] = False
    return matrix


def truncate_after_eos_with_padding(completions: torch.Tensor, eos_token_id: int, pad_token_id: int,
                                    additional_tokens: Optional[Sequence[int]] = None) -> torch.Tensor:
    """generator_models/generator.py:244-273 — cut after the first EOS; every additional stop
    id that is present OVERRIDES the cut (later list entries win, even if they occur later
    in the row than EOS); the rest of the row becomes pad."""
    rows = completions.tolist()
    for r, row in enumerate(rows):
        end = row.index(eos_token_id) if eos_token_id in row else None
        for tok in (additional_tokens or ()):
            if tok in row:
                end = row.index(tok)
        if end is not None:
            rows[r] = row[: end + 1] + [pad_token_id] * (len(row) - end - 1)
    return torch.tensor(rows, dtype=torch.long)


def sample_filter(logits: torch.Tensor, temperature: float, top_k: int, top_p: float) -> torch.Tensor:
    """HF generation order temperature -> top-k -> top-p (SURVEY.md B7;
    online_generator.py:292-309).  Returns filtered logits (-inf = removed)."""
    z = logits.float() / temperature
    if top_k and top_k > 0:
        k = min(top_k, z.size(-1))
        kth = torch.topk(z, k, dim=-1).values[..., -1:]
        z = z.masked_fill(z < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        s, idx = torch.sort(z, descending=False, dim=-1)
        cum = s.softmax(-1).cumsum(-1)
        remove = cum <= (1 - top_p)
        remove[..., -1:] = False
        z = z.masked_fill(remove.scatter(-1, idx, remove), float("-inf"))
    return z


def add_eos(ids: torch.Tensor, eos: int = EOS_ID, pad: int = PAD_ID) -> torch.Tensor:
    """utils/data_utils_dpo.py:75-88 — first pad cell of a right-padded row becomes EOS
    (rows with no pad are left alone)."""
    out = ids.clone()
    for r in range(out.size(0)):
        pads = (out[r] == pad).nonzero()
        if pads.numel():
            out[r, pads[0, 0]] = eos
    return out


def grad_accum_arith(rollout_batch_size, step_batch_size, rollout_per_device_batch_size,
                     step_per_device_batch_size, world_size):
    """opadpo/opadpo_train.py:383-433 — TrainingArguments.__post_init__ accumulation arithmetic."""
    assert rollout_batch_size % (rollout_per_device_batch_size * world_size) == 0
    assert step_batch_size % (step_per_device_batch_size * world_size) == 0
    return (rollout_batch_size // rollout_per_device_batch_size // world_size,
            step_batch_size // step_per_device_batch_size // world_size)
