"""AutoregressivePolicy-compatible wrapper over the HIP engine.

Same call contract as opadpo/dpo_models/rl_models.py:75-144:

    policy(images=, queries=, queries_attn_masks=, temperature=, **{<name>_response: ids [B,T]})
        -> {<name>_response_logprobs: [B,T], <name>_response_entropies: [B,T]}

with the same kwarg filter ("response" in key, no "_mask"/"scores"/"image_relations"), the same
stacking order of the response keys on the batch dimension and the same masking.  Outputs are
fp32 (the reference returns the model dtype, bf16 at run time).  The log-probs carry a grad_fn
when the adapter is trainable and torch grad mode is on; backward runs the HIP LoRA backward and
ACCUMULATES into adapter.grad (the flat fp32 gradient buffer), like autograd's .grad +=.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .dims import IMAGE_TOKEN_INDEX, PAD_ID
from .model import LlavaEngine, LoraAdapter, SeqBatch


def response_keys(kwargs) -> List[str]:
    return [k for k in kwargs
            if "response" in k and "_mask" not in k and "scores" not in k and "image_relations" not in k]


def host_row_plan(queries: torch.Tensor, queries_attn_masks: torch.Tensor, responses: Dict[str, torch.Tensor]):
    """Ragged-row plan of a collated batch, computed where the batch is born - on the HOST (data_utils_dpo.py:363-365 pads there):
    -> (lead int32 [B], {response key: valid length int32 [B]}), CPU tensors.  lead = leading masked query positions before the image
    token, valid length = position after the last non-pad token of a response.  Handing these to the policy (`row_lead=`, `row_lens=`)
    keeps the pass free of any device->host read; without them the policy derives the same numbers from the tensors it is given (on
    the host when they are host tensors, else with two small device reductions and ONE read)."""
    q, m = queries.cpu(), queries_attn_masks.cpu().bool()
    Q = q.shape[1]
    if m.shape[1] != Q:                                     # CoPO 'attention': [image mask | query mask]
        m = m[:, -Q:]
    img_pos = (q == IMAGE_TOKEN_INDEX).int().argmax(dim=1)
    lead = torch.minimum((m.int().cumsum(1) == 0).sum(1), img_pos).to(torch.int32)
    lens = {}
    for k, r in responses.items():
        r = r.cpu()
        lens[k] = (r.shape[1] - ((r != PAD_ID).flip(1).int().cumsum(1) == 0).sum(1)).to(torch.int32)
    return lead, lens


class _SeqLogprobs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, policy, batch, feats, temperature):
        logp, ent, saved = policy.engine.seq_logprobs_fwd(policy.adapter, batch, feats, temperature, train=True)
        ctx.policy, ctx.saved = policy, saved
        policy._pending_bwd += 1
        ctx.mark_non_differentiable(ent)
        return logp, ent

    @staticmethod
    def backward(ctx, dlogp, _dent):
        pol = ctx.policy
        pol._pending_bwd -= 1
        # the per-layer "gradient is final" hook only fires in the LAST pending backward pass of the micro-batch (CoPO runs two
        # policy forwards, clean + masked image: a layer's gradient is complete after both passes have left it)
        hook = pol.layer_done_hook if pol._pending_bwd == 0 else None
        pol.engine.seq_logprobs_bwd(pol.adapter, ctx.saved, dlogp, layer_done=hook)
        ctx.saved = None
        return None, None, None, None, None


class AutoregressivePolicy(torch.nn.Module):
    def __init__(self, engine: LlavaEngine, adapter: LoraAdapter, response_len: int, temperature: float = 1.0,
                 adapter_name: Optional[str] = None, pack_responses: bool = True):
        super().__init__()
        # pack_responses: the K response keys of a call become ONE row per sample, [image+query | r_0 | ... | r_{K-1}], with
        # segment-masked attention, so the shared image + query prefix (703 of 1087 positions at seq512) is computed once
        # instead of K times.  Same log-probs / gradients as the reference's K stacked sequences (causal attention makes the
        # prefix states independent of the response); False keeps the reference's layout.
        self.pack_responses = pack_responses
        self.engine = engine
        self.adapter = adapter
        self.adapter_name = adapter_name
        self.response_len = response_len
        self.temperature = temperature
        # autograd anchor: a leaf that requires grad so that the HIP forward gets a grad_fn
        self._anchor = torch.zeros(1, device=engine.dev, requires_grad=True) if adapter.trainable else None
        self._pending_bwd = 0             # training forwards whose backward has not run yet
        self.layer_done_hook = None       # callable(layer) set by the trainer on the gradient-sync micro-batch (optim.launch_bucket)

    def build_batch(self, queries, queries_attn_masks, responses: Dict[str, torch.Tensor], row_lead=None, row_lens=None):
        """-> (response keys, SeqBatch).  The reference policy and the trained policy of a step see the SAME collated tensors
        (dpo_trainer.py rollout() / compute_policy_loss): the device-side concatenations and - for ragged rows - the one host read of
        the row plan are done once per distinct input (cache on the engine, keyed by storage, version counter and shape)."""
        keys = response_keys(responses)
        if not keys:
            raise ValueError("no *response* tensors passed to the policy")
        if (row_lead is None) != (row_lens is None):      # a plan is a (lead, lens) PAIR: half a plan is dropped, the other half derived with it
            row_lead = row_lens = None
        # the caller's row plan is part of the key (a SeqBatch built from one plan must not be handed out for another, or for none)
        _tl = lambda x: tuple(x.tolist()) if hasattr(x, "tolist") else tuple(int(v) for v in x)      # host tensors / lists of S ints
        plan_key = None if row_lead is None else (_tl(row_lead), tuple(_tl(row_lens[k]) for k in keys))
        ck = (self.pack_responses, self.response_len, plan_key) + tuple(
            (t.data_ptr(), t._version, tuple(t.shape), str(t.device)) for t in [queries, queries_attn_masks] + [responses[k] for k in keys])
        cache = self.engine.__dict__.setdefault("_batch_cache", {})
        hit = cache.get(ck)
        if hit is not None:           # only the SeqBatch is shared: the key NAMES are the caller's (the same tensors may arrive under other names)
            return keys, hit[0]
        batch = self._build_batch(keys, queries, queries_attn_masks, responses, row_lead, row_lens)
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        cache[ck] = (batch, queries, queries_attn_masks, [responses[k] for k in keys])      # the inputs stay alive with their entry: a recycled address cannot alias it
        return keys, batch

    def _build_batch(self, keys, queries, queries_attn_masks, responses: Dict[str, torch.Tensor], row_lead=None, row_lens=None):
        dev = self.engine.dev
        ragged = getattr(self.engine, "ragged", False)
        if ragged and row_lead is None and not queries.is_cuda and all(not responses[k].is_cuda for k in keys):
            row_lead, row_lens = host_row_plan(queries, queries_attn_masks, {k: responses[k] for k in keys})     # host tensors: no device read at all
        d = self.engine.d
        B, Q = queries.shape
        queries = queries.to(dev)
        qm = queries_attn_masks.to(dev).bool()
        ids, masks = [], []
        image_mask = None
        P = d.n_patches
        if qm.size(1) == Q:                                   # rl_models.py:100-102
            qmask_txt = qm
        else:                                                 # CoPO 'attention' (:103-105): [image mask | query mask]
            assert qm.size(1) == P + Q
            qmask_txt = qm[:, P:]
            image_mask = qm[:, :P]
        for k in keys:
            r = responses[k].to(dev)
            ids.append(torch.cat([queries, r], dim=1))
            masks.append(torch.cat([qmask_txt, r != PAD_ID], dim=1))
        K = len(keys)
        T = responses[keys[0]].shape[1]
        assert T == self.response_len, "policy slices with args.response_len (rl_models.py:121-123, Quirk Q7)"
        plan_q = plan_r = None
        if ragged and row_lead is not None:
            # the collator's own numbers (host_row_plan): nothing is read back from the device
            plan_q = torch.as_tensor(row_lead, dtype=torch.int32).cpu()
            plan_r = [torch.as_tensor(row_lens[k], dtype=torch.int32).cpu() for k in keys]
        elif ragged:
            # ragged rows (ctx.CtxEngine): leading masked query positions before the image token and the trailing padding of every
            # response are not rows of the pass.  Fallback for callers that hand over DEVICE tensors without a plan: two small
            # reductions on the device + ONE host read per batch (cached on the batch).
            img_pos = (queries == IMAGE_TOKEN_INDEX).int().argmax(dim=1)
            lead = torch.minimum((qmask_txt.int().cumsum(1) == 0).sum(1), img_pos)
            vlen = [T - ((responses[k].to(dev) != PAD_ID).flip(1).int().cumsum(1) == 0).sum(1) for k in keys]
            plan_q, plan_r = lead.to(torch.int32), [v.to(torch.int32) for v in vlen]
        if self.pack_responses and K > 1:
            batch = SeqBatch(
                ids=torch.cat([queries] + [responses[k].to(dev) for k in keys], 1).to(torch.int32).contiguous(),
                text_mask=torch.cat([qmask_txt] + [responses[k].to(dev) != PAD_ID for k in keys], 1).to(torch.uint8).contiguous(),
                feat_row=torch.arange(B, device=dev, dtype=torch.int32),
                image_mask=None if image_mask is None else image_mask.to(torch.uint8).contiguous(),
                T=T, K=K)
            if plan_q is not None:
                batch.row_plan = torch.stack([plan_q] + plan_r, 1).cpu().contiguous()
            return batch
        batch = SeqBatch(
            ids=torch.cat(ids, 0).to(torch.int32).contiguous(),
            text_mask=torch.cat(masks, 0).to(torch.uint8).contiguous(),
            feat_row=torch.arange(B, device=dev, dtype=torch.int32).repeat(K).contiguous(),
            image_mask=None if image_mask is None else image_mask.to(torch.uint8).repeat(K, 1).contiguous(),
            T=T)
        if plan_q is not None:          # stacked layout: K sequences per sample, one response each
            batch.row_plan = torch.stack([plan_q.repeat(K), torch.cat(plan_r, 0)], 1).cpu().contiguous()
        return batch

    def forward(self, images: Optional[torch.Tensor] = None, queries: torch.Tensor = None,
                queries_attn_masks: torch.Tensor = None, temperature: Optional[float] = None,
                image_feats: Optional[torch.Tensor] = None, mode: Optional[str] = None, row_lead: Optional[torch.Tensor] = None,
                row_lens: Optional[Dict[str, torch.Tensor]] = None, **kwargs) -> Dict[str, torch.Tensor]:
        """row_lead / row_lens (optional, HOST tensors from `host_row_plan`): the ragged-row plan of this batch as the collator knows
        it; row_lens is keyed like the response kwargs (a key missing there falls back to the derived plan)."""
        if temperature is None:
            temperature = self.temperature
        if row_lens is not None and any(k not in row_lens for k in response_keys(kwargs)):
            row_lead = row_lens = None
        keys, batch = self.build_batch(queries, queries_attn_masks, kwargs, row_lead, row_lens)
        B = queries.shape[0]
        if image_feats is None:
            image_feats = self.engine.encode_images(images)   # once per image, shared by all response keys
        feats = image_feats.contiguous()
        if self.adapter.trainable and torch.is_grad_enabled():
            logp, ent = _SeqLogprobs.apply(self._anchor, self, batch, feats, float(temperature))
        else:
            with torch.no_grad():
                logp, ent, _ = self.engine.seq_logprobs_fwd(self.adapter, batch, feats, float(temperature), train=False)
        out = {}
        for i, k in enumerate(keys):
            out[k + "_logprobs"] = logp[i * B:(i + 1) * B]
            out[k + "_entropies"] = ent[i * B:(i + 1) * B]
        return out
