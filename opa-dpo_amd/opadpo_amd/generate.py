"""On-policy rollout: prefill + KV-cache sampling decode on the HIP kernels.

Replaces `policy.generate(inputs=queries, images=, attention_mask=, do_sample=True, max_new_tokens=response_len,
top_p, top_k, temperature, pad_token_id)` + `truncate_after_eos_with_padding(.., eos, pad, [1577, 29973])` of
opadpo/generator_models/online_generator.py:292-323 (HF generate: temperature -> top-k -> top-p -> multinomial;
finished rows emit pad; stop when every row hit EOS or max_new_tokens).  Replicas only: prompts are sharded by rank,
no collective (the reference's synced_gpus flag is only needed for ZeRO-3).

Queries are left-padded (utils/data_utils_online_gpt4v.py:100-107) so every row of a batch has the same length and
the new token of every row sits at the same position: one RoPE position and one cache slot per step.
"""
from __future__ import annotations

import types
from typing import List, Optional, Sequence

import torch

from . import lib as L
from .dims import EOS_ID, IMAGE_TOKEN_INDEX, PAD_ID
from .model import BF, LlavaEngine, LoraAdapter, Saved


def truncate_after_eos_with_padding(completions: torch.Tensor, eos_token_id: int, pad_token_id: int,
                                    additional_tokens: Optional[Sequence[int]] = None) -> torch.Tensor:
    """generator_models/generator.py:244-273: cut after the first EOS; every additional stop id present in the row
    overrides the cut position (later list entries win, even when they occur after EOS); the tail becomes pad."""
    rows = completions.tolist()
    for r, row in enumerate(rows):
        end = row.index(eos_token_id) if eos_token_id in row else None
        for tok in (additional_tokens or ()):
            if tok in row:
                end = row.index(tok)
        if end is not None:
            rows[r] = row[: end + 1] + [pad_token_id] * (len(row) - end - 1)
    return torch.tensor(rows, dtype=torch.long, device=completions.device)


class _WeightsOnlyAdapter:
    """A 'merged adapter' that only carries re-laid-out base weights (the SwiGLU-pair copy of gate|up): nothing trainable, no LoRA."""
    trainable = False

    def __init__(self, merged):
        self.merged = merged


class Generator:
    def __init__(self, engine: LlavaEngine, adapter: Optional[LoraAdapter] = None, use_graph: Optional[bool] = None, merge_adapter: bool = False,
                 fuse_swiglu: bool = False):
        """use_graph: replay one captured decode step per token (hipGraph) or launch the step's ~230 kernels one by one.  None = the
        faster form of the engine: the op-level engine launches from Python (20 us per launch > the step's GPU time) and needs the graph;
        the context launches from a C++ loop that keeps the queue full, where plain launches measured FASTER than graph replay (7B,
        B = 4: 3.45 vs 3.72 ms per step, B = 64: see DESIGN.md section 6 - graph nodes pay ~1 us more per kernel boundary).
        merge_adapter: fold a FROZEN adapter into its own bf16 copy of the projections (LoraAdapter.merge_into_base) - the
        rollout / evaluation policy does not change while it generates, so the 4 LoRA down-projections and the K-concatenated
        tails of every layer and step disappear (13 -> 8 launches per layer; the gate|up projection fuses SwiGLU)."""
        if use_graph is None:
            use_graph = not hasattr(engine, "ctx")
        self.engine, self.adapter, self.use_graph = engine, adapter, use_graph
        if merge_adapter and adapter is not None and not adapter.trainable and adapter.merged is None:
            adapter.merge_into_base(engine.base)
        d = engine.d
        if adapter is None and fuse_swiglu and d.ffn % 128 == 0:
            # adapter-free rollout (the shipped config): a second copy of gate|up with its rows arranged per 128 as [64 gate | 64 up]
            # lets the projection's epilogue apply SwiGLU (lib.ACT_SWIGLU_PAIR) - one launch and one [B, 2F] round trip less per
            # layer and step.  Carried like a merged adapter (weights only, nothing trainable).
            F, H = d.ffn, d.hidden
            sw = [{"wgu_sw": torch.stack([w["wgu"][:F].view(F // 64, 64, H), w["wgu"][F:].view(F // 64, 64, H)], dim=1)
                   .reshape(2 * F, H).contiguous()} for w in engine.base.layers]
            self.adapter = _WeightsOnlyAdapter(sw)

    @torch.no_grad()
    def generate(self, queries: torch.Tensor, query_attn_masks: torch.Tensor, images: Optional[torch.Tensor] = None, *,
                 image_feats: Optional[torch.Tensor] = None, max_new_tokens: int, temperature: float = 1.0, top_k: int = 0,
                 top_p: float = 1.0, seed: int = 0, eos_token_id: int = EOS_ID, pad_token_id: int = PAD_ID,
                 suppress_eos: bool = False) -> torch.Tensor:
        """-> responses [B, max_new_tokens] int64 (pad after a row finished)."""
        eng, d, b = self.engine, self.engine.d, self.engine.base
        dev = eng.dev
        if hasattr(eng, "ctx"):      # CtxEngine: prefill, KV cache, the per-token launch loop (optionally a hipGraph) and the sampler live below the C ABI
            if image_feats is None:
                image_feats = eng.encode_images(images)
            return eng.generate(self.adapter, queries, query_attn_masks, image_feats, max_new_tokens=max_new_tokens, temperature=temperature,
                                top_k=top_k, top_p=top_p, seed=seed, eos_token_id=eos_token_id, pad_token_id=pad_token_id,
                                suppress_eos=suppress_eos, use_graph=self.use_graph)
        # op-level sequencing of the same kernels (LlavaEngine): kept as the cross-check of the context path (tests/test_ctx_gpu.py)
        st = L.stream()
        B, Q = queries.shape
        P, H, F, r, nh, hd, V = d.n_patches, d.hidden, d.ffn, d.lora_r, d.n_heads, d.head_dim, d.vocab
        Lp = Q + P - 1
        max_ctx = Lp + max_new_tokens
        if image_feats is None:
            image_feats = eng.encode_images(images)
        ids = queries.to(dev).to(torch.int32).contiguous()
        tmask = query_attn_masks.to(dev).to(torch.uint8).contiguous()
        feat_row = torch.arange(B, device=dev, dtype=torch.int32)
        e = lambda shape, dtype=BF: torch.empty(*shape, dtype=dtype, device=dev)
        # ---- prefill: full-sequence kernels, K/V of every layer copied into the cache -----------------------
        sv = Saved()
        M = B * Lp
        sv.x = e((3, M, H), torch.float32)
        sv.n1, sv.qkv, sv.t_qkv, sv.attn = e((1, M, H)), e((1, M, 3 * H)), e((1, M, 3 * r)), e((1, M, H))
        sv.lse, sv.t_o, sv.h, sv.n2 = e((1, B, nh, Lp), torch.float32), e((1, M, r)), e((1, M, H), torch.float32), e((1, M, H))
        sv.t_gu, sv.gu, sv.act, sv.t_d = e((1, M, 2 * r)), e((1, M, 2 * F)), e((1, M, F)), e((1, M, r))
        sv.rstd1, sv.rstd2 = e((1, M), torch.float32), e((1, M), torch.float32)
        key_mask = torch.zeros(B, max_ctx, dtype=torch.uint8, device=dev)
        km_prefill = e((B, Lp), torch.uint8)
        L.call("opadpo_embed_splice", L.ptr(ids), L.ptr(tmask), L.ptr(b.embed), L.ptr(image_feats.contiguous()), L.ptr(feat_row),
               None, L.ptr(sv.x[0]), 1, L.ptr(km_prefill), B, Q, P, H, IMAGE_TOKEN_INDEX, st)
        key_mask[:, :Lp] = km_prefill
        kc = e((d.n_layers, B, nh, max_ctx, hd))      # head-major: one (sequence, head) is one contiguous key stream
        vc = e((d.n_layers, B, nh, max_ctx, hd))

        def kv_hook(i, qkv):      # cache fill = strided device copy (plumbing)
            q5 = qkv.view(B, Lp, 3, nh, hd)
            kc[i, :, :, :Lp].copy_(q5[:, :, 1].transpose(1, 2))
            vc[i, :, :, :Lp].copy_(q5[:, :, 2].transpose(1, 2))

        cos, sin = b.rope_tables(max_ctx)
        Yp, yin = sv.x[2], None
        for i in range(d.n_layers):
            eng.layer_fwd(i, self.adapter, sv.x[0] if i == 0 else sv.h[0], yin, sv.x[i & 1], Yp, sv, 0, B, Lp, km_prefill, cos, sin, kv_hook)
            yin = Yp
        last = (torch.arange(B, device=dev, dtype=torch.int32) * Lp + (Lp - 1)).contiguous()
        # ---- decode: ONE step captured in a HIP graph and replayed per token -----------------------------------
        # Everything that changes from step to step lives in device memory (position / step counter,
        # current tokens, finished flags), so the 7-12 launches x n_layers of a step (7: adapter-free or merged adapter with the fused SwiGLU projection) are recorded once and
        # replayed with a single hipGraphLaunch (the eager loop was host-launch-bound: ~480 Python->C calls per token).
        key_mask[:, Lp:] = 1                       # future slots: valid as soon as ctx (device counter) reaches them
        x = e((B, H), torch.float32)
        x2 = e((B, H), torch.float32)
        hs, hn = e((B, H), torch.float32), e((B, H))
        n1, qkv, t_qkv, att, t_o = e((B, H)), e((B, 3 * H)), e((B, 3 * r)), e((B, H)), e((B, r))
        hb, n2, t_gu, gu, act, t_d = e((B, H), torch.float32), e((B, H)), e((B, 2 * r)), e((B, 2 * F)), e((B, F)), e((B, r))
        rstd = e((B,), torch.float32)
        emb = e((B, H))
        ws_bytes = int(L.load().opadpo_attn_decode_workspace_bytes(B, nh, hd, max_ctx))
        ws = torch.zeros(max(ws_bytes, 4), dtype=torch.uint8, device=dev)         # split-KV scratch
        logits = e((B, V), torch.float32)
        cur_tok = torch.zeros(B, dtype=torch.int32, device=dev)
        finished = torch.zeros(B, dtype=torch.uint8, device=dev)
        history = torch.full((max_new_tokens, B), pad_token_id, dtype=torch.int32, device=dev)
        step_d = torch.zeros(1, dtype=torch.int32, device=dev)                     # tokens sampled so far
        pos_d = torch.full((1,), Lp - 1, dtype=torch.int32, device=dev)            # position of the newest cached key
        s = d.lora_scale
        ad = self.adapter
        eos_arg = -1 if suppress_eos else eos_token_id

        def head(src_f32, add_f32=None):
            if add_f32 is not None:      # prefill: last position x = h + y (down-projection product, residual deferred)
                L.call("opadpo_rmsnorm_sum_fwd", L.ptr(src_f32), 1, L.ptr(add_f32), 1, B * H, L.ptr(b.norm), L.ptr(x2), L.ptr(hn), L.ptr(rstd), B, H, d.rms_eps, L.stream())
            else:
                L.call("opadpo_rmsnorm_fwd", L.ptr(src_f32), 1, L.ptr(b.norm), L.ptr(hn), L.ptr(rstd), B, H, d.rms_eps, L.stream())
            L.gemm_nt(hn, b.lm_head, logits)
            if suppress_eos:
                logits[:, eos_token_id] = float("-inf")
            L.call("opadpo_sample", L.ptr(logits), V, B, V, float(temperature), int(top_k), float(top_p), int(seed), 0,
                   L.ptr(step_d), L.ptr(finished), pad_token_id, eos_arg, L.ptr(cur_tok), L.ptr(history), L.stream())
            step_d.add_(1)

        def decode_step():
            st = L.stream()
            pos_d.add_(1)
            L.call("opadpo_gather_rows", L.ptr(b.embed), H, L.ptr(cur_tok), L.ptr(emb), B, H, st)
            cur, nx = emb, x
            for i in range(d.n_layers):
                w = b.layers[i]
                ad = self.adapter
                if ad is not None and ad.merged is not None:      # frozen adapter folded into its own weight copy
                    w, ad = dict(w, **ad.merged[i]), None
                L.call("opadpo_rmsnorm_fwd", L.ptr(cur), int(cur.dtype == torch.float32), L.ptr(w["ln1"]), L.ptr(n1), L.ptr(rstd), B, H, d.rms_eps, st)
                if ad is not None:
                    L.gemm_nt(n1, ad.w(i, "a_qkv"), t_qkv, alpha=s)
                    L.gemm_nt(n1, w["wqkv"], qkv, a2=t_qkv, b2=ad.w(i, "b_qkv"), a2_group_n=H, a2_group_stride=r)
                else:
                    L.gemm_nt(n1, w["wqkv"], qkv)
                # RoPE of q / k at the device-resident position, KV append and attention over keys 0..pos: one launch
                L.call("opadpo_attn_decode_fused", L.ptr(qkv), 3 * H, L.ptr(cos), L.ptr(sin), L.ptr(kc[i]), L.ptr(vc[i]), L.ptr(att),
                       L.ptr(key_mask), B, nh, hd, L.ptr(pos_d), max_ctx, hd ** -0.5, L.ptr(ws), ws_bytes, st)
                if ad is not None:
                    L.gemm_nt(att, ad.w(i, "a_o"), t_o, alpha=s)
                    L.gemm_nt(att, w["wo"], hb, a2=t_o, b2=ad.w(i, "b_o"), residual=cur)
                else:
                    L.gemm_nt(att, w["wo"], hb, residual=cur)
                eng.mlp_fwd(i, self.adapter, hb, nx, n2, t_gu, gu, act, t_d, rstd, B)
                cur, nx = nx, (x2 if nx is x else x)
            head(cur)

        L.call("opadpo_gather_rows", L.ptr(sv.h[0]), 2 * H, L.ptr(last), L.ptr(hs), B, 2 * H, st)
        L.call("opadpo_gather_rows", L.ptr(Yp), 2 * H, L.ptr(last), L.ptr(x), B, 2 * H, st)
        with L.decode_schedule():                  # M = B <= 64 GEMMs: weight-streaming kernel
            head(hs, x)                            # token 0 from the prefill logits
            graph = None
            for step in range(1, max_new_tokens):
                if step == 2 and self.use_graph and max_new_tokens > 3:
                    torch.cuda.synchronize()       # step 1 ran eagerly (warm-up of every kernel of the step)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        decode_step()              # recorded, not executed
                if graph is not None:
                    graph.replay()
                else:
                    decode_step()
                if step % 32 == 0 and bool(finished.all()):
                    break
        return history.t().contiguous().to(torch.int64)

    def rollout(self, queries, query_attn_masks, images, *, response_len: int, temperature: float = 1.0, top_k: int = 30,
                top_p: float = 0.95, seed: int = 0, additional_stop_ids: Sequence[int] = (1577, 29973)) -> torch.Tensor:
        """Online_Generator.rollout's tensor part (online_generator.py:292-323): sample, then cut at EOS / '?'."""
        resp = self.generate(queries, query_attn_masks, images, max_new_tokens=response_len, temperature=temperature,
                             top_k=top_k, top_p=top_p, seed=seed)
        return truncate_after_eos_with_padding(resp, EOS_ID, PAD_ID, list(additional_stop_ids))
