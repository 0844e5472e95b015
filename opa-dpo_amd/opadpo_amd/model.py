"""LLaVA-1.5 (CLIP-ViT-L/14 + mlp2x_gelu + Llama-2 with LoRA) forward / backward on MI355X.

This is the replacement for the third-party call ``self.base_model(**inputs)`` in
AutoregressivePolicy.forward (opadpo/dpo_models/rl_models.py:114-120) and for
``accelerator.backward(loss)`` (opadpo/dpo_models/rl_trainer.py:162): Python only sequences
kernel launches of libopadpo_hip.so on torch-owned device memory; there is no torch math on
the path and no CPU fallback.

Layout decisions (MI355X, 288 GB HBM3E):
  * one bf16 copy of the frozen base + a K-major transposed copy of every LLM weight so that
    forward AND dgrad are both "NT" MFMA GEMMs (no transposed-operand kernel variant);
  * q|k|v and gate|up are fused along N; every LoRA branch is fused into its base GEMM by
    K-concatenation (C = [x | t] . [W | B]^T, t = s * x A^T);
  * both adapters (lora_policy trainable, lora_ref_policy frozen) share the single base copy
    (opadpo/dpo_models/qlora_model.py:66-93); CLIP/projector LoRA is frozen and identical in both
    adapters (dpo_trainer.py:1022-1030) so it is merged into the vision weights at load time and
    image features are computed once per image;
  * all activations needed by backward stay resident in HBM (no gradient checkpointing).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import lib as L
from .dims import (IMAGE_TOKEN_INDEX, LLM_LINEARS, LLM_PREFIX, PEFT_PREFIX, VIS_LINEARS, VIS_PREFIX, LlavaDims,
                   llm_linear_shape)

BF = torch.bfloat16


_FUSE_ROPE = os.environ.get("OPADPO_FUSE_ROPE", "0") == "1"


def _dev(t: torch.Tensor, device, dtype=BF) -> torch.Tensor:
    return t.detach().to(device=device, dtype=dtype).contiguous()


class BaseWeights:
    """Frozen base model on one GPU (shared by every adapter)."""

    def __init__(self, dims: LlavaDims, state: Dict[str, torch.Tensor], device, *, need_backward: bool = True,
                 vision_lora: Optional[Dict[str, torch.Tensor]] = None):
        dims.validate()
        self.dims = d = dims
        self.device = device
        self.need_backward = need_backward
        g = lambda k: state[k]
        self.embed = _dev(g(LLM_PREFIX + "embed_tokens.weight"), device)
        self.norm = _dev(g(LLM_PREFIX + "norm.weight"), device)
        self.lm_head = _dev(g("lm_head.weight"), device)
        self.lm_head_t = self.lm_head.t().contiguous() if need_backward else None
        self.layers: List[dict] = []
        for i in range(d.n_layers):
            p = f"{LLM_PREFIX}layers.{i}."
            w = {}
            w["wqkv"] = _dev(torch.cat([g(p + f"self_attn.{n}_proj.weight") for n in "qkv"], 0), device)
            w["wo"] = _dev(g(p + "self_attn.o_proj.weight"), device)
            w["wgu"] = _dev(torch.cat([g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")], 0), device)
            w["wd"] = _dev(g(p + "mlp.down_proj.weight"), device)
            w["ln1"] = _dev(g(p + "input_layernorm.weight"), device)
            w["ln2"] = _dev(g(p + "post_attention_layernorm.weight"), device)
            if need_backward:
                for k in ("wqkv", "wo", "wgu", "wd"):
                    w[k + "_t"] = w[k].t().contiguous()
            self.layers.append(w)
        # ---- vision tower + projector (frozen in DPO; LoRA merged: W_eff = W + s * B A) ----------
        s = d.lora_scale

        def merged(key: str) -> torch.Tensor:
            w = g(key + ".weight").float()
            if vision_lora is not None:
                a = vision_lora.get(PEFT_PREFIX + key + ".lora_A.weight")
                if a is not None:
                    b = vision_lora[PEFT_PREFIX + key + ".lora_B.weight"]
                    w = w + s * (b.float() @ a.float())
            return w

        v = VIS_PREFIX
        pw = g(v + "embeddings.patch_embedding.weight").float().reshape(d.v_hidden, d.patch_k)
        pwp = torch.zeros(d.v_hidden, d.patch_kpad)
        pwp[:, : d.patch_k] = pw
        self.patch_w = _dev(pwp, device)
        self.cls = _dev(g(v + "embeddings.class_embedding"), device)
        self.pos = _dev(g(v + "embeddings.position_embedding.weight"), device)
        self.pre_ln_w = _dev(g(v + "pre_layrnorm.weight"), device)
        self.pre_ln_b = _dev(g(v + "pre_layrnorm.bias"), device)
        self.vlayers: List[dict] = []
        for j in range(d.v_used_layers):
            p = f"{v}encoder.layers.{j}."
            w = {}
            for ln in ("layer_norm1", "layer_norm2"):
                w[ln + "_w"] = _dev(g(p + ln + ".weight"), device)
                w[ln + "_b"] = _dev(g(p + ln + ".bias"), device)
            w["wqkv"] = _dev(torch.cat([merged(p + f"self_attn.{n}_proj") for n in "qkv"], 0), device)
            w["bqkv"] = _dev(torch.cat([g(p + f"self_attn.{n}_proj.bias") for n in "qkv"], 0), device)
            w["wo"] = _dev(merged(p + "self_attn.out_proj"), device)
            w["bo"] = _dev(g(p + "self_attn.out_proj.bias"), device)
            w["fc1"] = _dev(merged(p + "mlp.fc1"), device)
            w["b1"] = _dev(g(p + "mlp.fc1.bias"), device)
            w["fc2"] = _dev(merged(p + "mlp.fc2"), device)
            w["b2"] = _dev(g(p + "mlp.fc2.bias"), device)
            self.vlayers.append(w)
        self.proj0 = _dev(merged(LLM_PREFIX + "mm_projector.0"), device)
        self.proj0_b = _dev(g(LLM_PREFIX + "mm_projector.0.bias"), device)
        self.proj2 = _dev(merged(LLM_PREFIX + "mm_projector.2"), device)
        self.proj2_b = _dev(g(LLM_PREFIX + "mm_projector.2.bias"), device)
        self._rope_cache: Dict[int, tuple] = {}

    def rope_tables(self, Lmax: int):
        """cos/sin [L, hd/2] fp32 (HF Llama rotary: inv_freq = theta^(-2i/d), positions arange(L))."""
        if Lmax not in self._rope_cache:
            hd = self.dims.head_dim
            inv = 1.0 / (self.dims.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
            f = torch.outer(torch.arange(Lmax, dtype=torch.float32), inv)
            self._rope_cache[Lmax] = (f.cos().to(self.device).contiguous(), f.sin().to(self.device).contiguous())
        return self._rope_cache[Lmax]


# names of the fused LoRA blocks of one decoder layer, in flat-buffer order: (name, rows, cols)
def lora_blocks(d: LlavaDims):
    H, F, r = d.hidden, d.ffn, d.lora_r
    return [("a_qkv", 3 * r, H), ("b_qkv", 3 * H, r), ("a_o", r, H), ("b_o", H, r),
            ("a_gu", 2 * r, H), ("b_gu", 2 * F, r), ("a_d", r, F), ("b_d", H, r)]


# fused block -> [(peft module name, lora_A|lora_B, row offset, rows)]
def _peft_map(d: LlavaDims):
    H, F, r = d.hidden, d.ffn, d.lora_r
    return {
        "a_qkv": [("self_attn.q_proj", "lora_A", 0, r), ("self_attn.k_proj", "lora_A", r, r), ("self_attn.v_proj", "lora_A", 2 * r, r)],
        "b_qkv": [("self_attn.q_proj", "lora_B", 0, H), ("self_attn.k_proj", "lora_B", H, H), ("self_attn.v_proj", "lora_B", 2 * H, H)],
        "a_o": [("self_attn.o_proj", "lora_A", 0, r)], "b_o": [("self_attn.o_proj", "lora_B", 0, H)],
        "a_gu": [("mlp.gate_proj", "lora_A", 0, r), ("mlp.up_proj", "lora_A", r, r)],
        "b_gu": [("mlp.gate_proj", "lora_B", 0, F), ("mlp.up_proj", "lora_B", F, F)],
        "a_d": [("mlp.down_proj", "lora_A", 0, r)], "b_d": [("mlp.down_proj", "lora_B", 0, H)],
    }


class LoraAdapter:
    """LLM LoRA adapter as ONE flat buffer (layer-major, block order of lora_blocks()).

    trainable: fp32 master + fp32 grad (+ AdamW state owned by optim.FlatAdamW) + bf16 working copy
    + bf16 transposed copies for dgrad/wgrad.  frozen (reference): bf16 working copy only.
    """

    def __init__(self, dims: LlavaDims, peft_state: Dict[str, torch.Tensor], device, trainable: bool):
        self.dims = d = dims
        self.device = device
        self.trainable = trainable
        blocks = lora_blocks(d)
        self.layer_numel = sum(r * c for _, r, c in blocks)
        self.numel = self.layer_numel * d.n_layers
        build_dev = next(iter(peft_state.values())).device      # assemble where the tensors already live
        flat = torch.empty(self.numel, dtype=torch.float32, device=build_dev)
        self.offsets: List[Dict[str, tuple]] = []
        pm = _peft_map(d)
        off = 0
        for i in range(d.n_layers):
            lo = {}
            for name, rows, cols in blocks:
                view = flat[off: off + rows * cols].view(rows, cols)
                for mod, ab, r0, nr in pm[name]:
                    key = f"{PEFT_PREFIX}{LLM_PREFIX}layers.{i}.{mod}.{ab}.weight"
                    view[r0: r0 + nr] = peft_state[key].float()
                lo[name] = (off, rows, cols)
                off += rows * cols
            self.offsets.append(lo)
        if trainable:
            self.master = flat.to(device)
            self.grad = torch.zeros_like(self.master)
            self.work = self.master.to(BF)
            self.work_t = torch.empty_like(self.work)
            self._tjobs = None
        else:
            self.master = None
            self.grad = None
            self.work = flat.to(device=device, dtype=BF)
            self.work_t = None
        self.merged: Optional[List[Dict[str, torch.Tensor]]] = None      # frozen adapters only, see merge_into_base()
        if trainable:
            self.refresh_transposed()

    def merge_into_base(self, base: "BaseWeights") -> None:
        """FROZEN adapter (the reference policy): fold s * B @ A into a second bf16 copy of the projection weights, once, at
        load time (fp32 sum, one rounding to bf16) - PEFT's merge for an adapter that never changes.  The no-grad reference
        forward then runs the bare-model path: no K-concatenated tail (6 % of each projection GEMM) and no r-wide x @ A^T
        GEMMs (the least efficient shapes of the step); costs one extra copy of the LLM projections (13 GB at 7B).
        Mathematically the same function; the rounding differs (weights rounded after the sum instead of t = s x A^T rounded
        to bf16 before the tail): measured max |delta logp| in tests/test_parity_gpu.py::test_merged_reference_adapter."""
        assert not self.trainable, "only a frozen adapter can be merged"
        d, s = self.dims, self.dims.lora_scale
        H, F, r = d.hidden, d.ffn, d.lora_r
        out = []
        for i in range(d.n_layers):
            w = base.layers[i]
            def add(wkey, b_name, a_name, groups, i=i, w=w):
                # W'[g] = bf16(W[g] + s * B[g] @ A[g]) per fused group: one gemm_nt each (X = B[g] [out, r], Y = A[g]^T [in, r], fp32
                # accumulation, alpha = s, bf16 residual = W[g], bf16 output -> a single rounding)
                W = w[wkey]
                A, B = self.w(i, a_name), self.w(i, b_name)                        # A [G*r, in], B [G*out, r]
                Wm = torch.empty_like(W)
                per = W.shape[0] // groups
                for gi in range(groups):
                    At = A[gi * r:(gi + 1) * r].t().contiguous()                   # [in, r]
                    L.gemm_nt(B[gi * per:(gi + 1) * per], At, Wm[gi * per:(gi + 1) * per], residual=W[gi * per:(gi + 1) * per], alpha=s)
                return Wm
            wgu = add("wgu", "b_gu", "a_gu", 2)
            m = {"wqkv": add("wqkv", "b_qkv", "a_qkv", 3), "wo": add("wo", "b_o", "a_o", 1), "wd": add("wd", "b_d", "a_d", 1)}
            if F % 128 == 0:
                # rows per 128: [64 gate | 64 up] -> the projection GEMM's epilogue applies SwiGLU (lib.ACT_SWIGLU_PAIR): the no-grad
                # pass writes `act` directly, no [M, 2F] pre-activation tensor, no silu_mul kernel
                m["wgu_sw"] = torch.stack([wgu[:F].view(F // 64, 64, H), wgu[F:].view(F // 64, 64, H)], dim=1).reshape(2 * F, H).contiguous()
            else:
                m["wgu"] = wgu
            out.append(m)
        self.merged = out

    def w(self, layer: int, name: str) -> torch.Tensor:
        off, rows, cols = self.offsets[layer][name]
        return self.work[off: off + rows * cols].view(rows, cols)

    def g(self, layer: int, name: str) -> torch.Tensor:
        off, rows, cols = self.offsets[layer][name]
        return self.grad[off: off + rows * cols].view(rows, cols)

    def wt(self, layer: int, name: str) -> torch.Tensor:
        """K-major copies used by backward.  A-type blocks [G*r, in] -> [in, G*r]; B-type blocks
        [G*out, r] -> G stacked [r, out] (one per fused group)."""
        off, rows, cols = self.offsets[layer][name]
        if name.startswith("a_"):
            return self.work_t[off: off + rows * cols].view(cols, rows)
        r = self.dims.lora_r
        groups = {"b_qkv": 3, "b_gu": 2}.get(name, 1)
        return self.work_t[off: off + rows * cols].view(groups, r, rows // groups)

    def refresh_transposed(self) -> None:
        """Re-derive the transposed bf16 copies after an optimizer step: ONE batched launch over all blocks of all layers."""
        if self._tjobs is None:
            jobs, mt = [], 0
            for lo in self.offsets:
                for name, (off, rows, cols) in lo.items():
                    groups = 1 if name.startswith("a_") else {"b_qkv": 3, "b_gu": 2}.get(name, 1)
                    per = rows // groups
                    for gi in range(groups):
                        jobs.append([off + gi * per * cols, off + gi * per * cols, per, cols])
                        mt = max(mt, ((per + 63) // 64) * ((cols + 63) // 64))
            self._tjobs = (torch.tensor(jobs, dtype=torch.int64, device=self.device), len(jobs), mt)
        tj, n, mt = self._tjobs
        L.call("opadpo_transpose_batched", L.ptr(self.work), L.ptr(self.work_t), L.ptr(tj), n, mt, L.stream())

    def to_peft_state(self) -> Dict[str, torch.Tensor]:
        """PEFT-0.5 key layout, bf16 like the reference's saved adapters."""
        out = {}
        pm = _peft_map(self.dims)
        src = self.work   # always complete (ZeRO-1 keeps fp32 master only for the local shard)
        for i, lo in enumerate(self.offsets):
            for name, (off, rows, cols) in lo.items():
                view = src[off: off + rows * cols].view(rows, cols)
                for mod, ab, r0, nr in pm[name]:
                    out[f"{PEFT_PREFIX}{LLM_PREFIX}layers.{i}.{mod}.{ab}.weight"] = view[r0: r0 + nr].to(BF).cpu().clone()
        return out


def _dbg(name: str, t: torch.Tensor) -> None:
    """OPADPO_DEBUG_NAN=1: report non-finite intermediates (diagnostics only)."""
    import os
    if os.environ.get("OPADPO_DEBUG_NAN"):
        torch.cuda.synchronize()
        f = t.float()
        bad = int((~torch.isfinite(f)).sum())
        print(f"[dbg] {name}: shape={tuple(t.shape)} nonfinite={bad} absmax={float(f[torch.isfinite(f)].abs().max()) if bad < f.numel() else float('nan'):.4g}", flush=True)


@dataclass
class SeqBatch:
    """Device-side description of S stacked sequences (rl_models.py:95-112)."""
    ids: torch.Tensor          # [S, n_txt] int32 (query | response), one IMAGE_TOKEN_INDEX per row
    text_mask: torch.Tensor    # [S, n_txt] uint8
    feat_row: torch.Tensor     # [S] int32: which image's features each sequence uses
    image_mask: Optional[torch.Tensor]  # [S, P] uint8 (CoPO 'attention') or None
    T: int                     # response length
    K: int = 1                 # responses packed per row: ids = [query | response_0 | ... | response_{K-1}] sharing one
                               # pass over the image + query prefix (K = 1: one response per row, the reference's layout)
    row_plan: Optional[torch.Tensor] = None     # CPU int32 [S, K+1]: dropped leading pad positions, valid length of every response
                                                # (opadpo_seq_logprobs_fwd's ragged rows; None = padded rows)


class Saved:
    """Activations kept for backward (all resident in HBM)."""
    pass


class LlavaEngine:
    """Kernel sequencing for one GPU: vision encode, sequence log-probs forward, LoRA backward."""

    def __init__(self, base: BaseWeights):
        self.base = base
        self.d = base.dims
        self.dev = base.device
        self._ws: Dict[tuple, torch.Tensor] = {}

    # -- scratch management (torch owns the memory; buffers are reused across calls) ----------------
    def buf(self, tag: str, shape, dtype=BF) -> torch.Tensor:
        key = (tag, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = torch.empty(*shape, dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t

    def release(self) -> None:
        self._ws.clear()

    # -- vision: CLIP-ViT hidden_states[-2][:,1:] -> mlp2x_gelu ------------------------------------------
    @torch.no_grad()
    def encode_images(self, pixels: torch.Tensor) -> torch.Tensor:
        """pixels [B,3,S,S] (any float dtype) -> projected features [B, P, H] bf16."""
        d, b = self.d, self.base
        st = L.stream()
        B = pixels.shape[0]
        P, vh, vf = d.n_patches, d.v_hidden, d.v_ffn
        px = pixels.to(device=self.dev, dtype=BF).contiguous()
        cols = self.buf("v_cols", (B * P, d.patch_kpad))
        L.call("opadpo_im2col", L.ptr(px), L.ptr(cols), B, d.image_size, d.patch, d.patch_kpad, st)
        patches = self.buf("v_patches", (B * P, vh))
        L.gemm_nt(cols, b.patch_w, patches)
        T = P + 1
        M = B * T
        # fp32 residual stream through the tower (round 4; same sequence as opadpo_vision_encode, bit for bit)
        x = self.buf("v_x32", (M, vh), torch.float32)
        L.call("opadpo_vision_embed_f32", L.ptr(patches), L.ptr(b.cls), L.ptr(b.pos), L.ptr(x), B, P, vh, st)
        x2 = self.buf("v_x2_32", (M, vh), torch.float32)
        L.call("opadpo_layernorm_fwd_f32", L.ptr(x), L.ptr(b.pre_ln_w), L.ptr(b.pre_ln_b), L.ptr(x2), 1, M, vh, d.v_eps, st)
        x, x2 = x2, x
        n = self.buf("v_n", (M, vh))
        qkv = self.buf("v_qkv", (M, 3 * vh))
        att = self.buf("v_att", (M, vh))
        f1 = self.buf("v_f1", (M, vf))
        hd = vh // d.v_heads
        for w in b.vlayers:
            L.call("opadpo_layernorm_fwd_f32", L.ptr(x), L.ptr(w["layer_norm1_w"]), L.ptr(w["layer_norm1_b"]), L.ptr(n), 0, M, vh, d.v_eps, st)
            L.gemm_nt(n, w["wqkv"], qkv, bias=w["bqkv"])
            L.call("opadpo_attn_fwd", L.ptr(qkv), qkv.data_ptr() + 2 * vh, qkv.data_ptr() + 4 * vh, 3 * vh, L.ptr(att), vh,
                   None, None, B, T, d.v_heads, hd, 0, hd ** -0.5, 0, 0, st)
            L.gemm_nt(att, w["wo"], x2, bias=w["bo"], residual=x)
            L.call("opadpo_layernorm_fwd_f32", L.ptr(x2), L.ptr(w["layer_norm2_w"]), L.ptr(w["layer_norm2_b"]), L.ptr(n), 0, M, vh, d.v_eps, st)
            L.gemm_nt(n, w["fc1"], f1, bias=w["b1"], act=L.ACT_QUICK_GELU)
            L.gemm_nt(f1, w["fc2"], x, bias=w["b2"], residual=x2)
        # drop CLS
        idx = (torch.arange(B, device=self.dev, dtype=torch.int32)[:, None] * T
               + torch.arange(1, T, device=self.dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
        tok = self.buf("v_tok", (B * P, vh))
        L.call("opadpo_f32_to_bf16", L.ptr(x), L.ptr(n), M * vh, st)
        L.call("opadpo_gather_rows", L.ptr(n), vh, L.ptr(idx), L.ptr(tok), B * P, vh, st)
        h0 = self.buf("v_h0", (B * P, d.hidden))
        L.gemm_nt(tok, b.proj0, h0, bias=b.proj0_b, act=L.ACT_GELU)
        feats = torch.empty(B * P, d.hidden, dtype=BF, device=self.dev)
        L.gemm_nt(h0, b.proj2, feats, bias=b.proj2_b)
        return feats.view(B, P, d.hidden)

    # -- LLM ---------------------------------------------------------------------------------------
    def _alloc_saved(self, S: int, Lp: int, T: int, train: bool, K: int = 1) -> Saved:
        """Activation storage of one forward.  Fresh torch allocations (the caching allocator recycles
        them): a Saved object owns its tensors until backward has consumed it, so several training
        forwards (clean + CoPO-masked) can be alive at once."""
        d = self.d
        M = S * Lp
        H, F, r, nl = d.hidden, d.ffn, d.lora_r, d.n_layers
        sv = Saved()
        sv.S, sv.L, sv.T, sv.M, sv.train = S, Lp, T, M, train
        nb = nl if train else 1
        e = lambda shape, dtype=BF: torch.empty(*shape, dtype=dtype, device=self.dev)
        sv.x = e((nl + 1 if train else 3, M, H), torch.float32)     # fp32 residual stream x_0..x_{nl-1} (train) / two alternating slots, + the branch product Y
        sv.n1 = e((nb, M, H))
        sv.rstd1 = e((nb, M), torch.float32)
        sv.qkv = e((nb, M, 3 * H))
        sv.t_qkv = e((nb, M, 3 * r))
        sv.attn = e((nb, M, H))
        sv.lse = e((nb, S, d.n_heads, Lp), torch.float32)
        sv.t_o = e((nb, M, r))
        sv.h = e((nb, M, H), torch.float32)
        sv.n2 = e((nb, M, H))
        sv.rstd2 = e((nb, M), torch.float32)
        sv.t_gu = e((nb, M, 2 * r))
        sv.gu = e((nb, M, 2 * F))
        sv.act = e((nb, M, F))
        sv.t_d = e((nb, M, r))
        R = S * T * K
        sv.key_mask = e((S, Lp), torch.uint8)
        sv.hs = e((R, H), torch.float32)
        sv.hn = e((R, H))
        sv.rstd_f = e((R,), torch.float32)
        sv.logits_buf = e((R * max(d.vocab, d.hidden),), torch.float32)      # also parks the [R,H] head rows of the deferred branch product
        sv.logits = sv.logits_buf[:R * d.vocab].view(R, d.vocab)
        sv.lse_head = e((R,), torch.float32)
        return sv

    def layer_fwd(self, i: int, adapter: Optional[LoraAdapter], res, yin, x, Y, sv, k: int, S: int, Lp: int, key_mask, cos, sin,
                  kv_hook=None, seg=(0, 0)) -> None:
        """One Llama decoder layer over M = S*Lp rows (csrc/ctx.hip layer_fwd).  Input x = res (+ yin: the previous layer's
        down-projection product, its residual add deferred to this layer's norm; yin None: x is res).  Leaves h = x + attention
        branch in sv.h[k] and the layer's own down-projection product in Y (fp32, no residual in the GEMM epilogue: a residual read
        at the end of a round of 256x256 tiles costs +0.21-0.28 ms per projection, the norm kernel adds it at 6 TB/s).
        `adapter=None` runs the bare base model (the shipped rollout config has no LoRA: run/online_generate.sh
        POLICY_LORA_DIR=none).  sv.<buf>[k] are the activation buffers; kv_hook(i, qkv) sees the post-RoPE q|k|v (KV-cache fill)."""
        d, w = self.d, self.base.layers[i]
        mlp_adapter = adapter
        if adapter is not None and adapter.merged is not None:      # frozen adapter folded into its own weight copy
            w = dict(w, **adapter.merged[i])
            adapter = None
        st = L.stream()
        H, F, r, nh, hd = d.hidden, d.ffn, d.lora_r, d.n_heads, d.head_dim
        M = S * Lp
        s = d.lora_scale
        n1, qkv, t_qkv, attn, t_o, h, n2, t_gu, gu, act, t_d = (sv.n1[k], sv.qkv[k], sv.t_qkv[k], sv.attn[k], sv.t_o[k],
                                                                 sv.h[k], sv.n2[k], sv.t_gu[k], sv.gu[k], sv.act[k], sv.t_d[k])
        if yin is not None:
            L.call("opadpo_rmsnorm_sum_fwd", L.ptr(res), 1, L.ptr(yin), 1, M * H, L.ptr(w["ln1"]), L.ptr(x), L.ptr(n1), L.ptr(sv.rstd1[k]), M, H, d.rms_eps, st)
        else:
            assert res.data_ptr() == x.data_ptr()
            L.call("opadpo_rmsnorm_fwd", L.ptr(x), int(x.dtype == torch.float32), L.ptr(w["ln1"]), L.ptr(n1), L.ptr(sv.rstd1[k]), M, H, d.rms_eps, st)
        # rotary embedding fused into the projection's epilogue (each 128-column block of q and k is one head): OPT-IN
        # (OPADPO_FUSE_ROPE=1).  Measured neutral at the bench shape: the epilogue pays 7.7 us per block for the fp32 cos / sin rows
        # (8 B of table per 2 B of output, latency-bound inside a one-block-per-CU kernel) - as much as the in-place kernel it saves.
        fuse_rope = _FUSE_ROPE and hd == 128 and H % 256 == 0 and (seg[1] == 0 or seg[1] >= 4) and Lp >= 4
        if adapter is not None:
            L.gemm_nt(n1, adapter.w(i, "a_qkv"), t_qkv, alpha=s)
            if fuse_rope:
                L.gemm_nt_rope(n1, w["wqkv"], qkv, cos, sin, Lp, 2 * H, seg, a2=t_qkv, b2=adapter.w(i, "b_qkv"), a2_group_n=H, a2_group_stride=r)
            else:
                L.gemm_nt(n1, w["wqkv"], qkv, a2=t_qkv, b2=adapter.w(i, "b_qkv"), a2_group_n=H, a2_group_stride=r)
        elif fuse_rope:
            L.gemm_nt_rope(n1, w["wqkv"], qkv, cos, sin, Lp, 2 * H, seg)
        else:
            L.gemm_nt(n1, w["wqkv"], qkv)
        if not fuse_rope:
            L.call("opadpo_rope", L.ptr(qkv), 3 * H, L.ptr(cos), L.ptr(sin), M, Lp, 2 * nh, hd, 0, None, seg[0], seg[1], st)
        if kv_hook is not None:
            kv_hook(i, qkv)
        L.call("opadpo_attn_fwd", L.ptr(qkv), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, L.ptr(attn), H,
               L.ptr(sv.lse[k]), L.ptr(key_mask), S, Lp, nh, hd, L.CAUSAL_SKIP_MASKED_Q, hd ** -0.5, seg[0], seg[1], st)
        if adapter is not None:
            L.gemm_nt(attn, adapter.w(i, "a_o"), t_o, alpha=s)
            L.gemm_nt(attn, w["wo"], Y, a2=t_o, b2=adapter.w(i, "b_o"))
        else:
            L.gemm_nt(attn, w["wo"], Y)
        L.call("opadpo_rmsnorm_sum_fwd", L.ptr(x), 1, L.ptr(Y), 1, M * H, L.ptr(w["ln2"]), L.ptr(h), L.ptr(n2), L.ptr(sv.rstd2[k]), M, H, d.rms_eps, st)
        self.mlp_fwd(i, mlp_adapter, h, None, n2, t_gu, gu, act, t_d, sv.rstd2[k], M, ydef=Y, norm_done=True)

    def mlp_fwd(self, i: int, adapter: Optional[LoraAdapter], h, xo, n2, t_gu, gu, act, t_d, rstd2, M: int, ydef=None, norm_done: bool = False) -> None:
        """ydef: the down projection writes its product there WITHOUT the residual (full-sequence passes, see layer_fwd); None
        (decode steps): residual in the epilogue, result in xo."""
        d, w = self.d, self.base.layers[i]
        if adapter is not None and adapter.merged is not None:
            w = dict(w, **adapter.merged[i])
            adapter = None
        st = L.stream()
        H, F, r, s = d.hidden, d.ffn, d.lora_r, d.lora_scale
        if not norm_done:
            L.call("opadpo_rmsnorm_fwd", L.ptr(h), int(h.dtype == torch.float32), L.ptr(w["ln2"]), L.ptr(n2), L.ptr(rstd2), M, H, d.rms_eps, st)
        if adapter is not None:
            L.gemm_nt(n2, adapter.w(i, "a_gu"), t_gu, alpha=s)
            L.gemm_nt(n2, w["wgu"], gu, a2=t_gu, b2=adapter.w(i, "b_gu"), a2_group_n=F, a2_group_stride=r)
        elif "wgu_sw" in w:                 # merged frozen adapter: SwiGLU fused into the projection's epilogue
            L.gemm_nt(n2, w["wgu_sw"], act, act=L.ACT_SWIGLU_PAIR)
        else:
            L.gemm_nt(n2, w["wgu"], gu)
        if adapter is not None or "wgu_sw" not in w:
            L.call("opadpo_silu_mul_fwd", L.ptr(gu), L.ptr(act), M, F, st)
        if adapter is not None:
            L.gemm_nt(act, adapter.w(i, "a_d"), t_d, alpha=s)
            L.gemm_nt(act, w["wd"], xo if ydef is None else ydef, a2=t_d, b2=adapter.w(i, "b_d"), residual=h if ydef is None else None)
        else:
            L.gemm_nt(act, w["wd"], xo if ydef is None else ydef, residual=h if ydef is None else None)

    def seq_logprobs_fwd(self, adapter: LoraAdapter, batch: SeqBatch, feats: torch.Tensor, temperature: float,
                         train: bool):
        """S stacked sequences -> (logp [S,T] fp32, entropy [S,T] fp32, Saved).  Mirrors
        rl_models.py:114-132: one batched LM forward, logits[:, -T-1:-1] / temperature, labels =
        ids[:, -T:], log-softmax gather + entropy, both masked by response != pad."""
        d, b = self.d, self.base
        st = L.stream()
        S, n_txt = batch.ids.shape
        P, H, F, r, nh, hd = d.n_patches, d.hidden, d.ffn, d.lora_r, d.n_heads, d.head_dim
        Lp = n_txt + P - 1
        T, K = batch.T, batch.K
        M = S * Lp
        s = d.lora_scale
        pfx = Lp - K * T                              # image + query positions shared by the K responses of a row
        seg = (pfx, T) if K > 1 else (0, 0)
        sv = self._alloc_saved(S, Lp, T, train, K)
        sv.batch = batch
        sv.temperature = temperature
        sv.seg, sv.K = seg, K
        x0 = sv.x[0]
        L.call("opadpo_embed_splice", L.ptr(batch.ids), L.ptr(batch.text_mask), L.ptr(b.embed), L.ptr(feats),
               L.ptr(batch.feat_row), L.ptr(batch.image_mask), L.ptr(x0), 1, L.ptr(sv.key_mask), S, n_txt, P, H,
               IMAGE_TOKEN_INDEX, st)
        cos, sin = b.rope_tables(Lp)
        Y = sv.x[d.n_layers if train else 2]          # branch product of the o / down projections (residual deferred to the next norm)
        res, yin = x0, None
        for i in range(d.n_layers):
            k = i if train else 0
            self.layer_fwd(i, adapter, res, yin, sv.x[i if train else (i & 1)], Y, sv, k, S, Lp, sv.key_mask, cos, sin, seg=seg)
            res, yin = sv.h[k], Y
        # response rows: the position before each response token predicts it (rl_models.py:121-123: logits[:, -T-1:-1]).
        # Packed rows: token 0 of EVERY response is predicted from the last prefix position, token t >= 1 of response k
        # from position pfx + k*T + t - 1.  Output order is [k][s][t] = the reference's stacking of the response keys.
        R = K * S * T
        ar = lambda n: torch.arange(n, device=self.dev, dtype=torch.int32)
        off = pfx + ar(K)[:, None, None] * T + ar(T)[None, None, :] - 1            # [K,1,T]
        off = torch.where(ar(T)[None, None, :] == 0, torch.full_like(off, pfx - 1), off)
        rows = (ar(S)[None, :, None] * Lp + off).reshape(-1).contiguous()            # [K,S,T]
        labels = batch.ids[:, n_txt - K * T:].reshape(S, K, T).transpose(0, 1).contiguous().view(-1)
        sv.rows, sv.labels = rows, labels
        # final hidden state x = h + y of the HEAD rows only (fp32 rows = 2H bf16 units; the y rows park in the logits buffer)
        L.call("opadpo_gather_rows", L.ptr(res), 2 * H, L.ptr(rows), L.ptr(sv.hs), R, 2 * H, st)
        L.call("opadpo_gather_rows", L.ptr(Y), 2 * H, L.ptr(rows), L.ptr(sv.logits), R, 2 * H, st)
        L.call("opadpo_rmsnorm_sum_fwd", L.ptr(sv.hs), 1, L.ptr(sv.logits), 1, R * H, L.ptr(b.norm), L.ptr(sv.hs), L.ptr(sv.hn), L.ptr(sv.rstd_f), R, H, d.rms_eps, st)
        L.gemm_nt(sv.hn, b.lm_head, sv.logits)
        logp = torch.empty(R, dtype=torch.float32, device=self.dev)
        ent = torch.empty(R, dtype=torch.float32, device=self.dev)
        L.call("opadpo_head_fwd", L.ptr(sv.logits), d.vocab, L.ptr(labels), 1.0 / temperature, L.ptr(logp), L.ptr(ent),
               L.ptr(sv.lse_head), R, d.vocab, st)
        sv.ent = ent
        return logp.view(K * S, T), ent.view(K * S, T), sv

    def seq_logprobs_bwd(self, adapter: LoraAdapter, sv: Saved, dlogp: torch.Tensor, d_feats: Optional[torch.Tensor] = None,
                         d_ent: Optional[torch.Tensor] = None, layer_done=None) -> None:
        """Accumulate d(loss)/d(LoRA A,B) into adapter.grad given dlogp [S,T] fp32.  d_feats (optional, fp32 [n_images, P, H],
        accumulated; layer_done(i): called once layer i's LoRA gradients are queued - the data-parallel exchange of a finished
        bucket of layers starts there): gradient w.r.t. the projected image features — only the OPA LoRA-SFT stage needs it (trainable vision /
        projector LoRA, vision_train.py); the DPO stage stops at the frozen layer-0 input.  d_ent (optional, [S,T]): gradient
        w.r.t. the per-token entropies (SFT entropy regulariser)."""
        assert sv.train and adapter.trainable and self.base.need_backward
        d, b = self.d, self.base
        st = L.stream()
        S, Lp, T, M = sv.S, sv.L, sv.T, sv.M
        H, F, r, nh, hd, V = d.hidden, d.ffn, d.lora_r, d.n_heads, d.head_dim, d.vocab
        s = d.lora_scale
        R = S * T * sv.K
        seg = sv.seg
        dlogp = dlogp.to(device=self.dev, dtype=torch.float32).contiguous().view(-1)
        dz = self.buf("bw_dz", (R, V))
        if d_ent is not None:
            d_ent = d_ent.to(device=self.dev, dtype=torch.float32).contiguous().view(-1)
        L.call("opadpo_head_bwd", L.ptr(sv.logits), V, L.ptr(sv.labels), L.ptr(sv.lse_head), L.ptr(dlogp),
               L.ptr(sv.ent) if d_ent is not None else None, L.ptr(d_ent), 1.0 / sv.temperature, L.ptr(dz), V, R, V, st)
        d_hn = self.buf("bw_dhn", (R, H))
        L.gemm_nt(dz, b.lm_head_t, d_hn)
        d_hs = self.buf("bw_dhs", (R, H), torch.float32)
        L.call("opadpo_rmsnorm_bwd", L.ptr(d_hn), L.ptr(sv.hs), 1, L.ptr(b.norm), L.ptr(sv.rstd_f), None, 0, L.ptr(d_hs), None, R, H, st)
        dX = self.buf("bw_dX", (M, H), torch.float32)      # fp32 gradient residual stream ...
        dXb = self.buf("bw_dXb", (M, H))                    # ... and its bf16 copy (GEMM operand)
        dX.zero_()
        if sv.K > 1:      # the last prefix row feeds token 0 of every response: contributions add
            L.call("opadpo_scatter_add_rows_f32", L.ptr(d_hs), L.ptr(sv.rows), L.ptr(dX), H, R, H, st)
        else:
            L.call("opadpo_scatter_rows", L.ptr(d_hs), L.ptr(sv.rows), L.ptr(dX), 2 * H, R, 2 * H, st)
        L.call("opadpo_f32_to_bf16", L.ptr(dX), L.ptr(dXb), M * H, st)
        _dbg("dz", dz); _dbg("d_hn", d_hn); _dbg("d_hs", d_hs); _dbg("dX0", dX)
        d_h = self.buf("bw_dh", (M, H), torch.float32)
        d_hb = self.buf("bw_dhb", (M, H))
        d_n = self.buf("bw_dn", (M, H))
        d_act = self.buf("bw_dact", (M, F))
        d_gu = self.buf("bw_dgu", (M, 2 * F))
        d_attn = self.buf("bw_dattn", (M, H))
        dqkv = self.buf("bw_dqkv", (M, 3 * H))
        delta = self.buf("bw_delta", (S, nh, Lp), torch.float32)
        dt_r = self.buf("bw_dt_r", (M, r))
        dt_2r = self.buf("bw_dt_2r", (M, 2 * r))
        dt_3r = self.buf("bw_dt_3r", (M, 3 * r))
        cos, sin = b.rope_tables(Lp)
        for i in range(d.n_layers - 1, -1, -1):
            w = b.layers[i]
            dY, dYf = dXb, dX
            # ---- MLP -------------------------------------------------------------------------------
            L.gemm_nt(dY, adapter.wt(i, "b_d")[0], dt_r, alpha=s)
            L.gemm_tn(dY, sv.t_d[i], adapter.g(i, "b_d"))
            L.gemm_tn(dt_r, sv.act[i], adapter.g(i, "a_d"))
            L.gemm_nt(dY, w["wd_t"], d_act, a2=dt_r, b2=adapter.wt(i, "a_d"))
            L.call("opadpo_silu_mul_bwd", L.ptr(d_act), L.ptr(sv.gu[i]), L.ptr(d_gu), M, F, st)
            _dbg(f"L{i} d_act", d_act); _dbg(f"L{i} d_gu", d_gu)
            L.gemm_nt(d_gu, adapter.wt(i, "b_gu").view(2 * r, F), dt_2r, alpha=s, k1=F, a1_group_n=r, a1_group_stride=F)
            L.gemm_tn(d_gu, sv.t_gu[i], adapter.g(i, "b_gu"), q_group_n1=F, q_group_stride=r)
            L.gemm_tn(dt_2r, sv.n2[i], adapter.g(i, "a_gu"))
            L.gemm_nt(d_gu, w["wgu_t"], d_n, a2=dt_2r, b2=adapter.wt(i, "a_gu"))
            L.call("opadpo_rmsnorm_bwd", L.ptr(d_n), L.ptr(sv.h[i]), 1, L.ptr(w["ln2"]), L.ptr(sv.rstd2[i]), L.ptr(dYf), 1,
                   L.ptr(d_h), L.ptr(d_hb), M, H, st)
            _dbg(f"L{i} d_n2", d_n); _dbg(f"L{i} d_h", d_h)
            # ---- attention -------------------------------------------------------------------------
            L.gemm_nt(d_hb, adapter.wt(i, "b_o")[0], dt_r, alpha=s)
            L.gemm_tn(d_hb, sv.t_o[i], adapter.g(i, "b_o"))
            L.gemm_tn(dt_r, sv.attn[i], adapter.g(i, "a_o"))
            L.gemm_nt(d_hb, w["wo_t"], d_attn, a2=dt_r, b2=adapter.wt(i, "a_o"))
            qkv = sv.qkv[i]
            L.call("opadpo_attn_bwd", L.ptr(qkv), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, L.ptr(sv.attn[i]),
                   L.ptr(d_attn), H, L.ptr(sv.lse[i]), L.ptr(sv.key_mask), L.ptr(dqkv), dqkv.data_ptr() + 2 * H,
                   dqkv.data_ptr() + 4 * H, None, L.ptr(delta), S, Lp, nh, hd, L.CAUSAL_SKIP_MASKED_Q, hd ** -0.5, seg[0], seg[1], st)
            _dbg(f"L{i} d_attn", d_attn); _dbg(f"L{i} attn", sv.attn[i]); _dbg(f"L{i} lse", sv.lse[i]); _dbg(f"L{i} delta", delta)
            _dbg(f"L{i} dq", dqkv[:, :H]); _dbg(f"L{i} dk", dqkv[:, H:2 * H]); _dbg(f"L{i} dv", dqkv[:, 2 * H:])
            L.call("opadpo_rope", L.ptr(dqkv), 3 * H, L.ptr(cos), L.ptr(sin), M, Lp, 2 * nh, hd, 1, None, seg[0], seg[1], st)
            _dbg(f"L{i} dqkv(after rope)", dqkv)
            L.gemm_nt(dqkv, adapter.wt(i, "b_qkv").view(3 * r, H), dt_3r, alpha=s, k1=H, a1_group_n=r, a1_group_stride=H)
            L.gemm_tn(dqkv, sv.t_qkv[i], adapter.g(i, "b_qkv"), q_group_n1=H, q_group_stride=r)
            L.gemm_tn(dt_3r, sv.n1[i], adapter.g(i, "a_qkv"))
            if i > 0 or d_feats is not None:   # layer-0 input is the frozen embedding / image features: no further dgrad (DPO)
                L.gemm_nt(dqkv, w["wqkv_t"], d_n, a2=dt_3r, b2=adapter.wt(i, "a_qkv"))
                L.call("opadpo_rmsnorm_bwd", L.ptr(d_n), L.ptr(sv.x[i]), 1, L.ptr(w["ln1"]), L.ptr(sv.rstd1[i]), L.ptr(d_h), 1,
                       L.ptr(dX), L.ptr(dXb), M, H, st)
            if layer_done is not None:           # every wgrad of layer i is queued: its slice of the flat gradient is final
                layer_done(i)
        if d_feats is not None:
            # splice backward: rows [img_pos, img_pos + P) of every sequence hold its image's features (dims.IMAGE_TOKEN_INDEX
            # sits at img_pos of the text ids); sequences sharing an image (stacked layout) accumulate.  Index plumbing in torch.
            P = d.n_patches
            ids = sv.batch.ids
            img_pos = (ids == IMAGE_TOKEN_INDEX).int().argmax(dim=1)                                   # [S]
            rows = (torch.arange(S, device=self.dev)[:, None] * Lp + img_pos[:, None] + torch.arange(P, device=self.dev)[None, :]).reshape(-1)
            g_rows = dX.view(M, H).index_select(0, rows).view(S, P, H)
            d_feats.index_add_(0, sv.batch.feat_row.long(), g_rows)
