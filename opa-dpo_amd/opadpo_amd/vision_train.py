"""CLIP-ViT + mm_projector with TRAINABLE LoRA: forward that keeps activations and backward to the LoRA tensors.

Groundwork for the OPA LoRA-SFT stage (opadpo/opa_train.py trains LoRA on every nn.Linear of the model
except lm_head — find_all_linear_names, :177-190 — i.e. also CLIP's q/k/v/out_proj/fc1/fc2 and mm_projector.0/.2, where the
DPO stage keeps them frozen and this build merges them into the weights).  Same kernel set as the LLM path: the LoRA branch
is fused into the base GEMM by K-concatenation (`[x | t].[W | B]^T`, fused q|k|v with group-dependent tail columns), dgrad
runs through K-major copies of the frozen weights, LoRA wgrads are `gemm_tn` reductions into the flat fp32 gradient, plus
`layernorm_bwd`, `act_fwd/act_bwd` on stored pre-activations and the non-causal `attn_bwd`.  The patch embedding (a Conv2d,
not a Linear -> no LoRA) and the LayerNorm affine parameters are frozen, so the backward stops at the first block's input.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import lib as L
from .dims import LLM_PREFIX, PEFT_PREFIX, VIS_PREFIX, LlavaDims
from .model import BF, BaseWeights


def vision_lora_blocks(d: LlavaDims):
    vh, vf, r = d.v_hidden, d.v_ffn, d.lora_r
    return [("a_qkv", 3 * r, vh), ("b_qkv", 3 * vh, r), ("a_o", r, vh), ("b_o", vh, r),
            ("a_f1", r, vh), ("b_f1", vf, r), ("a_f2", r, vf), ("b_f2", vh, r)]


def projector_lora_blocks(d: LlavaDims):
    vh, H, r = d.v_hidden, d.hidden, d.lora_r
    return [("a_p0", r, vh), ("b_p0", H, r), ("a_p2", r, H), ("b_p2", H, r)]


def _vis_peft_map(d: LlavaDims):
    vh, r = d.v_hidden, d.lora_r
    return {
        "a_qkv": [("self_attn.q_proj", "lora_A", 0, r), ("self_attn.k_proj", "lora_A", r, r), ("self_attn.v_proj", "lora_A", 2 * r, r)],
        "b_qkv": [("self_attn.q_proj", "lora_B", 0, vh), ("self_attn.k_proj", "lora_B", vh, vh), ("self_attn.v_proj", "lora_B", 2 * vh, vh)],
        "a_o": [("self_attn.out_proj", "lora_A", 0, r)], "b_o": [("self_attn.out_proj", "lora_B", 0, vh)],
        "a_f1": [("mlp.fc1", "lora_A", 0, r)], "b_f1": [("mlp.fc1", "lora_B", 0, d.v_ffn)],
        "a_f2": [("mlp.fc2", "lora_A", 0, r)], "b_f2": [("mlp.fc2", "lora_B", 0, vh)],
    }


class VisionLoraAdapter:
    """CLIP (used layers) + projector LoRA as ONE flat buffer: fp32 master / grad, bf16 working copy, K-major bf16 copies."""

    def __init__(self, dims: LlavaDims, peft_state: Dict[str, torch.Tensor], device):
        self.dims = d = dims
        self.device = device
        vb, pb = vision_lora_blocks(d), projector_lora_blocks(d)
        self.numel = sum(r * c for _, r, c in vb) * d.v_used_layers + sum(r * c for _, r, c in pb)
        flat = torch.empty(self.numel, dtype=torch.float32)
        self.offsets: List[Dict[str, tuple]] = []
        pm = _vis_peft_map(d)
        off = 0
        for j in range(d.v_used_layers):
            lo = {}
            for name, rows, cols in vb:
                view = flat[off: off + rows * cols].view(rows, cols)
                for mod, ab, r0, nr in pm[name]:
                    view[r0: r0 + nr] = peft_state[f"{PEFT_PREFIX}{VIS_PREFIX}encoder.layers.{j}.{mod}.{ab}.weight"].float().cpu()
                lo[name] = (off, rows, cols)
                off += rows * cols
            self.offsets.append(lo)
        lo = {}
        for name, rows, cols in pb:
            mod = "mm_projector.0" if name.endswith("p0") else "mm_projector.2"
            ab = "lora_A" if name.startswith("a_") else "lora_B"
            flat[off: off + rows * cols].view(rows, cols).copy_(peft_state[f"{PEFT_PREFIX}{LLM_PREFIX}{mod}.{ab}.weight"].float().cpu())
            lo[name] = (off, rows, cols)
            off += rows * cols
        self.offsets.append(lo)                      # index v_used_layers = projector
        self.master = flat.to(device)
        self.grad = torch.zeros_like(self.master)
        self.work = self.master.to(BF)
        self.work_t = torch.empty_like(self.work)
        self.refresh_transposed()

    def _view(self, buf, layer, name):
        off, rows, cols = self.offsets[layer][name]
        return buf[off: off + rows * cols].view(rows, cols)

    def w(self, layer: int, name: str) -> torch.Tensor:
        return self._view(self.work, layer, name)

    def g(self, layer: int, name: str) -> torch.Tensor:
        return self._view(self.grad, layer, name)

    def wt(self, layer: int, name: str) -> torch.Tensor:
        """A-type [G*r, in] -> [in, G*r]; B-type [G*out, r] -> G stacked [r, out]."""
        off, rows, cols = self.offsets[layer][name]
        if name.startswith("a_"):
            return self.work_t[off: off + rows * cols].view(cols, rows)
        groups = 3 if name == "b_qkv" else 1
        return self.work_t[off: off + rows * cols].view(groups, self.dims.lora_r, rows // groups)

    def refresh_transposed(self) -> None:
        st = L.stream()
        for lo in self.offsets:
            for name, (off, rows, cols) in lo.items():
                src, dst = self.work[off: off + rows * cols], self.work_t[off: off + rows * cols]
                groups = 3 if name == "b_qkv" else 1
                per = rows // groups
                for gi in range(groups):
                    L.call("opadpo_transpose", L.ptr(src[gi * per * cols:]), L.ptr(dst[gi * per * cols:]), per, cols, st)


    def to_peft_state(self) -> Dict[str, torch.Tensor]:
        """PEFT key layout of the CLIP / projector LoRA tensors (what the DPO stage merges into the vision weights at load)."""
        out = {}
        pm = _vis_peft_map(self.dims)
        for j in range(self.dims.v_used_layers):
            for name in self.offsets[j]:
                view = self.w(j, name)
                for mod, ab, r0, nr in pm[name]:
                    out[f"{PEFT_PREFIX}{VIS_PREFIX}encoder.layers.{j}.{mod}.{ab}.weight"] = view[r0: r0 + nr].to(BF).cpu().clone()
        for name in self.offsets[self.dims.v_used_layers]:
            mod = "mm_projector.0" if name.endswith("p0") else "mm_projector.2"
            ab = "lora_A" if name.startswith("a_") else "lora_B"
            out[f"{PEFT_PREFIX}{LLM_PREFIX}{mod}.{ab}.weight"] = self.w(self.dims.v_used_layers, name).to(BF).cpu().clone()
        return out


class VisionTrainPath:
    """Forward with saved activations + LoRA backward for the vision tower and the projector."""

    def __init__(self, base: BaseWeights, adapter: VisionLoraAdapter):
        self.base, self.ad, self.d, self.dev = base, adapter, base.dims, base.device
        # K-major copies of the frozen weights for dgrad (built once; base was loaded WITHOUT merged vision LoRA)
        self.wt: List[dict] = [{k: w[k].t().contiguous() for k in ("wqkv", "wo", "fc1", "fc2")} for w in base.vlayers]
        self.proj0_t, self.proj2_t = base.proj0.t().contiguous(), base.proj2.t().contiguous()

    def _lora_linear(self, x, w, bias, a, b, out, t, *, groups=1, residual=None):
        """out = x W^T + bias + s (x A^T) B^T (+ residual); t = s x A^T is kept for the wgrad."""
        d = self.d
        L.gemm_nt(x, a, t, alpha=d.lora_scale)
        kw = dict(a2_group_n=w.shape[0] // groups, a2_group_stride=d.lora_r) if groups > 1 else {}
        L.gemm_nt(x, w, out, a2=t, b2=b, bias=bias, residual=residual, **kw)

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor):
        """pixels [B,3,S,S] -> (feats [B*P, H] bf16, saved)."""
        d, b, ad = self.d, self.base, self.ad
        st = L.stream()
        B = pixels.shape[0]
        P, vh, vf, r, H = d.n_patches, d.v_hidden, d.v_ffn, d.lora_r, d.hidden
        T = P + 1
        M = B * T
        hd = vh // d.v_heads
        e = lambda *shape, dtype=BF: torch.empty(*shape, dtype=dtype, device=self.dev)
        px = pixels.to(device=self.dev, dtype=BF).contiguous()
        cols = e(B * P, d.patch_kpad)
        L.call("opadpo_im2col", L.ptr(px), L.ptr(cols), B, d.image_size, d.patch, d.patch_kpad, st)
        patches = e(B * P, vh)
        L.gemm_nt(cols, b.patch_w, patches)
        x0 = e(M, vh)
        L.call("opadpo_vision_embed", L.ptr(patches), L.ptr(b.cls), L.ptr(b.pos), L.ptr(x0), B, P, vh, st)
        x = e(M, vh)
        L.call("opadpo_layernorm_fwd", L.ptr(x0), L.ptr(b.pre_ln_w), L.ptr(b.pre_ln_b), L.ptr(x), M, vh, d.v_eps, st)
        sv = {"B": B, "layers": []}
        for j, w in enumerate(b.vlayers):
            s = dict(x=x, n1=e(M, vh), qkv=e(M, 3 * vh), t_qkv=e(M, 3 * r), att=e(M, vh), lse=e(B, d.v_heads, T, dtype=torch.float32),
                     t_o=e(M, r), x2=e(M, vh), n2=e(M, vh), t_1=e(M, r), z1=e(M, vf), f1=e(M, vf), t_2=e(M, r))
            L.call("opadpo_layernorm_fwd", L.ptr(x), L.ptr(w["layer_norm1_w"]), L.ptr(w["layer_norm1_b"]), L.ptr(s["n1"]), M, vh, d.v_eps, st)
            self._lora_linear(s["n1"], w["wqkv"], w["bqkv"], ad.w(j, "a_qkv"), ad.w(j, "b_qkv"), s["qkv"], s["t_qkv"], groups=3)
            qkv = s["qkv"]
            L.call("opadpo_attn_fwd", L.ptr(qkv), qkv.data_ptr() + 2 * vh, qkv.data_ptr() + 4 * vh, 3 * vh, L.ptr(s["att"]), vh,
                   L.ptr(s["lse"]), None, B, T, d.v_heads, hd, 0, hd ** -0.5, 0, 0, st)
            self._lora_linear(s["att"], w["wo"], w["bo"], ad.w(j, "a_o"), ad.w(j, "b_o"), s["x2"], s["t_o"], residual=x)
            L.call("opadpo_layernorm_fwd", L.ptr(s["x2"]), L.ptr(w["layer_norm2_w"]), L.ptr(w["layer_norm2_b"]), L.ptr(s["n2"]), M, vh, d.v_eps, st)
            self._lora_linear(s["n2"], w["fc1"], w["b1"], ad.w(j, "a_f1"), ad.w(j, "b_f1"), s["z1"], s["t_1"])
            L.call("opadpo_act_fwd", L.ptr(s["z1"]), L.ptr(s["f1"]), M * vf, L.ACT_QUICK_GELU, st)
            x = e(M, vh)
            self._lora_linear(s["f1"], w["fc2"], w["b2"], ad.w(j, "a_f2"), ad.w(j, "b_f2"), x, s["t_2"], residual=s["x2"])
            sv["layers"].append(s)
        idx = (torch.arange(B, device=self.dev, dtype=torch.int32)[:, None] * T
               + torch.arange(1, T, device=self.dev, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
        tok = e(B * P, vh)
        L.call("opadpo_gather_rows", L.ptr(x), vh, L.ptr(idx), L.ptr(tok), B * P, vh, st)
        pj = self.d.v_used_layers
        sp = dict(idx=idx, tok=tok, t_p0=e(B * P, r), zp=e(B * P, H), h0=e(B * P, H), t_p2=e(B * P, r))
        self._lora_linear(tok, b.proj0, b.proj0_b, ad.w(pj, "a_p0"), ad.w(pj, "b_p0"), sp["zp"], sp["t_p0"])
        L.call("opadpo_act_fwd", L.ptr(sp["zp"]), L.ptr(sp["h0"]), B * P * H, L.ACT_GELU, st)
        feats = e(B * P, H)
        self._lora_linear(sp["h0"], b.proj2, b.proj2_b, ad.w(pj, "a_p2"), ad.w(pj, "b_p2"), feats, sp["t_p2"])
        sv["proj"] = sp
        return feats, sv

    def _lora_linear_bwd(self, dY, x, t, w_t, layer, a_name, b_name, dx, *, groups=1):
        """dY [M,N] -> LoRA grads (accumulated) and dx [M,K] = dY W + (s dY B) A."""
        d, ad = self.d, self.ad
        r, s = d.lora_r, d.lora_scale
        M = dY.shape[0]
        dt = torch.empty(M, groups * r, dtype=BF, device=self.dev)
        bt = ad.wt(layer, b_name)
        if groups > 1:
            n_g = dY.shape[1] // groups
            L.gemm_nt(dY, bt.view(groups * r, n_g), dt, alpha=s, k1=n_g, a1_group_n=r, a1_group_stride=n_g)
            L.gemm_tn(dY, t, ad.g(layer, b_name), q_group_n1=n_g, q_group_stride=r)
        else:
            L.gemm_nt(dY, bt[0], dt, alpha=s)
            L.gemm_tn(dY, t, ad.g(layer, b_name))
        L.gemm_tn(dt, x, ad.g(layer, a_name))
        if dx is not None:
            L.gemm_nt(dY, w_t, dx, a2=dt, b2=ad.wt(layer, a_name))

    @torch.no_grad()
    def backward(self, sv, d_feats: torch.Tensor) -> None:
        """Accumulate d(loss)/d(vision + projector LoRA) into adapter.grad given d_feats [B*P, H] (bf16 or fp32)."""
        d, b = self.d, self.base
        st = L.stream()
        B = sv["B"]
        P, vh, vf, H = d.n_patches, d.v_hidden, d.v_ffn, d.hidden
        T = P + 1
        M = B * T
        hd = vh // d.v_heads
        e = lambda *shape, dtype=BF: torch.empty(*shape, dtype=dtype, device=self.dev)
        dF = d_feats.to(device=self.dev, dtype=BF).contiguous()
        sp, pj = sv["proj"], d.v_used_layers
        d_h0 = e(B * P, H)
        self._lora_linear_bwd(dF, sp["h0"], sp["t_p2"], self.proj2_t, pj, "a_p2", "b_p2", d_h0)
        d_zp = e(B * P, H)
        L.call("opadpo_act_bwd", L.ptr(d_h0), L.ptr(sp["zp"]), L.ptr(d_zp), B * P * H, L.ACT_GELU, st)
        d_tok = e(B * P, vh)
        self._lora_linear_bwd(d_zp, sp["tok"], sp["t_p0"], self.proj0_t, pj, "a_p0", "b_p0", d_tok)
        dx = torch.zeros(M, vh, dtype=BF, device=self.dev)                    # CLS rows get no gradient from the projector
        L.call("opadpo_scatter_rows", L.ptr(d_tok), L.ptr(sp["idx"]), L.ptr(dx), vh, B * P, vh, st)
        for j in range(len(sv["layers"]) - 1, -1, -1):
            s, w, wt = sv["layers"][j], b.vlayers[j], self.wt[j]
            d_f1 = e(M, vf)
            self._lora_linear_bwd(dx, s["f1"], s["t_2"], wt["fc2"], j, "a_f2", "b_f2", d_f1)
            d_z1 = e(M, vf)
            L.call("opadpo_act_bwd", L.ptr(d_f1), L.ptr(s["z1"]), L.ptr(d_z1), M * vf, L.ACT_QUICK_GELU, st)
            d_n2 = e(M, vh)
            self._lora_linear_bwd(d_z1, s["n2"], s["t_1"], wt["fc1"], j, "a_f1", "b_f1", d_n2)
            d_x2 = e(M, vh)
            L.call("opadpo_layernorm_bwd", L.ptr(d_n2), L.ptr(s["x2"]), L.ptr(w["layer_norm2_w"]), L.ptr(dx), L.ptr(d_x2), M, vh, d.v_eps, st)
            d_att = e(M, vh)
            self._lora_linear_bwd(d_x2, s["att"], s["t_o"], wt["wo"], j, "a_o", "b_o", d_att)
            qkv = s["qkv"]
            dqkv = e(M, 3 * vh)
            delta = e(B, d.v_heads, T, dtype=torch.float32)
            L.call("opadpo_attn_bwd", L.ptr(qkv), qkv.data_ptr() + 2 * vh, qkv.data_ptr() + 4 * vh, 3 * vh, L.ptr(s["att"]),
                   L.ptr(d_att), vh, L.ptr(s["lse"]), None, L.ptr(dqkv), dqkv.data_ptr() + 2 * vh, dqkv.data_ptr() + 4 * vh,
                   None, L.ptr(delta), B, T, d.v_heads, hd, 0, hd ** -0.5, 0, 0, st)
            need_dx = j > 0                      # the first block's input depends on no trainable tensor
            d_n1 = e(M, vh) if need_dx else None
            self._lora_linear_bwd(dqkv, s["n1"], s["t_qkv"], wt["wqkv"], j, "a_qkv", "b_qkv", d_n1, groups=3)
            if need_dx:
                dx = e(M, vh)
                L.call("opadpo_layernorm_bwd", L.ptr(d_n1), L.ptr(s["x"]), L.ptr(w["layer_norm1_w"]), L.ptr(d_x2), L.ptr(dx), M, vh, d.v_eps, st)
