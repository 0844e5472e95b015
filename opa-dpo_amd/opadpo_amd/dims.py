"""Model geometry of the LLaVA-1.5 family the OPA-DPO recipe trains."""
from __future__ import annotations

from dataclasses import dataclass

PAD_ID = 0                 # tokenizer.pad_token_id == unk (opadpo/opadpo_train.py:680-698)
EOS_ID = 2
IMAGE_TOKEN_INDEX = -200   # utils/constants.py:28

LLM_LINEARS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
               "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")
VIS_LINEARS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj",
               "mlp.fc1", "mlp.fc2")
LLM_PREFIX = "model."
VIS_PREFIX = "model.vision_tower.vision_tower.vision_model."
PEFT_PREFIX = "base_model.model."


@dataclass
class LlavaDims:
    hidden: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    head_dim: int = 128
    ffn: int = 11008
    vocab: int = 32000
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    v_hidden: int = 1024
    v_layers: int = 24          # checkpoint layers; v_layers-1 run (mm_vision_select_layer = -2)
    v_heads: int = 16
    v_ffn: int = 4096
    image_size: int = 336
    patch: int = 14
    v_eps: float = 1e-5
    lora_r: int = 256
    lora_alpha: float = 512.0

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def v_used_layers(self) -> int:
        return self.v_layers - 1

    @property
    def lora_scale(self) -> float:
        return self.lora_alpha / self.lora_r

    @property
    def patch_k(self) -> int:
        return 3 * self.patch * self.patch

    @property
    def patch_kpad(self) -> int:
        return (self.patch_k + 63) // 64 * 64

    def validate(self) -> None:
        """Shape rules of the gfx950 kernels (128-column GEMM tiles, 64-deep K steps)."""
        H, F, r = self.hidden, self.ffn, self.lora_r
        assert H == self.n_heads * self.head_dim and self.head_dim in (64, 128)
        assert H % 128 == 0 and F % 128 == 0 and self.vocab % 128 == 0, "H, FFN, vocab must be multiples of 128"
        assert r % 128 == 0, "LoRA rank must be a multiple of 128 (r=256 in the OPA-DPO recipe)"
        assert self.v_hidden % 128 == 0 and self.v_ffn % 128 == 0
        assert self.v_hidden // self.v_heads in (64, 128)
        assert self.image_size % self.patch == 0

    @staticmethod
    def llava15_7b() -> "LlavaDims":
        return LlavaDims()

    @staticmethod
    def llava15_13b() -> "LlavaDims":
        return LlavaDims(hidden=5120, n_layers=40, n_heads=40, ffn=13824)

    @staticmethod
    def tiny(**kw) -> "LlavaDims":
        d = dict(hidden=256, n_layers=2, n_heads=2, head_dim=128, ffn=384, vocab=512,
                 v_hidden=128, v_layers=3, v_heads=2, v_ffn=256, image_size=56, patch=14,
                 lora_r=128, lora_alpha=256.0)
        d.update(kw)
        return LlavaDims(**d)


def llm_linear_shape(d: LlavaDims, name: str):
    H, F = d.hidden, d.ffn
    return {"self_attn.q_proj": (H, H), "self_attn.k_proj": (H, H), "self_attn.v_proj": (H, H),
            "self_attn.o_proj": (H, H), "mlp.gate_proj": (F, H), "mlp.up_proj": (F, H),
            "mlp.down_proj": (H, F)}[name]


def vis_linear_shape(d: LlavaDims, name: str):
    h, f = d.v_hidden, d.v_ffn
    return {"mlp.fc1": (f, h), "mlp.fc2": (h, f)}.get(name, (h, h))


def lora_param_count(d: LlavaDims) -> int:
    """Trainable LLM LoRA parameters (7B r=256: 639.6 M)."""
    n = 0
    for lin in LLM_LINEARS:
        o, i = llm_linear_shape(d, lin)
        n += d.lora_r * (o + i)
    return n * d.n_layers


def pair_flops(d: LlavaDims, q_len: int, t_len: int) -> float:
    """Algorithmic FLOPs of one preference pair: 4 sequence forwards (policy x2,
    reference x2), dgrad + LoRA wgrad for the 2 policy sequences, vision once per image."""
    H, F, V, nl, r = d.hidden, d.ffn, d.vocab, d.n_layers, d.lora_r
    L = q_len + t_len + d.n_patches - 1
    p_lin = nl * (4 * H * H + 3 * H * F)
    p_lora = lora_param_count(d)
    f_seq = 2 * (p_lin + p_lora) * L + nl * 2 * L * L * H + 2 * V * H * t_len
    vh, vf, P1 = d.v_hidden, d.v_ffn, d.n_patches + 1
    f_img = d.v_used_layers * (2 * (4 * vh * vh + 2 * vh * vf) * P1 + 4 * P1 * P1 * vh) \
        + 2 * d.n_patches * (d.patch_k * vh + vh * H + H * H)
    dgrad = 2 * (p_lin + p_lora) * L + 2 * nl * 2 * L * L * H
    wgrad = 4 * p_lora * L
    return 4 * f_seq + 2 * dgrad + 2 * wgrad + f_img


def pair_flops_packed(d: LlavaDims, q_len: int, t_len: int, K: int = 2, ref_merged: bool = False) -> float:
    """FLOPs EXECUTED for one preference pair when the K responses of a sample are packed on a shared image + query
    prefix (policy.pack_responses): one row of pfx + K*t_len positions per policy / reference pass instead of K rows of
    pfx + t_len.  Same conventions as pair_flops (attention counted on the causal/segment-masked (q, k) pairs).
    ref_merged: the frozen reference adapter is folded into its weights (LoraAdapter.merge_into_base): no LoRA FLOPs in that pass."""
    H, F, V, nl, r = d.hidden, d.ffn, d.vocab, d.n_layers, d.lora_r
    pfx = q_len + d.n_patches - 1
    Lp = pfx + K * t_len
    p_lin = nl * (4 * H * H + 3 * H * F)
    p_lora = lora_param_count(d)
    pairs = pfx * pfx / 2 + K * (t_len * pfx + t_len * t_len / 2)          # attended (query, key) pairs
    f_row = 2 * (p_lin + p_lora) * Lp + nl * 4 * H * pairs + 2 * V * H * t_len * K
    vh, vf, P1 = d.v_hidden, d.v_ffn, d.n_patches + 1
    f_img = d.v_used_layers * (2 * (4 * vh * vh + 2 * vh * vf) * P1 + 4 * P1 * P1 * vh) \
        + 2 * d.n_patches * (d.patch_k * vh + vh * H + H * H)
    dgrad = 2 * (p_lin + p_lora) * Lp + 2 * nl * 4 * H * pairs
    wgrad = 4 * p_lora * Lp
    return 2 * f_row + dgrad + wgrad + f_img - (2 * p_lora * Lp if ref_merged else 0)


def pair_flops_ragged(d: LlavaDims, prefix_rows, resp_rows, ref_merged: bool = False, compact_top: bool = True) -> float:
    """FLOPs EXECUTED for one preference pair on ragged rows (ctx.CtxEngine default: padding positions are not rows of any kernel):
    `prefix_rows` valid image + query positions, `resp_rows` = valid lengths of the K responses packed behind it.  Same conventions
    as pair_flops_packed; the head (lm_head + log-softmax) runs on the valid response tokens only (compact head rows)."""
    H, V, nl = d.hidden, d.vocab, d.n_layers
    rows = prefix_rows + sum(resp_rows)
    p_lin = nl * (4 * H * H + 3 * H * d.ffn)
    p_lora = lora_param_count(d)
    pairs = prefix_rows * prefix_rows / 2 + sum(v * prefix_rows + v * v / 2 for v in resp_rows)
    f_row = 2 * (p_lin + p_lora) * rows + nl * 4 * H * pairs + 2 * V * H * sum(resp_rows)
    vh, vf, P1 = d.v_hidden, d.v_ffn, d.n_patches + 1
    f_img = d.v_used_layers * (2 * (4 * vh * vh + 2 * vh * vf) * P1 + 4 * P1 * P1 * vh) \
        + 2 * d.n_patches * (d.patch_k * vh + vh * H + H * H)
    dgrad = 2 * (p_lin + p_lora) * rows + 2 * nl * 4 * H * pairs
    wgrad = 4 * p_lora * rows
    total = 2 * f_row + dgrad + wgrad + f_img - (2 * p_lora * rows if ref_merged else 0)
    if compact_top:
        # top decoder layer: o-projection and MLP (and their LoRA) only on the rows the head reads (last prefix row + response rows)
        skipped = rows - (1 + sum(max(v - 1, 0) for v in resp_rows)) if any(v > 0 for v in resp_rows) else rows
        p_om = H * H + 3 * H * d.ffn
        l_om = sum(d.lora_r * sum(llm_linear_shape(d, lin)) for lin in ("self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"))
        total -= skipped * (2 * (p_om + l_om) + 2 * (p_om + (0 if ref_merged else l_om)) + 2 * (p_om + l_om) + 4 * l_om)
    return total
