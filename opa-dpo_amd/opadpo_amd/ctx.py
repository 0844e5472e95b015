"""Host side of the sequence-level C entry points (include/opadpo_hip.h "Context API", csrc/ctx.hip).

`CtxEngine` is what `AutoregressivePolicy` / `DPOTrainer` / `Generator` drive by default: ONE C call per pass -

    opadpo_vision_encode        <- get_vision_tower() + mm_projector
    opadpo_seq_logprobs_fwd     <- self.base_model(**inputs) + the logits slice / log-prob / entropy  (rl_models.py:114-132)
    opadpo_seq_logprobs_bwd     <- accelerator.backward(loss)                                          (rl_trainer.py:162)
    opadpo_decode_begin / _run  <- policy.generate(do_sample=True, ...)                                (online_generator.py:292-309)

- with the layer loop, the workspace, the saved activations and the KV cache below the ABI (the context owns them; its
allocator hooks point at torch's caching allocator so the process keeps one memory pool and `torch.cuda.max_memory_allocated`
still sees everything).  Python only hands over raw device pointers of torch-owned weights / ids / outputs and the stream.
`LlavaEngine` (model.py) remains the op-level sequencing of the same kernels: the OPA LoRA-SFT stage builds on it, and
tests/test_ctx_gpu.py checks that both give bit-identical log-probs.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, Optional

import torch

from . import lib as L
from .dims import LlavaDims
from .model import BF, BaseWeights, LlavaEngine, LoraAdapter, SeqBatch

_u16p = C.c_void_p


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("hidden", "n_layers", "n_heads", "head_dim", "ffn", "vocab")] + \
               [("rms_eps", C.c_float), ("rope_theta", C.c_float)] + \
               [(n, C.c_int) for n in ("v_hidden", "v_used_layers", "v_heads", "v_ffn", "image_size", "patch")] + \
               [("v_eps", C.c_float), ("lora_r", C.c_int), ("lora_alpha", C.c_float)]


class LayerWeights(C.Structure):
    _fields_ = [(n, _u16p) for n in ("wqkv", "wo", "wgu", "wd", "ln1", "ln2", "wqkv_t", "wo_t", "wgu_t", "wd_t")]


class VisionLayerWeights(C.Structure):
    _fields_ = [(n, _u16p) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv", "bqkv", "wo", "bo", "fc1", "b1", "fc2", "b2")]


class VisionWeights(C.Structure):
    _fields_ = [(n, _u16p) for n in ("patch_w", "cls", "pos", "pre_ln_w", "pre_ln_b", "proj0", "proj0_b", "proj2", "proj2_b")]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class CtxSaved:
    """Handle of the activations one training forward left in the context (released by the backward or on garbage collection)."""

    def __init__(self, engine: "CtxEngine", handle: int, batch: SeqBatch):
        self.engine, self.handle, self.batch = engine, handle, batch

    def release(self) -> None:
        if self.handle:
            self.engine._call("opadpo_saved_release", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class CtxEngine(LlavaEngine):
    def __init__(self, base: BaseWeights, torch_allocator: bool = True, ragged: bool = True):
        """ragged: run the LLM passes on the valid rows only (SeqBatch.row_plan: left pad of the query and right pad of every response
        are not rows of any kernel); False keeps the reference's padded rows."""
        super().__init__(base)
        self.ragged = ragged
        lib = L.load()
        d = base.dims
        self._dims = Dims(d.hidden, d.n_layers, d.n_heads, d.head_dim, d.ffn, d.vocab, d.rms_eps, d.rope_theta, d.v_hidden, d.v_used_layers,
                          d.v_heads, d.v_ffn, d.image_size, d.patch, d.v_eps, d.lora_r, d.lora_alpha)
        h = C.c_void_p()
        rc = lib.opadpo_ctx_create(C.byref(self._dims), self.dev.index or 0, C.byref(h))
        if rc != 0:
            raise L.OpadpoError(f"opadpo_ctx_create failed ({rc}): dims rejected")
        self.ctx = h
        self._lib = lib
        if torch_allocator:
            dev_index = self.dev.index or 0

            def _alloc(nbytes, stream, _user):
                try:
                    return torch.cuda.caching_allocator_alloc(int(nbytes), dev_index, int(stream or 0))
                except Exception:          # out of memory: the C side reports it with a message
                    return None

            def _free(p, _user):
                torch.cuda.caching_allocator_delete(int(p))

            self._alloc_cb, self._free_cb = ALLOC_FN(_alloc), FREE_FN(_free)       # keep the thunks alive
            self._call("opadpo_ctx_set_allocator", C.cast(self._alloc_cb, C.c_void_p), C.cast(self._free_cb, C.c_void_p), None)
        # ---- borrowed weights -----------------------------------------------------------------------------------------
        layers = (LayerWeights * d.n_layers)()
        for i, w in enumerate(base.layers):
            for k in ("wqkv", "wo", "wgu", "wd", "ln1", "ln2"):
                setattr(layers[i], k, _ptr(w[k]))
            for k in ("wqkv_t", "wo_t", "wgu_t", "wd_t"):
                setattr(layers[i], k, _ptr(w.get(k)))
        self._layers = layers
        self._call("opadpo_ctx_set_llm_weights", _ptr(base.embed), _ptr(base.norm), _ptr(base.lm_head), _ptr(base.lm_head_t), layers, d.n_layers)
        vl = (VisionLayerWeights * d.v_used_layers)()
        for j, w in enumerate(base.vlayers):
            for dst, src in (("ln1_w", "layer_norm1_w"), ("ln1_b", "layer_norm1_b"), ("ln2_w", "layer_norm2_w"), ("ln2_b", "layer_norm2_b"),
                             ("wqkv", "wqkv"), ("bqkv", "bqkv"), ("wo", "wo"), ("bo", "bo"), ("fc1", "fc1"), ("b1", "b1"), ("fc2", "fc2"), ("b2", "b2")):
                setattr(vl[j], dst, _ptr(w[src]))
        vw = VisionWeights(_ptr(base.patch_w), _ptr(base.cls), _ptr(base.pos), _ptr(base.pre_ln_w), _ptr(base.pre_ln_b), _ptr(base.proj0),
                           _ptr(base.proj0_b), _ptr(base.proj2), _ptr(base.proj2_b))
        self._vl, self._vw = vl, vw
        self._call("opadpo_ctx_set_vision_weights", C.byref(vw), vl, d.v_used_layers)
        self._rope_len = 0
        self._adapters: Dict[int, tuple] = {}        # id(adapter object) -> (slot 1..7, weakref, signature, keep-alive); slot 0 = bare base
        self._call("opadpo_ctx_set_adapter", 0, None, None, None)

    # ---- plumbing -------------------------------------------------------------------------------------------------------
    def _call(self, name: str, *args) -> None:
        rc = getattr(self._lib, name)(self.ctx, *args)
        if rc != 0:
            raise L.OpadpoError(f"{name} failed: {self._lib.opadpo_ctx_last_error(self.ctx).decode()}")

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self._lib.opadpo_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release(self) -> None:
        super().release()
        if getattr(self, "ctx", None):
            self._call("opadpo_ctx_trim")

    def set_flags(self, gemm_variant: int = -1, use_tr: int = -1) -> None:
        """Kernel-variant switches of THIS context (lib.set_flags documents the values); -1 = process default."""
        self._call("opadpo_ctx_set_flags", int(gemm_variant), int(use_tr))

    def profile(self, enable: bool) -> None:
        """Bracket every gemm_nt launch of the context with HIP events on the launch stream (bench.py roofline record)."""
        self._call("opadpo_ctx_profile", int(enable))

    def profile_read(self):
        """-> (sum of algorithmic FLOPs, sum of launch durations in ms, launches) since the last read; waits for the events."""
        f, ms, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        self._call("opadpo_ctx_profile_read", C.byref(f), C.byref(ms), C.byref(n))
        return f.value, ms.value, n.value

    def wgrad_deterministic(self) -> int:
        """How the last backward flushed its LoRA wgrads: 1 ordered reduce (bit-reproducible), 0 fp32 atomics (use_tr bit 12, or
        lora_r % 256 != 0 - the 128x128 wgrad kernel has no ordered flush), -1 no backward yet."""
        return int(self._lib.opadpo_ctx_wgrad_deterministic(self.ctx))

    def _ensure_rope(self, n_pos: int) -> None:
        if n_pos > self._rope_len:
            n = max(n_pos, 2048)
            self._rope = self.base.rope_tables(n)           # torch's tables: the same numbers the op-level path uses
            self._call("opadpo_ctx_set_rope_tables", _ptr(self._rope[0]), _ptr(self._rope[1]), n)
            self._rope_len = n

    def adapter_slot(self, adapter) -> int:
        """Register (or refresh) an adapter with the context and return its slot: None -> 0 (bare base model); a LoraAdapter keeps
        its slot for life, the pointers are re-sent when its buffers or merged copy changed."""
        if adapter is None:
            return 0
        merged = getattr(adapter, "merged", None)
        if merged is not None:
            sig = ("merged", id(merged))
        else:
            sig = ("lora", adapter.work.data_ptr(), _ptr(adapter.work_t), _ptr(adapter.grad))
        ent = self._adapters.get(id(adapter))
        if ent is not None and ent[1]() is not adapter:       # the id was recycled by a new object
            ent = None
        if ent is not None and ent[2] == sig:
            return ent[0]
        if ent is None:
            dead = [k for k, e in self._adapters.items() if e[1]() is None]      # slots of adapters that no longer exist are reused
            free = sorted(self._adapters.pop(k)[0] for k in dead)
            used = {e[0] for e in self._adapters.values()}
            cand = [s_ for s_ in range(1, 8) if s_ not in used]
            if not cand:
                raise L.OpadpoError("more than 7 live adapters registered with one context")
            slot = cand[0]
            del free
        else:
            slot = ent[0]
        keep = None
        if merged is not None:
            arr = (LayerWeights * self.d.n_layers)()
            sw = 0
            for i, m in enumerate(merged):
                gu = m.get("wgu_sw")
                sw = int(gu is not None)
                arr[i].wgu = _ptr(gu if gu is not None else m.get("wgu"))
                for k in ("wqkv", "wo", "wd"):
                    setattr(arr[i], k, _ptr(m.get(k)))
            self._call("opadpo_ctx_set_merged_adapter", slot, arr, self.d.n_layers, sw)
            keep = (arr, merged)
        else:
            self._call("opadpo_ctx_set_adapter", slot, _ptr(adapter.work), _ptr(adapter.work_t), _ptr(adapter.grad))
        self._adapters[id(adapter)] = (slot, weakref.ref(adapter), sig, keep)
        return slot

    # ---- the three passes -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_images(self, pixels: torch.Tensor) -> torch.Tensor:
        d = self.d
        px = pixels.to(device=self.dev, dtype=BF).contiguous()
        B = px.shape[0]
        feats = torch.empty(B, d.n_patches, d.hidden, dtype=BF, device=self.dev)
        self._call("opadpo_vision_encode", _ptr(px), B, _ptr(feats), L.stream())
        return feats

    def seq_logprobs_fwd(self, adapter: Optional[LoraAdapter], batch: SeqBatch, feats: torch.Tensor, temperature: float, train: bool):
        d = self.d
        S, n_txt = batch.ids.shape
        T, K = batch.T, batch.K
        self._ensure_rope(n_txt + d.n_patches - 1)
        slot = self.adapter_slot(adapter)
        R = K * S * T
        logp = torch.empty(R, dtype=torch.float32, device=self.dev)
        ent = torch.empty(R, dtype=torch.float32, device=self.dev)
        handle = C.c_void_p()
        feats = feats.contiguous()
        plan = getattr(batch, "row_plan", None) if self.ragged else None       # CPU int32 [S, K+1] (policy.build_batch) or None = padded rows
        self._call("opadpo_seq_logprobs_fwd", slot, _ptr(batch.ids), _ptr(batch.text_mask), _ptr(batch.feat_row), _ptr(batch.image_mask),
                   _ptr(feats), S, n_txt, T, K, float(temperature), int(train), _ptr(logp), _ptr(ent), C.byref(handle),
                   plan.data_ptr() if plan is not None else None, L.stream())
        sv = CtxSaved(self, handle.value, batch) if train else None
        return logp.view(K * S, T), ent.view(K * S, T), sv

    def seq_logprobs_bwd(self, adapter: LoraAdapter, sv: CtxSaved, dlogp: torch.Tensor, d_feats: Optional[torch.Tensor] = None,
                         d_ent: Optional[torch.Tensor] = None, layer_done=None) -> None:
        assert sv is not None and sv.handle, "activations already consumed"
        dlogp = dlogp.to(device=self.dev, dtype=torch.float32).contiguous().view(-1)
        if d_ent is not None:
            d_ent = d_ent.to(device=self.dev, dtype=torch.float32).contiguous().view(-1)
        nl = self.d.n_layers
        st = L.stream()
        try:
            if layer_done is None:
                self._call("opadpo_seq_logprobs_bwd", sv.handle, _ptr(dlogp), _ptr(d_ent), _ptr(d_feats), nl - 1, 0, st)
            else:       # ranged calls: the host starts the gradient exchange of a finished bucket between two of them
                for i in range(nl - 1, -1, -1):
                    self._call("opadpo_seq_logprobs_bwd", sv.handle, _ptr(dlogp), _ptr(d_ent), _ptr(d_feats), i, i, st)
                    layer_done(i)
        finally:
            sv.release()

    # ---- rollout -----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, adapter, queries: torch.Tensor, query_attn_masks: torch.Tensor, image_feats: torch.Tensor, *, max_new_tokens: int,
                 temperature: float, top_k: int, top_p: float, seed: int, eos_token_id: int, pad_token_id: int, suppress_eos: bool,
                 use_graph: bool) -> torch.Tensor:
        """-> responses [B, max_new_tokens] int64 (pad after a row finished); prefill, KV cache and the per-token launch loop (or hipGraph) live in
        the context (opadpo_decode_begin / opadpo_decode_run)."""
        dev, d = self.dev, self.d
        B, Q = queries.shape
        self._ensure_rope(Q + d.n_patches - 1 + max_new_tokens)
        slot = self.adapter_slot(adapter)
        ids = queries.to(dev).to(torch.int32).contiguous()
        tmask = query_attn_masks.to(dev).to(torch.uint8).contiguous()
        feats = image_feats.contiguous()
        history = torch.empty(max_new_tokens, B, dtype=torch.int32, device=dev)
        st = L.stream()
        self._call("opadpo_decode_begin", slot, _ptr(ids), _ptr(tmask), _ptr(feats), B, Q, int(max_new_tokens), float(temperature), int(top_k),
                   float(top_p), int(seed), int(eos_token_id), int(pad_token_id), int(suppress_eos), _ptr(history), st)
        try:
            left = max_new_tokens - 1
            fin = C.c_int(0)
            while left > 0:
                n = min(32, left)
                self._call("opadpo_decode_run", n, int(use_graph), st)
                left -= n
                if left > 0 and not suppress_eos:
                    self._call("opadpo_decode_all_finished", C.byref(fin), st)
                    if fin.value:
                        break
            torch.cuda.current_stream().synchronize()
        finally:
            self._call("opadpo_decode_end")
        return history.t().contiguous().to(torch.int64)
