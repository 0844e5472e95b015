"""GPU: the binding INTEGRATION.md tells a reference maintainer to write - ctypes against the C symbols of include/opadpo_hip.h ONLY
(own CDLL handle, own struct declarations, the library's default hipMalloc arena instead of torch's allocator) - run end to end:
context, weights, adapter, vision encode, sequence log-probs on ragged rows, LoRA backward, rollout.  The host classes of this
repository (opadpo_amd.ctx.CtxEngine, which is the same binding plus bookkeeping) serve as the expected values only: every result of
the raw binding must equal theirs bit for bit (gradients: to the order of the fp32 atomics)."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp, ci, cf = C.c_void_p, C.c_int, C.c_float


class Dims(C.Structure):          # opadpo_dims
    _fields_ = [(n, ci) for n in ("hidden", "n_layers", "n_heads", "head_dim", "ffn", "vocab")] + [("rms_eps", cf), ("rope_theta", cf)] + \
               [(n, ci) for n in ("v_hidden", "v_used_layers", "v_heads", "v_ffn", "image_size", "patch")] + [("v_eps", cf), ("lora_r", ci), ("lora_alpha", cf)]


class Layer(C.Structure):         # opadpo_layer_weights
    _fields_ = [(n, vp) for n in ("wqkv", "wo", "wgu", "wd", "ln1", "ln2", "wqkv_t", "wo_t", "wgu_t", "wd_t")]


class VLayer(C.Structure):        # opadpo_vision_layer_weights
    _fields_ = [(n, vp) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv", "bqkv", "wo", "bo", "fc1", "b1", "fc2", "b2")]


class Vision(C.Structure):        # opadpo_vision_weights
    _fields_ = [(n, vp) for n in ("patch_w", "cls", "pos", "pre_ln_w", "pre_ln_b", "proj0", "proj0_b", "proj2", "proj2_b")]


def _bind():
    lib = C.CDLL(os.path.join(REPO, "opa-dpo_amd", "lib", "libopadpo_hip.so"))
    lib.opadpo_ctx_last_error.restype = C.c_char_p
    lib.opadpo_ctx_last_error.argtypes = [vp]
    lib.opadpo_ctx_create.argtypes = [C.POINTER(Dims), ci, C.POINTER(vp)]
    lib.opadpo_ctx_destroy.argtypes = [vp]
    lib.opadpo_ctx_destroy.restype = None
    lib.opadpo_ctx_set_llm_weights.argtypes = [vp, vp, vp, vp, vp, C.POINTER(Layer), ci]
    lib.opadpo_ctx_set_vision_weights.argtypes = [vp, C.POINTER(Vision), C.POINTER(VLayer), ci]
    lib.opadpo_ctx_set_rope_tables.argtypes = [vp, vp, vp, ci]
    lib.opadpo_ctx_set_adapter.argtypes = [vp, ci, vp, vp, vp]
    lib.opadpo_vision_encode.argtypes = [vp, vp, ci, vp, vp]
    lib.opadpo_seq_logprobs_fwd.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp, vp, C.POINTER(vp), vp, vp]
    lib.opadpo_seq_logprobs_bwd.argtypes = [vp, vp, vp, vp, vp, ci, ci, vp]
    lib.opadpo_decode_begin.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, cf, ci, cf, C.c_uint64, ci, ci, ci, vp, vp]
    lib.opadpo_decode_run.argtypes = [vp, ci, ci, vp]
    lib.opadpo_decode_end.argtypes = [vp]
    return lib


def test_raw_c_binding_end_to_end():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd.ctx import CtxEngine                      # expected values only
    from opadpo_amd.dims import LlavaDims
    from opadpo_amd.generate import Generator
    from opadpo_amd.model import BaseWeights, LoraAdapter
    from opadpo_amd.policy import AutoregressivePolicy
    from opadpo_amd.synth import init_lora, init_weights, synth_pairs
    dev = torch.device("cuda:0")
    d = LlavaDims.tiny()
    base = BaseWeights(d, init_weights(d, seed=0, std=0.05, device=dev), dev, need_backward=True)      # torch tensors = the storage of the weights
    lora = init_lora(d, seed=1, b_std=0.03, device=dev)
    ad_want, ad_raw = LoraAdapter(d, lora, dev, True), LoraAdapter(d, lora, dev, True)
    p = synth_pairs(d, 3, 16, 24, seed=5, device=dev)
    T, K = 24, 2
    st = torch.cuda.current_stream().cuda_stream

    lib = _bind()

    def check(rc, ctx):
        assert rc == 0, lib.opadpo_ctx_last_error(ctx).decode()
    dims = Dims(d.hidden, d.n_layers, d.n_heads, d.head_dim, d.ffn, d.vocab, d.rms_eps, d.rope_theta, d.v_hidden, d.v_used_layers, d.v_heads,
                d.v_ffn, d.image_size, d.patch, d.v_eps, d.lora_r, d.lora_alpha)
    ctx = vp()
    assert lib.opadpo_ctx_create(C.byref(dims), 0, C.byref(ctx)) == 0
    layers = (Layer * d.n_layers)()
    for i, w in enumerate(base.layers):
        for k, _ in Layer._fields_:
            setattr(layers[i], k, w[k].data_ptr())
    check(lib.opadpo_ctx_set_llm_weights(ctx, base.embed.data_ptr(), base.norm.data_ptr(), base.lm_head.data_ptr(), base.lm_head_t.data_ptr(), layers, d.n_layers), ctx)
    vl = (VLayer * d.v_used_layers)()
    names = {"ln1_w": "layer_norm1_w", "ln1_b": "layer_norm1_b", "ln2_w": "layer_norm2_w", "ln2_b": "layer_norm2_b"}
    for j, w in enumerate(base.vlayers):
        for k, _ in VLayer._fields_:
            setattr(vl[j], k, w[names.get(k, k)].data_ptr())
    vw = Vision(*[getattr(base, k).data_ptr() for k, _ in Vision._fields_])
    check(lib.opadpo_ctx_set_vision_weights(ctx, C.byref(vw), vl, d.v_used_layers), ctx)
    cos, sin = base.rope_tables(2048)
    check(lib.opadpo_ctx_set_rope_tables(ctx, cos.data_ptr(), sin.data_ptr(), 2048), ctx)
    check(lib.opadpo_ctx_set_adapter(ctx, 1, ad_raw.work.data_ptr(), ad_raw.work_t.data_ptr(), ad_raw.grad.data_ptr()), ctx)

    # ---- expected values: this repository's host classes ------------------------------------------------------------------------
    eng = CtxEngine(base)
    pol = AutoregressivePolicy(eng, ad_want, T, pack_responses=True)
    feats_want = eng.encode_images(p["images"])
    _, batch = pol.build_batch(p["queries"], p["queries_attn_masks"], {"chosen_response": p["chosen"], "rejected_response": p["rejected"]})
    logp_want, ent_want, sv = eng.seq_logprobs_fwd(ad_want, batch, feats_want, 0.9, train=True)
    g = torch.Generator().manual_seed(2)
    dlogp = torch.randn(logp_want.shape, generator=g).to(dev)
    ad_want.grad.zero_()
    eng.seq_logprobs_bwd(ad_want, sv, dlogp)
    gen = Generator(eng, ad_want)
    toks_want = gen.generate(p["queries"], p["queries_attn_masks"], image_feats=feats_want, max_new_tokens=7, temperature=0.8, top_k=20, top_p=0.9, seed=11)
    torch.cuda.synchronize()

    # ---- the raw binding ------------------------------------------------------------------------------------------------------------
    px = p["images"].to(torch.bfloat16).contiguous()
    feats = torch.empty_like(feats_want)
    check(lib.opadpo_vision_encode(ctx, px.data_ptr(), px.shape[0], feats.data_ptr(), st), ctx)
    S, n_txt = batch.ids.shape
    logp = torch.empty(K * S * T, dtype=torch.float32, device=dev)
    ent = torch.empty_like(logp)
    saved = vp()
    plan = batch.row_plan                         # host int32 [S, K+1]: dropped left pads, valid length of every response
    check(lib.opadpo_seq_logprobs_fwd(ctx, 1, batch.ids.data_ptr(), batch.text_mask.data_ptr(), batch.feat_row.data_ptr(), None, feats.data_ptr(),
                                      S, n_txt, T, K, 0.9, 1, logp.data_ptr(), ent.data_ptr(), C.byref(saved), plan.data_ptr(), st), ctx)
    ad_raw.grad.zero_()
    check(lib.opadpo_seq_logprobs_bwd(ctx, saved, dlogp.contiguous().data_ptr(), None, None, d.n_layers - 1, 0, st), ctx)
    B, Q = p["queries"].shape
    hist = torch.empty(7, B, dtype=torch.int32, device=dev)
    q_ids, q_mask = p["queries"].to(torch.int32).contiguous(), p["queries_attn_masks"].to(torch.uint8).contiguous()      # borrowed by the library: keep them alive
    check(lib.opadpo_decode_begin(ctx, 1, q_ids.data_ptr(), q_mask.data_ptr(), feats.data_ptr(), B, Q, 7, 0.8, 20, 0.9, 11, 2, 0, 0, hist.data_ptr(), st), ctx)
    check(lib.opadpo_decode_run(ctx, 6, 0, st), ctx)
    torch.cuda.synchronize()
    check(lib.opadpo_decode_end(ctx), ctx)

    assert torch.equal(feats, feats_want)
    assert torch.equal(logp.view_as(logp_want), logp_want) and torch.equal(ent.view_as(ent_want), ent_want)
    assert float((ad_raw.grad - ad_want.grad).norm() / ad_want.grad.norm()) < 1e-5 and float(ad_want.grad.norm()) > 0
    assert torch.equal(hist.t().long(), toks_want)
    # errors come back as a code + text, never as a crash: an out-of-range adapter id
    rc = lib.opadpo_seq_logprobs_fwd(ctx, 99, batch.ids.data_ptr(), batch.text_mask.data_ptr(), batch.feat_row.data_ptr(), None, feats.data_ptr(),
                                     S, n_txt, T, K, 0.9, 0, logp.data_ptr(), ent.data_ptr(), None, None, st)
    assert rc != 0 and b"adapter" in lib.opadpo_ctx_last_error(ctx)
    lib.opadpo_ctx_destroy(ctx)
    eng.close()
