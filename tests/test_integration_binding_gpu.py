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


def test_exchange_entry_points_on_a_callers_rccl_communicator():
    """opadpo_allreduce_grads / _reduce_scatter_grads / _all_gather_params: the data-parallel exchange for a binder that owns its
    ncclComm_t and never touches torch.distributed.  The library resolves the collectives from the RCCL already loaded in the process - here the
    one PyTorch ships - so the test creates a communicator with THAT library through ctypes (world size 1: one GPU per box, RCCL refuses two ranks
    per device) and drives the ZeRO-1 sequence of include/opadpo_hip.h through the raw symbols: reduce-scatter -> sum of squares -> AdamW on the
    shard -> all-gather; at world 1 every collective is the identity, so the result must equal the plain opadpo_adamw step (to the summation order of the norm's fp32 atomics)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import glob
    torch.zeros(1, device="cuda:0")                       # the HIP context exists
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["/opt/rocm/lib/librccl.so"]
    rccl = C.CDLL(cands[0], mode=C.RTLD_GLOBAL)

    class UID(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UID(), vp()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UID)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(vp), ci, UID, ci]
    rccl.ncclCommDestroy.argtypes = [vp]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    lib = C.CDLL(os.path.join(REPO, "opa-dpo_amd", "lib", "libopadpo_hip.so"))
    lib.opadpo_last_error.restype = C.c_char_p
    sz, dbl = C.c_size_t, C.c_double
    lib.opadpo_allreduce_grads.argtypes = [vp, vp, sz, ci, vp]
    lib.opadpo_reduce_scatter_grads.argtypes = [vp, vp, vp, sz, ci, vp]
    lib.opadpo_all_gather_params.argtypes = [vp, vp, vp, sz, ci, vp]
    lib.opadpo_sumsq.argtypes = [vp, sz, vp, vp]
    lib.opadpo_adamw.argtypes = [vp, vp, vp, vp, vp, sz, dbl, dbl, dbl, dbl, dbl, ci, vp, dbl, dbl, vp]
    st = vp(torch.cuda.current_stream().cuda_stream)
    try:
        n = 1 << 20
        g = torch.Generator(device="cpu").manual_seed(9)
        grad = (torch.randn(n, generator=g) * 0.01).cuda()
        p0 = (torch.randn(n, generator=g) * 0.02).cuda()

        def check(rc):
            assert rc == 0, lib.opadpo_last_error().decode()
        # all-reduce, fp32 and bf16: the sum over ONE rank is the buffer itself; a vector of ones reads back the world size
        for dt, t in ((0, grad.clone()), (1, grad.to(torch.bfloat16))):
            want = t.clone()
            check(lib.opadpo_allreduce_grads(comm, t.data_ptr(), n, dt, st))
            torch.cuda.synchronize()
            assert torch.equal(t, want)
        ones = torch.ones(4, device="cuda")
        check(lib.opadpo_allreduce_grads(comm, ones.data_ptr(), 4, 0, st))
        torch.cuda.synchronize()
        assert ones.tolist() == [1.0] * 4
        # ZeRO-1 through the raw symbols against the plain step
        def adamw(p, gr, m, v, w, ss):
            check(lib.opadpo_adamw(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), w.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, ss.data_ptr(), 1.0, 1.0, st))
        pa, ma, va, wa, ssa = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.empty(n, dtype=torch.bfloat16, device="cuda"), torch.zeros(1, device="cuda")
        check(lib.opadpo_sumsq(grad.data_ptr(), n, ssa.data_ptr(), st))
        adamw(pa, grad, ma, va, wa, ssa)
        pb, mb, vb, ssb = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
        shard, wshard, wb = torch.empty(n, device="cuda"), torch.empty(n, dtype=torch.bfloat16, device="cuda"), torch.empty(n, dtype=torch.bfloat16, device="cuda")
        check(lib.opadpo_reduce_scatter_grads(comm, grad.data_ptr(), shard.data_ptr(), n, 0, st))
        check(lib.opadpo_sumsq(shard.data_ptr(), n, ssb.data_ptr(), st))
        check(lib.opadpo_allreduce_grads(comm, ssb.data_ptr(), 1, 0, st))
        adamw(pb, shard, mb, vb, wshard, ssb)
        check(lib.opadpo_all_gather_params(comm, wshard.data_ptr(), wb.data_ptr(), n, 1, st))
        torch.cuda.synchronize()
        # (opadpo_sumsq adds its block sums with fp32 atomics: the two norms agree to summation order, not to the bit, and so does the clip factor)
        assert abs(float(ssa) - float(ssb)) <= 1e-5 * float(ssa)
        assert float((pa - pb).abs().max()) <= 1e-6 * float(pa.abs().max()) and float((wa.float() - wb.float()).abs().max()) <= 2.0 ** -7 * float(wa.float().abs().max())
        assert not torch.equal(pa, p0)
        # loud on misuse
        assert lib.opadpo_allreduce_grads(None, grad.data_ptr(), n, 0, st) != 0 and b"null communicator" in lib.opadpo_last_error()
        assert lib.opadpo_allreduce_grads(comm, grad.data_ptr(), n, 5, st) != 0
    finally:
        rccl.ncclCommDestroy(comm)
