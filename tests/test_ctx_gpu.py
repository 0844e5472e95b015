"""GPU: the sequence-level C entry points (opadpo_ctx, csrc/ctx.hip; include/opadpo_hip.h "Context API") against the
op-level sequencing of the same kernels (model.LlavaEngine / generate.Generator's Python loop): the context moves the layer loop,
the workspace, the saved activations and the KV cache below the C ABI and must not change a single bit of the log-probs
(same launches, same order); LoRA gradients agree to fp32 accumulation order (the wgrad kernels add with atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd import lib
    from opadpo_amd.ctx import CtxEngine
    from opadpo_amd.dims import LlavaDims
    from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter
    from opadpo_amd.synth import init_lora, init_weights, synth_pairs
    lib.load()
    dev = torch.device("cuda:0")
    d = LlavaDims.tiny()
    base = BaseWeights(d, init_weights(d, seed=0, std=0.05, device=dev), dev, need_backward=True)
    op, cx, cxr = LlavaEngine(base), CtxEngine(base, ragged=False), CtxEngine(base)      # op-level, context on padded rows, context on ragged rows (default)
    sd_pol, sd_ref = init_lora(d, seed=1, b_std=0.03, device=dev), init_lora(d, seed=2, b_std=0.03, device=dev)
    ads = dict(pol=LoraAdapter(d, sd_pol, dev, True), ref=LoraAdapter(d, sd_ref, dev, False), merged=LoraAdapter(d, sd_ref, dev, False))
    ads["merged"].merge_into_base(base)
    p = synth_pairs(d, 3, 16, 24, seed=5, device=dev)
    yield dict(d=d, dev=dev, base=base, op=op, cx=cx, cxr=cxr, ads=ads, p=p)
    cx.close()
    cxr.close()


def _policy(eng, ad, T, pack):
    from opadpo_amd.policy import AutoregressivePolicy
    return AutoregressivePolicy(eng, ad, T, pack_responses=pack)


def _kw(p, eng):
    return dict(images=p["images"], queries=p["queries"], queries_attn_masks=p["queries_attn_masks"], chosen_response=p["chosen"],
                rejected_response=p["rejected"])


def test_vision_encode_is_bit_identical(env):
    a, b = env["op"].encode_images(env["p"]["images"]), env["cx"].encode_images(env["p"]["images"])
    torch.cuda.synchronize()
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("pack", [True, False])
@pytest.mark.parametrize("which", ["ref", "merged", "pol"])
def test_forward_is_bit_identical(env, which, pack):
    ad = env["ads"].get(which)
    outs = []
    for eng in (env["op"], env["cx"]):
        with torch.no_grad():
            outs.append(_policy(eng, ad, 24, pack)(**_kw(env["p"], eng), temperature=0.8))
    torch.cuda.synchronize()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{which} pack={pack}: {k} differs, max {float((outs[0][k] - outs[1][k]).abs().max())}"
    lp = outs[1]["chosen_response_logprobs"]
    assert bool((lp[env["p"]["chosen"] == 0] == 0).all()) and float(lp.min()) < 0


@pytest.mark.parametrize("pack", [True, False])
def test_training_forward_backward_matches_op_level(env, pack):
    ad, p = env["ads"]["pol"], env["p"]
    g = torch.Generator().manual_seed(3)
    w = {k: torch.randn(3, 24, generator=g).to(env["dev"]) for k in ("chosen_response", "rejected_response")}
    res = []
    for eng, hook in ((env["op"], None), (env["cx"], None), (env["cx"], "ranged")):
        ad.grad.zero_()
        pol = _policy(eng, ad, 24, pack)
        seen = []
        pol.layer_done_hook = (lambda i: seen.append(i)) if hook else None
        out = pol(**_kw(p, eng))
        loss = sum((out[k + "_logprobs"] * w[k]).sum() for k in w)
        loss.backward()
        torch.cuda.synchronize()
        if hook:
            assert seen == list(range(env["d"].n_layers - 1, -1, -1)), "layer-done hook must fire top-down, once per layer"
        res.append((out["chosen_response_logprobs"].detach().clone(), ad.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][0], res[2][0])
    for j in (1, 2):
        rel = float((res[j][1] - res[0][1]).norm() / res[0][1].norm())
        assert rel < 1e-5, f"context backward (variant {j}) differs from the op-level backward: rel {rel}"
    assert float(res[0][1].abs().max()) > 0
    ad.grad.zero_()


def test_two_pending_forwards_like_copo(env):
    """Two training forwards alive at once (clean + masked image), backward of both: activations of each are separate handles."""
    ad, p, cx = env["ads"]["pol"], env["p"], env["cx"]
    ad.grad.zero_()
    pol = _policy(cx, ad, 24, True)
    o1 = pol(**_kw(p, cx))
    p2 = dict(p, images=(p["images"].float() * 0.5).to(BF))
    o2 = pol(**_kw(p2, cx))
    (o1["chosen_response_logprobs"].sum() + o2["rejected_response_logprobs"].sum()).backward()
    torch.cuda.synchronize()
    g_both = ad.grad.clone()
    ad.grad.zero_()
    pol(**_kw(p, cx))["chosen_response_logprobs"].sum().backward()
    pol(**_kw(p2, cx))["rejected_response_logprobs"].sum().backward()
    torch.cuda.synchronize()
    assert float((g_both - ad.grad).norm() / ad.grad.norm()) < 1e-5
    ad.grad.zero_()


def test_default_allocator_and_per_context_flags(env):
    """hipMalloc-backed context (no torch allocator hooks) and a context forced onto the 128x128 GEMM: same numbers as the default one
    (the 256x256 and 128x128 kernels accumulate in a different order -> close, not equal); flags of one context do not leak."""
    from opadpo_amd.ctx import CtxEngine
    ad, p = env["ads"]["ref"], env["p"]
    with torch.no_grad():
        want = _policy(env["cx"], ad, 24, True)(**_kw(p, env["cx"]))
        plain = CtxEngine(env["base"], torch_allocator=False, ragged=False)
        got = _policy(plain, ad, 24, True)(**_kw(p, plain))
        small = CtxEngine(env["base"], ragged=False)
        small.set_flags(gemm_variant=4)
        got4 = _policy(small, ad, 24, True)(**_kw(p, small))
        again = _policy(env["cx"], ad, 24, True)(**_kw(p, env["cx"]))
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(want[k], got[k]) and torch.equal(want[k], again[k])
        assert float((want[k] - got4[k]).abs().max()) < 5e-2
    plain.close()
    small.close()


@pytest.mark.parametrize("mode", ["none", "ref", "merged", "fused"])
def test_generation_matches_op_level_loop(env, mode):
    from opadpo_amd.generate import Generator
    p, d = env["p"], env["d"]
    ad = {"none": None, "ref": env["ads"]["ref"], "merged": env["ads"]["merged"], "fused": None}[mode]
    feats = env["op"].encode_images(p["images"])
    for kw in (dict(top_k=1, top_p=1.0, seed=1), dict(temperature=0.8, top_k=20, top_p=0.9, seed=5), dict(top_k=30, top_p=0.95, seed=2, suppress_eos=True)):
        outs = []
        for eng, graph in ((env["op"], True), (env["cx"], True), (env["cx"], False)):
            gen = Generator(eng, ad, use_graph=graph, fuse_swiglu=mode == "fused")
            outs.append(gen.generate(p["queries"], p["queries_attn_masks"], image_feats=feats, max_new_tokens=40, **kw))
        torch.cuda.synchronize()
        assert outs[1].shape == (3, 40) and outs[1].dtype == torch.int64
        assert torch.equal(outs[0], outs[1]), f"{mode} {kw}: context rollout differs from the op-level loop"
        assert torch.equal(outs[1], outs[2]), f"{mode} {kw}: graph replay differs from eager steps"
        for row in outs[1].tolist():                     # pad after the first EOS
            if 2 in row:
                assert all(t == 0 for t in row[row.index(2) + 1:])


def test_wide_7b_forward_is_bit_identical():
    """LLaVA-1.5-7B width, 2 layers, enough rows for the 256x256 kernels: context path == op-level path, bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd.ctx import CtxEngine
    from opadpo_amd.dims import LlavaDims
    from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter
    from opadpo_amd.synth import init_lora, init_weights, synth_pairs
    dev = torch.device("cuda:0")
    d = LlavaDims(hidden=4096, n_layers=2, n_heads=32, head_dim=128, ffn=11008, vocab=32000, v_hidden=128, v_layers=2, v_heads=2, v_ffn=256,
                  image_size=56, patch=14)
    base = BaseWeights(d, init_weights(d, seed=0, device=dev), dev, need_backward=True)
    op, cx = LlavaEngine(base), CtxEngine(base, ragged=False)
    cx.set_flags(use_tr=1 | 1024)          # bit 10: the un-chunked head (one [rows, vocab] logits buffer), what the op-level sequencing runs
    ad = LoraAdapter(d, init_lora(d, seed=1, device=dev), dev, True)
    p = synth_pairs(d, 6, 128, 384, seed=7, device=dev)
    g = torch.Generator().manual_seed(1)
    w = {k: torch.randn(6, 384, generator=g).to(dev) for k in ("chosen_response", "rejected_response")}
    res = []
    for eng in (op, cx):
        ad.grad.zero_()
        out = _policy(eng, ad, 384, True)(**_kw(p, eng))
        sum((out[k + "_logprobs"] * w[k]).sum() for k in w).backward()
        torch.cuda.synchronize()
        res.append((out["chosen_response_logprobs"].detach().clone(), out["rejected_response_entropies"].clone(), ad.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float((res[0][2] - res[1][2]).norm() / res[0][2].norm()) < 1e-5
    # opt-in flag bit 6: SwiGLU backward inside the epilogue of the down projection's dgrad (OPADPO_ACT_SWIGLU_BWD) - same bits
    # into the wgrads, so the gradient differs from the two-kernel form only by the order of the fp32 atomics
    cx.set_flags(use_tr=1 | 64 | 1024)
    ad.grad.zero_()
    out = _policy(cx, ad, 384, True)(**_kw(p, cx))
    sum((out[k + "_logprobs"] * w[k]).sum() for k in w).backward()
    torch.cuda.synchronize()
    assert torch.equal(out["chosen_response_logprobs"].detach(), res[1][0])
    assert float((ad.grad - res[1][2]).norm() / res[1][2].norm()) < 1e-5
    # round 5 defaults against the forms they replaced: bit 13 = residual adds deferred to the RMSNorm pass (default: in the o / down projections'
    # direct epilogue), bit 14 = SwiGLU backward as its own kernel (default: in the dgrad's direct epilogue).  Same fp32 / bf16 operations on the
    # same operands: the log-probs, the entropies and (the LoRA wgrads being ordered reductions) the gradient keep their bits
    for bits in (8192, 16384, 8192 | 16384):
        cx.set_flags(use_tr=1 | bits | 1024)
        ad.grad.zero_()
        out = _policy(cx, ad, 384, True)(**_kw(p, cx))
        sum((out[k + "_logprobs"] * w[k]).sum() for k in w).backward()
        torch.cuda.synchronize()
        assert torch.equal(out["chosen_response_logprobs"].detach(), res[1][0]), bits
        assert torch.equal(out["rejected_response_entropies"], res[1][1]), bits
        assert float((ad.grad - res[1][2]).norm() / res[1][2].norm()) < 1e-6, bits
    # CHUNKED head (default when the logits of the batch shape reach 4 GiB; bit 9 forces it): lm_head + online log-sum-exp + label gather + entropy over 4096
    # vocabulary columns at a time, logits recomputed chunk by chunk in the backward - no [rows, vocab] buffer.  Same function in another
    # fp32 association: log-probs / entropies to fp32 rounding, gradients to the bf16 rounding of d_hn (accumulated over the chunks in fp32)
    peak0 = cx._lib.opadpo_ctx_bytes_peak(cx.ctx)
    cx.set_flags(use_tr=1 | 512)
    ad.grad.zero_()
    out = _policy(cx, ad, 384, True)(**_kw(p, cx))
    sum((out[k + "_logprobs"] * w[k]).sum() for k in w).backward()
    torch.cuda.synchronize()
    assert float((out["chosen_response_logprobs"].detach() - res[1][0]).abs().max()) < 2e-5
    assert float((out["rejected_response_entropies"] - res[1][1]).abs().max()) < 2e-4
    valid = p["chosen"] != 0
    assert bool((out["chosen_response_logprobs"].detach()[~valid] == 0).all())
    assert float((ad.grad - res[1][2]).norm() / res[1][2].norm()) < 1e-2      # measured 5e-3: bf16 rounding flips of d_hn, amplified by two layers
    cx.close()
    op.release()


@pytest.mark.parametrize("pack", [True, False])
@pytest.mark.parametrize("which", ["ref", "merged", "pol"])
def test_ragged_rows_equal_padded_rows(env, which, pack):
    """Default context path: the padding rows (left pad of the query, right pad of every response) are not rows of any kernel.
    Valid tokens: same log-probs / entropies as the padded layout up to the attention's tile partition (rows move relative to the
    64-row tiles: fp32 summation order -> occasional bf16 rounding flips); pad cells: exactly -0.0 / 0; LoRA gradients agree."""
    ad, p = env["ads"][which], env["p"]
    assert int((p["chosen"] == 0).sum()) > 0 and int((~p["queries_attn_masks"]).sum()) > 0, "the batch must contain padding"
    outs = []
    for eng in (env["cx"], env["cxr"]):
        with torch.no_grad():
            outs.append(_policy(eng, ad, 24, pack)(**_kw(p, eng), temperature=0.9))
    torch.cuda.synchronize()
    for key, ids in (("chosen_response", p["chosen"]), ("rejected_response", p["rejected"])):
        valid = ids != 0
        a, b = outs[0][key + "_logprobs"], outs[1][key + "_logprobs"]
        assert bool((b[~valid] == 0).all()) and bool((outs[1][key + "_entropies"][~valid] == 0).all())
        rel = ((a - b).abs()[valid] / a.abs()[valid].clamp_min(1e-3))
        assert float(rel.mean()) < 5e-4 and float(rel.max()) < 1e-2, (float(rel.mean()), float(rel.max()))
        assert float((outs[0][key + "_entropies"] - outs[1][key + "_entropies"]).abs().max()) < 2e-2


@pytest.mark.parametrize("pack", [True, False])
def test_compact_top_layer_equals_full_top_layer(env, pack):
    """Ragged default: the TOP decoder layer's o-projection and MLP run only on the rows the head reads (last prefix row + response
    rows), forward and backward.  Against the same context with flag bit 7 (top layer on every row): log-probs and LoRA gradients
    agree to the tile-order noise of GEMMs with another row count - the rows left out are read by nothing."""
    from opadpo_amd.ctx import CtxEngine
    ad, p = env["ads"]["pol"], env["p"]
    full = CtxEngine(env["base"])
    full.set_flags(use_tr=1 | 128)
    g = torch.Generator().manual_seed(5)
    w = {k: torch.randn(3, 24, generator=g).to(env["dev"]) for k in ("chosen_response", "rejected_response")}
    res = []
    for eng in (full, env["cxr"]):
        ad.grad.zero_()
        out = _policy(eng, ad, 24, pack)(**_kw(p, eng), temperature=0.9)
        sum((out[k + "_logprobs"] * w[k]).sum() for k in w).backward()
        torch.cuda.synchronize()
        res.append(({k: v.detach().clone() for k, v in out.items()}, ad.grad.clone()))
    for key, ids in (("chosen_response", p["chosen"]), ("rejected_response", p["rejected"])):
        valid = ids != 0
        a, b = res[0][0][key + "_logprobs"], res[1][0][key + "_logprobs"]
        assert bool((b[~valid] == 0).all())
        rel = ((a - b).abs()[valid] / a.abs()[valid].clamp_min(1e-3))
        assert float(rel.mean()) < 5e-4 and float(rel.max()) < 1e-2, (float(rel.mean()), float(rel.max()))
        assert float((res[0][0][key + "_entropies"] - res[1][0][key + "_entropies"]).abs().max()) < 2e-2
    rel = float((res[0][1] - res[1][1]).norm() / res[0][1].norm())
    assert rel < 1e-2, rel
    # block by block: the top layer's own blocks (computed on compact rows) and the layer below it (fed through the scattered gradients)
    d = env["d"]
    n = ad.layer_numel
    for i in range(d.n_layers):
        a, b = res[0][1][i * n:(i + 1) * n], res[1][1][i * n:(i + 1) * n]
        assert float((a - b).norm() / a.norm()) < 1.5e-2, i
    full.close()


@pytest.mark.parametrize("pack", [True, False])
def test_ragged_rows_backward(env, pack):
    ad, p = env["ads"]["pol"], env["p"]
    g = torch.Generator().manual_seed(3)
    w = {k: torch.randn(3, 24, generator=g).to(env["dev"]) for k in ("chosen_response", "rejected_response")}
    res = []
    for eng, hook in ((env["cx"], None), (env["cxr"], None), (env["cxr"], "ranged")):
        ad.grad.zero_()
        pol = _policy(eng, ad, 24, pack)
        pol.layer_done_hook = (lambda i: None) if hook else None
        out = pol(**_kw(p, eng))
        sum((out[k + "_logprobs"] * w[k]).sum() for k in w).backward()
        torch.cuda.synchronize()
        res.append(ad.grad.clone())
    for j in (1, 2):
        rel = float((res[j] - res[0]).norm() / res[0].norm())
        assert rel < 1e-2, f"ragged backward (variant {j}) vs padded backward: rel {rel}"
    assert float((res[1] - res[2]).norm() / res[1].norm()) < 1e-5
    ad.grad.zero_()


def test_ragged_rows_edge_cases(env):
    """No padding at all, a response that is only EOS, a query left-padded down to [image, one token], CoPO 'attention' key mask
    (interior masked image rows stay rows): ragged == padded on the valid tokens."""
    d, dev = env["d"], env["dev"]
    B, Q, T = 3, 10, 7
    g = torch.Generator().manual_seed(77)
    images = torch.randn(B, 3, d.image_size, d.image_size, generator=g).to(BF).to(dev)
    queries = torch.randint(3, d.vocab, (B, Q), generator=g)
    qmask = torch.ones(B, Q, dtype=torch.bool)
    queries[0, 0] = -200
    queries[1, :Q - 2] = 0; qmask[1, :Q - 2] = False
    queries[1, Q - 2] = -200
    queries[2, Q - 1] = -200
    a = torch.randint(3, d.vocab, (B, T), generator=g)
    a[0, 0] = 2; a[0, 1:] = 0
    b = torch.randint(3, d.vocab, (B, T), generator=g)
    b[:, 3] = 2; b[:, 4:] = 0
    resp = {"standard_response": a, "original_generate_response": b}
    im = torch.ones(B, d.n_patches, dtype=torch.bool)
    im[0, :5] = False; im[1, 3] = False; im[2, d.n_patches - 4:] = False
    for qm in (qmask, torch.cat([im, qmask], 1)):
        outs = []
        for eng in (env["cx"], env["cxr"]):
            with torch.no_grad():
                outs.append(_policy(eng, env["ads"]["ref"], T, True)(images=images, queries=queries, queries_attn_masks=qm, **resp))
        torch.cuda.synchronize()
        for k, ids in resp.items():
            valid = (ids != 0).to(dev)
            x, y = outs[0][k + "_logprobs"], outs[1][k + "_logprobs"]
            assert bool(torch.isfinite(y).all()) and bool((y[~valid] == 0).all())
            assert float((x - y).abs()[valid].max()) < 3e-2, (k, float((x - y).abs()[valid].max()))
