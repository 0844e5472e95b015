"""GPU, BASELINE.json's FULL size (LLaVA-1.5-7B: 32 layers, H 4096, FFN 11008, V 32000, LoRA r 256; query 128 + response
384 -> L = 1087): the CPU oracle cannot run here in seconds, so parity is pinned through size-independent properties of
the path (tolerances are bf16 noise levels measured against the oracle at 7B WIDTH in test_parity_gpu.test_wide_model_parity):

  P1  layout invariance   packed ([prefix | chosen | rejected], prefix computed once) == stacked (the reference's two
                          sequences): per-token log-probs, entropies and LoRA gradients
  P2  mask placement      pad cells are exactly 0 (-0.0) in both layouts (Quirk Q4: downstream masks compare with 0)
  P3  batch independence  a pair's log-probs do not depend on which other pairs share the micro-batch: BIT-equal
  P4  backward linearity  grad(2 * dlogp) == 2 * grad(dlogp) up to bf16 rounding; two identical backward passes give BIT-identical gradients
  P5  causality           changing a response token changes only log-probs at and after its position, and nothing of
                          the other response
  P6  rollout             graph-replayed decode == eager decode token for token (same seed), and is deterministic
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd import lib
    from opadpo_amd.dims import LlavaDims
    from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter
    from opadpo_amd.synth import init_lora, init_weights, synth_pairs
    lib.load()
    dev = torch.device("cuda:0")
    d = LlavaDims.llava15_7b()
    base = BaseWeights(d, init_weights(d, seed=0, device=dev), dev, need_backward=True)
    from opadpo_amd.ctx import CtxEngine
    eng = CtxEngine(base)                   # product path: opadpo_ctx on ragged rows
    ad = LoraAdapter(d, init_lora(d, seed=1, device=dev), dev, trainable=True)
    p = synth_pairs(d, 3, 128, 384, seed=5, device=dev)
    feats = eng.encode_images(p["images"])
    yield dict(d=d, eng=eng, ad=ad, p=p, feats=feats, dev=dev)
    eng.release()
    torch.cuda.empty_cache()


def _run(s, pack, sel=slice(None), weights=None, grad=False, p=None):
    from opadpo_amd.policy import AutoregressivePolicy
    p = p or s["p"]
    pol = AutoregressivePolicy(s["eng"], s["ad"], 384, pack_responses=pack)
    kw = dict(queries=p["queries"][sel], queries_attn_masks=p["queries_attn_masks"][sel], image_feats=s["feats"][sel],
              chosen_response=p["chosen"][sel], rejected_response=p["rejected"][sel])
    if not grad:
        with torch.no_grad():
            return pol(**kw)
    s["ad"].grad.zero_()
    out = pol(**kw)
    loss = sum((out[k + "_logprobs"] * weights[k]).sum() for k in ("chosen_response", "rejected_response"))
    loss.backward()
    torch.cuda.synchronize()
    return out, s["ad"].grad.clone()


def _meanrel(a, b, valid):
    return float(((a - b).abs()[valid] / b.abs()[valid].clamp_min(1e-3)).mean())


def test_p1_p2_packed_equals_stacked_at_full_size(full):
    s, p = full, full["p"]
    g = torch.Generator().manual_seed(1)
    w = {k: torch.randn(3, 384, generator=g).to(s["dev"]) for k in ("chosen_response", "rejected_response")}
    op, gp = _run(s, True, weights=w, grad=True)
    os_, gs = _run(s, False, weights=w, grad=True)
    for k, ids in (("chosen_response", p["chosen"]), ("rejected_response", p["rejected"])):
        valid = ids != 0
        for out in (op, os_):          # P2
            lp, en = out[k + "_logprobs"].detach(), out[k + "_entropies"]
            assert bool((lp[~valid] == 0).all()) and bool((en[~valid] == 0).all()) and bool(torch.isfinite(lp).all())
        e = _meanrel(op[k + "_logprobs"].detach(), os_[k + "_logprobs"].detach(), valid)
        assert e < 2.5e-3, f"{k}: packed vs stacked log-probs mean rel {e}"     # two bf16 evaluation orders of the same math
        assert float((op[k + "_entropies"] - os_[k + "_entropies"]).abs().max()) < 5e-2
    cos = float((gp * gs).sum() / (gp.norm() * gs.norm()))
    rel = float((gp - gs).norm() / gs.norm())
    assert cos > 0.999 and rel < 4e-2, (cos, rel)


def test_p3_batch_independence(full):
    s = full
    all3 = _run(s, True)
    one = _run(s, True, sel=slice(1, 2))
    # BIT-equal (round 3 again): every output element of every GEMM is summed in the same k order whichever kernel / tile takes it (the
    # tail of a partly filled round of 256x256 tiles runs as quarter tiles over the full K range; round 2's split-K tail re-associated the
    # sums of the tail tiles, and which tiles are tail tiles depends on the batch's row count), attention and the row-wise kernels never
    # mix sequences: a pair's log-probs do not depend on who shares its micro-batch.
    for k in ("chosen_response", "rejected_response"):
        a, b = all3[k + "_logprobs"][1:2], one[k + "_logprobs"]
        assert torch.equal(a, b), f"{k}: a pair's log-probs depend on its batch neighbours (max abs {float((a - b).abs().max())})"
        assert torch.equal(all3[k + "_entropies"][1:2], one[k + "_entropies"])


def test_p3b_bench_microbatch_22_pairs_equals_3_pair_batch_bit_for_bit(full):
    """Round 6 (VERDICT r05 Missing #2): parity AT THE BENCHMARKED MICRO-BATCH.  bench.py's step runs 22 pairs = ~24.5 k ragged rows per pass: q|k|v / o / the
    plain products take the STREAMING walk of gemm_nt_w4s over ~2 000 tiles per launch, the partial last round its quarter / sixteenth tiles - a walk
    the oracle-checked shapes (P7: 2 pairs; the 4-layer width tests: 7 pairs) never take at full depth.  Here the bench micro-batch runs once through the
    product path exactly as bench.py runs it (context API, packed ragged rows; the POLICY pass as a training forward with the K-concatenated adapter,
    the REFERENCE pass no-grad on the merged copy with the SwiGLU-pair epilogue) and pairs 0-2 must be BIT-equal to the same three pairs run as a
    3-pair batch - the batch size P7 ties to the oracle at full depth.  Every GEMM sums each output element in one k order whatever tile takes it,
    attention and the row-wise kernels never mix sequences: so the 22-pair numbers ARE the oracle-checked numbers."""
    from opadpo_amd.model import LoraAdapter
    from opadpo_amd.policy import AutoregressivePolicy
    from opadpo_amd.synth import init_lora, synth_pairs
    s = full
    d, eng, dev = s["d"], s["eng"], s["dev"]
    p22 = synth_pairs(d, 22, 128, 384, seed=1000, device=dev)          # bench.py's pool entry 0 of rank 0 (seed = 1000 * rank + i)
    feats = eng.encode_images(p22["images"])
    ref_ad = LoraAdapter(d, init_lora(d, seed=2, device=dev), dev, trainable=False)
    ref_ad.merge_into_base(eng.base)

    def passes(sel):
        kw = dict(queries=p22["queries"][sel], queries_attn_masks=p22["queries_attn_masks"][sel], image_feats=feats[sel],
                  chosen_response=p22["chosen"][sel], rejected_response=p22["rejected"][sel])
        with torch.no_grad():
            r = AutoregressivePolicy(eng, ref_ad, 384, pack_responses=True)(**kw)
        o = AutoregressivePolicy(eng, s["ad"], 384, pack_responses=True)(**kw)      # grad mode on: the training forward bench.py times
        out = {("ref", k): v.detach().clone() for k, v in r.items()}
        out.update({("pol", k): v.detach().clone() for k, v in o.items()})
        del r, o
        return out
    big = passes(slice(None))
    rows = int(p22["queries_attn_masks"].sum()) + 22 * (d.n_patches - 1) + int((p22["chosen"] != 0).sum()) + int((p22["rejected"] != 0).sum())
    assert rows >= 256 * 2 * 256 // (d.hidden // 256), rows      # >= 2 tiles per CU on the N = 4096 products: the streaming kernel walks them
    small = passes(slice(0, 3))
    torch.cuda.synchronize()
    for key in big:
        a, b = big[key][:3], small[key]
        assert torch.equal(a, b), f"{key}: pairs 0-2 of the 22-pair bench micro-batch differ from the 3-pair batch (max abs {float((a - b).abs().max())})"
    for k, ids in (("chosen_response", p22["chosen"]), ("rejected_response", p22["rejected"])):
        lp = big[("pol", k + "_logprobs")]
        assert bool((lp[ids == 0] == 0).all()) and bool(torch.isfinite(lp).all()) and float(lp[ids != 0].mean()) < -5.0
    ref_ad.merged = None
    del big, small, feats, ref_ad
    eng.release()                   # opadpo_ctx_trim: the context keeps its activation arena (here: the 22-pair training forward's ~90 GB) for the next batch of the same shape
    torch.cuda.empty_cache()


def test_p4_backward_is_linear_in_dlogp(full):
    s = full
    g = torch.Generator().manual_seed(2)
    w = {k: torch.randn(3, 384, generator=g).to(s["dev"]) for k in ("chosen_response", "rejected_response")}
    _, g1 = _run(s, True, weights=w, grad=True)
    _, g2 = _run(s, True, weights={k: 2 * v for k, v in w.items()}, grad=True)
    rel = float((g2 - 2 * g1).norm() / (2 * g1).norm())
    assert rel < 2e-3, rel          # bf16 activation-gradient rounding is scale-invariant
    # round 4: the LoRA wgrads flush through a workspace + ordered reduce instead of fp32 atomics - the whole gradient of the full-size
    # model is BIT-reproducible run to run (the last non-deterministic kernel of the training step is gone)
    _, g1b = _run(s, True, weights=w, grad=True)
    assert torch.equal(g1, g1b), f"LoRA gradient differs between two identical backward passes: rel {float((g1 - g1b).norm() / g1.norm())}"


def test_p5_causality_and_response_isolation(full):
    s, p = full, full["p"]
    base = _run(s, True, sel=slice(0, 1))
    q = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in p.items()}       # (the host row plan entries are not tensors to edit)
    pos = 40
    assert int(q["chosen"][0, pos]) != 0
    q["chosen"][0, pos] = 3 + (int(q["chosen"][0, pos]) - 2) % 1000
    mod = _run(s, True, sel=slice(0, 1), p=q)
    c0, c1 = base["chosen_response_logprobs"][0], mod["chosen_response_logprobs"][0]
    assert torch.equal(c0[:pos], c1[:pos]), "tokens before the edit changed (causality)"
    assert not torch.equal(c0[pos:], c1[pos:])
    assert torch.equal(base["rejected_response_logprobs"], mod["rejected_response_logprobs"]), "the other response saw the edit"


def test_p6_rollout_graph_equals_eager(full):
    from opadpo_amd.generate import Generator
    s, p = full, full["p"]
    outs = []
    for use_graph in (True, False, True):
        gen = Generator(s["eng"], None, use_graph=use_graph)
        outs.append(gen.generate(p["queries"], p["queries_attn_masks"], image_feats=s["feats"], max_new_tokens=12, top_k=30, top_p=0.95,
                                 seed=4, suppress_eos=True))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert int((outs[0] == 2).sum()) == 0      # EOS suppressed; ids 0 / 1 are ordinary vocabulary entries of a random-init model


# ---- P7: the benchmarked path against the oracle AT THE BENCHMARKED DEPTH ---------------------------------------------------
def _oracle_full_pass(LR, W, lora, od, images, queries, qmask, resp, emulate_bf16):
    """One evaluation of oracle/llava_ref.py on the stacked sequences (rl_models.py:95-132), keeping the residual stream after every
    decoder layer.  -> ({key_logprobs}, [x after layer i: [K*B, L, H]])"""
    from oracle.dpo_ref import policy_head, response_keys, stack_policy_inputs
    keys = response_keys(resp)
    ids, mask = stack_policy_inputs(queries, qmask, resp)
    B, Q = queries.shape
    T = resp[keys[0]].shape[1]
    feats = LR.image_features(images, W, None, od, emulate_bf16).repeat(len(keys), 1, 1)
    x, km = LR.splice(ids, mask, feats, W[LR.LLM_PREFIX + "embed_tokens.weight"], od.n_patches, None)
    h, layers = LR.llama_decoder(x, km, W, lora, od, emulate_bf16, return_layers=True)
    lp, _ = policy_head(LR.lm_logits(h, W, od, emulate_bf16), ids, Q, T, 1.0)
    return {k: lp[i * B:(i + 1) * B] for i, k in enumerate(keys)}, layers


def _stats(got, want, valid):
    r = ((got - want).abs()[valid] / want.abs()[valid].clamp_min(1e-3)).double()
    return {"mean": float(r.mean()), "p99": float(torch.quantile(r, 0.99)), "max": float(r.max())}


def _p7_body(full, od, report_name, model_name):
    """BASELINE.json's configuration at its REAL depth: LLaVA-1.5-7B, 32 decoder layers, CLIP-L/14-336, 2 synthetic pairs at seq512,
    the product path (context API, packed ragged rows, trained adapter K-concatenated, frozen adapter merged + SwiGLU-pair) against
    oracle/llava_ref.py evaluated in fp32 and with bf16 emulated at the HIP pipeline's HBM write points
    (rl_models.py:114-132, utils/common_utils.py:112-118).  Records, per response token: relative log-prob error mean / p99 / max;
    the drift curve of the fp32 residual stream layer by layer; and the error of the LOG-RATIO pi - ref when policy and reference hold the
    SAME adapter (exactly 0 in the reference, which runs both through identical PEFT code; here the merged copy rounds differently) -
    the quantity dpo_loss consumes (dpo_trainer.py:444-449).  Report -> gpurun_out/parity_fulldepth.json (committed copy:
    profiles/r*_parity_fulldepth.json, quoted by bench.py's `parity` record).

    north_star asks for 1e-3 relative.  A 32-layer random-init bf16 pipeline does not reach that against fp32 arithmetic - the
    oracle's OWN bf16 emulation does not either - so the assertions pin the HIP path to the oracle's bf16 realisation (never further
    from fp32 than 1.35 x that realisation is) and cap the absolute numbers; the measured distances are what the report states."""
    import ctypes as C
    import json
    import os
    import time
    from opadpo_amd import lib as L
    from opadpo_amd.model import LoraAdapter
    from opadpo_amd.policy import AutoregressivePolicy, host_row_plan
    from opadpo_amd.synth import init_lora, init_weights, synth_pairs
    from oracle import llava_ref as LR
    s = full
    d, eng, dev = s["d"], s["eng"], s["dev"]
    assert (od.hidden, od.n_layers, od.ffn, od.vocab, od.v_layers, od.image_size) == (d.hidden, d.n_layers, d.ffn, d.vocab, d.v_layers, d.image_size)
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    t_start = time.time()
    # the fixture's weights again (device generators are deterministic), brought to the host for the oracle
    # The oracle (oracle/llava_ref.py, the same torch code the CPU tests pin against transformers) is evaluated on ODEV: the accelerator by
    # default - three full-depth passes cost 350 s of host time at 7B and do not exist at 13B (2 x 52 GB of fp32 weights); torch's fp32
    # kernels there, not this library's (tests/test_parity_gpu.py::test_oracle_is_device_independent).  OPADPO_ORACLE_DEVICE=cpu restores the host.
    odev = torch.device(os.environ.get("OPADPO_ORACLE_DEVICE", "cuda:0"))
    Wd = init_weights(d, seed=0, device=dev)
    W = {k: v.to(odev).float() for k, v in Wd.items()}
    del Wd
    lora_d = init_lora(d, seed=1, device=dev)
    lora = {k: v.to(odev).float() for k, v in lora_d.items()}
    ref_ad = LoraAdapter(d, lora_d, dev, trainable=False)      # the SAME adapter as the policy's, frozen and merged (bench.py / CLI default)
    assert torch.equal(s["ad"].work[:65536].cpu(), ref_ad.work[:65536].cpu()), "the fixture's policy adapter is not init_lora(seed=1) any more"
    del lora_d
    B, Q, T = 2, 128, 384
    p = synth_pairs(d, B, Q, T, seed=21)
    images, queries, qmask = p["images"].float(), p["queries"], p["queries_attn_masks"]
    resp = {"chosen_response": p["chosen"], "rejected_response": p["rejected"]}
    # ---- HIP: the reference pass UNMERGED (the frozen adapter through the same K-concatenated kernels as the policy: --no-merge-ref /
    # --merge_ref_adapter 0), then MERGED (bench.py default; opadpo_train --merge_ref_adapter 1), then the training forward of the policy (activations kept)
    kw = dict(images=p["images"].to(dev), queries=queries, queries_attn_masks=qmask, **resp)
    with torch.no_grad():
        r_unm = {k: v.cpu() for k, v in AutoregressivePolicy(eng, ref_ad, T, pack_responses=True)(**kw).items()}
    ref_ad.merge_into_base(eng.base)
    with torch.no_grad():
        r_out = {k: v.cpu() for k, v in AutoregressivePolicy(eng, ref_ad, T, pack_responses=True)(**kw).items()}
    pol = AutoregressivePolicy(eng, s["ad"], T, pack_responses=True)
    keys, batch = pol.build_batch(queries, qmask, resp)
    assert batch.row_plan is not None and batch.K == 2
    feats = eng.encode_images(kw["images"])
    logp, _, sv = eng.seq_logprobs_fwd(s["ad"], batch, feats, 1.0, train=True)
    p_out = {k + "_logprobs": logp[i * B:(i + 1) * B].cpu() for i, k in enumerate(keys)}
    lead, lens = host_row_plan(queries, qmask, resp)
    assert torch.equal(batch.row_plan[:, 0], lead) and torch.equal(batch.row_plan[:, 1], lens["chosen_response"])
    pfx = Q + d.n_patches - 1
    n_rows = C.c_int(0)
    eng._call("opadpo_saved_residual", sv.handle, 0, None, C.byref(n_rows), L.stream())
    M = n_rows.value
    assert M == sum(pfx - int(lead[b]) + int(lens["chosen_response"][b]) + int(lens["rejected_response"][b]) for b in range(B))
    hip_x = []
    buf = torch.empty(M, d.hidden, dtype=torch.float32, device=dev)
    for i in range(1, d.n_layers):                 # x entering layer i = the residual stream after layer i - 1
        eng._call("opadpo_saved_residual", sv.handle, i, buf.data_ptr(), None, L.stream())
        hip_x.append(buf.cpu())
    sv.release()
    # compact row -> (stacked oracle sequence, padded position): prefix rows from the chosen sequence, response a from sequence a*B + b
    seq_idx, pos_idx = [], []
    for b in range(B):
        ld = int(lead[b])
        seq_idx += [b] * (pfx - ld)
        pos_idx += list(range(ld, pfx))
        for a, k in enumerate(keys):
            n = int(lens[k][b])
            seq_idx += [a * B + b] * n
            pos_idx += list(range(pfx, pfx + n))
    seq_idx, pos_idx = torch.tensor(seq_idx), torch.tensor(pos_idx)
    assert len(seq_idx) == M
    t_hip = time.time()
    # ---- oracle: fp32, and bf16 emulated at the HBM write points ----
    rep = {"model": model_name, "oracle_device": str(odev), "layers": d.n_layers, "pairs": B, "query_len": Q, "response_len": T, "rows": M,
           "path": "opadpo_ctx, packed ragged rows; policy = K-concatenated LoRA, reference = merged copy + SwiGLU-pair epilogue"}
    oracle_lp = {}
    with torch.no_grad():
        # every run measures its own floors: the fp32 pass, the bf16-emulating pass and the merged emulation (no committed constants)
        o_in = (images.to(odev), queries.to(odev), qmask.to(odev), {k: v.to(odev) for k, v in resp.items()})
        sidx, pidx = seq_idx.to(odev), pos_idx.to(odev)
        for name, emu in (("fp32", False), ("emu_bf16", True)):
            t0 = time.time()
            lp, layers = _oracle_full_pass(LR, W, lora, od, *o_in, emu)
            oracle_lp[name] = {k: v.cpu() for k, v in lp.items()}
            drift = []
            for i in range(d.n_layers - 1):
                want = layers[i][sidx, pidx].cpu()
                drift.append(float((hip_x[i] - want).norm() / want.norm()))
            rep[f"residual_drift_vs_{name}"] = drift
            if name == "fp32":
                f32_layers = [l_[sidx, pidx].cpu() for l_ in layers[:-1]]
            else:
                rep["residual_drift_emu_vs_fp32"] = [float((layers[i][sidx, pidx].cpu() - f32_layers[i]).norm() / f32_layers[i].norm())
                                                     for i in range(d.n_layers - 1)]
            rep[f"oracle_{name}_seconds"] = time.time() - t0
            del layers
        if True:
            # the oracle's OWN merged-vs-unmerged distance under bf16 emulation (weights W + s B A rounded once to bf16, like the HIP merge): the
            # yardstick for the log-ratio noise a merged reference copy puts under a policy that holds the same adapter
            t0 = time.time()
            Wm, rest = LR.merge_llm_lora(W, lora, od, emulate_bf16=True)
            lp_m, layers = _oracle_full_pass(LR, Wm, rest, od, *o_in, True)
            del layers, Wm
            oracle_lp["emu_bf16_merged"] = {k: v.cpu() for k, v in lp_m.items()}
            rep["oracle_emu_merged_seconds"] = time.time() - t0
    worst = {}
    for k in keys:
        valid = resp[k] != 0
        kk = k
        cmp_ = [("policy_vs_fp32", p_out[k + "_logprobs"], oracle_lp["fp32"][kk]), ("ref_merged_vs_fp32", r_out[k + "_logprobs"], oracle_lp["fp32"][kk])]
        if "emu_bf16" in oracle_lp:
            cmp_ += [("policy_vs_emu", p_out[k + "_logprobs"], oracle_lp["emu_bf16"][kk]),
                     ("ref_merged_vs_emu_unmerged", r_out[k + "_logprobs"], oracle_lp["emu_bf16"][kk]),
                     ("oracle_emu_vs_fp32", oracle_lp["emu_bf16"][kk], oracle_lp["fp32"][kk])]
        for name, got, want in cmp_:
            assert bool((got[~valid] == 0).all()) and bool((want[~valid] == 0).all())            # exact zeros on pad cells (Quirk Q4)
            st = _stats(got, want, valid)
            rep[f"{name}_{k}"] = st
            w = worst.setdefault(name, {"mean": 0.0, "p99": 0.0, "max": 0.0})
            for f in w:
                w[f] = max(w[f], st[f])
    rep["worst"] = worst
    # log-ratio at policy == reference adapter: the reference computes exactly 0 (dpo_trainer.py:444-449 cl = pi_c - ref_c)
    lr = {}
    dl = []
    for k in keys:
        valid = resp[k] != 0
        dlt = (p_out[k + "_logprobs"] - r_out[k + "_logprobs"])[valid].double()
        lr[k] = {"mean_abs": float(dlt.abs().mean()), "p99_abs": float(torch.quantile(dlt.abs(), 0.99)), "max_abs": float(dlt.abs().max()),
                 "mean_signed": float(dlt.mean())}
        dl.append(p_out[k + "_logprobs"] - r_out[k + "_logprobs"])
    both = (resp["chosen_response"] != 0) & (resp["rejected_response"] != 0)
    z = 0.1 * (dl[0] - dl[1])[both].double()                # beta * (chosen log-ratio - rejected log-ratio): the sigmoid's argument (0 in the reference)
    lr["dpo_logit_beta_0.1"] = {"mean_abs": float(z.abs().mean()), "max_abs": float(z.abs().max()),
                                "loss_shift_mean": float((torch.nn.functional.softplus(-z) - 0.6931471805599453).mean())}
    rep["logratio_policy_eq_reference"] = lr
    # the same quantity with the reference pass UNMERGED: policy and reference run the same kernels on the same adapter bits
    lru = {}
    for k in keys:
        valid = resp[k] != 0
        dlt = (p_out[k + "_logprobs"] - r_unm[k + "_logprobs"])[valid].double()
        lru[k] = {"mean_abs": float(dlt.abs().mean()), "max_abs": float(dlt.abs().max())}
    rep["logratio_policy_eq_reference_unmerged"] = lru
    if "emu_bf16_merged" in oracle_lp:
        fl = {}
        for k in keys:
            valid = resp[k] != 0
            dlt = (oracle_lp["emu_bf16"][k] - oracle_lp["emu_bf16_merged"][k])[valid].double()
            fl[k] = {"mean_abs": float(dlt.abs().mean()), "p99_abs": float(torch.quantile(dlt.abs(), 0.99)), "max_abs": float(dlt.abs().max())}
        rep["logratio_oracle_emu_merged_vs_unmerged"] = fl
    rep["seconds_total"] = time.time() - t_start
    rep["seconds_hip_side"] = t_hip - t_start
    rep["bench_line"] = {"layers": d.n_layers, "pairs": B, "vs": "oracle/llava_ref.py fp32", "mean": worst["policy_vs_fp32"]["mean"],
                         "p99": worst["policy_vs_fp32"]["p99"], "max": worst["policy_vs_fp32"]["max"],
                         "reference_pass": worst["ref_merged_vs_fp32"], "oracle_bf16_vs_fp32": worst.get("oracle_emu_vs_fp32"),
                         "vs_bf16_oracle": worst.get("policy_vs_emu"), "logratio_equal_adapters_mean_abs": max(lr[k]["mean_abs"] for k in keys),
                         "logratio_equal_adapters_unmerged_max_abs": max(lru[k]["max_abs"] for k in keys),
                         "logratio_oracle_emulation_merged_vs_unmerged_mean_abs": (max(v["mean_abs"] for v in rep["logratio_oracle_emu_merged_vs_unmerged"].values())
                                                                                   if "logratio_oracle_emu_merged_vs_unmerged" in rep else None),
                         "north_star_tolerance": 1e-3}
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, report_name), "w") as f:
        json.dump(rep, f, indent=1)
    print("[p7]", json.dumps({"worst": worst, "logratio": lr, "logratio_unmerged": lru, "t": rep["seconds_total"]}))
    ref_ad.merged = None
    torch.cuda.empty_cache()
    # the HIP path is a bf16 realisation of the oracle's function: never further from fp32 than 1.35 x the oracle's own bf16 emulation,
    # measured in THIS run on the same inputs
    floor = worst["oracle_emu_vs_fp32"]
    for a in ("policy_vs_fp32", "ref_merged_vs_fp32"):
        assert worst[a]["mean"] <= 1.35 * floor["mean"] + 1e-4, (a, worst[a], floor)
        assert worst[a]["p99"] <= 1.35 * floor["p99"] + 5e-4, (a, worst[a], floor)
    assert worst["policy_vs_fp32"]["max"] < 0.05
    assert max(rep["residual_drift_vs_fp32"]) < 5e-2
    # log-ratio at policy == reference adapter (0 in the reference, dpo_trainer.py:444-449, 997-1016).  UNMERGED: the two passes run the same
    # kernels on the same bits - exactly 0 here too.  MERGED (default): bounded by 1.5 x what the oracle's own bf16 emulation puts between a
    # merged and an unmerged evaluation of the same adapter (measured in this run)
    assert max(lru[k]["max_abs"] for k in keys) == 0.0, lru
    lfloor = max(v["mean_abs"] for v in rep["logratio_oracle_emu_merged_vs_unmerged"].values())
    assert max(lr[k]["mean_abs"] for k in keys) <= 1.5 * lfloor, (lr, lfloor)


def test_p7_full_depth_32_layers_against_the_oracle(full):
    from oracle import llava_ref as LR
    # round 4's figures of the same measurement (profiles/r04_parity_fulldepth.json): the emulation 2.06e-3 / 6.7e-3 from fp32, the HIP policy pass
    # 2.09e-3 / 7.7e-3, the merged reference pass 2.65e-3 / 8.4e-3; the oracle's merged-vs-unmerged emulation 0.0322 nat per token, the HIP log-ratio 0.0335
    _p7_body(full, LR.LlavaDims(), "parity_fulldepth.json", "LLaVA-1.5-7B")


def test_p7_13b_full_depth_40_layers_against_the_oracle():
    """BASELINE.json configs[3]'s model at its real depth (LLaVA-1.5-13B: 40 layers, H 5120, 40 heads, FFN 13824), same measurement as P7 with
    the SAME independent floors: the oracle's fp32 pass, its bf16-emulating pass and its merged emulation all run in this test (on the
    accelerator: 52 GB of fp32 weights per copy; round 4 derived this model's floors from the HIP path's own distances).  Report:
    gpurun_out/parity_fulldepth_13b.json."""
    from opadpo_amd import lib
    from opadpo_amd.ctx import CtxEngine
    from opadpo_amd.dims import LlavaDims
    from opadpo_amd.model import BaseWeights, LoraAdapter
    from opadpo_amd.synth import init_lora, init_weights
    from oracle import llava_ref as LR
    lib.load()
    dev = torch.device("cuda:0")
    d = LlavaDims.llava15_13b()
    base = BaseWeights(d, init_weights(d, seed=0, device=dev), dev, need_backward=True)
    eng = CtxEngine(base)
    ad = LoraAdapter(d, init_lora(d, seed=1, device=dev), dev, trainable=True)
    try:
        _p7_body(dict(d=d, eng=eng, ad=ad, dev=dev), LR.LlavaDims(hidden=d.hidden, n_layers=d.n_layers, n_heads=d.n_heads, ffn=d.ffn),
                 "parity_fulldepth_13b.json", "LLaVA-1.5-13B")
    finally:
        eng.release()
        torch.cuda.empty_cache()
