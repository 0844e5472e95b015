"""CPU: host-side logic of the product package (no kernels): the loss / statistics mirror against the
reference's golden vectors, batch stacking, schedules, checkpoint naming, geometry arithmetic."""
import json
import os
import types

import numpy as np
import pytest
import torch

from opadpo_amd import dims as DM
from opadpo_amd import losses as LS
from opadpo_amd.optim import cosine_lr, shard_bounds


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_product_dpo_loss_matches_reference(golden_dir):
    g = load(golden_dir, "ref_dpo_loss.npz")
    pc, pr, rc, rr, sc, sr = (t(g[k]) for k in ("pc", "pr", "rc", "rr", "sc", "sr"))
    for i, meta in enumerate(g["meta"]):
        fdiv, ls, rf, scores = str(meta).split("|")
        a = LS.DPOArgs(f_divergence_type=fdiv, label_smoothing=float(ls), reference_free=bool(int(rf)))
        out = LS.dpo_loss(a, pc, pr, rc, rr, sc if int(scores) else None, sr if int(scores) else None)
        for got, key in zip(out, ("losses", "c", "r")):
            np.testing.assert_allclose(got.numpy(), g[f"case{i}_{key}"], rtol=1e-6, atol=1e-7)


def test_product_policy_loss_matches_reference(golden_dir):
    g = load(golden_dir, "ref_policy_loss.npz")
    for ci, meta in enumerate(g["meta"]):
        CoPO, AncPO, mdpo, detailed = (bool(int(x)) for x in str(meta).split("|"))
        pre = f"c{ci}_"
        rollouts = {k[len(pre) + 3:]: t(v) for k, v in g.items() if k.startswith(pre + "in_")}
        pol = {k[len(pre) + 4:]: t(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith(pre + "pol_")}
        a = LS.DPOArgs(CoPO=CoPO, AncPO=AncPO, mDPO_anchor=mdpo, detailed_report=detailed)
        loss, stats = LS.policy_loss(a, rollouts, {k: v for k, v in pol.items() if not k.startswith("mask_")},
                                     {k: v for k, v in pol.items() if k.startswith("mask_")} if CoPO else None)
        np.testing.assert_allclose(loss.item(), g[pre + "loss"], rtol=1e-6)
        loss.backward()
        for k, v in pol.items():
            want = g[pre + "grad_" + k]
            got = v.grad.numpy() if v.grad is not None else np.zeros_like(want)
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-8)
        for k, v in g.items():
            if k.startswith(pre + "stat_"):
                np.testing.assert_allclose(stats[k[len(pre) + 5:].replace("__", "/")].numpy(), v, rtol=1e-5, atol=1e-7)
        assert len(stats) == 32


def test_mask_image_matches_reference(golden_dir):
    g = load(golden_dir, "ref_mask_image.npz")
    torch.manual_seed(99)
    np.testing.assert_array_equal(LS.mask_single_image(t(g["img"]), 0.3, "random").numpy(), g["random"])
    torch.manual_seed(99)
    np.testing.assert_array_equal(LS.mask_single_image(t(g["img"]), 0.3, "blockwise").numpy(), g["blockwise"])


def test_build_batch_stacking(golden_dir):
    """kwarg filter, key order on the batch dimension and masks of rl_models.py:91-112."""
    from opadpo_amd.policy import AutoregressivePolicy, response_keys
    kw = dict(standard_response=1, standard_response_attention_mask=2, AI_pseudo_response_scores=3,
              AI_pseudo_response_image_relations=4, original_generate_response=5, mask_standard_response=6)
    assert response_keys(kw) == ["standard_response", "original_generate_response", "mask_standard_response"]
    eng = types.SimpleNamespace(dev=torch.device("cpu"), d=DM.LlavaDims.tiny())
    ad = types.SimpleNamespace(trainable=False)
    pol = AutoregressivePolicy(eng, ad, response_len=5, pack_responses=False)      # the reference's stacked layout
    g = load(golden_dir, "ref_policy_forward.npz")
    queries, qmask = t(g["queries"]), t(g["qmask"]).bool()
    resp = {k[5:]: t(v) for k, v in g.items() if k.startswith("resp_")}
    keys, b = pol.build_batch(queries, qmask, resp)
    from oracle import dpo_ref as D
    ids, mask = D.stack_policy_inputs(queries, qmask, resp)
    assert keys == D.response_keys(resp)
    assert torch.equal(b.ids.long(), ids) and torch.equal(b.text_mask.bool(), mask)
    assert b.feat_row.tolist() == [0, 1, 0, 1] and b.T == 5
    # CoPO 'attention': [image mask (P) | query mask (Q)]
    P = eng.d.n_patches
    im = torch.ones(2, P, dtype=torch.bool)
    im[0, 3] = False
    keys, b2 = pol.build_batch(queries, torch.cat([im, qmask], 1), resp)
    assert torch.equal(b2.text_mask.bool(), mask) and torch.equal(b2.image_mask.bool(), im.repeat(2, 1))
    # packed layout (default): ONE row per sample = [query | response_0 | response_1 | ...]; the stacked rows of the
    # reference are recovered by slicing segment k out of it
    K, B, Q, T = len(keys), queries.shape[0], queries.shape[1], 5
    polp = AutoregressivePolicy(eng, ad, response_len=5)
    keys_p, bp = polp.build_batch(queries, torch.cat([im, qmask], 1), resp)
    assert keys_p == keys and bp.K == K and bp.T == T and tuple(bp.ids.shape) == (B, Q + K * T)
    assert bp.feat_row.tolist() == list(range(B)) and torch.equal(bp.image_mask.bool(), im)
    for k in range(K):
        seg = slice(Q + k * T, Q + (k + 1) * T)
        assert torch.equal(torch.cat([bp.ids[:, :Q], bp.ids[:, seg]], 1).long(), ids[k * B:(k + 1) * B])
        assert torch.equal(torch.cat([bp.text_mask[:, :Q], bp.text_mask[:, seg]], 1).bool(), mask[k * B:(k + 1) * B])
    assert abs(DM.pair_flops_packed(DM.LlavaDims(), 128, 384, 2) / DM.pair_flops(DM.LlavaDims(), 128, 384) - 0.682) < 0.01


def test_host_row_plan_and_ragged_batches():
    """The ragged-row plan is computed on the HOST from the collated tensors (policy.host_row_plan): leading masked query positions in
    front of the image token, response lengths up to the last non-pad token; a ragged engine receives it through build_batch without any
    device tensor being read (host inputs -> derived there; `row_lead` / `row_lens` -> taken as given; cache hits keep the caller's key
    names; a key missing from row_lens falls back to the derived plan)."""
    from opadpo_amd.policy import AutoregressivePolicy, host_row_plan
    from opadpo_amd.synth import synth_pairs
    d = DM.LlavaDims.tiny()
    p = synth_pairs(d, 5, 16, 24, seed=11)
    q, qm = p["queries"].clone(), p["queries_attn_masks"].clone()
    q[0, :] = 7; q[0, 0] = DM.IMAGE_TOKEN_INDEX; qm[0, :] = True; qm[0, 0] = False      # image token at slot 0 behind a masked slot: lead = min(pads, image position) = 0
    resp = {"chosen_response": p["chosen"].clone(), "rejected_response": p["rejected"].clone()}
    resp["chosen_response"][1, :] = 0                                                    # an all-pad response: length 0
    resp["rejected_response"][2, 5] = 0                                                  # a pad INSIDE a response does not end it
    lead, lens = host_row_plan(q, qm, resp)
    for b in range(5):
        img = int((q[b] == DM.IMAGE_TOKEN_INDEX).nonzero()[0])
        n_pad = 0
        while n_pad < 16 and not bool(qm[b, n_pad]):
            n_pad += 1
        assert int(lead[b]) == min(n_pad, img)
        for k, ids in resp.items():
            nz = (ids[b] != 0).nonzero()
            assert int(lens[k][b]) == (int(nz[-1]) + 1 if nz.numel() else 0)
    assert int(lead[0]) == 0 and int(lens["chosen_response"][1]) == 0 and int(lens["rejected_response"][2]) > 5
    eng = types.SimpleNamespace(dev=torch.device("cpu"), d=d, ragged=True)
    ad = types.SimpleNamespace(trainable=False)
    for pack in (True, False):
        pol = AutoregressivePolicy(eng, ad, response_len=24, pack_responses=pack)
        keys, b1 = pol.build_batch(q, qm, resp)                                        # host tensors: plan derived on the host
        want = torch.stack([lead, lens["chosen_response"], lens["rejected_response"]], 1) if pack else \
            torch.stack([lead.repeat(2), torch.cat([lens["chosen_response"], lens["rejected_response"]])], 1)
        assert b1.row_plan.dtype == torch.int32 and torch.equal(b1.row_plan, want)
        eng.__dict__.pop("_batch_cache", None)
        fake = (lead + 1, {k: v - 1 for k, v in lens.items()})                           # a GIVEN plan is taken as is
        _, b2 = pol.build_batch(q, qm, resp, fake[0], fake[1])
        assert torch.equal(b2.row_plan[:, 0], fake[0] if pack else fake[0].repeat(2))
        # cache: same tensors under other names -> the caller's names, the cached batch
        other = {"mask_standard_response": resp["chosen_response"], "mask_AI_pseudo_response": resp["rejected_response"]}
        fake_other = {"mask_standard_response": fake[1]["chosen_response"], "mask_AI_pseudo_response": fake[1]["rejected_response"]}
        keys3, b3 = pol.build_batch(q, qm, other, fake[0], fake_other)
        assert keys3 == list(other) and b3 is b2
        # ... but the row plan is part of the key: the same tensors WITHOUT a plan (or with another one) get their own batch
        _, b4 = pol.build_batch(q, qm, other)
        assert b4 is not b2 and torch.equal(b4.row_plan, want)
        # half a plan is no plan: lead without lens falls back to the derived plan instead of failing on None
        _, b5 = pol.build_batch(q, qm, other, fake[0], None)
        assert b5 is b4
        eng.__dict__.pop("_batch_cache", None)


def test_schedule_shards_and_arith():
    assert cosine_lr(0, 1e-6, 5, 300) == 0.0 and abs(cosine_lr(5, 1e-6, 5, 300) - 1e-6) < 1e-18
    assert cosine_lr(300, 1e-6, 5, 300) < 1e-12
    n = 639_631_360
    covered = 0
    for r in range(8):
        lo, hi, per = shard_bounds(n, 8, r)
        assert per % 256 == 0 and lo == min(n, r * per)
        covered += hi - lo
    assert covered == n
    d = DM.LlavaDims()
    assert DM.lora_param_count(d) == 639_631_360
    assert DM.lora_param_count(DM.LlavaDims.llava15_13b()) == 1_001_390_080
    assert abs(DM.pair_flops(d, 128, 384) / 1e12 - 101.8) < 0.5
    d.validate()
    DM.LlavaDims.llava15_13b().validate()
    DM.LlavaDims.tiny().validate()


def test_checkpoint_layout(tmp_path, golden_dir):
    from opadpo_amd.trainer import get_last_checkpoint, save_adapter
    g = load(golden_dir, "ref_last_checkpoint.npz")
    run = tmp_path / "run"
    assert [str(x) for x in get_last_checkpoint(str(tmp_path / "nope"))] == list(g["first"])
    (run / "checkpoint-75").mkdir(parents=True)
    (run / "checkpoint-150").mkdir()
    (run / "checkpoint-final").mkdir()
    path, done = get_last_checkpoint(str(run))
    assert [os.path.basename(path), str(done)] == list(g["found"])
    (run / "completed").touch()
    assert [str(x) for x in get_last_checkpoint(str(run))] == list(g["done"])

    d = DM.LlavaDims.tiny()

    class FakeAdapter:
        def to_peft_state(self):
            return {"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": torch.zeros(d.lora_r, d.hidden, dtype=torch.bfloat16)}

    out = tmp_path / "ckpt" / "adapter_model" / "lora_policy"
    save_adapter(FakeAdapter(), str(out), d, "llava-1.5-7b")
    cfg = json.load(open(out / "adapter_config.json"))
    assert cfg["peft_type"] == "LORA" and cfg["r"] == d.lora_r and cfg["inference_mode"] is True
    assert cfg["target_modules"] == ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]
    sd = torch.load(out / "adapter_model.bin")
    assert list(sd) == ["base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"]


def test_lora_flat_layout_roundtrip():
    """PEFT key layout <-> fused flat buffer (CPU tensors; no kernels involved for a frozen adapter)."""
    from opadpo_amd.model import LoraAdapter
    from opadpo_amd.synth import init_lora
    d = DM.LlavaDims.tiny()
    st = init_lora(d, seed=3)
    ad = LoraAdapter(d, st, torch.device("cpu"), trainable=False)
    assert ad.numel == DM.lora_param_count(d)
    back = ad.to_peft_state()
    assert set(back) == set(st)
    for k in st:
        assert torch.equal(back[k], st[k].to(torch.bfloat16)), k
    r, H = d.lora_r, d.hidden
    assert torch.equal(ad.w(1, "a_qkv")[r:2 * r], st["base_model.model.model.layers.1.self_attn.k_proj.lora_A.weight"])
    assert torch.equal(ad.w(0, "b_gu")[d.ffn:], st["base_model.model.model.layers.0.mlp.up_proj.lora_B.weight"])


def test_collator_matches_reference(golden_dir):
    """DataCollatorForCausalLM against the reference's own collator (golden, toy HF-style tokenizer)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from toy_tokenizer import ToyTokenizer, collator_instances
    from opadpo_amd import data as DT
    g = load(golden_dir, "ref_collator.npz")
    for detailed in (False, True):
        coll = DT.DataCollatorForCausalLM(tokenizer=ToyTokenizer(), query_len=24, response_len=40, detailed_report=detailed)
        batch = coll(collator_instances())
        keys = {k[3:] for k in g if k.startswith(f"d{int(detailed)}_")}
        assert set(batch) == keys, set(batch) ^ keys
        for k in keys:
            got = batch[k]
            got = got.to(torch.uint8).numpy() if got.dtype == torch.bool else got.numpy()
            np.testing.assert_array_equal(got, g[f"d{int(detailed)}_{k}"], err_msg=f"detailed={detailed} {k}")
    assert (batch["queries"] == -200).sum() == 2            # one image token per row
    assert list(DT.complete_copied_content("a b c. d e f. g", ["a b", "d e f.", ""])) == list(g["h_complete"])
    assert list(DT.complete_copied_content("a b c", ["zzz", "a"])) == list(g["h_complete_fail"])
    # malformed report -> plain tokenisation with all-zero weights (data_utils_dpo.py:259-278)
    inst = collator_instances()
    inst[0]["AI_json_report"] = '{"Sentence 1": {"score": 4}}'
    b2 = DT.DataCollatorForCausalLM(tokenizer=ToyTokenizer(), query_len=24, response_len=40, detailed_report=True)(inst)
    assert float(b2["AI_pseudo_response_scores"].abs().sum()) == 0.0


def test_dataset_and_image_preprocessing():
    import base64
    import io
    from PIL import Image
    from opadpo_amd import data as DT
    img = Image.new("RGB", (50, 30), (200, 10, 10))
    buf = io.BytesIO()
    img.save(buf, format="PNG")
    rows = [{"queries": "<image>\nWhat is this?", "image_bytes": base64.b64encode(buf.getvalue()).decode(), "standard_response": "s",
             "original_generate_response": "o", "AI_pseudo_response": "a", "AI_json_report": "{}"}]
    item = DT.DPODataset(rows, image_size=28)[0]
    assert item["images"].shape == (3, 28, 28)
    assert item["queries"].startswith("<s> A chat between a curious user") and "图 \nWhat is this? ASSISTANT: " in item["queries"]
    px = item["images"]
    # centre = the red image, top/bottom bands = CLIP-mean padding (normalises to ~0)
    assert abs(float(px[:, 0, 14].abs().max())) < 0.05 and float(px[0, 14, 14]) > 1.0
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={"shortest_edge": 28}, crop_size={"height": 28, "width": 28})
    sq = Image.new("RGB", (50, 50), tuple(int(x * 255) for x in DT.CLIP_MEAN))
    sq.paste(img, (0, 10))
    want = torch.from_numpy(proc.preprocess(sq, return_tensors="np")["pixel_values"][0])
    assert float((px - want).abs().max()) < 2e-2


def test_truncate_and_generate_helpers(golden_dir):
    from opadpo_amd.generate import truncate_after_eos_with_padding
    g = load(golden_dir, "ref_truncate.npz")
    comp = t(g["completions"])
    np.testing.assert_array_equal(truncate_after_eos_with_padding(comp, 2, 0).numpy(), g["plain"])
    np.testing.assert_array_equal(truncate_after_eos_with_padding(comp, 2, 0, [1577, 29973]).numpy(), g["with_stops"])


def test_cli_surface_and_quirks(tmp_path, monkeypatch):
    from opadpo_amd import cli
    p = cli.make_parser()
    ns = p.parse_args([])
    # Quirk Q8: store_false flags default to True
    for f in ("bf16", "tf32", "use_flash_attention", "resume_from_training", "do_train", "clean_tokens_after_eos"):
        assert getattr(ns, f) is True
    assert p.parse_args(["--bf16"]).bf16 is False
    # the shipped script's batch arithmetic at 4 GPUs (opadpo_train.py:383-433)
    ns = p.parse_args("--rollout_batch_size 64 --step_batch_size 32 --rollout_per_device_batch_size 2 "
                      "--step_per_device_batch_size 2 --CoPO False --AncPO True".split())
    a = cli.build_args(ns, world_size=4)
    assert (a.rollout_accumulation_steps, a.gradient_accumulation_steps) == (8, 4)
    assert a.CoPO is False and a.AncPO is False          # Quirk Q5: AncPO follows --CoPO
    with pytest.raises(ValueError):
        cli.build_args(p.parse_args(["--rollout_batch_size", "30"]), world_size=4)
    # PPO-era flags are accepted and ignored (Quirk Q9)
    p.parse_args("--kl_coef 0.2 --cliprange 0.1 --gamma 0.9 --lam 0.9 --mm_vision_select_layer -1 --value_head_mode linear".split())
    # YAML fills only what the command line left alone
    y = tmp_path / "c.yaml"
    y.write_text("training:\n  response_len: 896\n  beta: 0.3\n")
    argv = ["--cfg", str(y), "--beta", "0.1"]
    ns = p.parse_args(argv)
    cli.load_yaml_defaults(ns, argv)
    assert ns.response_len == 896 and ns.beta == 0.1


def test_checkpoint_io_roundtrip(tmp_path):
    from opadpo_amd import checkpoint_io as CK
    from opadpo_amd.synth import init_lora, init_weights
    d = DM.LlavaDims.tiny()
    W = init_weights(d, seed=0)
    llm = {k: v for k, v in W.items() if "vision_tower" not in k}
    vis = {"vision_model." + k[len(DM.VIS_PREFIX):]: v for k, v in W.items() if k.startswith(DM.VIS_PREFIX)}
    base, clip = tmp_path / "llava", tmp_path / "clip"
    base.mkdir(); clip.mkdir()
    keys = sorted(llm)
    half = len(keys) // 2
    torch.save({k: llm[k] for k in keys[:half]}, base / "pytorch_model-00001-of-00002.bin")
    torch.save({k: llm[k] for k in keys[half:]}, base / "pytorch_model-00002-of-00002.bin")
    wm = {k: ("pytorch_model-00001-of-00002.bin" if i < half else "pytorch_model-00002-of-00002.bin") for i, k in enumerate(keys)}
    (base / "pytorch_model.bin.index.json").write_text(json.dumps({"weight_map": wm}))
    (base / "config.json").write_text(json.dumps({"hidden_size": d.hidden, "num_hidden_layers": d.n_layers, "num_attention_heads": d.n_heads,
                                                   "intermediate_size": d.ffn, "vocab_size": d.vocab, "image_checkpoint": str(clip)}))
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in vis.items()}, str(clip / "model.safetensors"))
    state = CK.load_llava_state(str(base))
    assert set(state) == set(W) and all(torch.equal(state[k], W[k]) for k in W)
    dd = CK.dims_from_config(str(base), lora_r=d.lora_r, lora_alpha=d.lora_alpha)
    assert (dd.hidden, dd.n_layers, dd.n_heads, dd.head_dim, dd.ffn, dd.vocab) == (d.hidden, d.n_layers, d.n_heads, d.head_dim, d.ffn, d.vocab)
    ad = init_lora(d, seed=1)
    torch.save(ad, tmp_path / "adapter_model.bin")
    back = CK.load_adapter(str(tmp_path))
    assert set(back) == set(ad)


def test_sft_batches_from_dpo_batch():
    from opadpo_amd.sft import sft_batches_from_dpo_batch
    b = dict(images=torch.zeros(2, 3, 4, 4), queries=torch.ones(2, 5, dtype=torch.long), queries_attention_mask=torch.ones(2, 5, dtype=torch.bool),
             standard_response=torch.full((2, 3), 7), AI_pseudo_response=torch.full((2, 3), 9), original_generate_response=torch.full((2, 3), 5))
    std, ai = sft_batches_from_dpo_batch(b)
    assert set(std) == {"images", "queries", "queries_attn_masks", "responses"}
    assert int(std["responses"][0, 0]) == 7 and int(ai["responses"][0, 0]) == 9 and std["queries"] is b["queries"]
