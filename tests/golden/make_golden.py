#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

* ref_*.npz  — outputs of the reference's own Python (``/root/reference``), imported with
  ``sys.modules`` stubs for the third-party packages that are absent here (llava, peft,
  loguru, bitsandbytes; recipe: SURVEY.md §8c).  Only inputs/outputs are stored — never
  reference source.
* hf_*.npz   — outputs of the *installed* transformers Llama / CLIP with seeded weights:
  the pin for oracle/llava_ref.py (the model arithmetic is third-party, not under
  /root/reference).

Nothing on the GPU box or in the product path depends on this script.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    import transformers.trainer_utils  # noqa: F401  (must precede the peft stub)
    import transformers.trainer  # noqa: F401

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    _stub("llava")
    _stub("llava.conversation", conv_templates={}, SeparatorStyle=_Any)
    _stub("llava.train")
    _stub("llava.train.train", DataArguments=_Any)
    _stub("llava.mm_utils", tokenizer_image_token=lambda *a, **k: None)
    _stub("llava.model", LlavaLlamaForCausalLM=_Any)
    _stub("llava.model.utils", resize_token_embeddings_with_mean=None, set_reproducibility=None)
    _stub("llava.utils", get_max_num_dataloaders=None)
    _stub("llava.constants", IGNORE_INDEX=-100, IMAGE_TOKEN_INDEX=-200, DEFAULT_IMAGE_TOKEN="<image>",
          DEFAULT_IM_START_TOKEN="<im_start>", DEFAULT_IM_END_TOKEN="<im_end>")
    _stub("llava.model.language_model")
    _stub("llava.model.language_model.llava_llama", LlavaLlamaForCausalLM=_Any)

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    _stub("loguru", logger=_Logger())
    _stub("peft", PeftModel=_Any, LoraConfig=_Any, get_peft_model=_Any, PeftModelForCausalLM=_Any,
          prepare_model_for_kbit_training=_Any, LoraModel=_Any)
    _stub("peft.utils", WEIGHTS_NAME="adapter_model.bin", get_peft_model_state_dict=_Any, CONFIG_NAME="adapter_config.json")
    _stub("peft.tuners")
    _stub("peft.tuners.lora", LoraLayer=_Any)
    _stub("bitsandbytes")
    _stub("openai", AzureOpenAI=_Any, OpenAI=_Any)
    _stub("bitsandbytes.nn", Linear4bit=_Any)
    sys.path.insert(0, REF)
    import utils.common_utils as cu
    import opadpo.dpo_models.rl_models as rl_models
    import opadpo.dpo_models.dpo_trainer as dpo_trainer
    import opadpo.dpo_models.rl_trainer as generator  # same truncate_after_eos_with_padding body as generator_models/generator.py:244-273
    import utils.lora_utils as lora_utils
    return cu, rl_models, dpo_trainer, generator, lora_utils


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu()
            if v.dtype == torch.bool:
                v = v.to(torch.uint8)
            v = v.numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, len(out), "arrays")


def rollout_fixture(g, B, T, with_scores=True):
    """A synthetic rollout dict with the reference's keys (dpo_trainer.py:389-418)."""
    def resp():
        lens = torch.randint(2, T, (B,), generator=g)
        ids = torch.randint(3, 50, (B, T), generator=g)
        for b in range(B):
            ids[b, lens[b]] = 2
            ids[b, lens[b] + 1:] = 0
        return ids

    def lp(ids):
        x = -torch.rand(B, T, generator=g) * 3 - 0.01
        return x * (ids != 0)

    r = {}
    for k in ("standard_response", "original_generate_response", "AI_pseudo_response"):
        r[k] = resp()
        r["ref_base_" + k + "_logprobs"] = lp(r[k])
    r["ref_mask_standard_response_logprobs"] = lp(r["standard_response"])
    r["ref_mask_AI_pseudo_response_logprobs"] = lp(r["AI_pseudo_response"])
    choices = torch.tensor([1.0, 1.5, 2.0, 2.5])
    for k in ("original_generate_response", "AI_pseudo_response"):
        sc = choices[torch.randint(0, 4, (B, T), generator=g)] * (r[k] != 0)
        rel = torch.tensor([1.0, 3.0])[torch.randint(0, 2, (B, T), generator=g)] * (r[k] != 0)
        r[k + "_scores"] = sc
        r[k + "_image_relations"] = rel
    r["queries"] = torch.randint(3, 50, (B, 4), generator=g)
    r["queries_attn_masks"] = torch.ones(B, 4, dtype=torch.bool)
    r["images"] = torch.zeros(B, 3, 2, 2)
    r["masked_images"] = torch.zeros(B, 3, 2, 2)
    pol = {k + "_logprobs": lp(r[k]).requires_grad_(True)
           for k in ("standard_response", "original_generate_response", "AI_pseudo_response")}
    polm = {"mask_" + k + "_logprobs": lp(r[k]).requires_grad_(True)
            for k in ("standard_response", "AI_pseudo_response")}
    return r, pol, polm


def main():
    cu, rl_models, dpo_trainer, generator, lora_utils = import_reference()
    g = torch.Generator().manual_seed(1234)

    # ---- G1/G2: compute_logprobs + entropy -------------------------------------------------
    logits = torch.randn(2, 5, 11, generator=g) * 2
    labels = torch.randint(1, 11, (2, 5), generator=g)
    labels[0, 3:] = 0
    labels[1, 4:] = 0
    lp = cu.compute_logprobs(logits, labels, ignore_index=0)
    ent = -(logits.softmax(dim=-1) * logits.log_softmax(dim=-1)).sum(dim=-1)   # rl_models.py:128 formula
    save("ref_logprobs.npz", logits=logits, labels=labels, logprobs=lp, entropies=ent,
         signbit=torch.signbit(lp))

    # ---- G3: dpo_loss variants ---------------------------------------------------------------
    T = dpo_trainer.DPOTrainer
    arrs = {}
    B, Tn = 3, 7
    pc, pr, rc, rr = [-torch.rand(B, Tn, generator=g) * 4 for _ in range(4)]
    sc = torch.rand(B, Tn, generator=g) * 2 + 0.5
    sr = torch.rand(B, Tn, generator=g) * 2 + 0.5
    arrs.update(pc=pc, pr=pr, rc=rc, rr=rr, sc=sc, sr=sr)
    idx = 0
    meta = []
    for fdiv in ("reverse_kl", "js_divergence", "alpha_divergence"):
        for ls in (0.0, 0.1):
            for rf in (False, True):
                for scores in (False, True):
                    t = T.__new__(T)
                    t.reference_free, t.f_divergence_type, t.loss_type = rf, fdiv, "sigmoid"
                    t.beta, t.label_smoothing = 0.1, ls
                    t.f_divergence_params = None
                    t.accelerator = types.SimpleNamespace(device=torch.device("cpu"))
                    out = t.dpo_loss(pc, pr, rc, rr, sc if scores else None, sr if scores else None)
                    arrs[f"case{idx}_losses"], arrs[f"case{idx}_c"], arrs[f"case{idx}_r"] = out
                    meta.append(f"{fdiv}|{ls}|{int(rf)}|{int(scores)}")
                    idx += 1
    arrs["meta"] = np.array(meta)
    save("ref_dpo_loss.npz", **arrs)

    # ---- G4: compute_policy_loss end to end with a fake policy --------------------------------
    class FakePolicy(torch.nn.Module):
        def __init__(self, clean, masked):
            super().__init__()
            self.clean, self.masked = clean, masked

        def forward(self, **kw):
            return self.masked if any(k.startswith("mask_") for k in kw) else self.clean

    arrs = {}
    metas = []
    ci = 0
    for CoPO in (False, True):
        for AncPO, mdpo in ((False, True), (True, True), (True, False)):
            for detailed in (False, True):
                r, pol, polm = rollout_fixture(g, B=2, T=9)
                t = T.__new__(T)
                t.reference_free, t.f_divergence_type, t.loss_type = False, "reverse_kl", "sigmoid"
                t.beta, t.label_smoothing, t.f_divergence_params = 0.1, 0.0, None
                t.accelerator = types.SimpleNamespace(device=torch.device("cpu"), num_processes=1)
                t.tokenizer = types.SimpleNamespace(pad_token_id=0)
                t.args = types.SimpleNamespace(
                    detailed_report=detailed, response_score=True, response_image_relation=True,
                    CoPO=CoPO, CoPO_method="random", CoPO_coef=0.2, AncPO=AncPO, mDPO_anchor=mdpo,
                    Anchor_value=0.0, Anchor_coef=1.0, standard_pair_coef=1.0, AI_pair_coef=1.0, temperature=1.0)
                t.policy = FakePolicy(pol, polm)
                loss, stats = t.compute_policy_loss(r)
                loss.backward()
                pre = f"c{ci}_"
                for k, v in r.items():
                    if "logprobs" in k or "scores" in k or "relations" in k:
                        arrs[pre + "in_" + k] = v
                for k, v in {**pol, **polm}.items():
                    arrs[pre + "pol_" + k] = v
                    arrs[pre + "grad_" + k] = v.grad if v.grad is not None else torch.zeros_like(v)
                arrs[pre + "loss"] = loss
                for k, v in stats.items():
                    arrs[pre + "stat_" + k.replace("/", "__")] = v
                metas.append(f"{int(CoPO)}|{int(AncPO)}|{int(mdpo)}|{int(detailed)}")
                ci += 1
    arrs["meta"] = np.array(metas)
    save("ref_policy_loss.npz", **arrs)

    # ---- G5: AutoregressivePolicy.forward over a tiny HF Llama (slicing / shift / masks) -------
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(7)
    cfg = LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=48, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=1e-5,
                      max_position_embeddings=64, attn_implementation="eager", tie_word_embeddings=False)
    lm = LlamaForCausalLM(cfg).eval()

    class Shim(torch.nn.Module):
        def __init__(self, lm):
            super().__init__()
            self.lm = lm
            self.config = lm.config

        def set_adapter(self, name):
            pass

        def prepare_inputs_for_generation(self, input_ids=None, attention_mask=None, images=None, use_cache=None):
            return dict(input_ids=input_ids, attention_mask=attention_mask)

        def forward(self, input_ids=None, attention_mask=None, output_hidden_states=False):
            return self.lm(input_ids=input_ids, attention_mask=attention_mask.long(),
                           output_hidden_states=output_hidden_states, use_cache=False)

    Q, Tn, B = 6, 5, 2
    queries = torch.randint(3, 64, (B, Q), generator=g)
    qmask = torch.ones(B, Q, dtype=torch.bool)
    queries[0, :2] = 0
    qmask[0, :2] = False
    resp = {}
    for k in ("standard_response", "original_generate_response"):
        ids = torch.randint(3, 64, (B, Tn), generator=g)
        ids[0, 3] = 2
        ids[0, 4:] = 0
        resp[k] = ids
    pol = rl_models.AutoregressivePolicy(
        types.SimpleNamespace(temperature=0.7, response_len=Tn), Shim(lm),
        types.SimpleNamespace(pad_token_id=0), adapter_name="lora_policy")
    with torch.no_grad():
        out = pol(images=torch.zeros(B, 1), queries=queries, queries_attn_masks=qmask, temperature=0.7,
                  standard_response_attention_mask=None, **resp)
        ids_all = torch.cat([torch.cat([queries, resp[k]], 1) for k in resp], 0)
        am = ids_all != 0
        am[:, :Q] = torch.cat([qmask, qmask], 0)
        full_logits = lm(input_ids=ids_all, attention_mask=am.long()).logits
    sd = {k: v for k, v in lm.state_dict().items()}
    save("ref_policy_forward.npz", queries=queries, qmask=qmask, full_logits=full_logits,
         **{"resp_" + k: v for k, v in resp.items()}, **{"out_" + k: v for k, v in out.items()},
         **{"w_" + k: v for k, v in sd.items()})

    # ---- G6: mask_single_image / mask_percentage_per_row under torch.manual_seed ---------------
    img = torch.randn(1, 3, 28, 28, generator=g)
    torch.manual_seed(99)
    m_rand = dpo_trainer.mask_single_image(img, 0.3, "random")
    torch.manual_seed(99)
    m_blk = dpo_trainer.mask_single_image(img, 0.3, "blockwise")
    torch.manual_seed(99)
    rowmask = dpo_trainer.mask_percentage_per_row(torch.ones(3, 16, dtype=torch.bool), 0.3)
    save("ref_mask_image.npz", img=img, random=m_rand, blockwise=m_blk, rowmask=rowmask)

    # ---- G8: truncate_after_eos_with_padding ------------------------------------------------
    comp = torch.tensor([[5, 6, 2, 7, 8, 9], [5, 1577, 6, 2, 7, 8], [5, 6, 7, 8, 9, 10],
                         [5, 2, 6, 29973, 7, 8], [1577, 29973, 2, 4, 4, 4]])
    t1 = generator.truncate_after_eos_with_padding(comp, 2, 0)
    t2 = generator.truncate_after_eos_with_padding(comp, 2, 0, additional_tokens=[1577, 29973])
    save("ref_truncate.npz", completions=comp, plain=t1, with_stops=t2)

    # ---- HF pins for oracle/llava_ref.py ------------------------------------------------------
    from oracle import llava_ref as LR
    d = LR.LlavaDims.tiny(hidden=64, n_layers=2, n_heads=2, head_dim=32, ffn=96, vocab=80,
                          v_hidden=32, v_layers=3, v_heads=2, v_ffn=64, image_size=28, lora_r=8, lora_alpha=16.0)
    W = LR.init_weights(d, seed=3, std=0.2)
    cfg = LlamaConfig(vocab_size=d.vocab, hidden_size=d.hidden, intermediate_size=d.ffn,
                      num_hidden_layers=d.n_layers, num_attention_heads=d.n_heads,
                      num_key_value_heads=d.n_heads, head_dim=d.head_dim, rms_norm_eps=d.rms_eps,
                      rope_theta=d.rope_theta, max_position_embeddings=128,
                      attn_implementation="eager", tie_word_embeddings=False)
    lm = LlamaForCausalLM(cfg).eval()
    missing = lm.load_state_dict({k: v for k, v in W.items() if k.startswith("model.layers") or k in
                                  ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")}, strict=False)
    assert not [m for m in missing.missing_keys if "rotary" not in m], missing
    S, Ltxt = 3, 12
    ids = torch.randint(3, d.vocab, (S, Ltxt), generator=g)
    mask = torch.ones(S, Ltxt, dtype=torch.bool)
    ids[0, :3] = 0
    mask[0, :3] = False
    ids[1, -2:] = 0
    mask[1, -2:] = False
    with torch.no_grad():
        hf_logits = lm(input_ids=ids, attention_mask=mask.long()).logits
    save("hf_llama.npz", ids=ids, mask=mask, logits=hf_logits, seed=3, std=0.2)

    from transformers import CLIPVisionConfig, CLIPVisionModel
    vcfg = CLIPVisionConfig(hidden_size=d.v_hidden, intermediate_size=d.v_ffn, num_hidden_layers=d.v_layers,
                            num_attention_heads=d.v_heads, image_size=d.image_size, patch_size=d.patch,
                            hidden_act="quick_gelu", layer_norm_eps=d.v_eps, attn_implementation="eager")
    vm = CLIPVisionModel(vcfg).eval()
    vsd = {k[len(LR.VIS_PREFIX):]: v for k, v in W.items() if k.startswith(LR.VIS_PREFIX)}  # transformers>=5: no "vision_model." prefix
    missing = vm.load_state_dict(vsd, strict=False)
    assert not [m for m in missing.missing_keys if "post_layernorm" not in m and "position_ids" not in m], missing
    pix = torch.randn(2, 3, d.image_size, d.image_size, generator=g)
    with torch.no_grad():
        hs = vm(pixel_values=pix, output_hidden_states=True).hidden_states
    save("hf_clip.npz", pixels=pix, feats=hs[-2][:, 1:], seed=3, std=0.2)

    # ---- G9: get_last_checkpoint -----------------------------------------------------------
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        a = lora_utils.get_last_checkpoint(os.path.join(td, "nope"))
        os.makedirs(os.path.join(td, "run", "checkpoint-75"))
        os.makedirs(os.path.join(td, "run", "checkpoint-150"))
        b = lora_utils.get_last_checkpoint(os.path.join(td, "run"))
        open(os.path.join(td, "run", "completed"), "w").close()
        c = lora_utils.get_last_checkpoint(os.path.join(td, "run"))
        save("ref_last_checkpoint.npz", first=np.array([str(a[0]), str(a[1])]),
             found=np.array([os.path.basename(b[0]), str(b[1])]), done=np.array([str(c[0]), str(c[1])]))


if __name__ == "__main__":
    main()


# ---- G7: collator (utils/data_utils_dpo.py) with a toy HF-style tokenizer ---------------------------------------
def make_collator_golden():
    import json as _json
    sys.path.insert(0, os.path.join(REPO, "tests"))
    sys.path.insert(0, REF)
    from toy_tokenizer import ToyTokenizer, collator_instances
    import utils.data_utils_dpo as ref_dd
    out = {}
    for detailed in (False, True):
        coll = ref_dd.DataCollatorForCausalLM(tokenizer=ToyTokenizer(), query_len=24, response_len=40, detailed_report=detailed)
        batch = coll(collator_instances())
        for k, v in batch.items():
            out[f"d{int(detailed)}_{k}"] = v
    # helper functions on their own
    out["h_complete"] = np.array(ref_dd.complete_copied_content("a b c. d e f. g", ["a b", "d e f.", ""]))
    out["h_complete_fail"] = np.array(ref_dd.complete_copied_content("a b c", ["zzz", "a"]))
    save("ref_collator.npz", **out)


if __name__ == "__main__":
    make_collator_golden()
