"""CPU oracle for the OPA-DPO hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the reported CPU
baseline.  The product path (``opa-dpo_amd/``) never imports it and fails
loudly when the HIP library is missing.

What it restates (plain torch fp32 on CPU, no HuggingFace / peft imports):

* ``dpo_ref``   – the arithmetic the reference OWNS: ``compute_logprobs``
  (utils/common_utils.py:112-118), the entropy formula
  (opadpo/dpo_models/rl_models.py:128,132), ``DPOTrainer.dpo_loss``
  (opadpo/dpo_models/dpo_trainer.py:429-473), ``compute_policy_loss``
  (dpo_trainer.py:475-802), ``mask_single_image`` (dpo_trainer.py:83-109),
  ``truncate_after_eos_with_padding`` (generator_models/generator.py:244-273),
  ``AutoregressivePolicy.forward`` slicing (rl_models.py:75-144).
  Pinned: golden vectors in ``tests/golden/ref_*.npz`` were produced by
  importing the reference's own Python in the build container
  (``tests/golden/make_golden.py``).
* ``llava_ref`` – the third-party model arithmetic reached from
  rl_models.py:114-120 (LLaVA-1.5 = CLIP-ViT-L/14-336 -> mlp2x_gelu -> splice ->
  Llama-2 + PEFT-LoRA).  That code is NOT under /root/reference
  (haotian-liu/LLaVA@817a4af + a missing patch, transformers==4.34.1,
  peft==0.5.0, flash-attn==2.5.3) so it is restated from the published
  algorithm and pinned against the *installed* transformers 5.15
  ``LlamaForCausalLM`` / ``CLIPVisionModel`` with seeded weights
  (``tests/golden/hf_*.npz``).  The multimodal splice is defined by this build
  (SURVEY.md §8c last row): parity for it is "unpinned" against the reference.
* ``optim_ref`` – AdamW / global-norm clip / cosine-with-warmup schedule
  (utils/trainer_utils.py:9-49, HF defaults).
"""
