"""GPU: the drop-in entry point end to end on a synthetic tiny model — rollout, CoPO/AncPO loss, HIP backward,
clip + AdamW, checkpoint cadence (save BEFORE stepping at multiples of save_steps), PEFT layout, resume."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_entry_point_checkpoints_and_resume(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd import cli
    from opadpo_amd.trainer import get_last_checkpoint
    out = str(tmp_path / "run")
    argv = ["--synthetic", "tiny", "--output_dir", out, "--rollout_batch_size", "4", "--step_batch_size", "4",
            "--rollout_per_device_batch_size", "2", "--step_per_device_batch_size", "2", "--query_len", "16",
            "--response_len", "16", "--max_step", "3", "--save_steps", "2", "--total_epochs", "2", "--noptepochs", "1",
            "--learning_rate", "1e-4", "--report_to", "none", "--cfg", "none"]
    cli.main(argv)
    ck2 = os.path.join(out, "checkpoint-2", "adapter_model", "lora_policy")
    assert os.path.exists(os.path.join(ck2, "adapter_model.bin")) and os.path.exists(os.path.join(ck2, "adapter_config.json"))
    assert os.path.exists(os.path.join(out, "checkpoint-2", "scheduler.pt"))
    fin = os.path.join(out, "checkpoint-final", "adapter_model", "lora_policy")
    cfg = json.load(open(os.path.join(fin, "adapter_config.json")))
    assert cfg["r"] == 128 and cfg["inference_mode"] is True
    sd2 = torch.load(os.path.join(ck2, "adapter_model.bin"))
    sdf = torch.load(os.path.join(fin, "adapter_model.bin"))
    # every checkpoint carries the LLM LoRA (layers x linears x (A,B)) AND the frozen CLIP-tower / projector LoRA of the adapter the
    # policy started from, like get_peft_model_state_dict(adapter_name='lora_policy') (dpo_trainer.py:1047-1095)
    llm = [k for k in sd2 if "vision_tower" not in k and "mm_projector" not in k]
    vis = [k for k in sd2 if "vision_tower" in k]
    assert set(sd2) == set(sdf) and len(llm) == 2 * 7 * 2 and len(vis) == 3 * 6 * 2 and sum("mm_projector" in k for k in sd2) == 4
    assert {"fc1", "fc2", "out_proj", "q_proj", "down_proj"} <= set(cfg["target_modules"])
    assert all(torch.equal(sd2[k], sdf[k]) for k in sd2 if k not in llm), "frozen vision / projector LoRA must be written back unchanged"
    moved = sum(float((sdf[k].float() - sd2[k].float()).abs().sum()) for k in llm)
    assert moved > 0, "training after checkpoint-2 did not change the adapter"
    assert get_last_checkpoint(out) == (None, True)                 # 'completed' marker -> nothing to resume
    os.remove(os.path.join(out, "completed"))
    path, done = get_last_checkpoint(out)
    assert path.endswith("checkpoint-2") and not done
    # only the newest optimizer.pt is kept (dpo_trainer.py:885-896): it moved on to checkpoint-final
    assert not os.path.exists(os.path.join(out, "checkpoint-2", "optimizer.pt"))
    optf = torch.load(os.path.join(out, "checkpoint-final", "optimizer.pt"))["optimizer"]
    assert optf["format"] == 2 and optf["master"].dtype == torch.float32 and optf["master"].numel() == optf["m"].numel() > 0
    cli.main(argv)                                                    # resumes from checkpoint-2
    assert os.path.exists(os.path.join(out, "completed"))
    # resume continues the trajectory: fp32 master + Adam moments + step restored.  Deterministic variant (no CoPO image masks,
    # which draw from the global RNG), lr large enough that an un-restored optimizer would show.  Run A: uninterrupted, steps 1-2.
    # Run B: stopped after step 1 (its final checkpoint = what checkpoint-2 holds: saved BEFORE step 2), then resumed for step 2.
    def variant(o, max_step):
        a = [x for x in argv]
        a[a.index("--output_dir") + 1] = o
        a[a.index("--learning_rate") + 1] = "1e-2"
        a[a.index("--max_step") + 1] = str(max_step)
        return a + ["--CoPO", "False"]
    out_a, out_b = str(tmp_path / "run_a"), str(tmp_path / "run_b")
    cli.main(variant(out_a, 3))
    cli.main(variant(out_b, 2))
    os.rename(os.path.join(out_b, "checkpoint-final"), os.path.join(out_b, "checkpoint-2"))
    os.remove(os.path.join(out_b, "completed"))
    assert os.path.exists(os.path.join(out_b, "checkpoint-2", "optimizer.pt"))
    cli.main(variant(out_b, 3))
    name = os.path.join("checkpoint-final", "adapter_model", "lora_policy", "adapter_model.bin")
    sda, sdb = torch.load(os.path.join(out_a, name)), torch.load(os.path.join(out_b, name))
    start = torch.load(os.path.join(out_b, "checkpoint-2", "adapter_model", "lora_policy", "adapter_model.bin"))
    worst = max(float((sda[k].float() - sdb[k].float()).abs().max()) for k in llm)
    step = max(float((sda[k].float() - start[k].float()).abs().max()) for k in llm)
    assert step > 1e-3 and worst < 0.1 * step, f"resumed run differs from the uninterrupted one by {worst} (the resumed step moved {step})"


def test_sft_entry_point(tmp_path):
    """opadpo/opa_train_custom.py flag surface (run/train_opa.sh) on a tiny synthetic model: accumulation, cosine schedule,
    periodic + final PEFT checkpoints holding LLM, CLIP and projector LoRA tensors."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd import cli_sft
    out = str(tmp_path / "opa")
    argv = ["--synthetic", "tiny", "--synthetic_samples", "16", "--output_dir", out, "--per_device_train_batch_size", "2",
            "--gradient_accumulation_steps", "2", "--num_train_epochs", "2", "--save_steps", "3", "--learning_rate", "1e-3",
            "--full_tune", "False", "--lora_tune", "True", "--tune_vision_tower", "True", "--entropy_loss", "True",
            "--entropy_mask_ratio", "0.5", "--entropy_loss_coef", "0.01", "--cfg", "none", "--bf16", "--tf32", "--deepspeed", "x.json"]
    cli_sft.main(argv)
    fin = os.path.join(out, "checkpoint-final")
    sd = torch.load(os.path.join(fin, "adapter_model.bin"))
    cfg = json.load(open(os.path.join(fin, "adapter_config.json")))
    assert cfg["r"] == 128 and "fc1" in cfg["target_modules"]
    n_vis = sum(1 for k in sd if "vision_tower" in k)
    assert n_vis == 2 * 6 * 2 and sum(1 for k in sd if "mm_projector" in k) == 4 and len(sd) == n_vis + 4 + 2 * 7 * 2
    ck3 = torch.load(os.path.join(out, "checkpoint-3", "adapter_model.bin"))
    assert set(ck3) == set(sd) and sum(float((ck3[k].float() - sd[k].float()).abs().sum()) for k in sd) > 0
    with pytest.raises(SystemExit):
        cli_sft.main(argv + ["--full_tune", "True"])


def test_rollout_entry_point_writes_step_files(tmp_path, monkeypatch):
    """opadpo/online_generation_custom.py surface on a synthetic tiny model: rows -> left-padded queries -> sampled responses ->
    step{N}_rank{R}.json, readable by the dataset builder (all records dropped by its first filter: no feedback model here)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd import cli_generate as cg
    from opadpo_amd.dataset_build import build_rows, load_rollout_records
    out = str(tmp_path / "gen")
    ns, _ = cg.rollout_parser().parse_known_args(
        ["--synthetic", "tiny", "--synthetic_rows", "7", "--output_dir", out, "--rollout_batch_size", "4", "--rollout_per_device_batch_size", "2",
         "--query_len", "64", "--response_len", "6", "--top_k", "10", "--base_model", "ignored", "--phase", "0", "--local-rank", "0"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    files = cg.run_rollout(ns, log=lambda *_: None)
    assert [os.path.basename(f) for f in files] == ["step0_rank0.json", "step1_rank0.json"]
    recs = load_rollout_records([os.path.join(out, "rollouts")], log=lambda *_: None)
    assert len(recs) == 7 and set(recs[0]) == {"query", "image_id", "standard_response", "original_generate_response", "AI_generate_response",
                                               "AI_pseudo_response", "AI_json_report", "image_bytes"}
    assert [r["image_id"] for r in recs] == [f"synthetic_{i}.png" for i in range(7)]
    assert all(r["query"].startswith("<image>\n") for r in recs)
    assert build_rows([os.path.join(out, "rollouts")], log=lambda *_: None) == []
    # a feedback callable fills the report fields
    fb = lambda urls, q, rsp, std: {"Pseudo_response": [s + " fixed" for s in std], "Generated_response": list(rsp),
                                    "report_json": [{"Sentence 1": {"score": 4}}] * len(rsp)}
    ns.output_dir = str(tmp_path / "gen_fb")
    ns.max_step = 1
    files = cg.run_rollout(ns, feedback=fb, log=lambda *_: None)
    assert len(files) == 1
    rows = build_rows([os.path.join(ns.output_dir, "rollouts")], log=lambda *_: None)
    assert 0 < len(rows) <= 4 and rows[0]["AI_pseudo_response"].endswith(" fixed")


def test_eval_entry_point(tmp_path):
    """eval_llava_rlhf_coco/model_vqa.py surface: question file -> answers file; an existing answers file is refused."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from PIL import Image
    from opadpo_amd import cli_generate as cg
    Image.new("RGB", (12, 7), (10, 120, 200)).save(tmp_path / "x.png")
    qf = tmp_path / "q.jsonl"
    qf.write_text("".join(json.dumps({"question_id": i, "image": "x.png", "text": f"what is item {i} ?"}) + "\n" for i in range(5)))
    ans = tmp_path / "out" / "a.jsonl"
    argv = ["--synthetic", "tiny", "--model-path", "tiny-model", "--use-qlora", "True", "--qlora-path", "synthetic", "--question-file", str(qf),
            "--image-folder", str(tmp_path), "--answers-file", str(ans), "--short_eval", "True", "--batch-size", "2", "--num-chunks", "2",
            "--chunk-idx", "0", "--test-prompt", ""]
    cg.main_eval(argv)
    lines = [json.loads(x) for x in open(ans)]
    assert [x["question_id"] for x in lines] == [0, 1, 2] and lines[0]["model_id"] == "tiny-model" and lines[0]["prompt"] == "what is item 0 ?"
    with pytest.raises(SystemExit):
        cg.main_eval(argv)
