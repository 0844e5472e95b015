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
    assert set(sd2) == set(sdf) and len(sd2) == 2 * 7 * 2          # layers x linears x (A,B)
    moved = sum(float((sdf[k].float() - sd2[k].float()).abs().sum()) for k in sd2)
    assert moved > 0, "training after checkpoint-2 did not change the adapter"
    assert get_last_checkpoint(out) == (None, True)                 # 'completed' marker -> nothing to resume
    os.remove(os.path.join(out, "completed"))
    path, done = get_last_checkpoint(out)
    assert path.endswith("checkpoint-2") and not done
    cli.main(argv)                                                    # resumes from checkpoint-2
    assert os.path.exists(os.path.join(out, "completed"))


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
