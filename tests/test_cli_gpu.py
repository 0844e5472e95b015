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
