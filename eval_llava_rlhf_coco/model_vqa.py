#!/usr/bin/env python3
"""Evaluation generation launcher, same path and flags as the reference's eval_llava_rlhf_coco/model_vqa.py
(run/eval_all_metrics.sh); the work happens in opadpo_amd.cli_generate.main_eval."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "opa-dpo_amd"))
from opadpo_amd.cli_generate import main_eval  # noqa: E402

if __name__ == "__main__":
    main_eval()
