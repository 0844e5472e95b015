#!/usr/bin/env python3
"""Rollout JSON -> OPA / OPA-DPO HF datasets: same command as the reference (`python base_operations/make_opadpo_dataset.py`,
run from the repository root); the work is opadpo_amd.dataset_build (paths via --json_dir / --opa_out / --opadpo_out, defaults
= the reference's LLaVA-1.5-7B locations)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd"))
from opadpo_amd.dataset_build import main  # noqa: E402

if __name__ == "__main__":
    main()
