#!/usr/bin/env python3
"""Rollout-generation stage launcher, same path and flags as the reference's opadpo/online_generation_custom.py
(run/online_generate.sh); the work happens in opadpo_amd.cli_generate.main_rollout."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "opa-dpo_amd"))
from opadpo_amd.cli_generate import main_rollout  # noqa: E402

if __name__ == "__main__":
    main_rollout()
