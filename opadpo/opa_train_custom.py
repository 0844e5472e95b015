#!/usr/bin/env python3
"""Entry point with the reference's name (opadpo/opa_train_custom.py, launched by run/train_opa.sh): OPA LoRA-SFT stage on
the MI355X-native kernels.  See opa-dpo_amd/opadpo_amd/cli_sft.py for the flag surface."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd"))
from opadpo_amd.cli_sft import main  # noqa: E402

if __name__ == "__main__":
    main()
