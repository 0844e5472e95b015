#!/usr/bin/env python3
"""Drop-in entry point: same path and flags as the reference's `opadpo/opadpo_train_custom.py`
(launched by run/train_opa_dpo.sh through torchrun); the work happens in opa-dpo_amd/opadpo_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd"))
from opadpo_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
