#!/usr/bin/env python3
"""Same path as the reference's base_operations/make_online_generation_dataset.py: RLAIF-V parquet shards -> the four stratified
2500-row subsets the rollout stage reads.  The work happens in opadpo_amd.dataset_build.make_online_generation_subsets."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "opa-dpo_amd"))
from opadpo_amd.dataset_build import make_online_generation_subsets  # noqa: E402

if __name__ == "__main__":
    root = "./base_datasets/LLaVA-RLAIF-Data"
    make_online_generation_subsets([f"{root}/RLAIF-V-Dataset_{i:03d}.parquet" for i in range(14)])
