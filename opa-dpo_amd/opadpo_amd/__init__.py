"""opadpo_amd — MI355X (gfx950) native OPA-DPO hot path: HIP kernels behind a C ABI
(libopadpo_hip.so, include/opadpo_hip.h) sequenced by a thin Python host on PyTorch-ROCm.

There is no CPU fallback: importing the kernels' users without the built library raises.
"""
from .dims import LlavaDims, IMAGE_TOKEN_INDEX, PAD_ID, EOS_ID, lora_param_count, pair_flops  # noqa: F401

__all__ = ["LlavaDims", "IMAGE_TOKEN_INDEX", "PAD_ID", "EOS_ID", "lora_param_count", "pair_flops"]
