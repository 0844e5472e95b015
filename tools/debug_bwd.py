"""Diagnostics: tiny backward with per-intermediate non-finite report (run on the GPU box)."""
import os
import sys

os.environ["OPADPO_DEBUG_NAN"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "opa-dpo_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from opadpo_amd.dims import LlavaDims  # noqa: E402
from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter  # noqa: E402
from opadpo_amd.policy import AutoregressivePolicy  # noqa: E402
from oracle import llava_ref as LR  # noqa: E402
from test_parity_gpu import make_inputs  # noqa: E402

BF = torch.bfloat16
d = LlavaDims.tiny()
od = LR.LlavaDims.tiny(lora_r=d.lora_r, lora_alpha=d.lora_alpha)
W = {k: v.to(BF).float() for k, v in LR.init_weights(od, seed=0, std=0.05).items()}
lp = {k: v.to(BF).float() for k, v in LR.init_lora(od, seed=1, b_std=0.02).items()}
dev = torch.device("cuda:0")
eng = LlavaEngine(BaseWeights(d, W, dev, need_backward=True))
ad = LoraAdapter(d, lp, dev, True)
print("work_t nonfinite:", int((~torch.isfinite(ad.work_t.float())).sum()))
B, Q, T = 2, 12, 9
images, queries, qmask, resp = make_inputs(d, B, Q, T, seed=5)
two = {k: resp[k] for k in ("standard_response", "original_generate_response")}
pol = AutoregressivePolicy(eng, ad, T)
out = pol(images=images.to(dev), queries=queries, queries_attn_masks=qmask, **two)
loss = sum(out[k + "_logprobs"].sum() for k in two)
loss.backward()
torch.cuda.synchronize()
print("grad nonfinite:", int((~torch.isfinite(ad.grad)).sum()), "of", ad.grad.numel())
