"""Full-size sanity run (diagnostic): LLaVA-1.5-7B LoRA DPO on ONE synthetic micro-batch for a few optimizer steps at a learning rate
large enough to see the loss move; prints loss / pre-clip gradient norm per step.  SOAK_STEPS, SOAK_PAIRS, SOAK_LR; SOAK_POOL > 1
cycles over that many different synthetic batches (different ragged row counts from step to step)."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd.dims import LlavaDims
from opadpo_amd.losses import DPOArgs, pair_loss
from opadpo_amd.ctx import CtxEngine
from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter
from opadpo_amd.optim import FlatAdamW
from opadpo_amd.policy import AutoregressivePolicy
from opadpo_amd.synth import init_lora, init_weights, synth_pairs


def main():
    dev = torch.device("cuda:0")
    steps, pairs, lr = int(os.environ.get("SOAK_STEPS", 10)), int(os.environ.get("SOAK_PAIRS", 22)), float(os.environ.get("SOAK_LR", 2e-5))
    d = LlavaDims.llava15_7b()
    base = BaseWeights(d, init_weights(d, seed=0, device=dev), dev, need_backward=True)
    eng = LlavaEngine(base) if os.environ.get("OPADPO_OP_LEVEL") == "1" else CtxEngine(base)
    pol = LoraAdapter(d, init_lora(d, seed=1, device=dev), dev, trainable=True)
    ref = LoraAdapter(d, init_lora(d, seed=1, device=dev), dev, trainable=False)          # same start as the policy: loss = log 2 at step 0
    ref.merge_into_base(base)
    policy, ref_policy = AutoregressivePolicy(eng, pol, 384), AutoregressivePolicy(eng, ref, 384)
    opt = FlatAdamW(pol.master, pol.grad, pol.work, lr=lr, max_grad_norm=1.0, mode="allreduce")
    pool = [synth_pairs(d, pairs, 128, 384, seed=7 + i, device=dev) for i in range(int(os.environ.get("SOAK_POOL", 1)))]
    largs = DPOArgs()
    for it in range(steps):
        t0 = time.time()
        b = pool[it % len(pool)]
        feats = eng.encode_images(b["images"])
        kw = dict(queries=b["queries"], queries_attn_masks=b["queries_attn_masks"], image_feats=feats,
                  chosen_response=b["chosen"], rejected_response=b["rejected"])
        with torch.no_grad():
            r = ref_policy(**kw)
        o = policy(**kw)
        loss, _, _ = pair_loss(largs, o["chosen_response_logprobs"], o["rejected_response_logprobs"],
                               r["chosen_response_logprobs"], r["rejected_response_logprobs"])
        loss.backward()
        assert bool(torch.isfinite(pol.grad).all()), "non-finite LoRA gradient"
        opt.step()
        gn = opt.grad_norm_post_clip()
        opt.zero_grad()
        pol.refresh_transposed()
        torch.cuda.synchronize()
        print(f"step {it}: loss {float(loss.detach()):.6f}  post-clip grad_norm {gn:.4f}  {time.time() - t0:.2f} s", flush=True)
        assert torch.isfinite(loss)


if __name__ == "__main__":
    main()
