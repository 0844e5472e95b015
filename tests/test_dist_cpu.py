"""CPU, world_size 2, gloo: the data-parallel step of FlatAdamW (bucketed gradient exchange launched per finished bucket,
global-norm clip, ZeRO-1 slice update + all-gather) equals the single-process step on the concatenated batch
(SURVEY.md §8e parity definition); ZeRO-1 checkpoints are world-size independent (save at world 2 -> resume at world 2 and
at world 1 == the uninterrupted run); and a MODEL-level check: oracle LoRA gradients of two half-batches exchanged by two
ranks give the 1-rank update on the concatenated batch.  The per-slice update functions are torch stand-ins for the HIP
kernels (same contract)."""
import math
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opadpo_amd.optim import FlatAdamW, layer_buckets, torch_cast


def torch_sumsq(g, out):
    out += (g.double() ** 2).sum().float()


def torch_adamw(p, g, m, v, p_bf16, *, lr, beta1, beta2, eps, weight_decay, step, sumsq, max_norm, grad_div):
    scale = grad_div
    if sumsq is not None and max_norm:
        norm = math.sqrt(float(sumsq)) * grad_div
        scale *= min(1.0, max_norm / (norm + 1e-6))
    gi = g * scale
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(gi, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
    denom = v.sqrt() / math.sqrt(1 - beta2 ** step) + eps
    p.addcdiv_(m, denom, value=-lr / (1 - beta1 ** step))
    p_bf16.copy_(p.to(p_bf16.dtype))


CPU_KW = dict(sumsq_fn=torch_sumsq, adamw_fn=torch_adamw, cast_fn=torch_cast)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N = 5000
BOUNDS = [0, 1300, 2600, 3777, N]        # 4 uneven "layer" buckets; 3777 - 2600 is odd: padded slices


def _grads(step):
    g = torch.Generator().manual_seed(100 + step)
    return [torch.randn(N, generator=g) * (3.0 if step == 0 else 0.01) for _ in range(2)]   # one per rank


def _by_value(items):
    """Tensors cross the queue as numpy arrays (pickled by value): a torch tensor travels as a file descriptor that the SENDER must
    keep alive until the parent has received it - a 1-rank worker exits first (ConnectionResetError in the parent)."""
    return tuple(t.numpy() if isinstance(t, torch.Tensor) else t for t in items)


def _from_queue(items):
    import numpy as np
    return tuple(torch.from_numpy(t) if isinstance(t, np.ndarray) else t for t in items)


def _spawn(fn, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_from_queue(q.get(timeout=180)) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker(rank, world, port, q, mode, wire, bucketed):
    _init(rank, world, port)
    torch.manual_seed(0)
    master = torch.randn(N)
    grad = torch.zeros(N)
    work = master.to(torch.bfloat16)
    opt = FlatAdamW(master, grad, work, lr=1e-2, max_grad_norm=1.0, mode=mode, bucket_bounds=BOUNDS if bucketed else None,
                    exchange_dtype=wire, **CPU_KW)
    # ZeRO-1: the full-size fp32 master is dropped (this rank's slices live in p_shard; state_dict() still assembles the whole tensor)
    import types
    holder = types.SimpleNamespace(master=master)
    opt.release_full_master(holder)
    assert (opt.master is None and holder.master is None) == (mode == "zero1")
    norms = []
    for step in range(3):
        grad.copy_(_grads(step)[rank])
        if bucketed:                                   # what the backward hook does: highest bucket first, as soon as it is final
            for bi in range(len(opt.buckets) - 1, 0, -1):
                opt.launch_bucket(bi)                  # bucket 0 is left to prepare()
        opt.step(grad_accum_div=1.0)
        norms.append(opt.grad_norm_post_clip())
        opt.zero_grad()
    full_master = opt.state_dict()["master"]
    q.put(_by_value((rank, work.float().clone(), full_master.clone(), opt.shard_ranges(), norms)))
    dist.barrier()
    dist.destroy_process_group()


def _single_rank_reference(steps=3):
    torch.manual_seed(0)
    master = torch.randn(N)
    grad = torch.zeros(N)
    work = master.to(torch.bfloat16)
    opt = FlatAdamW(master, grad, work, lr=1e-2, max_grad_norm=1.0, **CPU_KW)
    norms = []
    for step in range(steps):
        g0, g1 = _grads(step)
        grad.copy_((g0 + g1) / 2)
        opt.step()
        norms.append(opt.grad_norm_post_clip())
        opt.zero_grad()
    return master, work, norms, opt


@pytest.mark.parametrize("mode,wire,bucketed", [("allreduce", None, False), ("allreduce", None, True), ("zero1", torch.float32, False),
                                                ("zero1", torch.float32, True), ("zero1", torch.bfloat16, True)])
def test_two_rank_step_equals_single_rank(mode, wire, bucketed):
    res = _spawn(_worker, 2, mode, wire, bucketed)
    master, work, ref_norms, _ = _single_rank_reference()
    exact = wire is not torch.bfloat16
    for rank, wk, full_master, ranges, norms in res:
        if exact:
            assert torch.equal(wk, work.float()), f"rank {rank}: bf16 working copy differs from the 1-rank step"
            torch.testing.assert_close(full_master, master, rtol=1e-6, atol=1e-7)
        else:       # bf16 on the wire: each rank's gradient is rounded once before the sum (2^-9 relative)
            # Adam is sign-like for near-zero gradient entries, so a few elements may move by up to ~lr per step differently
            diff = (full_master - master).abs()
            assert float((diff > 2e-3).float().mean()) < 0.01 and float(diff.max()) < 3 * 1e-2 * 3, (float(diff.max()))
            assert float((wk - work.float()).abs().max()) < 0.1
        for a, b in zip(norms, ref_norms):
            assert abs(a - b) < (1e-4 if exact else 1e-2) * max(1.0, b)
    assert torch.equal(res[0][1], res[1][1]), "ranks disagree on the working copy after the all-gather"
    if mode == "zero1":       # the two ranks' slices are a disjoint cover of the flat buffer
        cover = torch.zeros(N, dtype=torch.int32)
        for _, _, _, ranges, _ in res:
            for lo, hi, _off in ranges:
                cover[lo:hi] += 1
        assert bool((cover == 1).all())


# ---- ZeRO-1 checkpoint: save at world 2, resume at world 2 and at world 1 ---------------------------------------------------
def _ckpt_worker(rank, world, port, q, path, phase):
    _init(rank, world, port)
    torch.manual_seed(0)
    master = torch.randn(N)
    grad = torch.zeros(N)
    work = master.to(torch.bfloat16)
    opt = FlatAdamW(master, grad, work, lr=1e-2, max_grad_norm=1.0, mode="zero1", bucket_bounds=BOUNDS, exchange_dtype=torch.float32, **CPU_KW)
    steps = range(0, 2) if phase == "save" else range(2, 4)
    if phase != "save":
        opt.load_state_dict(torch.load(path, map_location="cpu"))
        assert opt.step_count == 2
    for step in steps:
        gs = _grads(step)
        grad.copy_(gs[rank] if world == 2 else (gs[0] + gs[1]) / 2)
        opt.step()
        opt.zero_grad()
    sd = opt.state_dict()                      # collective: every rank calls it
    if phase == "save" and rank == 0:
        torch.save(sd, path)
    q.put(_by_value((rank, work.float().clone(), sd["master"].clone(), sd["m"].clone(), sd["v"].clone())))
    dist.barrier()
    dist.destroy_process_group()


def test_zero1_save_resume_equals_uninterrupted_run_and_reshards():
    master, work, _, opt = _single_rank_reference(steps=4)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "optimizer.pt")
        _spawn(_ckpt_worker, 2, path, "save")
        sd = torch.load(path, map_location="cpu")
        assert sd["format"] == 2 and sd["m"].numel() == N and sd["master"].numel() == N and sd["step"] == 2
        for world in (2, 1):                   # same world size, and a changed one (re-sharding)
            res = _spawn(_ckpt_worker, world, path, "resume")
            for rank, wk, full_master, m, v in res:
                assert torch.equal(wk, work.float()), f"world {world} rank {rank}: resumed run differs from the uninterrupted one"
                torch.testing.assert_close(full_master, master, rtol=1e-6, atol=1e-7)
                torch.testing.assert_close(m, opt.m, rtol=1e-5, atol=1e-8)
                torch.testing.assert_close(v, opt.v, rtol=1e-5, atol=1e-10)


def test_old_rank0_only_optimizer_state_is_refused():
    master = torch.randn(64)
    opt = FlatAdamW(master, torch.zeros(64), master.to(torch.bfloat16), lr=1e-3, **CPU_KW)
    with pytest.raises(ValueError):
        opt.load_state_dict({"m": torch.zeros(32), "v": torch.zeros(32), "step": 3, "lo": 0, "hi": 32})


# ---- model level: oracle LoRA gradients of two half-batches on two ranks == the concatenated batch on one rank ----------
def _model_setup():
    from oracle import llava_ref as LR
    d = LR.LlavaDims.tiny(n_layers=2, hidden=64, n_heads=1, head_dim=64, ffn=128, vocab=96, v_hidden=64, v_heads=1, v_ffn=64,
                          v_layers=2, image_size=28, lora_r=8, lora_alpha=16.0)
    W = LR.init_weights(d, seed=0, std=0.05)
    lora = LR.init_lora(d, seed=1, b_std=0.05, with_vision=False)
    g = torch.Generator().manual_seed(3)
    B, Q, T = 4, 6, 5
    images = torch.randn(B, 3, d.image_size, d.image_size, generator=g)
    queries = torch.randint(3, d.vocab, (B, Q), generator=g)
    queries[:, 1] = -200
    qmask = torch.ones(B, Q, dtype=torch.bool)
    resp = {"chosen_response": torch.randint(3, d.vocab, (B, T), generator=g), "rejected_response": torch.randint(3, d.vocab, (B, T), generator=g)}
    resp["chosen_response"][:, -1] = 0
    ref = {k: torch.randn(B, T, generator=g) * 0.1 - 4.0 for k in resp}
    return LR, d, W, lora, (images, queries, qmask, resp, ref)


def _pair_loss_grads(LR, d, W, lora, batch, sel):
    """Mean token-level DPO loss (dpo_trainer.py:429-473) of the pairs `sel`, gradient w.r.t. every LoRA tensor, flattened."""
    from oracle import dpo_ref as D
    images, queries, qmask, resp, ref = batch
    lo = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
    out = LR.policy_forward(images[sel], queries[sel], qmask[sel], {k: v[sel] for k, v in resp.items()}, W, lo, d)
    losses, _, _ = D.dpo_loss(D.DPOConfig(), out["chosen_response_logprobs"], out["rejected_response_logprobs"],
                              ref["chosen_response"][sel], ref["rejected_response"][sel])
    losses.mean().backward()
    keys = sorted(lo)
    return torch.cat([lo[k].grad.flatten() for k in keys]), torch.cat([lora[k].flatten() for k in keys])


def _model_worker(rank, world, port, q):
    _init(rank, world, port)
    torch.set_num_threads(2)
    LR, d, W, lora, batch = _model_setup()
    sel = slice(rank * 2, rank * 2 + 2)                        # pairs sharded by rank
    g, p0 = _pair_loss_grads(LR, d, W, lora, batch, sel)
    master, grad = p0.clone(), g.clone()
    work = master.to(torch.bfloat16)
    n = master.numel()
    opt = FlatAdamW(master, grad, work, lr=1e-2, max_grad_norm=1.0, mode="zero1", bucket_bounds=[0, n // 3, n], exchange_dtype=torch.float32,
                    align=8, **CPU_KW)
    opt.launch_bucket(1)
    opt.step()
    q.put(_by_value((rank, opt.state_dict()["master"].clone(), opt.grad_norm_post_clip())))
    dist.barrier()
    dist.destroy_process_group()


def test_model_level_two_rank_gradient_equals_concatenated_batch():
    res = _spawn(_model_worker, 2)
    LR, d, W, lora, batch = _model_setup()
    g, p0 = _pair_loss_grads(LR, d, W, lora, batch, slice(0, 4))        # 1 rank, the concatenated batch (mean over all 4 pairs)
    master = p0.clone()
    opt = FlatAdamW(master, g.clone(), master.to(torch.bfloat16), lr=1e-2, max_grad_norm=1.0, **CPU_KW)
    opt.step()
    assert float(g.norm()) > 0
    for rank, full_master, norm in res:
        torch.testing.assert_close(full_master, master, rtol=1e-5, atol=1e-6)
        assert abs(norm - opt.grad_norm_post_clip()) < 1e-5
