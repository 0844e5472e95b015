"""CPU, world_size 2, gloo: the data-parallel step of FlatAdamW (gradient exchange, global-norm clip,
ZeRO-1 shard update + all-gather) equals the single-process step on the concatenated batch.
The per-shard update functions are torch stand-ins for the HIP kernels (same contract)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from opadpo_amd.optim import FlatAdamW


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N = 5000


def _grads(step):
    g = torch.Generator().manual_seed(100 + step)
    return [torch.randn(N, generator=g) * (3.0 if step == 0 else 0.01) for _ in range(2)]   # one per rank


def _worker(rank, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    torch.manual_seed(0)
    master = torch.randn(N)
    grad = torch.zeros(N)
    work = master.to(torch.bfloat16)
    opt = FlatAdamW(master, grad, work, lr=1e-2, max_grad_norm=1.0, mode=mode, sumsq_fn=torch_sumsq, adamw_fn=torch_adamw)
    norms = []
    for step in range(3):
        grad.copy_(_grads(step)[rank])
        opt.step(grad_accum_div=1.0)
        norms.append(opt.grad_norm_post_clip())
        opt.zero_grad()
    q.put((rank, work.float().clone(), master[opt.lo:opt.hi].clone(), (opt.lo, opt.hi), norms))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "zero1"])
def test_two_rank_step_equals_single_rank(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the averaged gradient
    torch.manual_seed(0)
    master = torch.randn(N)
    grad = torch.zeros(N)
    work = master.to(torch.bfloat16)
    opt = FlatAdamW(master, grad, work, lr=1e-2, max_grad_norm=1.0, sumsq_fn=torch_sumsq, adamw_fn=torch_adamw)
    ref_norms = []
    for step in range(3):
        g0, g1 = _grads(step)
        grad.copy_((g0 + g1) / 2)
        opt.step()
        ref_norms.append(opt.grad_norm_post_clip())
        opt.zero_grad()
    for rank, wk, mshard, (lo, hi), norms in res:
        assert torch.equal(wk, work.float()), f"rank {rank}: bf16 working copy differs from the 1-rank step"
        torch.testing.assert_close(mshard, master[lo:hi], rtol=1e-6, atol=1e-7)
        for a, b in zip(norms, ref_norms):
            assert abs(a - b) < 1e-4 * max(1.0, b)
    if mode == "zero1":
        assert res[0][3][1] == res[1][3][0] and res[0][3][0] == 0 and res[1][3][1] == N   # disjoint cover
