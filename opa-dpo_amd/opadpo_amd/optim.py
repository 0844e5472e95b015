"""Clip + AdamW over the flat LoRA buffer, data-parallel gradient exchange, cosine schedule.

Replaces accelerate.clip_grad_norm_ + bitsandbytes paged AdamW + DDP of the reference
(opadpo/dpo_models/rl_trainer.py:164-175, utils/trainer_utils.py:9-49).  The reference never
syncs gradients (every backward runs under no_sync — SURVEY.md Quirk Q1); this build does what
north_star mandates: ONE collective exchange of the flat LoRA gradient per optimizer step.

Two exchange modes (both over RCCL on GPUs, gloo in CPU tests):
  * "allreduce": all-reduce(SUM) of the flat fp32 gradient, every rank updates everything;
  * "zero1"    : reduce-scatter -> each rank clips + AdamW-updates its 1/N shard (fp32 master, m, v
                 live only for that shard) -> all-gather of the bf16 working copy.
The per-shard update function is injected so that the partitioning / collective logic is testable
on CPU (tests/test_dist_cpu.py) while the product passes the HIP kernels.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.distributed as dist


def cosine_lr(step: int, base_lr: float, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    """HF get_scheduler('cosine') multiplier * base_lr after `step` scheduler steps."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * prog)))


def shard_bounds(numel: int, world: int, rank: int, align: int = 256):
    """Contiguous shard [lo, hi) of a flat buffer padded to world*align elements."""
    per = (numel + world * align - 1) // (world * align) * align
    lo = min(numel, rank * per)
    hi = min(numel, lo + per)
    return lo, hi, per


def hip_sumsq(g: torch.Tensor, out: torch.Tensor) -> None:
    from . import lib as L
    L.call("opadpo_sumsq", L.ptr(g), g.numel(), L.ptr(out), L.stream())


def hip_adamw(p, g, m, v, p_bf16, *, lr, beta1, beta2, eps, weight_decay, step, sumsq, max_norm, grad_div) -> None:
    from . import lib as L
    L.call("opadpo_adamw", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(p_bf16), p.numel(), lr, beta1, beta2, eps,
           weight_decay, step, L.ptr(sumsq), max_norm if max_norm is not None else 0.0, grad_div, L.stream())


class FlatAdamW:
    """AdamW(beta=(0.9,0.999), eps=1e-8, wd=0) + global-norm clip on flat fp32 master / grad buffers."""

    def __init__(self, master: torch.Tensor, grad: torch.Tensor, work_bf16: torch.Tensor, *, lr: float,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, max_grad_norm: Optional[float] = 1.0,
                 mode: str = "allreduce", group=None, sumsq_fn: Callable = hip_sumsq, adamw_fn: Callable = hip_adamw):
        assert mode in ("allreduce", "zero1")
        self.master, self.grad, self.work = master, grad, work_bf16
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.mode = mode
        self.group = group
        self.sumsq_fn, self.adamw_fn = sumsq_fn, adamw_fn
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # OPADPO_FORCE_COLLECTIVES=1 (diagnostics): run the exchange code path even in a 1-rank group
        self._collective = self.world > 1 or (dist.is_available() and dist.is_initialized()
                                                and __import__("os").environ.get("OPADPO_FORCE_COLLECTIVES") == "1")
        n = master.numel()
        if mode == "zero1" and self._collective:
            self.lo, self.hi, self.per = shard_bounds(n, self.world, self.rank)
        else:
            self.lo, self.hi, self.per = 0, n, n
        sh = self.hi - self.lo
        self.m = torch.zeros(sh, dtype=torch.float32, device=master.device)
        self.v = torch.zeros(sh, dtype=torch.float32, device=master.device)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=master.device)
        self.step_count = 0
        self.last_grad_norm = None     # lazily materialised POST-clip norm (Quirk Q15)
        if mode == "zero1" and self._collective:
            pad_n = self.per * self.world
            self._gpad = torch.zeros(pad_n, dtype=torch.float32, device=master.device)
            self._gshard = torch.zeros(self.per, dtype=torch.float32, device=master.device)
            self._wpad = torch.zeros(pad_n, dtype=work_bf16.dtype, device=master.device)

    # ---- gradient exchange --------------------------------------------------------------------------
    def _exchange(self) -> torch.Tensor:
        """Returns the (summed over ranks) gradient slice this rank updates."""
        if not self._collective:
            return self.grad
        if self.mode == "allreduce":
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)
            return self.grad
        n = self.grad.numel()
        self._gpad[:n].copy_(self.grad)
        backend = dist.get_backend(self.group)
        if backend == "gloo":   # gloo has no reduce_scatter: all-reduce then slice (CPU tests only)
            dist.all_reduce(self._gpad, op=dist.ReduceOp.SUM, group=self.group)
            self._gshard.copy_(self._gpad[self.rank * self.per:(self.rank + 1) * self.per])
        else:
            dist.reduce_scatter_tensor(self._gshard, self._gpad, op=dist.ReduceOp.SUM, group=self.group)
        return self._gshard[: self.hi - self.lo]

    def step(self, grad_accum_div: float = 1.0) -> None:
        """One optimizer step.  Effective gradient = sum over ranks / (world * grad_accum_div)."""
        self.prepare(grad_accum_div)
        self.apply()

    def prepare(self, grad_accum_div: float = 1.0) -> None:
        """Phase 1: gradient exchange + sum of squares of the (summed) gradient.  Several FlatAdamW instances that must be
        clipped by ONE global norm (SFT stage: LLM LoRA + vision LoRA buffers) call prepare() on each, add their `sumsq`
        tensors, write the total back into each (`share_sumsq`) and then call apply()."""
        self.step_count += 1
        self._g = self._exchange()
        self._grad_div = 1.0 / (self.world * grad_accum_div)
        self.sumsq.zero_()
        if self.max_grad_norm is not None:
            self.sumsq_fn(self._g, self.sumsq)
            if self._collective and self.mode == "zero1":
                dist.all_reduce(self.sumsq, op=dist.ReduceOp.SUM, group=self.group)

    @staticmethod
    def share_sumsq(*opts: "FlatAdamW") -> None:
        total = sum(o.sumsq for o in opts)
        for o in opts:
            o.sumsq.copy_(total)

    def apply(self) -> None:
        """Phase 2: clip by the norm in `sumsq`, AdamW on this rank's slice, (ZeRO-1) all-gather of the working copy."""
        g, grad_div = self._g, self._grad_div
        p = self.master[self.lo:self.hi]
        wk = self.work[self.lo:self.hi]
        self.adamw_fn(p, g, self.m, self.v, wk, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                      weight_decay=self.wd, step=self.step_count,
                      sumsq=self.sumsq if self.max_grad_norm is not None else None,
                      max_norm=self.max_grad_norm, grad_div=grad_div)
        if self._collective and self.mode == "zero1":
            n = self.work.numel()
            self._wpad[self.lo:self.hi].copy_(wk)
            shard = self._wpad[self.rank * self.per:(self.rank + 1) * self.per]
            if dist.get_backend(self.group) == "gloo":
                parts = [torch.empty_like(shard) for _ in range(self.world)]
                dist.all_gather(parts, shard.clone(), group=self.group)
                self._wpad.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(self._wpad, shard.clone(), group=self.group)
            self.work.copy_(self._wpad[:n])

    def zero_grad(self) -> None:
        self.grad.zero_()

    def grad_norm_post_clip(self) -> float:
        """loss/grad_norm of the reference is computed AFTER clipping (rl_trainer.py:165-171)."""
        if self.max_grad_norm is None:
            return float("nan")
        norm = math.sqrt(float(self.sumsq.item())) * self._grad_div
        return norm * min(1.0, self.max_grad_norm / (norm + 1e-6))

    def state_dict(self):
        return {"m": self.m.cpu(), "v": self.v.cpu(), "step": self.step_count, "lo": self.lo, "hi": self.hi}

    def load_state_dict(self, sd):
        assert sd["lo"] == self.lo and sd["hi"] == self.hi, "optimizer shard layout changed (world size differs)"
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
