"""Flat parameter arenas + fused clip_grad_norm_ / Adam (reference src/training.py:24-26,194-199).

A ``ParamArena`` re-homes a list of ``nn.Parameter``s into one contiguous float32 buffer (each
parameter becomes a view, so state-dicts / checkpoints keep the reference's keys) with a matching
flat gradient buffer.  One flat buffer per model means: one norm reduction, one Adam kernel, and one
RCCL all-reduce per optimizer under data parallelism.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.nn as nn

from . import engine


class ParamArena:
    def __init__(self, params: Iterable[nn.Parameter]):
        self.params: List[nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("ParamArena needs at least one parameter")
        dev = self.params[0].device
        engine.require_gpu(*self.params)
        self.sizes = [p.numel() for p in self.params]
        # 16-byte aligned slots so every view can be a 16-B vector-load operand
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                if p.dtype != torch.float32:
                    raise ValueError("master weights must be float32")
                view = self.flat[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + n].view(p.shape)
        engine.bump_param_epoch()

    def span(self, params) -> Optional[tuple]:
        """(start, end) of the arena slots of ``params`` if they are one contiguous run (in any order), else None."""
        want = {id(p) for p in params}
        idx = sorted(i for i, p in enumerate(self.params) if id(p) in want)
        if len(idx) != len(want) or not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        end = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.params) else self.numel
        return self.offsets[idx[0]], end

    def grad_views(self) -> List[torch.Tensor]:
        return [self.grad[o:o + n].view(p.shape) for p, o, n in zip(self.params, self.offsets, self.sizes)]

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, g in zip(self.params, self.grad_views()):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g


class FusedClipAdam:
    """``opt.zero_grad(); loss.backward(); clip_grad_norm_(params, clip); opt.step()`` of the reference's
    ``optimize`` (training.py:194-199) with clip + Adam as two kernels over the arena.  Several optimizers
    may share one arena with separate moments (pretrain_opt / gen_opt, training.py:24-25)."""

    def __init__(self, arena: ParamArena, lr: float, clip_norm: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.arena, self.lr, self.clip_norm, self.betas, self.eps = arena, lr, clip_norm, betas, eps
        dev = arena.flat.device
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)       # pre-clip global L2 norm of the last step
        self._partials = torch.zeros(engine.clip_adam_partials(arena.numel), dtype=torch.float32, device=dev)

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero_grad()

    def step(self) -> None:
        a = self.arena
        engine.clip_adam(a.flat, a.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                         self.clip_norm, self.step_count, self.grad_norm, self._partials)

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count, "lr": self.lr}

    def load_state_dict(self, sd) -> None:
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count.copy_(sd["step"])
        self.lr = sd.get("lr", self.lr)
