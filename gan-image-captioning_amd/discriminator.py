"""RelGAN-style multi-representation CNN discriminator with the reference's module API and
state-dict keys (src/discriminator.py:9-86), computing through libgicap.so.

``forward(inp)`` accepts the reference's dense ``[B, L, V]`` float tensor (soft captions or a
one-hot) and, as an extension, ``int64 [B, L]`` token ids (the one-hot product evaluated as a
gather).  Gradients flow to the parameters and to a dense ``inp``.
"""
from __future__ import annotations

import contextlib
import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import engine
from .generator import SEEDS, _LinearParams, _compute_dtype


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, train, keep_mask, seed, want_param_grads, inp, *params):
        dparams = [p.detach() for p in params]
        is_ids = inp.dtype == torch.int64
        soft = None if is_ids else eng.soft_input(inp.detach())
        ids = inp if is_ids else None
        logits, st = eng.fwd(dparams, soft, ids, train, keep_mask, seed)
        ctx.eng, ctx.train, ctx.st, ctx.dparams = eng, train, st, dparams
        ctx.soft, ctx.ids = soft, ids
        ctx.in_dtype = inp.dtype
        ctx.want_param_grads = want_param_grads
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        want_inp = ctx.needs_input_grad[5] and ctx.soft is not None
        want_par = ctx.want_param_grads and any(ctx.needs_input_grad[6:])
        grads, d_inp = ctx.eng.bwd(ctx.dparams, ctx.st, ctx.soft, ctx.ids, ctx.train, d_logits, want_par, want_inp)
        ctx.st = None
        if d_inp is not None and d_inp.dtype != ctx.in_dtype:
            d_inp = d_inp.to(ctx.in_dtype)
        pg = grads if grads is not None else [None] * len(ctx.dparams)
        return (None, None, None, None, None, d_inp, *pg)


class _ConvParams(nn.Module):
    """nn.Conv2d(1, n, (f, s), stride=(1, s)) parameter container (discriminator.py:22-25)."""

    def __init__(self, n: int, f: int, s: int):
        super().__init__()
        k = 1.0 / math.sqrt(f * s)
        self.weight = nn.Parameter(torch.empty(n, 1, f, s).uniform_(-k, k))
        self.bias = nn.Parameter(torch.empty(n).uniform_(-k, k))


class Discriminator(nn.Module):
    def __init__(self, args, gpu=False, dropout=0.2):
        super().__init__()
        if not 0.0 <= float(dropout) < 1.0:            # nn.Dropout(dropout), discriminator.py:10,30 (p = 1 would zero every feature)
            raise ValueError("dropout probability has to be in [0, 1), but got {}".format(dropout))
        self.dropout_p = float(dropout)
        self.vocab_size = args.vocab_size
        self.embed_dim = args.disc_embed_dim
        self.padding_idx = args.padding_idx
        self.feature_dim = sum(args.disc_num_filters)
        self.emb_dim_single = int(args.disc_embed_dim / args.disc_num_rep)
        self.gpu = gpu
        self.embeddings = _LinearParams(self.vocab_size, self.embed_dim, bias=False)
        self.convs = nn.ModuleList([_ConvParams(n, f, self.emb_dim_single)
                                    for n, f in zip(args.disc_num_filters, args.disc_filter_sizes)])
        self.highway = _LinearParams(self.feature_dim, self.feature_dim)
        self.feature2out = _LinearParams(self.feature_dim, 100)
        self.out2logits = _LinearParams(100, 1)
        self.args = args
        self._engine: Optional[engine.DiscEngine] = None
        self._param_grads = True
        self.init_params()

    def engine(self) -> engine.DiscEngine:
        if self._engine is None:
            a = self.args
            self._engine = engine.DiscEngine(a.vocab_size, a.disc_embed_dim, a.disc_num_rep, a.disc_filter_sizes,
                                             a.disc_num_filters, _compute_dtype(a), dropout=self.dropout_p)
        return self._engine

    def param_list(self) -> List[nn.Parameter]:
        ps = [self.embeddings.weight]
        for c in self.convs:
            ps += [c.weight, c.bias]
        return ps + [self.highway.weight, self.highway.bias, self.feature2out.weight, self.feature2out.bias,
                     self.out2logits.weight, self.out2logits.bias]

    @contextlib.contextmanager
    def input_grad_only(self):
        """Inside this context forward() records no parameter gradients: the generator's path through D
        (training.py:164,169) only needs d(loss)/d(input); the parameter gradients the reference computes
        there are discarded by the next zero_grad (training.py:195)."""
        prev, self._param_grads = self._param_grads, False
        try:
            yield self
        finally:
            self._param_grads = prev

    def forward(self, inp, keep_mask=None):
        """inp: float [B, L, V] (or int64 ids [B, L]) -> logits [B * num_rep] (discriminator.py:34-62)."""
        seed = 0 if keep_mask is not None else SEEDS.next()
        return _DiscFn.apply(self.engine(), self.training, keep_mask, seed, self._param_grads, inp, *self.param_list())

    def get_feature(self, inp):
        raise NotImplementedError("get_feature is unused by the reference trainer and broken there for num_rep > 1 "
                                  "(discriminator.py:64-77)")

    def init_params(self):
        """discriminator.py:79-86."""
        for param in self.parameters():
            if param.requires_grad and len(param.shape) > 0:
                if self.args.disc_init == "uniform":
                    torch.nn.init.uniform_(param, a=-0.05, b=0.05)
                elif self.args.disc_init == "normal":
                    torch.nn.init.normal_(param, std=1 / math.sqrt(param.shape[0]))
