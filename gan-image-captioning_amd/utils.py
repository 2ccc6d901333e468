"""Losses, temperature schedules, logger: the public surface of the reference's
``src/utils.py`` (get_losses :10-53, get_fixed_temperature :55-76, create_logger :78-103).

``get_losses`` computes on the GPU through ``gic_gan_losses`` (one fused reduction +
gradient kernel) and is differentiable with respect to the three logit vectors.
"""
from __future__ import annotations

import logging
import math
import sys
from time import gmtime, strftime

import torch

from . import engine

LOSS_TYPES = ("standard", "JS", "KL", "hinge", "tv", "rsgan")


class _GanLossFn(torch.autograd.Function):
    """One of the two losses as its own autograd node (which = 0: g_loss, 1: d_loss), so that
    ``d_loss.backward()`` never walks the generator's graph and vice versa."""

    @staticmethod
    def forward(ctx, loss_type, which, d_out_real, d_out_fake, g_out):
        need = any(ctx.needs_input_grad[2:])
        losses, grads = engine.gan_losses(loss_type, d_out_real.detach(), d_out_fake.detach(), g_out.detach(), want_grads=need)
        ctx.grads, ctx.which = grads, which
        return losses[which]

    @staticmethod
    def backward(ctx, up):
        g = ctx.grads
        # the upstream gradient is a scalar; the per-logit gradients were produced by the loss kernel
        if ctx.which == 1:
            return None, None, up * g["dd_real"], up * g["dd_fake"], None
        return None, None, up * g["dg_real"], up * g["dg_fake"], up * g["dg_out"]


def get_losses(d_out_real, d_out_fake, g_out, loss_type="JS", detach_d_for_g=False):
    """Returns (g_loss, d_loss) like the reference (utils.py:10,53).

    'standard', 'JS', 'KL', 'rsgan' follow utils.py:14-33,46-48.  'hinge' and 'tv' raise
    TypeError in the reference (nn.ReLU / nn.Tanh called with a tensor, utils.py:36-37,43-44);
    here they are implemented with the evident intent (element-wise relu / tanh).

    detach_d_for_g: cut g_loss's dependence on d_out_real / d_out_fake (only 'rsgan' has one, and in the
    reference trainer the discriminator gradients it produces are discarded, training.py:195).
    """
    if loss_type not in LOSS_TYPES:
        raise NotImplementedError("Divergence '%s' is not implemented" % loss_type)
    d_loss = _GanLossFn.apply(loss_type, 1, d_out_real, d_out_fake, g_out.detach())
    if detach_d_for_g:
        g_loss = _GanLossFn.apply(loss_type, 0, d_out_real.detach(), d_out_fake.detach(), g_out)
    else:
        g_loss = _GanLossFn.apply(loss_type, 0, d_out_real, d_out_fake, g_out)
    return g_loss, d_loss


def get_fixed_temperature(temper, i, N, adapt):
    """Temperature control policies (utils.py:55-76), float64 host arithmetic."""
    if adapt == "no":
        return 1.0
    if adapt == "lin":
        return 1 + i / (N - 1) * (temper - 1)
    if adapt == "exp":
        return temper ** (i / N)
    if adapt == "log":
        return 1 + (temper - 1) / math.log(N) * math.log(i + 1)
    if adapt == "sigmoid":
        return (temper - 1) * 1 / (1 + math.exp((N / 2 - i) * 20 / N)) + 1
    if adapt == "quad":
        return (temper - 1) / (N - 1) ** 2 * i ** 2 + 1
    if adapt == "sqrt":
        return (temper - 1) / math.sqrt(N - 1) * math.sqrt(i) + 1
    raise Exception("Unknown adapt type!")


def create_logger(name, silent=False, to_disk=False, log_file=None):
    """Message-only logger to stdout and/or file(s) (utils.py:78-103)."""
    log = logging.getLogger(name)
    log.setLevel(logging.DEBUG)
    log.propagate = False
    fmt = logging.Formatter(fmt="%(message)s", datefmt="%Y/%m/%d %I:%M:%S")
    if not silent:
        handler = logging.StreamHandler(sys.stdout)
        handler.setLevel(logging.DEBUG)
        handler.setFormatter(fmt)
        log.addHandler(handler)
    if to_disk:
        if log_file is None:
            log_file = strftime("log/log_%m%d_%H%M.txt", gmtime())
        for filename in ([log_file] if isinstance(log_file, str) else list(log_file)):
            handler = logging.FileHandler(filename, mode="w")
            handler.setLevel(logging.INFO)
            handler.setFormatter(fmt)
            log.addHandler(handler)
    return log
