"""Import the reference's own classes from /root/reference/src (read-only).

TEST INFRASTRUCTURE; only usable in the build container (the reference does not
travel to the GPU box).  The reference modules import ``torchvision.models``
(unused on the decoder/discriminator path) and ``SummaryWriter`` (unused by
get_losses) at module scope; both packages are absent here, so inert stand-in
modules are registered *for those two unused imports only* before importing.
Nothing from the reference is copied: the classes are used in place.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import torch

REFERENCE_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "generator.py"))


def load():
    """Returns (generator_module, discriminator_module, utils_module)."""
    if not available():
        raise RuntimeError("reference sources not present (only in the build container)")
    sys.dont_write_bytecode = True          # the reference tree is read-only
    for name in ("torchvision", "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:                # never instantiated on the oracle path
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import generator as ref_generator       # noqa: E402
    import discriminator as ref_discriminator  # noqa: E402
    import utils as ref_utils               # noqa: E402
    return ref_generator, ref_discriminator, ref_utils


def make_args(vocab, embed, hidden, layers, temperature=100, disc_embed_dim=64, disc_num_rep=64,
              filter_sizes=(3, 4, 5), num_filters=(300, 300, 300)) -> argparse.Namespace:
    """The fields the reference modules read from args."""
    return argparse.Namespace(
        vocab_size=vocab, gen_embed_dim=embed, gen_hidden_dim=hidden, gen_num_layers=layers,
        max_seq_len=34, temperature=temperature, device=torch.device("cpu"), gen_init="uniform",
        disc_embed_dim=disc_embed_dim, disc_num_rep=disc_num_rep, padding_idx=0,
        disc_num_filters=list(num_filters), disc_filter_sizes=list(filter_sizes), disc_init="uniform",
        conditional_gan=0, cgan=0)
