"""Command-line / config surface, drop-in for the reference's ``src/args.py``.

Every flag of the reference keeps its name, type and default (args.py:12-257); the parsed
Namespace doubles as the model-config object exactly as there.  ``get_args()`` keeps the
side effects too: a fresh ``<save-dir>/<expt-name>_<n>/`` with a ``models/`` sub-directory
(args.py:261-273) and ``args.device`` resolved to a ``torch.device`` (args.py:275-278).

New flags (all optional, defaults preserve the reference behaviour where one exists) select the
MI355X-specific knobs: compute dtype, synthetic data, fused step, data-parallel options.
"""
from __future__ import annotations

import argparse
import os

import torch

# (flag, type, default, help[, extra kwargs])  -- reference flags, args.py line numbers in comments
_MODEL = [
    ("--gen-hidden-dim", int, 512, "hidden dimension of generator"),                      # :12
    ("--gen-embed-dim", int, 32, "embedding dimension of generator"),                     # :17
    ("--gen-num-layers", int, 1, "number of layers in generator"),                        # :22
    ("--gen-init", str, "uniform", "Initialization strategy for generator weights"),      # :27
    ("--disc-embed-dim", int, 64, "embeddings dimension to use in discriminator"),        # :34
    ("--disc-num-rep", int, 64, "number of representations to use for CNN discriminator"),  # :39
    # the reference declares these two with type=list (args.py:44-52), which only works at the defaults;
    # here a comma-separated list is accepted as well
    ("--disc-filter-sizes", "intlist", [3, 4, 5], "Layer wise filter sizes to use in discriminator"),
    ("--disc-num-filters", "intlist", [300, 300, 300], "number of filters to use in discriminator per layer"),
    ("--disc-init", str, "uniform", "init strategy for discriminator weights"),            # :54
    ("--conditional-gan", int, 0, "is the gan conditional?", {"choices": [0, 1]}),         # :61
]
_DATA = [
    ("--vocab-size", int, -1, "vocab size for training"),                                  # :78
    ("--max-seq-len", int, 34, "maximum sequence length of captions"),                     # :83
    ("--padding-idx", int, 0, "index of padding token in vocab"),                          # :88
    ("--image-size", int, 256, "resize dim of images"),                                    # :96
    ("--captions-per-image", int, 1, "no of captions to use per image"),                   # :101
    ("--dataset_percent", float, 1.0, "percentage of dataset to use for training"),        # :108
]
_TRAIN = [
    ("--pretrain-lr", float, 1e-2, "learning rate for pretraining generator"),             # :123
    ("--pretrain-epochs", int, 0, "number of epochs for pretraining generator"),           # :128
    ("--pre-train-batch-size", int, 64, "batch size of pretrain training"),                # :133
    ("--pre-eval-batch-size", int, 64, "batch size of pretrain evaluation"),               # :138
    ("--gen-lr", float, 1e-4, "learning rate for adversarial training of generator"),      # :145
    ("--disc-lr", float, 1e-4, "learning rate for adversarial training of discriminator"),  # :150
    ("--disc-train-freq", int, 1, "ratio of training steps of disc vs gen (parsed, unused as in the reference)"),  # :155
    ("--adv-epochs", int, 30, "number of epochs for adversarial training"),                # :160
    ("--adv-train-batch-size", int, 64, "batch size of adversarial training"),             # :165
    ("--adv-eval-batch-size", int, 64, "batch size of adversarial evaluation"),            # :170
    ("--adv-loss-type", str, "standard", "Loss function to use for adversarial training"),  # :175
    ("--temperature", int, 100, "Temperature for rel gan training"),                       # :180
    ("--temp-adpt", str, "exp", "Temperature adoption strategy"),                          # :185
    ("--clip-norm", float, 5.0, "Gradient clipping threshold"),                            # :190
]
_GLOBAL = [
    ("--device", str, "cuda", "device to use for training (cpu|cuda)"),                    # :208
    ("--device-ids", int, 0, "device id (parsed, unused as in the reference)"),            # :213
    ("--expt-name", str, "debug", "Name of the experiment"),                               # :218
    ("--model-dir", str, "models", "directory to save models"),                            # :223
    ("--data-dir", str, "./data", "directory where data is stored"),                       # :228
    ("--save-dir", str, "./save", "directory to save the expt logs and tensorboard logs"),  # :233
    ("--adv-log-step", int, 1, "Log step frequency for adversarial training"),             # :238
    ("--pre-log-step", int, 1, "Log step frequency for pretraining"),                      # :243
    ("--test-log-step", int, 1, "Log step frequency for testing (parsed, unused)"),        # :248
    ("--log-file", str, "log", "Log file to save logs"),                                   # :253
]
# MI355X-native additions (no reference counterpart)
_NATIVE = [
    ("--compute-dtype", str, "bf16", "MFMA operand dtype: bf16 (perf) or fp32 (parity)", {"choices": ["bf16", "fp32"]}),
    ("--encoder-arch", str, "resnet18", "trunk shape: resnet18 (the reference's) or resnet50", {"choices": ["resnet18", "resnet50"]}),
    ("--step-impl", str, "fused", "adversarial step driver: fused (direct kernel sequence) or autograd (module API)",
     {"choices": ["fused", "autograd"]}),
    ("--adv-mode", str, "relgan", "generator update of the adversarial loop: relgan = Gumbel-softmax relaxation through D (the reference, "
                                  "training.py:144-169); seqgan = policy gradient with Monte-Carlo roll-outs scored by D", {"choices": ["relgan", "seqgan"]}),
    ("--mc-rollouts", int, 16, "Monte-Carlo roll-outs per prefix (--adv-mode seqgan)"),
    ("--decoder", str, "lstm", "caption decoder: lstm (the reference, generator.py:27-96) or attention (visual attention over the trunk's "
                               "feature map, Show-Attend-Tell style; needs --conditional-gan 1)", {"choices": ["lstm", "attention"]}),
    ("--attn-dim", int, 512, "width of the additive-attention hidden layer (--decoder attention)"),
    ("--real-as-ids", int, 1, "feed real captions to D as token ids (gather) instead of a dense one-hot", {"choices": [0, 1]}),
    ("--synthetic", int, 0, "use synthetic (image, caption) batches instead of COCO", {"choices": [0, 1]}),
    ("--synthetic-batches", int, 8, "batches per epoch of the synthetic dataset"),
    ("--synthetic-caption-len", int, 20, "caption length (incl. <S>/<E>) of synthetic batches"),
    ("--resume", str, "", "checkpoint to start from: adv_model.ckpt ({generator, discriminator}) or pretrained_model.ckpt "
                          "(generator state dict), in the reference's format (training.py:118,225-226); the reference cannot resume"),
    ("--seed", int, 1008, "RNG seed (src/main.py:14 fixes 1008)"),
    ("--num-workers", int, 4, "DataLoader workers (training.py:28-32 uses 4)"),
]


def _intlist(text):
    if isinstance(text, (list, tuple)):
        return [int(v) for v in text]
    return [int(v) for v in str(text).replace("[", "").replace("]", "").split(",") if v.strip()]


def _add(parser, rows):
    for row in rows:
        flag, typ, default, helptext = row[:4]
        extra = dict(row[4]) if len(row) > 4 else {}
        if typ == "intlist":
            typ = _intlist
        parser.add_argument(flag, type=typ, default=default, help=helptext, **extra)


def add_model_args(parser):
    _add(parser, _MODEL)


def add_data_args(parser):
    _add(parser, _DATA)


def add_training_args(parser):
    _add(parser, _TRAIN)


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser("NLP GAN args")
    add_training_args(parser)
    add_data_args(parser)
    add_model_args(parser)
    _add(parser, _GLOBAL)
    _add(parser, _NATIVE)
    return parser


def resolve_device(args) -> None:
    """args.py:275-278: 'cuda' -> cuda:0 when available else cpu.  Under torchrun each rank takes
    cuda:LOCAL_RANK (one process per GPU)."""
    if str(args.device).startswith("cuda") and torch.cuda.is_available():
        args.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    else:
        args.device = torch.device("cpu")


def make_experiment_dirs(args) -> None:
    """args.py:261-273: first unused <save_dir>/<expt_name>_<n>, plus its model dir and log file path."""
    n = 1
    while os.path.exists(os.path.join(args.save_dir, f"{args.expt_name}_{n}")):
        n += 1
    args.expt_name = f"{args.expt_name}_{n}"
    args.save_dir = os.path.join(args.save_dir, args.expt_name)
    os.makedirs(args.save_dir)
    args.model_dir = os.path.join(args.save_dir, args.model_dir)
    os.mkdir(args.model_dir)
    args.log_file = os.path.join(args.save_dir, args.log_file)


def get_args(argv=None, make_dirs: bool = True):
    args = build_parser().parse_args(argv)
    if make_dirs:
        make_experiment_dirs(args)
    resolve_device(args)
    return args


def default_args(**overrides):
    """Config object with every default, no directory side effects (for tests / bench / library use)."""
    args = build_parser().parse_args([])
    for k, v in overrides.items():
        if not hasattr(args, k):
            raise AttributeError(f"unknown config field {k}")
        setattr(args, k, v)
    if not isinstance(args.device, torch.device):
        resolve_device(args)
    return args
