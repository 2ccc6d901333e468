"""Script entry, drop-in for the reference's ``src/main.py``: seeds (main.py:14-23), datasets
(main.py:36-40), ``GANInstructor(...)._run()`` (main.py:46-48).  ``--synthetic 1`` swaps COCO for
``SyntheticCaptionData`` so the same entry point drives the synthetic configs.

Single GPU:   python -m gan_image_captioning_amd.main --synthetic 1 --vocab-size 10000 ...
Multi GPU:    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m gan_image_captioning_amd.main ...
"""
from __future__ import annotations

import random

import numpy as np
import torch

from .args import get_args
from .tasks import COCO_data, SyntheticCaptionData
from .training import GANInstructor


def main(argv=None):
    args = get_args(argv)
    seed = args.seed                     # the reference hard-codes 1008 (main.py:14)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if args.synthetic:
        if args.vocab_size <= 0:
            args.vocab_size = 10000
        n_train = args.synthetic_batches * args.adv_train_batch_size
        train = SyntheticCaptionData(n_train, args.vocab_size, args.image_size, args.synthetic_caption_len, seed)
        val = SyntheticCaptionData(max(args.adv_eval_batch_size, n_train // 4), args.vocab_size, args.image_size,
                                   args.synthetic_caption_len, seed + 1)
    else:
        train = COCO_data(args.data_dir + "/dataset_coco.json", args.data_dir, "train", args.image_size,
                          args.captions_per_image, dataset_percent=args.dataset_percent)
        args.vocab_size = train.vocab_size                                       # main.py:38
        if getattr(args, "decoder", "lstm") == "attention" and args.vocab_size % 4:
            # the attention kernels (no reference counterpart) move vocabulary rows in 16-byte pieces: V is rounded up to a multiple
            # of 4; the extra classes never occur in the data (the LSTM decoder and the discriminator take any V)
            args.vocab_size += 4 - args.vocab_size % 4
        val = COCO_data(args.data_dir + "/dataset_coco.json", args.data_dir, "val", args.image_size, args.captions_per_image,
                        vocab_dicts=(train.word_to_index, train.index_to_word), dataset_percent=args.dataset_percent)
    inst = GANInstructor(args, train, val)
    if getattr(args, "resume", ""):
        inst.load_checkpoint(args.resume)
    inst._run()
    return inst


if __name__ == "__main__":
    main()
