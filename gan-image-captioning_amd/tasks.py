"""Data side of the trainer: the reference's ``COCO_data`` / ``collate_fn`` contract
(src/tasks.py:18-158) plus a synthetic dataset of the same batch shape.

Batch contract (tasks.py:138-158): ``(images f32[B,3,S,S], captions i64[B,Lmax], lengths i32[B],
max_caption_len)`` with caption rows ``[<S>=1] + tokens + [<E>=2] + [<PAD>=0]*`` and
``max_caption_len = longest caption + 2``.  Specials: <PAD>=0, <S>=1, <E>=2, <UNK>=3 (tasks.py:42-45).

Image decoding is host work outside the timed hot path; torchvision is not required (the resize /
to-tensor / normalise pipeline of tasks.py:92-100 is restated with PIL + numpy).
"""
from __future__ import annotations

import json
import os
import pickle
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

SPECIALS = ("<PAD>", "<S>", "<E>", "<UNK>")
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def load_image(path: str, size: int) -> torch.Tensor:
    """Resize((S,S), bilinear) -> [0,1] CHW -> grey to 3 channels -> ImageNet normalise (tasks.py:92-100)."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("RGB", "L"):
            im = im.convert("RGB")
        im = im.resize((size, size), resample=Image.BILINEAR)
        arr = np.asarray(im, dtype=np.float32) / 255.0
    if arr.ndim == 2:
        arr = np.repeat(arr[:, :, None], 3, axis=2)
    arr = (arr - IMAGENET_MEAN) / IMAGENET_STD
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))


class COCO_data(Dataset):
    """Karpathy-split COCO captions (dataset_coco.json); vocabulary built from the training split."""

    def __init__(self, captions_path, image_path, split, image_size=256, captions_per_image=5, vocab_dicts=None,
                 dataset_percent=1.0):
        assert split in {"train", "val", "test"}
        self.split, self.image_path, self.image_size = split, image_path, image_size
        self.dataset_percent = dataset_percent
        cache = os.path.join(image_path, f"{split}_{captions_per_image}.pkl")      # tasks.py:30,88
        if os.path.exists(cache):
            print("Loading from saved dict")
            with open(cache, "rb") as fh:
                saved = pickle.load(fh)
            self.captions, self.word_to_index, self.index_to_word = saved["captions"], saved["w2i"], saved["i2w"]
        else:
            print("Creating and saving dict")
            with open(captions_path, "r") as fh:
                entries = json.load(fh)["images"]
            if vocab_dicts is None:
                self.word_to_index = {w: i for i, w in enumerate(SPECIALS)}
                self.index_to_word = {i: w for i, w in enumerate(SPECIALS)}
            else:
                self.word_to_index, self.index_to_word = vocab_dicts
            self.captions = []
            for row in entries:
                if split not in row["filepath"]:                                    # tasks.py:60
                    continue
                meta = {k: v for k, v in row.items() if not isinstance(v, list)}
                for sent in row["sentences"][:captions_per_image]:
                    self.captions.append({**meta, **sent})
                    if vocab_dicts is None:
                        for word in sent["tokens"]:
                            if word not in self.word_to_index:
                                idx = len(self.word_to_index)
                                self.word_to_index[word] = idx
                                self.index_to_word[idx] = word
            with open(cache, "wb+") as fh:
                pickle.dump({"captions": self.captions, "w2i": self.word_to_index, "i2w": self.index_to_word}, fh)
        self.vocab_size = len(self.word_to_index)

    def __len__(self):
        return int(self.dataset_percent * len(self.captions))

    def __getitem__(self, index):
        entry = self.captions[index]
        image = load_image(os.path.join(self.image_path, entry["filepath"], entry["filename"]), self.image_size)
        unk = self.word_to_index["<UNK>"]
        tokens = [t if isinstance(t, int) else self.word_to_index.get(t, unk) for t in entry["tokens"]]
        return image, tokens


class SyntheticCaptionData(Dataset):
    """Deterministic stand-in for COCO: normal-distributed "normalised pixels" and random token ids in
    [4, V) (no specials inside a caption), every caption the same length so max_caption_len is exact
    (SURVEY.md §8(d))."""

    def __init__(self, num_items: int, vocab_size: int, image_size: int = 224, caption_len: int = 20, seed: int = 1008,
                 ragged: bool = False):
        self.vocab_size, self.image_size, self.n = vocab_size, image_size, num_items
        g = torch.Generator().manual_seed(seed)
        self.seeds = torch.randint(0, 2 ** 31 - 1, (num_items,), generator=g).tolist()
        self.body_len = caption_len - 2
        self.ragged = ragged
        self.word_to_index = {w: i for i, w in enumerate(SPECIALS)}
        self.index_to_word = {i: w for i, w in enumerate(SPECIALS)}

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seeds[index])
        image = torch.randn(3, self.image_size, self.image_size, generator=g)
        n = self.body_len if not self.ragged else int(torch.randint(1, self.body_len + 1, (1,), generator=g))
        tokens = torch.randint(4, self.vocab_size, (n,), generator=g).tolist()
        return image, tokens


def collate_fn(batch: Sequence[Tuple[torch.Tensor, List[int]]]):
    """tasks.py:138-158."""
    size = batch[0][0].shape[-1]
    images = torch.zeros(len(batch), 3, size, size)
    max_caption_len = max(len(tokens) for _, tokens in batch) + 2
    captions = torch.zeros(len(batch), max_caption_len, dtype=torch.long)
    lengths = torch.zeros(len(batch), dtype=torch.int)
    for i, (image, tokens) in enumerate(batch):
        images[i] = image
        row = [1] + list(tokens) + [2]
        captions[i, :len(row)] = torch.tensor(row, dtype=torch.long)
        lengths[i] = len(row)
    return images, captions, lengths, max_caption_len


def synthetic_batch(batch: int, vocab_size: int, image_size: int, caption_len: int, seed: int = 1008, device=None,
                    with_images: bool = True):
    """One fixed batch straight on the device (bench / smoke): (images|None, captions, lengths, L)."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch, 3, image_size, image_size, generator=g) if with_images else None
    body = torch.randint(4, vocab_size, (batch, caption_len - 2), generator=g)
    captions = torch.cat([torch.ones(batch, 1, dtype=torch.long), body, torch.full((batch, 1), 2, dtype=torch.long)], 1)
    lengths = torch.full((batch,), caption_len, dtype=torch.int)
    if device is not None:
        images = images.to(device) if images is not None else None
        captions, lengths = captions.to(device), lengths.to(device)
    return images, captions, lengths, caption_len
