"""ResNet-18 / ResNet-50 shaped image trunk (forward only, BatchNorm on batch statistics) and the
trainable encoder head, replacing ``torchvision.models.resnet18`` minus ``fc`` + Linear +
BatchNorm1d of the reference's Encoder (src/generator.py:9-25).

PARITY UNPINNED for the trunk: torchvision is not part of the reference tree and no version is
pinned, so the architecture below is the published ResNet (He et al. 2015; v1.5 stride placement
for the bottleneck) restated by this build; its CPU oracle is oracle/cpu_encoder.py.

Parameter / buffer names reproduce ``nn.Sequential(*list(resnet.children())[:-1])`` so reference
checkpoints load: ``0`` conv1, ``1`` bn1, ``4``..``7`` layer1..layer4 with ``<block>.convK``,
``<block>.bnK``, ``<block>.downsample.0`` (conv) and ``<block>.downsample.1`` (bn).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

ARCHS = {
    # name: (block kind, blocks per stage, stage widths, expansion)
    "resnet18": ("basic", (2, 2, 2, 2), (64, 128, 256, 512), 1),
    "resnet50": ("bottleneck", (3, 4, 6, 3), (64, 128, 256, 512), 4),
}


class _ConvParams(nn.Module):
    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        self.weight = nn.Parameter(w)


class _BNParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, 1e-5, 0.1
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Block(nn.Module):
    def __init__(self, kind, cin, width, stride, expansion):
        super().__init__()
        self.kind, self.stride = kind, stride
        cout = width * expansion
        if kind == "basic":
            self.conv1, self.bn1 = _ConvParams(cin, width, 3, stride, 1), _BNParams(width)
            self.conv2, self.bn2 = _ConvParams(width, width, 3, 1, 1), _BNParams(width)
        else:
            self.conv1, self.bn1 = _ConvParams(cin, width, 1, 1, 0), _BNParams(width)
            self.conv2, self.bn2 = _ConvParams(width, width, 3, stride, 1), _BNParams(width)
            self.conv3, self.bn3 = _ConvParams(width, cout, 1, 1, 0), _BNParams(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_ConvParams(cin, cout, 1, stride, 0), _BNParams(cout))
        self.cout = cout


class ResNetTrunk(nn.Module):
    def __init__(self, arch: str = "resnet18"):
        super().__init__()
        if arch not in ARCHS:
            raise ValueError(f"unknown encoder arch {arch!r}")
        kind, counts, widths, exp = ARCHS[arch]
        self.arch = arch
        mods: Dict[str, nn.Module] = {"0": _ConvParams(3, 64, 7, 2, 3), "1": _BNParams(64)}
        cin = 64
        for si, (n, w) in enumerate(zip(counts, widths)):
            blocks = []
            for bi in range(n):
                blk = _Block(kind, cin, w, (1 if si == 0 else 2) if bi == 0 else 1, exp)
                cin = blk.cout
                blocks.append(blk)
            mods[str(4 + si)] = nn.Sequential(*blocks)
        for k, v in mods.items():              # indices 2 (relu), 3 (maxpool), 8 (avgpool) hold no state
            self.add_module(k, v)
        self.out_features = cin
        self._plan = None
        # bumped whenever tensors of this module may have MOVED (Module._apply: .to() / .float() / .cuda(); load_state_dict): the
        # execution plan keys the pointers baked into its replayed hipGraph and its device tables on it instead of calling
        # data_ptr() on ~320 tensors per forward (0.1 ms of host time per step).  Values changing in place need no such signal:
        # BatchNorm affine / running tensors are read through their pointers, conv weights carry their own _version.
        self._ptr_epoch = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._bump_ptr_epoch())

    def _bump_ptr_epoch(self) -> None:
        self._ptr_epoch += 1

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._ptr_epoch += 1
        return out

    def stages(self) -> List[nn.Sequential]:
        return [getattr(self, str(i)) for i in (4, 5, 6, 7)]

    def forward(self, images: torch.Tensor, dtype: int, training=None) -> torch.Tensor:
        from . import encoder_engine
        if self._plan is None or self._plan.dtype != dtype:
            self._plan = encoder_engine.TrunkPlan(self, dtype)
        return self._plan.forward(images, self.training if training is None else training)

    def state_dict(self, *a, **k):
        if self._plan is not None:
            self._plan.sync_counters()
        return super().state_dict(*a, **k)


def encoder_head_fwd(dtype, feat, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps):
    from . import encoder_engine
    return encoder_engine.head_fwd(dtype, feat, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps)


def encoder_head_bwd(dtype, saved, weight, gamma, d_out, grads=None):
    from . import encoder_engine
    return encoder_engine.head_bwd(dtype, saved, weight, gamma, d_out, grads=grads)
