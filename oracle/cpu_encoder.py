"""CPU restatement of the image trunk of the reference's Encoder (src/generator.py:9-25):
``nn.Sequential(*list(torchvision.models.resnet18().children())[:-1])`` run under no_grad with every
BatchNorm in train mode (batch statistics; ``gen.train()``, src/training.py:216).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: torchvision is not part of the
reference tree and no version is pinned, so this is the published ResNet-18 / ResNet-50 (v1.5)
architecture restated in functional torch; the HIP trunk is checked against THIS, not against
reference outputs.  Parameter names follow the reference's state-dict (``encoder.resnet.<idx>...``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2), (64, 128, 256, 512), 1),
    "resnet50": ("bottleneck", (3, 4, 6, 3), (64, 128, 256, 512), 4),
}
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def out_features(arch: str) -> int:
    _, _, widths, exp = ARCHS[arch]
    return widths[-1] * exp


def layer_specs(arch: str) -> List[Tuple[str, int, int, int, int, int]]:
    """[(name, cin, cout, k, stride, pad)] for every conv, in execution order; BN of conv X is named by
    replacing 'conv' with 'bn' ('0'->'1', 'downsample.0'->'downsample.1')."""
    kind, counts, widths, exp = ARCHS[arch]
    specs = [("0", 3, 64, 7, 2, 3)]
    cin = 64
    for si, (n, w) in enumerate(zip(counts, widths)):
        for bi in range(n):
            stride = (1 if si == 0 else 2) if bi == 0 else 1
            p = f"{4 + si}.{bi}."
            cout = w * exp
            if kind == "basic":
                specs += [(p + "conv1", cin, w, 3, stride, 1), (p + "conv2", w, w, 3, 1, 1)]
            else:
                specs += [(p + "conv1", cin, w, 1, 1, 0), (p + "conv2", w, w, 3, stride, 1), (p + "conv3", w, cout, 1, 1, 0)]
            if stride != 1 or cin != cout:
                specs.append((p + "downsample.0", cin, cout, 1, stride, 0))
            cin = cout
    return specs


def bn_name(conv_name: str) -> str:
    if conv_name == "0":
        return "1"
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    return conv_name.replace("conv", "bn")


def make_trunk_params(arch: str, gen: torch.Generator, prefix: str = "encoder.resnet.", init: str = "uniform") -> Dict[str, torch.Tensor]:
    """init="uniform": U(-0.05, 0.05) for conv weights AND BatchNorm affine parameters (Generator.init_params, generator.py:116-123:
    what the reference trains from).  init="kaiming": the published ResNet initialisation (He et al. 2015: N(0, 2 / fan_out) conv
    weights, gamma = 1, beta = 0) -- what a trunk looks like before init_params overwrites it, and the shape of any trained trunk's
    statistics; used to show which part of the bf16 error budget is an artefact of the degenerate uniform init."""
    p: Dict[str, torch.Tensor] = {}
    for name, cin, cout, k, _s, _p in layer_specs(arch):
        b = prefix + bn_name(name)
        if init == "uniform":
            p[prefix + name + ".weight"] = torch.empty(cout, cin, k, k).uniform_(-0.05, 0.05, generator=gen)
            p[b + ".weight"] = torch.empty(cout).uniform_(-0.05, 0.05, generator=gen)
            p[b + ".bias"] = torch.empty(cout).uniform_(-0.05, 0.05, generator=gen)
        elif init == "kaiming":
            p[prefix + name + ".weight"] = torch.empty(cout, cin, k, k).normal_(0.0, (2.0 / (cout * k * k)) ** 0.5, generator=gen)
            p[b + ".weight"] = torch.ones(cout)
            p[b + ".bias"] = torch.zeros(cout)
        else:
            raise ValueError(f"unknown init {init!r}")
    return p


def _r16(x: torch.Tensor) -> torch.Tensor:
    """Round to bfloat16 and back (what storing a tensor in the bf16 compute dtype does to it)."""
    return x.bfloat16().float()


def _bn(x, tp, name, training, running, stats_out, x_stored=None):
    """x: the convolution output the statistics are taken from (f32 accumulators); x_stored: the tensor the affine map is applied to
    (the same, or its bf16-rounded copy in the storage-emulating mode)."""
    g, b = tp[name + ".weight"], tp[name + ".bias"]
    if training:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        if stats_out is not None:
            stats_out[name] = (mean, var)
        if running is not None:
            n = x.numel() / x.shape[1]
            running[name + ".running_mean"] = (1 - BN_MOMENTUM) * running[name + ".running_mean"] + BN_MOMENTUM * mean
            running[name + ".running_var"] = (1 - BN_MOMENTUM) * running[name + ".running_var"] + BN_MOMENTUM * var * n / max(n - 1, 1)
    else:
        mean, var = running[name + ".running_mean"], running[name + ".running_var"]
    scale = g / torch.sqrt(var + BN_EPS)
    xs = x if x_stored is None else x_stored
    return xs * scale[None, :, None, None] + (b - mean * scale)[None, :, None, None]


def trunk_forward(tp: Dict[str, torch.Tensor], images: torch.Tensor, arch: str, training: bool = True,
                  running: Optional[Dict[str, torch.Tensor]] = None, prefix: str = "encoder.resnet.",
                  stats_out: Optional[dict] = None, taps: Optional[dict] = None, emulate_bf16: bool = False,
                  unstored: Optional[set] = None) -> torch.Tensor:
    """images [N,3,S,S] -> [N, out_features] (global average pool squeezed).

    ``emulate_bf16``: the same network with every tensor that the bf16 compute mode STORES rounded to bfloat16 at the point where it
    is stored -- the packed image, the convolution weights, every raw convolution output (the BatchNorm statistics are still taken
    from the f32 accumulators, as the kernels' epilogues do), every normalised + ReLU'd activation fed to the next convolution, every
    block output, the pooled feature -- and everything else (products, accumulation, the affine maps, the residual sums) in f32.  It
    separates what bf16 STORAGE does to this network (a property of the data: a pre-BatchNorm tensor whose per-channel |mean| is
    many times its spread loses that factor in relative precision when the mean is subtracted) from what the kernels do.
    ``unstored``: names of convolutions ("4.0.conv3", ...) whose raw output the kernels never store -- it is normalised from the f32
    accumulators (the conv3 -> block output -> next conv1 launch, TrunkPlan.unstored_convs()): no rounding there."""
    kind, counts, widths, exp = ARCHS[arch]
    P = prefix
    r = _r16 if emulate_bf16 else (lambda t: t)

    def conv(x, name, stride, pad):
        return F.conv2d(x, r(tp[P + name + ".weight"]), None, stride, pad)

    def bn(y, name, conv_name=None):
        keep = unstored is not None and conv_name in unstored
        return _bn(y, tp, P + name, training, running, stats_out, x_stored=r(y) if (emulate_bf16 and not keep) else None)

    x = torch.relu(bn(conv(r(images), "0", 2, 3), "1"))
    x = r(F.max_pool2d(x, 3, 2, 1))
    if taps is not None:
        taps["stem"] = x
    cin = 64
    for si, (n, w) in enumerate(zip(counts, widths)):
        for bi in range(n):
            stride = (1 if si == 0 else 2) if bi == 0 else 1
            p = f"{4 + si}.{bi}."
            cout = w * exp
            idt = x
            if kind == "basic":
                y = r(torch.relu(bn(conv(x, p + "conv1", stride, 1), p + "bn1")))
                y = bn(conv(y, p + "conv2", 1, 1), p + "bn2")
            else:
                y = r(torch.relu(bn(conv(x, p + "conv1", 1, 0), p + "bn1")))
                y = r(torch.relu(bn(conv(y, p + "conv2", stride, 1), p + "bn2")))
                y = bn(conv(y, p + "conv3", 1, 0), p + "bn3", p + "conv3")
            if stride != 1 or cin != cout:
                idt = bn(conv(x, p + "downsample.0", stride, 0), p + "downsample.1")
            x = r(torch.relu(y + idt))
            cin = cout
        if taps is not None:
            taps[f"stage{si}"] = x
    return r(x.mean((2, 3)))


def macs(arch: str, S: int) -> int:
    """Multiply-accumulates of the trunk's convolutions for one SxS image."""
    total = 0
    h = S
    sizes = {}
    x = (S + 6 - 7) // 2 + 1
    total += x * x * 64 * 3 * 49
    x = (x + 2 - 3) // 2 + 1
    kind, counts, widths, exp = ARCHS[arch]
    cin = 64
    for si, (n, w) in enumerate(zip(counts, widths)):
        for bi in range(n):
            stride = (1 if si == 0 else 2) if bi == 0 else 1
            xo = (x - 1) // stride + 1
            cout = w * exp
            if kind == "basic":
                total += xo * xo * w * cin * 9 + xo * xo * w * w * 9
            else:
                total += x * x * w * cin + xo * xo * w * w * 9 + xo * xo * cout * w
            if stride != 1 or cin != cout:
                total += xo * xo * cout * cin
            x, cin = xo, cout
    return total
