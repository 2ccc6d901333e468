"""Host wrappers of the encoder kernels (image trunk + Linear/BatchNorm1d head).  Filled in with the
trunk kernels; until then the conditional path raises instead of computing anywhere else."""
from __future__ import annotations


class TrunkPlan:
    def __init__(self, trunk, dtype):
        self.dtype = dtype
        raise NotImplementedError("encoder trunk kernels are not built yet: use --conditional-gan 0")


def head_fwd(*a, **k):
    raise NotImplementedError("encoder head kernels are not built yet: use --conditional-gan 0")


def head_bwd(*a, **k):
    raise NotImplementedError("encoder head kernels are not built yet: use --conditional-gan 0")
