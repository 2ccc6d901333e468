"""MI355X-native adversarial image-captioning train step.

Host side (this package, plain Python + PyTorch-ROCm for device memory, streams and
``torch.distributed``) mirrors the reference's module API (``args``, ``tasks``,
``generator``, ``discriminator``, ``utils``, ``training``); all compute on the hot path
runs in hand-written HIP kernels for gfx950 behind the C ABI of ``include/gicap.h``
(``csrc/`` -> ``libgicap.so``, bound with ctypes in ``_lib.py``).  There is no CPU or
eager-PyTorch fallback: importing a compute module without the built library, or
running it on a tensor that is not on an AMD GPU, raises.
"""
__version__ = "0.1.0"
