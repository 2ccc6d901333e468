"""Kernel-name classification shared by the PMC tools (tools only)."""
import re

TRUNK_CONV_LAUNCHES = 53      # convolutions of one ResNet-50 trunk pass (oracle/cpu_encoder.layer_specs)


def is_trunk_conv(name: str) -> bool:
    """True for the kernels that run the trunk's convolutions: the dedicated conv kernels, and tile8_kernel<T, BN, EPI, CONV, ...> with
    CONV = true only -- the other tile8 instantiations are the discriminator's highway products (mangled: ...tile8_kernelIDF16bLi128ELi1ELb0E...;
    demangled: tile8_kernel<float, 128, 0, false, ...>)."""
    if any(k in name for k in ("conv3x3_patch_kernel", "conv1x1_stream_kernel", "conv1x1_panel_kernel", "conv_stem_kernel", "conv3x3_s2_kernel",
                               "conv1x1_small_kernel", "conv_b2b_kernel", "conv1x1_pix_kernel")):
        return True
    if "tile8_kernel" not in name:
        return False
    m = re.search(r"tile8_kernelI\w+?Li\d+ELi\d+ELb([01])E", name)
    if m:
        return m.group(1) == "1"
    m = re.search(r"tile8_kernel<[^,]+, *\d+, *\d+, *(true|false)", name)
    return bool(m) and m.group(1) == "true"
