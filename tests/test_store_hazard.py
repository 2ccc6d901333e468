"""Build-machine check (no GPU): the kernels that store 16 bytes per lane through buffer descriptors carry no "wide store, next
instruction overwrites its data registers" site in their gfx950 assembly (hipcc 7.2 leaves one behind a buffer
store with an SGPR offset, and on gfx950 the overwrite reaches memory).  tools/store_hazard_scan.py scans every source."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_buffer_store_kernels_have_no_store_data_hazard_site():
    spec = importlib.util.spec_from_file_location("store_hazard_scan", os.path.join(ROOT, "tools", "store_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["conv_b2b.hip", "conv1x1_pix.hip"]) == 0
