"""CPU-side checks of the drop-in boundary: libgicap.so loads and exports every symbol that
include/gicap.h declares; argument validation returns status codes (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gicap.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gic_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_bound_symbols():
    from gan_image_captioning_amd import _lib
    assert set(declared_symbols()) == set(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from gan_image_captioning_amd import _lib
    lib = ctypes.CDLL(_lib.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), f"libgicap.so does not export {name}"
    assert _lib.load().gic_abi_version() == _lib.ABI_VERSION == 4


def test_argument_validation_returns_status_not_crash():
    from gan_image_captioning_amd import _lib
    lib = _lib.load()
    assert lib.gic_gemm(None, None, None, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, None, 0, 1.0, None) == -1
    assert b"null operand" in lib.gic_last_error()
    with pytest.raises(ValueError):
        _lib.check(lib.gic_gan_losses(99, None, None, None, 0, None, None, None, None, None, None, None), "gan_losses")
    d = _lib.DecoderDims(4, 5, 50, 8, 16, 9, 0)          # NL out of range
    assert lib.gic_decoder_prepare(ctypes.byref(d), None, None, None) == -1
    assert b"gen_num_layers" in lib.gic_last_error()


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors of the C structs: pointer arrays sized by GIC_MAX_LAYERS / GIC_MAX_CONVS."""
    from gan_image_captioning_amd import _lib
    P = ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.DecoderDims) == 7 * 4
    assert ctypes.sizeof(_lib.DecoderParams) == (3 + 4 * _lib.MAX_LAYERS) * P
    assert ctypes.sizeof(_lib.DecoderGrads) == (4 + 4 * _lib.MAX_LAYERS) * P
    assert ctypes.sizeof(_lib.DecoderState) == (4 + 3 * _lib.MAX_LAYERS) * P
    assert ctypes.sizeof(_lib.DecoderSampleOpts) == 10 * P         # 4 pointers, int32 (padded), pointer, int32 (padded), 2 pointers, int32 (padded)
    assert ctypes.sizeof(_lib.StepScalars) == 8 + 8 * _lib.STEP_SEEDS
    assert ctypes.sizeof(_lib.DiscDims) == (6 + 2 * _lib.MAX_CONVS + 3) * 4 + 4       # + float drop_p
    assert ctypes.sizeof(_lib.DiscParams) == (7 + 2 * _lib.MAX_CONVS) * P


def test_cpu_tensors_are_refused():
    import torch
    from gan_image_captioning_amd import engine
    from gan_image_captioning_amd._lib import GicError
    with pytest.raises(GicError):
        engine.embedding_fwd(torch.zeros(4, 4), torch.zeros(2, dtype=torch.long))


def test_workspace_size_queries_match_the_host_allocations():
    """gic_*_bytes (host-only) against the buffers the Python engines allocate (on the meta device: shapes only)."""
    import torch
    from gan_image_captioning_amd import _lib, engine
    lib = _lib.load()
    ML = _lib.MAX_LAYERS
    for dt in (0, 1):
        B, Lc, V, E, H, NL = 6, 9, 70, 24, 32, 2
        dec = engine.DecoderEngine(V, E, H, NL, dt)
        st, ws = dec.alloc_state(B, Lc, "meta"), dec.alloc_bwd_ws(B, Lc, "meta")
        nb = lambda t: t.numel() * t.element_size()
        d = dec.dims(B, Lc)
        out = (ctypes.c_uint64 * (3 * ML + 4))()
        assert lib.gic_decoder_state_bytes(ctypes.byref(d), out) == 0
        want = [nb(t) for t in st["xh"]] + [0] * (ML - NL) + [nb(t) for t in st["gates"]] + [0] * (ML - NL) + \
               [nb(t) for t in st["c"]] + [0] * (ML - NL) + [nb(st["hout"]), nb(st["logits"]), nb(st["gpre"]), nb(st["part"])]
        assert list(out) == want
        out = (ctypes.c_uint64 * (2 + 3 * ML))()
        assert lib.gic_decoder_bwd_ws_bytes(ctypes.byref(d), out) == 0
        want = [nb(ws["dlogits"]), nb(ws["dhout"])] + [nb(t) for t in ws["dgates"]] + [0] * (ML - NL) + \
               [nb(t) for t in ws["dxh"]] + [0] * (ML - NL) + [nb(t) for t in ws["dc"]] + [0] * (ML - NL)
        assert list(out) == want
        den = engine.DiscEngine(V, 64, 64, [3, 4, 5], [300, 300, 300], dt)
        st, ws = den.alloc_state(B, Lc, "meta"), den.alloc_bwd_ws(B, Lc, "meta")
        dd = den.dims(B, Lc)
        out = (ctypes.c_uint64 * 7)()
        assert lib.gic_disc_state_bytes(ctypes.byref(dd), out) == 0
        assert list(out) == [nb(st[k]) for k in ("emb", "pooled", "argmax", "hpre", "keep", "ydrop", "feat")]
        out = (ctypes.c_uint64 * 5)()
        assert lib.gic_disc_bwd_ws_bytes(ctypes.byref(dd), out) == 0
        assert list(out) == [nb(ws[k]) for k in ("dfeat", "dh", "dydrop", "dpooled", "demb")]
    assert lib.gic_decoder_state_bytes(None, None) == -1


def test_fused_rollout_query_drives_the_rollout_buffers():
    """gic_decoder_fused_rollout_rows (host-only) is the library's own path selection: shapes the fused step kernels decline
    (V % 4, E % 8, H % 8) or batches beyond the row limit get the generic products' scratch, never a NULL the C side then rejects."""
    from gan_image_captioning_amd import engine
    for dt in (0, 1):
        ok = engine.DecoderEngine(64, 32, 64, 1, dt)
        assert ok.fused_rollout_rows() == 512
        st = ok.alloc_rollout_state(8, 5, "meta")
        assert st["part"] is not None and st["logits"] is None and st["gpre"] is None
        st = ok.alloc_rollout_state(513, 5, "meta")
        assert st["part"] is None and st["logits"] is not None and st["gpre"] is not None
        for V, E, H in ((63, 32, 64), (64, 30, 64), (64, 32, 36), (9487, 300, 512)):
            odd = engine.DecoderEngine(V, E, H, 1, dt)
            assert odd.fused_rollout_rows() == 0
            st = odd.alloc_rollout_state(8, 5, "meta")
            assert st["part"] is None and tuple(st["logits"].shape) == (8, V) and tuple(st["gpre"].shape) == (8, 4 * H)
