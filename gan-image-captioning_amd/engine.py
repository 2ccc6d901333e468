"""Tensor-level host wrappers around the C ABI: buffer planning + ctypes calls.

``DecoderEngine`` / ``DiscEngine`` own nothing but the derived compute-dtype weight
images; every other buffer (outputs, saved-for-backward state, workspaces, grads) is a
torch tensor allocated here through PyTorch's caching allocator and handed to the
library as a raw device pointer.  All launches go to the current PyTorch HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L

TORCH_DTYPE = {L.F32: torch.float32, L.BF16: torch.bfloat16}
DTYPE_BY_NAME = {"fp32": L.F32, "f32": L.F32, "float32": L.F32, "bf16": L.BF16, "bfloat16": L.BF16}

_param_epoch = 0   # bumped by optimizers that update weights through raw pointers


def bump_param_epoch() -> None:
    global _param_epoch
    _param_epoch += 1


class capture_guard:
    """Bracket for a hipGraph capture region: no cyclic garbage collection may run inside it.

    A generational collection that fires between two captured launches finalises whatever HIP-owning garbage the process has
    accumulated (stale ``CUDAGraph``s with their private pools, streams, events of dropped plans): a destructor that frees device
    memory or destroys a graph while a capture is open throws inside a C++ destructor -> ``std::terminate`` (the rc=134 abort of
    round 2).  torch >= 2.6 no longer collects before a capture (``force_cudagraph_gc`` is off), so this
    does: collect BEFORE the region (the garbage dies outside it), disable the collector inside, restore it afterwards."""

    def __enter__(self):
        import gc
        self._gc = gc
        self._was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self._was_enabled:
            self._gc.enable()
        return False


class on_stream:
    """``with torch.cuda.stream(s)`` without its per-entry device query (torch's StreamContext asks the runtime for the device count
    on every construction: ~20 us, a dozen times per step).  Single device per process (one process per GPU)."""
    __slots__ = ("s", "prev")

    def __init__(self, stream):
        self.s = stream

    def __enter__(self):
        self.prev = torch.cuda.current_stream(self.s.device)
        torch.cuda.set_stream(self.s)
        return self.s

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


def parse_dtype(x) -> int:
    if isinstance(x, int):
        return x
    try:
        return DTYPE_BY_NAME[str(x).lower()]
    except KeyError:
        raise ValueError(f"unknown compute dtype {x!r} (use 'bf16' or 'fp32')")


def require_gpu(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise L.GicError("tensor is not on a GPU: the hot path runs only through the HIP library "
                             "(libgicap.so) on an AMD GPU and has no CPU fallback")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _arr(ctype, n, values):
    a = (ctype * n)()
    for i, v in enumerate(values):
        a[i] = v
    return a


def _key(params: Sequence[torch.Tensor]):
    return tuple((p.data_ptr(), p._version) for p in params) + (_param_epoch,)


class StepScalarsBuffer:
    """Device-resident gic_step_scalars (gicap.h): the decoder's temperature and the seeds of the step's device noise streams, read
    by the kernels from device memory so that a captured step graph replays with unchanged launch arguments."""

    def __init__(self, device):
        self.buf = torch.zeros(C.sizeof(L.StepScalars), dtype=torch.uint8, device=device)
        self.ptr = self.buf.data_ptr()

    def set(self, temperature: float, seeds) -> None:
        """Enqueue the update on the current stream (values travel as kernel arguments)."""
        v = L.StepScalars()
        v.temperature = float(temperature)
        for i, sd in enumerate(seeds):
            v.seed[i] = int(sd) & (2 ** 64 - 1)
        L.check(L.load().gic_step_scalars_set(self.ptr, C.byref(v), stream_ptr()), "gic_step_scalars_set")


# ------------------------------------------------------------------------------------------ generic ops
def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=True, b_kc=True, bias=None, accumulate=False, alpha=1.0):
    require_gpu(A, B, Cout)
    in_dt = L.F32 if A.dtype == torch.float32 else L.BF16
    out_dt = L.F32 if Cout.dtype == torch.float32 else L.BF16
    L.check(L.load().gic_gemm(ptr(A), ptr(B), ptr(Cout), M, N, K, lda, ldb, ldc, int(a_kc), int(b_kc), in_dt, out_dt,
                              ptr(bias), int(accumulate), float(alpha), stream_ptr()), "gic_gemm")
    return Cout


def cast2d(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, lds: int, ldd: int):
    require_gpu(src, dst)
    sd = L.F32 if src.dtype == torch.float32 else L.BF16
    dd = L.F32 if dst.dtype == torch.float32 else L.BF16
    L.check(L.load().gic_cast2d(ptr(src), sd, lds, ptr(dst), dd, ldd, rows, cols, stream_ptr()), "gic_cast2d")
    return dst


def embedding_fwd(weight: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    require_gpu(weight, ids)
    ids = ids.contiguous()
    out = torch.empty(ids.numel(), weight.shape[1], device=weight.device, dtype=torch.float32)
    L.check(L.load().gic_embedding_fwd(ptr(weight), ptr(ids), ptr(out), ids.numel(), weight.shape[0], weight.shape[1],
                                       stream_ptr()), "gic_embedding_fwd")
    return out.view(*ids.shape, weight.shape[1])


def embedding_bwd(d_out: torch.Tensor, ids: torch.Tensor, V: int, d_weight: Optional[torch.Tensor] = None,
                  zero_first: bool = True) -> torch.Tensor:
    require_gpu(d_out, ids)
    E = d_out.shape[-1]
    d_out = d_out.contiguous()
    ids = ids.contiguous()
    if d_weight is None:
        d_weight = torch.empty(V, E, device=d_out.device, dtype=torch.float32)
    L.check(L.load().gic_embedding_bwd(ptr(d_out), ptr(ids), ptr(d_weight), ids.numel(), V, E, int(zero_first), stream_ptr()),
            "gic_embedding_bwd")
    return d_weight


def gan_losses(loss_type: str, d_real, d_fake, g_out, want_grads: bool = True):
    """Returns (losses[2] device tensor: [g_loss, d_loss], grads dict or None)."""
    if loss_type not in L.LOSS_TYPES:
        raise NotImplementedError("Divergence '%s' is not implemented" % loss_type)   # utils.py:50-51
    require_gpu(d_real, d_fake, g_out)
    d_real, d_fake, g_out = d_real.contiguous(), d_fake.contiguous(), g_out.contiguous()
    n = d_real.numel()
    losses = torch.empty(2, device=d_real.device, dtype=torch.float32)
    g = None
    if want_grads:
        buf = torch.empty(5, n, device=d_real.device, dtype=torch.float32)
        g = {"dd_real": buf[0], "dd_fake": buf[1], "dg_out": buf[2], "dg_real": buf[3], "dg_fake": buf[4],
             "dd_real_fake": buf[:2].view(-1)}          # [real ; fake] contiguous: one backward over both passes
    L.check(L.load().gic_gan_losses(L.LOSS_TYPES[loss_type], ptr(d_real), ptr(d_fake), ptr(g_out), n, ptr(losses),
                                    ptr(g["dd_real"]) if g else None, ptr(g["dd_fake"]) if g else None,
                                    ptr(g["dg_out"]) if g else None, ptr(g["dg_real"]) if g else None,
                                    ptr(g["dg_fake"]) if g else None, stream_ptr()), "gic_gan_losses")
    return losses, g


def xent(logits: torch.Tensor, targets: torch.Tensor, want_grad: bool = True, row_weight: Optional[torch.Tensor] = None):
    """CrossEntropyLoss(mean over all rows). logits [rows,V] (f32/bf16, contiguous). Returns (loss[1], d_logits|None).
    ``row_weight`` f32 [rows]: weighted form (policy-gradient loss, gicap.h)."""
    require_gpu(logits, targets, row_weight)
    rows, V = logits.shape
    dt = L.F32 if logits.dtype == torch.float32 else L.BF16
    buf = torch.empty(1 + rows, device=logits.device, dtype=torch.float32)
    dl = torch.empty_like(logits) if want_grad else None
    if row_weight is not None:
        row_weight = row_weight.contiguous().float()
        if row_weight.numel() != rows:
            raise ValueError("row_weight must hold one weight per row")
    L.check(L.load().gic_xent(ptr(logits), dt, rows, V, ptr(targets.contiguous()), ptr(buf), ptr(dl), ptr(row_weight), stream_ptr()),
            "gic_xent")
    return buf[:1], dl


def rollout_rewards(mc_logits: Optional[torch.Tensor], full_logits: torch.Tensor, B: int, Lc: int, N: int, R: int) -> torch.Tensor:
    """gic_rollout_rewards: f32 [B, L] Monte-Carlo rewards from D's logits on the roll-outs (gicap.h)."""
    require_gpu(mc_logits, full_logits)
    out = torch.empty(B, Lc, device=full_logits.device, dtype=torch.float32)
    L.check(L.load().gic_rollout_rewards(ptr(mc_logits), ptr(full_logits.contiguous()), ptr(out), B, Lc, N, R, stream_ptr()),
            "gic_rollout_rewards")
    return out


# ------------------------------------------------------------------------------------------ decoder
class DecoderEngine:
    """Decoder.sample forward/backward (reference src/generator.py:55-96) on the HIP library."""

    def __init__(self, vocab: int, embed: int, hidden: int, layers: int, dtype: int):
        if not 1 <= layers <= L.MAX_LAYERS:
            raise ValueError(f"gen_num_layers must be in 1..{L.MAX_LAYERS}")
        self.V, self.E, self.H, self.NL, self.dt = vocab, embed, hidden, layers, dtype
        self.act = TORCH_DTYPE[dtype]
        self._shadow: Optional[Dict[str, object]] = None
        self._shadow_key = None

    def din(self, l: int) -> int:
        return self.E if l == 0 else self.H

    def ldx(self, l: int) -> int:
        return self.din(l) + self.H

    def dims(self, B: int, Lc: int) -> L.DecoderDims:
        return L.DecoderDims(B, Lc, self.V, self.E, self.H, self.NL, self.dt)

    # params: [embed, (w_ih, w_hh, b_ih, b_hh) * NL, w_out, b_out]
    def _pstruct(self, params, cls=L.DecoderParams, extra=None):
        nl = self.NL
        s = cls()
        s.embed = ptr(params[0])
        s.w_ih = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(params[1 + 4 * l]) for l in range(nl)])
        s.w_hh = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(params[2 + 4 * l]) for l in range(nl)])
        s.b_ih = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(params[3 + 4 * l]) for l in range(nl)])
        s.b_hh = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(params[4 + 4 * l]) for l in range(nl)])
        s.w_out = ptr(params[1 + 4 * nl])
        s.b_out = ptr(params[2 + 4 * nl])
        if extra is not None:
            s.features = ptr(extra)
        return s

    def check_params(self, params) -> None:
        if len(params) != 3 + 4 * self.NL:
            raise ValueError("decoder expects embed, 4 tensors per LSTM layer, linear weight and bias")
        require_gpu(*params)
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("decoder parameters must be contiguous float32 (master weights)")

    def prepare(self, params) -> Dict[str, object]:
        """Refresh the compute-dtype weight images if any master weight changed."""
        key = _key(params)
        if self._shadow is not None and key == self._shadow_key:
            return self._shadow
        dev = params[0].device
        if self._shadow is None or self._shadow["wcat"][0].device != dev:
            self._shadow = {
                "wcat": [torch.empty(4 * self.H, self.ldx(l), device=dev, dtype=self.act) for l in range(self.NL)],
                "bsum": [torch.empty(4 * self.H, device=dev, dtype=torch.float32) for l in range(self.NL)],
                "wout": None if self.dt == L.F32 else torch.empty(self.V, self.H, device=dev, dtype=self.act),
                # Wcat^T: the k-contiguous weight operand of the BPTT products d[x|h] = d_gates Wcat (fused BPTT step kernel)
                "wcat_t": [torch.empty(self.ldx(l), 4 * self.H, device=dev, dtype=self.act) for l in range(self.NL)],
            }
        sh = self._shadow
        s = self._shadow_struct(params)
        d = self.dims(1, 1)
        L.check(L.load().gic_decoder_prepare(C.byref(d), C.byref(self._pstruct(params)), C.byref(s), stream_ptr()),
                "gic_decoder_prepare")
        self._shadow_key = key
        return sh

    def _shadow_struct(self, params) -> L.DecoderShadow:
        sh = self._shadow
        s = L.DecoderShadow()
        s.wcat = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in sh["wcat"]])
        s.bsum = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in sh["bsum"]])
        s.wout = ptr(params[1 + 4 * self.NL]) if sh["wout"] is None else ptr(sh["wout"])
        if sh["wcat_t"] is not None:
            s.wcat_t = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in sh["wcat_t"]])
        return s

    def alloc_state(self, B: int, Lc: int, dev) -> Dict[str, object]:
        f32 = torch.float32
        return {
            "xh": [torch.empty(Lc + 1, B, self.ldx(l), device=dev, dtype=self.act) for l in range(self.NL)],
            "gates": [torch.empty(Lc, B, 4 * self.H, device=dev, dtype=f32) for _ in range(self.NL)],
            "c": [torch.empty(Lc + 1, B, self.H, device=dev, dtype=f32) for _ in range(self.NL)],
            "hout": torch.empty(B, Lc, self.H, device=dev, dtype=self.act),
            "logits": torch.empty(B, self.V, device=dev, dtype=f32),
            "gpre": torch.empty(B, 4 * self.H, device=dev, dtype=f32),
            # per-tile softmax partials of the fused step kernels ([3][L][B][ceil(V/64)], decoder_step.h)
            "part": torch.empty(self.part_floats(B, Lc), device=dev, dtype=f32),
        }

    def part_floats(self, B: int, Lc: int) -> int:
        """Floats of the fused step kernels' scratch: [2][L][B][ceil(V/64)] tile partials + [L][B] 64-bit argmax keys + two reserved
        words (gicap.h; = gic_decoder_state_bytes' figure)."""
        return 2 * Lc * B * ((self.V + 63) // 64) + 2 * Lc * B + 4

    def fused_rollout_rows(self) -> int:
        """Largest batch the fused step kernels take for this decoder's shapes; 0: they decline them (gic_decoder_fused_rollout_rows:
        the library's own path selection, so buffer planning here cannot disagree with it)."""
        out = C.c_int32(0)
        L.check(L.load().gic_decoder_fused_rollout_rows(C.byref(self.dims(1, 1)), C.byref(out)), "gic_decoder_fused_rollout_rows")
        return int(out.value)

    def alloc_rollout_state(self, B: int, Lc: int, dev) -> Dict[str, object]:
        """State of an inference roll-out (``no_state``): recurrent buffers only, nothing saved for a backward pass."""
        f32 = torch.float32
        fused = B <= self.fused_rollout_rows()
        return {
            "xh": [torch.empty(Lc + 1, B, self.ldx(l), device=dev, dtype=self.act) for l in range(self.NL)],
            "gates": [None] * self.NL,
            "c": [torch.empty(Lc + 1, B, self.H, device=dev, dtype=f32) for _ in range(self.NL)],
            "hout": None,
            # where the fused step kernels decline (row limit, V % 4, E % 8, H % 8) the roll-out runs as generic products: their scratch
            "logits": None if fused else torch.empty(B, self.V, device=dev, dtype=f32),
            "gpre": None if fused else torch.empty(B, 4 * self.H, device=dev, dtype=f32),
            "part": torch.empty(self.part_floats(B, Lc), device=dev, dtype=f32) if fused else None,
        }

    def _state_struct(self, st) -> L.DecoderState:
        s = L.DecoderState()
        s.xh = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in st["xh"]])
        s.gates = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in st["gates"]])
        s.c = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in st["c"]])
        s.hout, s.logits, s.gpre = ptr(st["hout"]), ptr(st["logits"]), ptr(st["gpre"])
        s.part = ptr(st.get("part"))
        return s

    def alloc_bwd_ws(self, B: int, Lc: int, dev) -> Dict[str, object]:
        f32 = torch.float32
        return {
            "dlogits": torch.empty(B, Lc, self.V, device=dev, dtype=self.act),
            "dhout": torch.empty(B, Lc, self.H, device=dev, dtype=f32),
            "dgates": [torch.empty(Lc, B, 4 * self.H, device=dev, dtype=self.act) for _ in range(self.NL)],
            "dxh": [torch.empty(Lc + 1, B, self.ldx(l), device=dev, dtype=f32) for l in range(self.NL)],
            "dc": [torch.empty(B, self.H, device=dev, dtype=f32) for _ in range(self.NL)],
        }

    def alloc_grads(self, params, B: int) -> List[torch.Tensor]:
        """[d_embed, per-layer grads..., d_w_out, d_b_out, d_features]"""
        g = [torch.empty_like(p) for p in params]
        g.append(torch.empty(B, self.E, device=params[0].device, dtype=torch.float32))
        return g

    def sample_fwd(self, params, features: torch.Tensor, Lc: int, temperature: float, pretrain: bool = False,
                   noise_u: Optional[torch.Tensor] = None, seed: int = 0, state=None, out=None, ids=None,
                   states=None, force_ids: Optional[torch.Tensor] = None, force_len: Optional[torch.Tensor] = None,
                   ids_only: bool = False, resume=None, dev_scalars=None, seed_slot: int = 0):
        """``dev_scalars`` (StepScalarsBuffer) / ``seed_slot``: temperature and seed are read from device memory (gicap.h
        gic_step_scalars; fused step kernels only).  ``states`` = (h0, c0), each f32 [NL, B, H] (generator.py:55,61).  ``force_ids`` int64 [B, L] (+ ``force_len`` int32 [B]):
        trajectory to follow (gicap.h).  ``ids_only``: inference roll-out, returns (None, ids, state) and saves nothing for backward.
        ``resume`` = (state of an earlier call, its batch size, active_rows list[L]): resumed roll-outs (gicap.h,
        gic_decoder_sample_opts.resume_from) -- rows sorted by prefix length, each starting at its prefix from that call's state."""
        self.check_params(params)
        require_gpu(features, noise_u, force_ids, force_len)
        B = features.shape[0]
        if features.shape != (B, self.E) or features.dtype != torch.float32:
            raise ValueError(f"features must be float32 [B,{self.E}], got {tuple(features.shape)} {features.dtype}")
        features = features.contiguous()
        dev = features.device
        if noise_u is not None:
            if tuple(noise_u.shape) != (Lc, B, self.V) or noise_u.dtype != torch.float32:
                raise ValueError(f"noise_u must be float32 [L={Lc},B={B},V={self.V}]")
            noise_u = noise_u.contiguous()
        self.prepare(params)
        if state is not None:
            st = state
        else:
            st = self.alloc_rollout_state(B, Lc, dev) if ids_only else self.alloc_state(B, Lc, dev)
        if not ids_only:
            out = out if out is not None else torch.empty(B, Lc, self.V, device=dev, dtype=self.act)
        ids = ids if ids is not None else torch.empty(B, Lc, device=dev, dtype=torch.int64)
        opts = None
        keep = []
        if states is not None or force_ids is not None or ids_only or dev_scalars is not None:
            opts = L.DecoderSampleOpts()
            if dev_scalars is not None:
                opts.dev_scalars, opts.seed_slot = dev_scalars.ptr, int(seed_slot)
            if states is not None:
                h0, c0 = (t.detach().to(torch.float32).contiguous() for t in states)
                if tuple(h0.shape) != (self.NL, B, self.H) or tuple(c0.shape) != (self.NL, B, self.H):
                    raise ValueError(f"states must be (h0, c0), each [num_layers={self.NL}, B={B}, H={self.H}]")
                require_gpu(h0, c0)
                opts.h0, opts.c0 = ptr(h0), ptr(c0)
                keep += [h0, c0]
            if force_ids is not None:
                if tuple(force_ids.shape) != (B, Lc) or force_ids.dtype != torch.int64:
                    raise ValueError(f"force_ids must be int64 [B={B}, L={Lc}]")
                force_ids = force_ids.contiguous()
                opts.force_ids = ptr(force_ids)
                if force_len is not None:
                    force_len = force_len.to(torch.int32).contiguous()
                    if tuple(force_len.shape) != (B,):
                        raise ValueError("force_len must hold one prefix length per caption")
                    opts.force_len = ptr(force_len)
                keep += [force_ids, force_len]
            opts.no_state = int(bool(ids_only))
        if resume is not None:
            src_state, src_B, active = resume
            if len(active) != Lc or force_ids is None or force_len is None or not ids_only or st.get("logits") is None:
                raise ValueError("resumed roll-outs: active_rows per step, force_ids, force_len, ids_only and a state with the generic "
                                 "products' scratch (more rows than the fused step kernels take)")
            src_struct = self._state_struct(src_state)
            act = _arr(C.c_int32, Lc, [int(v) for v in active])
            opts.resume_from = C.cast(C.pointer(src_struct), C.c_void_p)
            opts.resume_B = int(src_B)
            opts.host_active_rows = C.cast(act, C.c_void_p)
            keep += [src_struct, act]
        d = self.dims(B, Lc)
        L.check(L.load().gic_decoder_sample_fwd(
            C.byref(d), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(st)),
            ptr(features), ptr(noise_u), int(seed) & (2 ** 64 - 1), float(temperature), int(bool(pretrain)), ptr(out) if not ids_only else None,
            ptr(ids), C.byref(opts) if opts is not None else None, stream_ptr()), "gic_decoder_sample_fwd")
        return (None if ids_only else out), ids, st

    def forward_tf(self, params, features: torch.Tensor, caps: torch.Tensor, lengths, temperature: float, pretrain: bool = False,
                   noise_u: Optional[torch.Tensor] = None, seed: int = 0, keep_state: bool = False):
        """gic_decoder_forward_tf: Decoder.forward (teacher forcing, generator.py:39-53).
        Returns (pred act [B, max(lengths), V], (h_n, c_n) f32 [NL, B, H]); with ``keep_state`` also what ``forward_tf_bwd`` needs."""
        self.check_params(params)
        require_gpu(features, caps, noise_u)
        B, Lc = caps.shape
        T = Lc + 1
        lens = [int(v) for v in (lengths.tolist() if torch.is_tensor(lengths) else lengths)]
        if len(lens) != B or min(lens) < 1 or max(lens) > T:
            raise ValueError(f"lengths must hold {B} values in 1..{T}")
        Tmax = max(lens)
        dev = features.device
        if features.shape != (B, self.E) or caps.dtype != torch.int64:
            raise ValueError("features must be [B, E] and caps int64 [B, L]")
        if noise_u is not None and tuple(noise_u.shape) != (B, Tmax, self.V):
            raise ValueError(f"noise_u must be [B, max(lengths)={Tmax}, V]")
        self.prepare(params)
        st = self.alloc_state(B, T, dev)
        out = torch.empty(B, Tmax, self.V, device=dev, dtype=self.act)
        h_n = torch.empty(self.NL, B, self.H, device=dev, dtype=torch.float32)
        c_n = torch.empty_like(h_n)
        logits_ws = torch.empty(B * Tmax, self.V, device=dev, dtype=torch.float32)
        ids_ws = torch.empty(B * Tmax, device=dev, dtype=torch.int64)
        len_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
        d = self.dims(B, T)
        L.check(L.load().gic_decoder_forward_tf(
            C.byref(d), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(st)),
            ptr(features.contiguous().float()), ptr(caps.contiguous()), ptr(len_dev), Tmax,
            ptr(noise_u.contiguous().float()) if noise_u is not None else None, int(seed) & (2 ** 64 - 1), float(temperature),
            int(bool(pretrain)), ptr(logits_ws), ptr(ids_ws), ptr(out), ptr(h_n), ptr(c_n), stream_ptr()), "gic_decoder_forward_tf")
        if keep_state:
            return out, (h_n, c_n), {"st": st, "caps": caps.contiguous(), "len_dev": len_dev, "Tmax": Tmax, "T": T}
        return out, (h_n, c_n)

    def forward_tf_bwd(self, params, saved, pred: torch.Tensor, d_pred: torch.Tensor, temperature: float, pretrain: bool = False,
                       ws=None, grads=None) -> List[torch.Tensor]:
        """gic_decoder_forward_tf_bwd: gradients of a loss on ``pred`` of the ``forward_tf(..., keep_state=True)`` call that returned
        ``saved``; the list is ordered as ``sample_bwd``'s (parameters, then d features)."""
        B, Tmax, dev = pred.shape[0], saved["Tmax"], pred.device
        if tuple(d_pred.shape) != tuple(pred.shape):
            raise ValueError("d_pred must have pred's shape")
        if d_pred.dtype != self.act:
            d_pred = self._cast_like(d_pred)
        d_pred = d_pred.contiguous()
        self.prepare(params)
        ws = ws if ws is not None else self.alloc_bwd_ws(B, Tmax, dev)
        grads = grads if grads is not None else self.alloc_grads(params, B)
        w = L.DecoderBwdWs()
        w.dlogits, w.dhout = ptr(ws["dlogits"]), ptr(ws["dhout"])
        w.dgates = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in ws["dgates"]])
        w.dxh = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in ws["dxh"]])
        w.dc = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in ws["dc"]])
        L.check(L.load().gic_decoder_forward_tf_bwd(
            C.byref(self.dims(B, saved["T"])), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)),
            C.byref(self._state_struct(saved["st"])), C.byref(w), ptr(pred), ptr(saved["caps"]), ptr(saved["len_dev"]), Tmax, ptr(d_pred),
            float(temperature), int(bool(pretrain)), C.byref(self._pstruct(grads[:-1], L.DecoderGrads, grads[-1])), stream_ptr()),
            "gic_decoder_forward_tf_bwd")
        return grads

    def sample_bwd(self, params, st, out: torch.Tensor, ids: torch.Tensor, d_out: torch.Tensor, temperature: float,
                   pretrain: bool = False, ws=None, grads=None, phases: int = 3, dev_scalars=None) -> List[torch.Tensor]:
        """phases: 1 = output layer only (w_out / b_out gradients complete), 2 = recurrent part, 3 = both; | 4 = also the gradient of
        the initial states, read back with ``state_grads(ws)`` (gicap.h)."""
        B, Lc = ids.shape
        dev = out.device
        if d_out.dtype != self.act:
            d_out = self._cast_like(d_out)
        d_out = d_out.contiguous()
        self.prepare(params)
        ws = ws if ws is not None else self.alloc_bwd_ws(B, Lc, dev)
        grads = grads if grads is not None else self.alloc_grads(params, B)
        w = L.DecoderBwdWs()
        w.dlogits, w.dhout = ptr(ws["dlogits"]), ptr(ws["dhout"])
        w.dgates = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in ws["dgates"]])
        w.dxh = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in ws["dxh"]])
        w.dc = _arr(C.c_void_p, L.MAX_LAYERS, [ptr(t) for t in ws["dc"]])
        d = self.dims(B, Lc)
        L.check(L.load().gic_decoder_sample_bwd(
            C.byref(d), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(st)),
            C.byref(w), ptr(out), ptr(ids), ptr(d_out), float(temperature), int(bool(pretrain)),
            C.byref(self._pstruct(grads[:-1], L.DecoderGrads, grads[-1])), int(phases),
            dev_scalars.ptr if dev_scalars is not None else None, stream_ptr()), "gic_decoder_sample_bwd")
        return grads

    def state_grads(self, ws):
        """(d_h0, d_c0), each f32 [NL, B, H], from the backward workspace of the sample_bwd call that just ran: slot 0 of the
        recurrent input-gradient buffers holds d[x_0 | h_-1], the cell-gradient carry ends at d c_-1."""
        d_h0 = torch.stack([ws["dxh"][l][0][:, self.din(l):] for l in range(self.NL)]).contiguous()
        d_c0 = torch.stack([ws["dc"][l] for l in range(self.NL)]).contiguous()
        return d_h0, d_c0

    def _cast_like(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        dst = torch.empty(t.shape, device=t.device, dtype=self.act)
        n = t.shape[-1]
        cast2d(t, dst, t.numel() // n, n, n, n)
        return dst


# ------------------------------------------------------------------------------------------ discriminator
class DiscEngine:
    """Discriminator.forward/backward (reference src/discriminator.py:34-62) on the HIP library."""

    OUT = 100
    OUT_PAD = 104

    def __init__(self, vocab: int, embed_dim: int, num_rep: int, filter_sizes: Sequence[int], num_filters: Sequence[int], dtype: int,
                 dropout: float = 0.2):
        if len(filter_sizes) != len(num_filters) or not 1 <= len(filter_sizes) <= L.MAX_CONVS:
            raise ValueError("disc_filter_sizes / disc_num_filters must have equal length in 1..%d" % L.MAX_CONVS)
        if embed_dim % num_rep:
            raise ValueError("disc_embed_dim must be a multiple of disc_num_rep")
        self.V, self.De, self.R = vocab, embed_dim, num_rep
        self.fs, self.nf = list(filter_sizes), list(num_filters)
        self.F = sum(self.nf)
        # leading dimension of the [B*R, F] activations: whole 128-byte lines per row in bf16 (64 elements), so that a 64-byte LDS-DMA piece of
        # a row never straddles two cache lines (the highway product over cfg5's 622 592 roll-out rows: 1.65 -> 1.45 ms; cfg5 10.46 -> 10.18 ms,
        # profiles/r03_gemm_tile16_experiment.txt)
        pad = int(os.environ.get("GIC_DISC_FP_ALIGN", "64"))
        self.Fp = (self.F + pad - 1) // pad * pad
        self.s = embed_dim // num_rep
        self.dt = dtype
        self.act = TORCH_DTYPE[dtype]
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError("dropout probability has to be in [0, 1), got %r" % (dropout,))
        self.drop_p = float(dropout)
        self._shadow = None
        self._shadow_key = None

    # params: [emb, (conv_w, conv_b)*nconv, hw_w, hw_b, f2o_w, f2o_b, o2l_w, o2l_b]
    def nparams(self) -> int:
        return 7 + 2 * len(self.fs)

    def dims(self, B: int, Lc: int) -> L.DiscDims:
        d = L.DiscDims()
        d.B, d.L, d.V, d.De, d.R, d.nconv = B, Lc, self.V, self.De, self.R, len(self.fs)
        d.fsize = _arr(C.c_int32, L.MAX_CONVS, self.fs)
        d.nfilt = _arr(C.c_int32, L.MAX_CONVS, self.nf)
        d.F, d.Fp, d.dtype, d.drop_p = self.F, self.Fp, self.dt, self.drop_p
        return d

    def _pstruct(self, params, cls=L.DiscParams):
        n = len(self.fs)
        s = cls()
        s.emb = ptr(params[0])
        s.conv_w = _arr(C.c_void_p, L.MAX_CONVS, [ptr(params[1 + 2 * k]) for k in range(n)])
        s.conv_b = _arr(C.c_void_p, L.MAX_CONVS, [ptr(params[2 + 2 * k]) for k in range(n)])
        o = 1 + 2 * n
        s.hw_w, s.hw_b, s.f2o_w, s.f2o_b, s.o2l_w, s.o2l_b = (ptr(params[o + i]) for i in range(6))
        return s

    def check_params(self, params) -> None:
        if len(params) != self.nparams():
            raise ValueError("discriminator expects %d parameter tensors" % self.nparams())
        require_gpu(*params)
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("discriminator parameters must be contiguous float32 (master weights)")

    def prepare(self, params):
        key = _key(params)
        if self._shadow is not None and key == self._shadow_key:
            return self._shadow
        dev = params[0].device
        if self._shadow is None or self._shadow["hw_w"].device != dev:
            self._shadow = {
                "emb": None if self.dt == L.F32 else torch.empty(self.De, self.V, device=dev, dtype=self.act),
                "hw_w": torch.empty(self.Fp, self.Fp, device=dev, dtype=self.act),
                "f2o_w": torch.empty(self.OUT_PAD, self.Fp, device=dev, dtype=self.act),
                # bf16 mode: highway^T so that the input-gradient product d_pooled += dh W runs on k-contiguous operands
                "hw_w_t": None if self.dt == L.F32 else torch.empty(self.Fp, self.Fp, device=dev, dtype=self.act),
            }
        d = self.dims(1, max(self.fs))
        L.check(L.load().gic_disc_prepare(C.byref(d), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)),
                                          stream_ptr()), "gic_disc_prepare")
        self._shadow_key = key
        return self._shadow

    def _shadow_struct(self, params) -> L.DiscShadow:
        sh = self._shadow
        s = L.DiscShadow()
        s.emb = ptr(params[0]) if sh["emb"] is None else ptr(sh["emb"])
        s.hw_w, s.f2o_w = ptr(sh["hw_w"]), ptr(sh["f2o_w"])
        s.hw_w_t = ptr(sh["hw_w_t"])
        return s

    def alloc_state(self, B: int, Lc: int, dev, forward_only: bool = False):
        """``forward_only``: an eval-mode forward that no backward follows (reward evaluation): no argmax / pre-activation / keep
        buffers (gic_disc_fwd then writes none), and only the pad columns of ``ydrop`` are zeroed."""
        f32, u8 = torch.float32, torch.uint8
        MR = B * self.R
        if forward_only:
            ydrop = torch.empty(MR, self.Fp, device=dev, dtype=self.act)
            if self.Fp > self.F:
                ydrop[:, self.F:].zero_()
            return {"emb": torch.empty(B * Lc, self.De, device=dev, dtype=f32), "pooled": torch.empty(MR, self.Fp, device=dev, dtype=self.act),
                    "argmax": None, "hpre": None, "keep": None, "ydrop": ydrop, "feat": torch.empty(MR, self.OUT, device=dev, dtype=f32)}
        return {
            "emb": torch.empty(B * Lc, self.De, device=dev, dtype=f32),
            "pooled": torch.empty(MR, self.Fp, device=dev, dtype=self.act),
            "argmax": torch.empty(MR, self.Fp, device=dev, dtype=u8),
            "hpre": torch.empty(MR, self.Fp, device=dev, dtype=f32),
            "keep": torch.empty(MR, self.Fp, device=dev, dtype=u8),
            "ydrop": torch.zeros(MR, self.Fp, device=dev, dtype=self.act),    # pad columns must stay zero
            "feat": torch.empty(MR, self.OUT, device=dev, dtype=f32),
        }

    def _state_struct(self, st) -> L.DiscState:
        s = L.DiscState()
        for k in ("emb", "pooled", "argmax", "hpre", "keep", "ydrop", "feat"):
            setattr(s, k, ptr(st.get(k)))
        return s

    def alloc_bwd_ws(self, B: int, Lc: int, dev):
        f32 = torch.float32
        MR = B * self.R
        return {
            "dfeat": torch.empty(MR, self.OUT_PAD, device=dev, dtype=self.act),
            "dh": torch.empty(MR, self.Fp, device=dev, dtype=self.act),
            "dydrop": torch.empty(MR, self.Fp, device=dev, dtype=f32),
            "dpooled": torch.empty(MR, self.Fp, device=dev, dtype=f32),
            "demb": torch.empty(B * Lc, self.De, device=dev, dtype=self.act),
        }

    def soft_input(self, inp: torch.Tensor) -> torch.Tensor:
        """[B,L,V] float tensor -> contiguous compute-dtype tensor (converted by gic_cast2d if needed)."""
        if inp.dim() != 3 or inp.shape[2] != self.V:
            raise ValueError(f"discriminator input must be [B, L, V={self.V}], got {tuple(inp.shape)}")
        require_gpu(inp)
        if inp.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("discriminator input must be float32 or bfloat16")
        inp = inp.contiguous()
        if inp.dtype != self.act:
            dst = torch.empty(inp.shape, device=inp.device, dtype=self.act)
            cast2d(inp, dst, inp.shape[0] * inp.shape[1], self.V, self.V, self.V)
            inp = dst
        return inp

    def fwd(self, params, inp_soft: Optional[torch.Tensor], inp_ids: Optional[torch.Tensor], train: bool,
            keep_mask: Optional[torch.Tensor] = None, seed: int = 0, state=None, logits=None, forward_only: bool = False,
            dev_scalars=None, seed_slot: int = 0):
        """``dev_scalars`` / ``seed_slot``: the dropout seed is read from device memory (gic_step_scalars).  ``forward_only`` (eval mode only): nothing is saved for a backward pass (see alloc_state)."""
        if forward_only and train:
            raise ValueError("forward_only is an eval-mode option: the train-mode forward saves its dropout mask for the backward")
        self.check_params(params)
        src = inp_soft if inp_soft is not None else inp_ids
        require_gpu(src, keep_mask)
        B, Lc = src.shape[0], src.shape[1]
        dev = src.device
        if inp_ids is not None:
            if inp_ids.dtype != torch.int64:
                raise ValueError("token ids must be int64")
            inp_ids = inp_ids.contiguous()
        if keep_mask is not None:
            if tuple(keep_mask.shape) != (B * self.R, self.F):
                raise ValueError(f"keep_mask must be [B*R={B * self.R}, F={self.F}]")
            keep_mask = keep_mask.to(torch.uint8).contiguous()
        self.prepare(params)
        st = state if state is not None else self.alloc_state(B, Lc, dev, forward_only=forward_only)
        logits = logits if logits is not None else torch.empty(B * self.R, device=dev, dtype=torch.float32)
        d = self.dims(B, Lc)
        L.check(L.load().gic_disc_fwd(
            C.byref(d), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(st)),
            ptr(inp_soft), self.V, ptr(inp_ids), int(bool(train)), ptr(keep_mask), int(seed) & (2 ** 64 - 1), ptr(logits),
            dev_scalars.ptr if dev_scalars is not None else None, int(seed_slot), stream_ptr()), "gic_disc_fwd")
        return logits, st

    def shared_state(self, src: dict, B: int, Lc: int, dev) -> dict:
        """State of a second forward on the same input: its own dropout / head buffers, the rest aliases ``src``."""
        st = dict(src)
        own = self.alloc_state(B, Lc, dev)
        for k in ("keep", "ydrop", "feat"):
            st[k] = own[k]
        return st

    def fwd_redrop(self, params, src_state: dict, dst_state: dict, train: bool, keep_mask: Optional[torch.Tensor] = None,
                   seed: int = 0, logits=None, dev_scalars=None, seed_slot: int = 0):
        """gic_disc_fwd_redrop: D on the same input as the forward that filled ``src_state``, under another dropout draw."""
        self.check_params(params)
        MR = src_state["pooled"].shape[0]
        dev = src_state["pooled"].device
        if keep_mask is not None:
            require_gpu(keep_mask)
            if tuple(keep_mask.shape) != (MR, self.F):
                raise ValueError(f"keep_mask must be [B*R={MR}, F={self.F}]")
            keep_mask = keep_mask.to(torch.uint8).contiguous()
        self.prepare(params)
        logits = logits if logits is not None else torch.empty(MR, device=dev, dtype=torch.float32)
        B = MR // self.R
        d = self.dims(B, src_state["emb"].shape[0] // B)
        L.check(L.load().gic_disc_fwd_redrop(
            C.byref(d), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(src_state)),
            C.byref(self._state_struct(dst_state)), int(bool(train)), ptr(keep_mask), int(seed) & (2 ** 64 - 1), ptr(logits),
            dev_scalars.ptr if dev_scalars is not None else None, int(seed_slot), stream_ptr()), "gic_disc_fwd_redrop")
        return logits, dst_state

    def split_state(self, st: dict):
        """Views of the first / second half of a state's rows (two forward passes of B/2 captions each written into one state)."""
        a, b = {}, {}
        for k, v in st.items():
            h = v.shape[0] // 2
            a[k], b[k] = v[:h], v[h:]
        return a, b

    def bwd(self, params, st, inp_soft, inp_ids, train: bool, d_logits: torch.Tensor, want_param_grads: bool,
            want_input_grad: bool, grads=None, accumulate: bool = False, ws=None, d_inp=None):
        """Both ``inp_soft`` and ``inp_ids`` given: mixed batch (gicap.h) -- ``st`` holds the ids pass in its first half of the rows
        and the soft pass in its second half; one backward serves both."""
        src = inp_soft if inp_soft is not None else inp_ids
        B, Lc = src.shape[0], src.shape[1]
        if inp_soft is not None and inp_ids is not None:
            if inp_soft.shape[:2] != inp_ids.shape[:2] or want_input_grad or not want_param_grads:
                raise ValueError("mixed batch: equal halves, parameter gradients only")
            B *= 2
        dev = src.device
        d_logits = d_logits.contiguous().float()
        self.prepare(params)
        ws = ws if ws is not None else self.alloc_bwd_ws(B, Lc, dev)
        if want_param_grads and grads is None:
            grads = [torch.empty_like(p) for p in params]
            accumulate = False
        if want_input_grad and d_inp is None:
            d_inp = torch.empty(B, Lc, self.V, device=dev, dtype=self.act)
        w = L.DiscBwdWs()
        for k in ("dfeat", "dh", "dydrop", "dpooled", "demb"):
            setattr(w, k, ptr(ws[k]))
        d = self.dims(B, Lc)
        gs = self._pstruct(grads, L.DiscGrads) if want_param_grads else None
        L.check(L.load().gic_disc_bwd(
            C.byref(d), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(st)),
            C.byref(w), ptr(inp_soft), self.V, ptr(inp_ids), int(bool(train)), ptr(d_logits),
            C.byref(gs) if gs is not None else None, int(bool(accumulate)), ptr(d_inp) if want_input_grad else None, self.V,
            stream_ptr()), "gic_disc_bwd")
        return (grads if want_param_grads else None), (d_inp if want_input_grad else None)


# ------------------------------------------------------------------------------------------ fused clip + Adam
def clip_adam_partials(n: int) -> int:
    return int(L.load().gic_clip_adam_partials(n))


def clip_adam(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, clip_norm, step_count, norm_out, partials):
    require_gpu(params, grads, exp_avg, exp_avg_sq, step_count, norm_out, partials)
    L.check(L.load().gic_clip_adam(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), float(lr), float(beta1),
                                   float(beta2), float(eps), float(clip_norm), ptr(step_count), ptr(norm_out), ptr(partials),
                                   stream_ptr()), "gic_clip_adam")
    bump_param_epoch()


# ------------------------------------------------------------------------------------------ visual-attention decoder
class AttnDecoderEngine:
    """gic_attn_sample_fwd / bwd (gicap.h): the reference's roll-out loop with soft attention over the trunk's feature map
    (BASELINE config 4; no reference counterpart, oracle/cpu_attention.py).
    params order: [embed, w_ih, w_hh, b_ih, b_hh, w_out, b_out, w_f, b_f, w_h, w_a]."""

    NAMES = ("embed", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out", "w_f", "b_f", "w_h", "w_a")

    def __init__(self, vocab: int, embed: int, hidden: int, feat_c: int, positions: int, attn: int, dtype: int):
        self.V, self.E, self.H, self.C, self.P, self.A, self.dt = vocab, embed, hidden, feat_c, positions, attn, dtype
        self.act = TORCH_DTYPE[dtype]
        self.ldx = embed + feat_c + hidden
        self._shadow = None
        self._shadow_key = None

    def dims(self, B: int, Lc: int) -> L.AttnDims:
        return L.AttnDims(B, Lc, self.V, self.E, self.H, self.C, self.P, self.A, self.dt)

    def _pstruct(self, params, cls=L.AttnParams, extra=None):
        s = cls()
        for n, p in zip(self.NAMES, params):
            setattr(s, n, ptr(p))
        if extra is not None:
            s.features = ptr(extra)
        return s

    def check_params(self, params) -> None:
        if len(params) != len(self.NAMES):
            raise ValueError("attention decoder expects %d parameter tensors" % len(self.NAMES))
        require_gpu(*params)
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("attention decoder parameters must be contiguous float32 (master weights)")

    def prepare(self, params):
        key = _key(params)
        if self._shadow is not None and key == self._shadow_key:
            return self._shadow
        dev = params[0].device
        if self._shadow is None:
            self._shadow = {
                "wcat": torch.empty(4 * self.H, self.ldx, device=dev, dtype=self.act),
                "bsum": torch.empty(4 * self.H, device=dev, dtype=torch.float32),
                "wout": None if self.dt == L.F32 else torch.empty(self.V, self.H, device=dev, dtype=self.act),
                "wcat_t": torch.empty(self.ldx, 4 * self.H, device=dev, dtype=self.act),
                "wf": torch.empty(self.A, self.C, device=dev, dtype=self.act),
                "wh": torch.empty(self.A, self.H, device=dev, dtype=self.act),
            }
        L.check(L.load().gic_attn_prepare(C.byref(self.dims(1, 1)), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)),
                                          stream_ptr()), "gic_attn_prepare")
        self._shadow_key = key
        return self._shadow

    def _shadow_struct(self, params) -> L.AttnShadow:
        sh = self._shadow
        s = L.AttnShadow()
        s.wcat, s.bsum, s.wcat_t, s.wf, s.wh = ptr(sh["wcat"]), ptr(sh["bsum"]), ptr(sh["wcat_t"]), ptr(sh["wf"]), ptr(sh["wh"])
        s.wout = ptr(params[5]) if sh["wout"] is None else ptr(sh["wout"])
        return s

    def alloc_state(self, B: int, Lc: int, dev):
        f32 = torch.float32
        nblk = (self.V + 63) // 64
        return {
            "xh": torch.empty(Lc + 1, B, self.ldx, device=dev, dtype=self.act),
            "gates": torch.empty(Lc, B, 4 * self.H, device=dev, dtype=f32),
            "c": torch.empty(Lc + 1, B, self.H, device=dev, dtype=f32),
            "hout": torch.empty(B, Lc, self.H, device=dev, dtype=self.act),
            "part": torch.empty(2 * Lc * B * nblk + 2 * Lc * B + 2, device=dev, dtype=f32),
            "fproj": torch.empty(B, self.P, self.A, device=dev, dtype=self.act),
            "alpha": torch.empty(Lc, B, self.P, device=dev, dtype=f32),
            "hproj": torch.empty(Lc, B, self.A, device=dev, dtype=f32),
        }

    def _state_struct(self, st) -> L.AttnState:
        s = L.AttnState()
        for k in ("xh", "gates", "c", "hout", "part", "fproj", "alpha", "hproj"):
            setattr(s, k, ptr(st[k]))
        return s

    def sample_fwd(self, params, features, fmap, Lc: int, temperature: float, pretrain: bool = False, noise_u=None, seed: int = 0,
                   state=None, out=None, ids=None, states=None, dev_scalars=None, seed_slot: int = 0):
        """``dev_scalars`` / ``seed_slot``: temperature and seed from device memory (StepScalarsBuffer).
        ``state`` / ``out`` / ``ids``: caller-owned buffers (alloc_state; the fused step driver pre-allocates them).
        ``states`` = (h0, c0), each [1, B, H] or [B, H]: initial LSTM state (constants of the backward pass)."""
        self.check_params(params)
        require_gpu(features, fmap, noise_u)
        B = features.shape[0]
        if tuple(features.shape) != (B, self.E) or features.dtype != torch.float32:
            raise ValueError(f"features must be float32 [B,{self.E}]")
        if tuple(fmap.shape) != (B, self.P, self.C):
            raise ValueError(f"fmap must be [B, P={self.P}, C={self.C}], got {tuple(fmap.shape)}")
        fmap = fmap.contiguous()
        if fmap.dtype != self.act:
            dst = torch.empty(fmap.shape, device=fmap.device, dtype=self.act)
            cast2d(fmap.float() if fmap.dtype not in (torch.float32, torch.bfloat16) else fmap, dst, B * self.P, self.C, self.C, self.C)
            fmap = dst
        if noise_u is not None:
            if tuple(noise_u.shape) != (Lc, B, self.V) or noise_u.dtype != torch.float32:
                raise ValueError(f"noise_u must be float32 [L={Lc},B={B},V={self.V}]")
            noise_u = noise_u.contiguous()
        dev = features.device
        self.prepare(params)
        st = dict(state) if state is not None else self.alloc_state(B, Lc, dev)
        out = out if out is not None else torch.empty(B, Lc, self.V, device=dev, dtype=self.act)
        ids = ids if ids is not None else torch.empty(B, Lc, device=dev, dtype=torch.int64)
        h0 = c0 = None
        if states is not None:
            h0, c0 = (t.detach().to(torch.float32).reshape(-1, self.H).contiguous() for t in states)
            if tuple(h0.shape) != (B, self.H) or tuple(c0.shape) != (B, self.H):
                raise ValueError(f"states must be (h0, c0), each [1, B={B}, H={self.H}]")
            require_gpu(h0, c0)
        L.check(L.load().gic_attn_sample_fwd(
            C.byref(self.dims(B, Lc)), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(st)),
            ptr(features.contiguous()), ptr(fmap), ptr(noise_u), int(seed) & (2 ** 64 - 1), float(temperature), int(bool(pretrain)),
            ptr(out), ptr(ids), ptr(h0), ptr(c0), dev_scalars.ptr if dev_scalars is not None else None, int(seed_slot), stream_ptr()),
            "gic_attn_sample_fwd")
        st["fmap"] = fmap
        return out, ids, st

    def alloc_bwd_ws(self, B: int, Lc: int, dev):
        f32 = torch.float32
        return {
            "dlogits": torch.empty(B, Lc, self.V, device=dev, dtype=self.act),
            "dhout": torch.empty(B, Lc, self.H, device=dev, dtype=f32),
            "dgates": torch.empty(Lc, B, 4 * self.H, device=dev, dtype=self.act),
            "dc": torch.empty(B, self.H, device=dev, dtype=f32),
            "dz": torch.empty(B, self.C, device=dev, dtype=f32),
            "dalpha": torch.empty(B, self.P, device=dev, dtype=f32),
            "dh_extra": torch.empty(B, self.H, device=dev, dtype=f32),
            "dhproj": torch.empty(Lc, B, self.A, device=dev, dtype=self.act),
            "dfproj": torch.empty(B, self.P, self.A, device=dev, dtype=f32),
            "dfproj_act": None if self.dt == L.F32 else torch.empty(B, self.P, self.A, device=dev, dtype=self.act),
            "dwa_rows": torch.empty(B, self.A, device=dev, dtype=f32),
            "dx": torch.empty(Lc * B, self.E, device=dev, dtype=f32),
        }

    def sample_bwd(self, params, st, out, ids, d_out, temperature: float, pretrain: bool = False, ws=None, grads=None, dev_scalars=None):
        """Returns [grads in NAMES order ..., d_features].  ``ws`` (alloc_bwd_ws) / ``grads`` (12 tensors): caller-owned buffers."""
        B, Lc = ids.shape
        dev = out.device
        f32 = torch.float32
        if d_out.dtype != self.act:
            t = d_out.contiguous()
            dst = torch.empty(t.shape, device=dev, dtype=self.act)
            cast2d(t, dst, t.numel() // self.V, self.V, self.V, self.V)
            d_out = dst
        d_out = d_out.contiguous()
        self.prepare(params)
        ws = ws if ws is not None else self.alloc_bwd_ws(B, Lc, dev)
        w = L.AttnBwdWs()
        for k, v in ws.items():
            setattr(w, k, ptr(v))
        grads = grads if grads is not None else [torch.empty_like(p) for p in params] + [torch.empty(B, self.E, device=dev, dtype=f32)]
        L.check(L.load().gic_attn_sample_bwd(
            C.byref(self.dims(B, Lc)), C.byref(self._pstruct(params)), C.byref(self._shadow_struct(params)), C.byref(self._state_struct(st)),
            C.byref(w), ptr(st["fmap"]), ptr(out), ptr(ids), ptr(d_out), float(temperature), int(bool(pretrain)),
            C.byref(self._pstruct(grads[:-1], L.AttnGrads, grads[-1])), dev_scalars.ptr if dev_scalars is not None else None, stream_ptr()),
            "gic_attn_sample_bwd")
        return grads
