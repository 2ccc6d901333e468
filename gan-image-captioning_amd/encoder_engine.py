"""Host side of the encoder kernels: execution plan of the ResNet trunk (buffers, packed weights,
BatchNorm statistics arena) and the Linear + BatchNorm1d head, all through the C ABI.

Trunk dataflow per convolution (NHWC, compute dtype):
    y = conv(x)            gic_conv2d: implicit GEMM on MFMA, BatchNorm sums reduced in the epilogue
    z = relu(bn(y) [+ r])  gic_bn_act: one read of y (+ residual) and one write
The stem fuses bn + relu + 3x3/2 max-pool (gic_bn_relu_maxpool).  All BatchNorm sums of one forward live
in ONE f32 arena that is zeroed with a single fill and consumed by ONE running-statistics launch.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import engine
from .engine import ptr, stream_ptr

AVAILABLE = True
STATS_REPLICAS = 8       # BatchNorm sum replicas (arena spacing): conv workgroup b adds into replica b % nrep (global f32 atomics serialise per address)


def _check(status, what):
    L.check(status, what)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


class _ConvTimer:
    __slots__ = ("trace", "name", "a")

    def __init__(self, trace, name):
        self.trace, self.name = trace, name

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.a.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        b = torch.cuda.Event(enable_timing=True)
        b.record(torch.cuda.current_stream())
        self.trace.append((self.name, self.a, b))
        return False


class _ConvStep:
    __slots__ = ("conv", "bn", "cin", "cout", "k", "stride", "pad", "stats_off", "fused_in", "w", "name")

    def __init__(self, name, conv, bn):
        self.name, self.conv, self.bn = name, conv, bn
        self.cin, self.cout, self.k, self.stride, self.pad = conv.cin, conv.cout, conv.k, conv.stride, conv.pad
        self.stats_off = 0
        self.fused_in = None     # None: not probed; True / False: gic_conv2d_bn_in accepted / refused this layer's shapes
        self.w = None


class TrunkPlan:
    def __init__(self, trunk, dtype: int):
        self.trunk, self.dtype = trunk, dtype
        self.act = engine.TORCH_DTYPE[dtype]
        self.stem = _ConvStep("0", getattr(trunk, "0"), getattr(trunk, "1"))
        self.blocks: List[dict] = []
        for si, stage in enumerate(trunk.stages()):
            for bi, blk in enumerate(stage):
                p = f"{4 + si}.{bi}."
                b = {"kind": blk.kind, "c1": _ConvStep(p + "conv1", blk.conv1, blk.bn1), "c2": _ConvStep(p + "conv2", blk.conv2, blk.bn2),
                     "c3": _ConvStep(p + "conv3", blk.conv3, blk.bn3) if blk.kind == "bottleneck" else None,
                     "ds": _ConvStep(p + "downsample.0", blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None}
                self.blocks.append(b)
        self.steps: List[_ConvStep] = [self.stem]
        for b in self.blocks:
            self.steps += [s for s in (b["c1"], b["c2"], b["c3"], b["ds"]) if s is not None]
        off = 0
        for s in self.steps:
            s.stats_off = off
            off += 2 * s.cout * STATS_REPLICAS
        self.stats_len = off
        self._wkey = None
        self._bufs: Dict[Tuple[int, int], dict] = {}
        self._graphs: Dict[tuple, "torch.cuda.CUDAGraph"] = {}
        self._warm: set = set()
        self.fuse_in = not os.environ.get("GIC_NO_FUSED_BN_IN")
        # block outputs formed on load by the next conv1 (gic_conv1x1_res_in): ON by default for the large grids (>= res_min_rows rows,
        # single-channel-tile conv1s); GIC_FUSED_RES_IN=0 restores the separate bn_act pass.  Measured at cfg2 with the parallel replica
        # folds: 3.01 -> 2.92 ms per step; parity: test_block_output_formed_on_load_equals_the_separate_pass
        self.fuse_res = os.environ.get("GIC_FUSED_RES_IN", "1") != "0"
        self.res_min_rows = int(os.environ.get("GIC_RES_IN_MIN_ROWS", "50000"))
        self.res_max_cout = int(os.environ.get("GIC_RES_IN_MAX_COUT", "128"))
        # ... and, where the kernels exist, WITHOUT conv3's output ever reaching memory: conv3 runs as a statistics-only pass and is
        # recomputed inside the launch that forms the block output and runs the next conv1 (gic_conv_b2b; GIC_NO_CONV_B2B=1: off)
        self.fuse_b2b = not os.environ.get("GIC_NO_CONV_B2B")
        self.b2b_only = int(os.environ.get("GIC_B2B_ONLY", "0"))     # tuning: 0 = every boundary the kernels take, 1 = the 28 x 28 blocks only,
        # 2 = all but the 56 x 56 boundary into 128 output channels
        self._nrep = {}
        self.use_graph = not os.environ.get("GIC_NO_GRAPH")
        self.pending_tracked = 0
        self.conv_trace = None      # measurement (bench.py): a list -> (layer name, start event, stop event) per convolution launch

    def sync_counters(self) -> None:
        """Fold the forward count into every BatchNorm's ``num_batches_tracked`` buffer (kept off the hot path)."""
        if self.pending_tracked:
            for s in self.steps:
                s.bn.num_batches_tracked += self.pending_tracked
            self.pending_tracked = 0

    # ---------------------------------------------------------------- weights (frozen: packed once per version)
    def _pack_weights(self, dev) -> None:
        key = (self.trunk._ptr_epoch, tuple(s.conv.weight._version for s in self.steps))      # (see ResNetTrunk._ptr_epoch)
        if key == self._wkey:
            return
        lib = L.load()
        for s in self.steps:
            w = s.conv.weight.detach()
            engine.require_gpu(w)
            if s is self.stem:
                cinp, kwp = 4, 8                       # [64,7,8,4]: the stem reads a zero-bordered NHWC4 image
            else:
                cinp, kwp = s.cin, s.k
            s.w = torch.empty(s.cout, s.k, kwp, cinp, device=dev, dtype=self.act)
            _check(lib.gic_repack_conv_weight(ptr(w.contiguous()), ptr(s.w), self.dtype, s.cout, s.cin, s.k, s.k, cinp, kwp, stream_ptr()),
                   "gic_repack_conv_weight")
        self._wkey = key

    # ---------------------------------------------------------------- buffers for one (batch, image size)
    def _buffers(self, N: int, S: int, dev) -> dict:
        key = (N, S)
        if key in self._bufs:
            self._nrep = self._bufs[key]["nrep"]
            self._running_table(self._bufs[key], dev)
            return self._bufs[key]
        act = self.act
        b: dict = {"stats": torch.zeros(self.stats_len, device=dev, dtype=torch.float32)}
        b["xin"] = torch.empty(N, S + 6, S + 6, 4, device=dev, dtype=act)
        h = (S + 6 - 7) // 2 + 1
        b["y0"] = torch.empty(N, h, h, 64, device=dev, dtype=act)
        hp = (h + 2 - 3) // 2 + 1
        b["x0"] = torch.empty(N, hp, hp, 64, device=dev, dtype=act)
        rows = {self.stem.name: N * h * h}
        cur = hp
        blocks = []
        for blk in self.blocks:
            c1, c2, c3, ds = blk["c1"], blk["c2"], blk["c3"], blk["ds"]
            e: dict = {"hin": cur}
            if blk["kind"] == "basic":
                ho = (cur + 2 - 3) // c1.stride + 1
                e["y1"] = torch.empty(N, ho, ho, c1.cout, device=dev, dtype=act); e["z1"] = torch.empty_like(e["y1"])
                e["y2"] = torch.empty(N, ho, ho, c2.cout, device=dev, dtype=act)
                rows[c1.name] = rows[c2.name] = N * ho * ho
                cout = c2.cout
            else:
                e["y1"] = torch.empty(N, cur, cur, c1.cout, device=dev, dtype=act); e["z1"] = torch.empty_like(e["y1"])
                rows[c1.name] = N * cur * cur
                ho = (cur + 2 - 3) // c2.stride + 1
                e["y2"] = torch.empty(N, ho, ho, c2.cout, device=dev, dtype=act); e["z2"] = torch.empty_like(e["y2"])
                e["y3"] = torch.empty(N, ho, ho, c3.cout, device=dev, dtype=act)
                rows[c2.name] = rows[c3.name] = N * ho * ho
                cout = c3.cout
            if ds is not None:
                e["yd"] = torch.empty(N, ho, ho, ds.cout, device=dev, dtype=act)
                rows[ds.name] = N * ho * ho
            e["out"] = torch.empty(N, ho, ho, cout, device=dev, dtype=act)
            e["hout"] = ho
            blocks.append(e)
            cur = ho
        b["blocks"] = blocks
        b["feat"] = torch.empty(N, self.trunk.out_features, device=dev, dtype=act)
        b["rows"] = rows
        # replicas actually used per layer: many workgroups -> 8 (atomic contention), few -> 2 (every bn_act block folds all of them)
        lo, mid = (int(v) for v in os.environ.get("GIC_STATS_NREP", "2,4").split(","))
        b["nrep"] = {}
        for s in self.steps:
            tiles = -(-rows[s.name] // 128) * -(-s.cout // 128)
            b["nrep"][s.name] = STATS_REPLICAS if tiles > 1024 else (mid if tiles > 256 else lo)
        self._running_table(b, dev)
        self._bufs[key] = b
        self._nrep = b["nrep"]
        return b

    def _running_table(self, b: dict, dev) -> None:
        """Device table for the one-launch running-statistics update.  It bakes in the BatchNorm buffers' addresses, so it is
        rebuilt whenever they move (module .to() / .float() / load into new storage), like the graph key of _launch_trunk."""
        key = self.trunk._ptr_epoch
        if b.get("table_key") == key:
            return
        rows = b["rows"]
        table = (L.BnRunningDesc * len(self.steps))()
        for i, s in enumerate(self.steps):
            table[i].stats = b["stats"].data_ptr() + 4 * s.stats_off
            table[i].running_mean = s.bn.running_mean.data_ptr()
            table[i].running_var = s.bn.running_var.data_ptr()
            table[i].count = float(rows[s.name])
            table[i].momentum = float(s.bn.momentum)
            table[i].C = s.cout
            table[i].nrep = b["nrep"][s.name]
        raw = bytes(table)
        b["table"] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        b["table_key"] = key

    # ---------------------------------------------------------------- kernels
    def _traced(self, name: str):
        """Context that brackets ONE convolution launch with HIP events on its launch stream while ``conv_trace`` is a list;
        a shared no-op object otherwise (nothing is allocated on the normal path: a hipGraph capture must not see garbage
        collection free device objects)."""
        return _NULL_CTX if self.conv_trace is None else _ConvTimer(self.conv_trace, name)

    def _conv(self, s: _ConvStep, x: torch.Tensor, y: torch.Tensor, stats: Optional[torch.Tensor], N, H, W, cin=None, kw=None, pad=None):
        st = None if stats is None else stats.data_ptr() + 4 * s.stats_off
        with self._traced(s.name):
            _check(L.load().gic_conv2d(ptr(x), ptr(s.w), ptr(y), st, self._nrep[s.name], self.dtype, N, H, W, cin if cin is not None else s.cin, s.cout, s.k,
                                       kw if kw is not None else s.k, s.stride, pad if pad is not None else s.pad, stream_ptr()), "gic_conv2d " + s.name)

    def _bn_relu_conv(self, prev: _ConvStep, y_prev, z_prev, s: _ConvStep, y, stats, training: bool, N, H, W, rows_prev) -> None:
        """z = relu(bn_prev(y_prev)); y = conv_s(z).  In bf16 training mode the normalisation rides in the convolution's A-operand
        path (gic_conv2d_bn_in: z_prev is never written); otherwise bn_act + convolution."""
        if training and self.fuse_in and s.fused_in is not False and self.dtype != L.F32:
            base = stats.data_ptr()
            with self._traced(s.name):
                status = L.load().gic_conv2d_bn_in(ptr(y_prev), base + 4 * prev.stats_off, self._nrep[prev.name], ptr(prev.bn.weight.detach()),
                                                   ptr(prev.bn.bias.detach()), float(rows_prev), ptr(s.w), ptr(y), base + 4 * s.stats_off,
                                                   self._nrep[s.name], self.dtype, N, H, W, s.cin, s.cout, s.k, s.k, s.stride, s.pad,
                                                   stream_ptr())
            if status == L.ERR_UNSUPPORTED and self.conv_trace:
                self.conv_trace.pop()
            if status == L.ERR_UNSUPPORTED:
                s.fused_in = False
            else:
                _check(status, "gic_conv2d_bn_in " + s.name)
                s.fused_in = True
                return
        self._bn_act(prev, y_prev, z_prev, stats, training, rows_prev)
        self._conv(s, z_prev, y, stats, N, H, W)

    def _conv1_res_in(self, pend, c1: _ConvStep, y1, stats, N, H, W) -> bool:
        """conv1 of a block whose input (the previous block's output) is still pending: formed on load and written to its buffer by
        the convolution itself.  pend = (last step, its raw output, shortcut tensor, shortcut step | None, rows, out buffer)."""
        last, ylast, res, res_step, rows, out = pend
        # measured (profiles/, cfg2): the fused launch beats bn_act + plain convolution where the grid is large (>= ~390 row tiles: the
        # 56x56 and 28x28 inputs at batch 64: -8..-15 us per block); on small grids the ring-less fused kernel is latency-bound
        # ... and where conv1 has ONE output-channel tile: every further tile would fetch and form the two A-side tiles again
        if c1.fused_in is False or c1.k != 1 or c1.stride != 1 or rows < self.res_min_rows or c1.cout > self.res_max_cout:
            return False
        base = stats.data_ptr()
        rs = base + 4 * res_step.stats_off if res_step is not None else None
        with self._traced(c1.name):
            status = L.load().gic_conv1x1_res_in(
                ptr(ylast), base + 4 * last.stats_off, self._nrep[last.name], ptr(last.bn.weight.detach()), ptr(last.bn.bias.detach()),
                ptr(res), rs, self._nrep[res_step.name] if res_step is not None else 1,
                ptr(res_step.bn.weight.detach()) if res_step is not None else None, ptr(res_step.bn.bias.detach()) if res_step is not None else None,
                float(rows), ptr(out), ptr(c1.w), ptr(y1), base + 4 * c1.stats_off, self._nrep[c1.name], self.dtype, N, H, W, c1.cin, c1.cout,
                stream_ptr())
        if status == L.ERR_UNSUPPORTED:
            if self.conv_trace:
                self.conv_trace.pop()
            c1.fused_in = False
            return False
        _check(status, "gic_conv1x1_res_in " + c1.name)
        c1.fused_in = True
        return True

    def unstored_convs(self) -> set:
        """Names of the convolutions whose raw output the last training pass never stored (conv3 recomputed inside gic_conv_b2b): what a
        storage-emulating reference has to leave unrounded."""
        return {blk["c3"].name for blk in self.blocks if blk.get("b2b") is True}

    def _b2b_ok(self, blk: dict, nxt: Optional[dict], rows_: int, training: bool) -> bool:
        """conv3 of `blk` + the block output + conv1 of `nxt` as gic_conv1x1_bn_in_stats + gic_conv_b2b (shapes the kernels exist for)."""
        if not (training and self.fuse_res and self.fuse_b2b and self.fuse_in and self.dtype != L.F32) or nxt is None or rows_ in blk.get("b2b_refused", ()):
            return False
        if blk["kind"] != "bottleneck" or nxt["kind"] != "bottleneck":
            return False
        c2, c3, c1n = blk["c2"], blk["c3"], nxt["c1"]
        if (self.b2b_only == 1 and c2.cout != 128) or (self.b2b_only == 2 and (c2.cout, c1n.cout) == (64, 128)):
            return False
        return (c3.k == 1 and c3.stride == 1 and c1n.k == 1 and c1n.stride == 1 and c3.cout == 4 * c2.cout and c1n.cin == c3.cout and
                (c2.cout, c1n.cout) in ((64, 64), (64, 128), (128, 128), (128, 256)) and rows_ % 128 == 0 and rows_ >= self.res_min_rows)

    def _conv3_stats_only(self, blk: dict, y2, stats, rows_c2: int, rows_: int) -> bool:
        c2, c3 = blk["c2"], blk["c3"]
        base = stats.data_ptr()
        with self._traced(c3.name):
            status = L.load().gic_conv1x1_bn_in_stats(ptr(y2), base + 4 * c2.stats_off, self._nrep[c2.name], ptr(c2.bn.weight.detach()),
                                                      ptr(c2.bn.bias.detach()), float(rows_c2), ptr(c3.w), base + 4 * c3.stats_off,
                                                      self._nrep[c3.name], self.dtype, rows_, c2.cout, c3.cout, stream_ptr())
        if status == L.ERR_UNSUPPORTED:
            if self.conv_trace:
                self.conv_trace.pop()
            blk["b2b"] = False
            blk.setdefault("b2b_refused", set()).add(rows_)        # (this row count only: another batch size is probed again)
            return False
        _check(status, "gic_conv1x1_bn_in_stats " + c3.name)
        blk["b2b"] = True
        return True

    def _conv_b2b(self, pb, c1: _ConvStep, y1, stats) -> None:
        """pb = (c2, y2, c3, shortcut tensor, shortcut step | None, rows, out buffer): the pending block's tail + this block's conv1."""
        c2, y2, c3, res, res_step, rows_, out = pb
        base = stats.data_ptr()
        with self._traced(c1.name):
            status = L.load().gic_conv_b2b(
                ptr(y2), base + 4 * c2.stats_off, self._nrep[c2.name], ptr(c2.bn.weight.detach()), ptr(c2.bn.bias.detach()), ptr(c3.w),
                base + 4 * c3.stats_off, self._nrep[c3.name], ptr(c3.bn.weight.detach()), ptr(c3.bn.bias.detach()), ptr(res),
                base + 4 * res_step.stats_off if res_step is not None else None, self._nrep[res_step.name] if res_step is not None else 1,
                ptr(res_step.bn.weight.detach()) if res_step is not None else None, ptr(res_step.bn.bias.detach()) if res_step is not None else None,
                float(rows_), ptr(out), ptr(c1.w), ptr(y1), base + 4 * c1.stats_off, self._nrep[c1.name], self.dtype, rows_, c2.cout, c1.cout,
                stream_ptr())
        _check(status, "gic_conv_b2b " + c1.name)      # (eligibility was decided by _b2b_ok: a refusal here would leave the block output unformed)

    def _bn_args(self, s: Optional[_ConvStep], stats: Optional[torch.Tensor], training: bool):
        """(stats, gamma, beta, run_mean, run_var) pointers of one BatchNorm; all None for 'no BN'."""
        if s is None:
            return (None,) * 5
        g, b = ptr(s.bn.weight.detach()), ptr(s.bn.bias.detach())
        if training:
            return (stats.data_ptr() + 4 * s.stats_off, g, b, None, None)
        return (None, g, b, ptr(s.bn.running_mean), ptr(s.bn.running_var))

    def _bn_act(self, s, y, out, stats, training, rows, relu=True, res=None, res_step=None):
        a = self._bn_args(s, stats, training)
        r = self._bn_args(res_step, stats, training)
        _check(L.load().gic_bn_act(ptr(y), *a, ptr(res), *r, self._nrep[s.name], float(rows), int(relu), ptr(out), self.dtype, rows, s.cout, stream_ptr()),
               "gic_bn_act " + s.name)

    def forward(self, images: torch.Tensor, training: bool) -> torch.Tensor:
        """images f32 [N,3,S,S] (NCHW, as the reference's collate_fn delivers them) -> features act [N, C]."""
        engine.require_gpu(images)
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != images.shape[3] or images.dtype != torch.float32:
            raise ValueError(f"images must be float32 [N,3,S,S], got {tuple(images.shape)} {images.dtype}")
        N, S = images.shape[0], images.shape[2]
        if S % 2:
            raise ValueError("image size must be even")
        dev = images.device
        lib = L.load()
        self._pack_weights(dev)
        b = self._buffers(N, S, dev)
        # the caller's image tensor changes from batch to batch: packed into the plan's own NHWC4 buffer outside the graph
        _check(lib.gic_pack_image(ptr(images.contiguous()), ptr(b["xin"]), self.dtype, N, S, 3, S + 6, stream_ptr()), "gic_pack_image")
        self._launch_trunk(b, N, S, training)
        if training:
            self.pending_tracked += 1          # num_batches_tracked buffers are brought up to date by sync_counters()
        return b["feat"]

    def last_map(self, N: int, S: int) -> torch.Tensor:
        """The last block's output [N, h, w, C] of the most recent pass at this shape (the plan's live buffer: clone to keep)."""
        return self._bufs[(N, S)]["blocks"][-1]["out"]

    def _launch_trunk(self, b: dict, N: int, S: int, training: bool) -> None:
        """Everything behind the packed image runs on the plan's own buffers with fixed arguments: ~105 launches that
        are replayed as ONE hipGraph (the second call with a given key captures it; GIC_NO_GRAPH=1 keeps eager launches).
        The key covers every pointer baked into the graph."""
        if not self.use_graph:
            return self._run_trunk(b, N, S, training)
        key = (N, S, bool(training), self._wkey)       # _wkey carries the trunk's pointer epoch: every pointer baked into the graph
        g = self._graphs.get(key)
        if g is not None:
            g.replay()
        elif key not in self._warm:
            self._warm.add(key)                # first call: eager (lazy code-object loads are not capturable)
            self._run_trunk(b, N, S, training)
        else:
            self._graphs.clear()               # at most one live graph per plan: stale pointers never replay (dies here, outside capture)
            g = torch.cuda.CUDAGraph()
            try:
                # capture_guard: the collector runs before the region and not inside it (engine.capture_guard; _run_trunk allocates
                # tuples / ctypes structs per launch, enough to trigger a generational collection mid-capture).
                # thread_local: other threads of the process (RCCL's watchdog polls events) may keep calling HIP during capture
                with engine.capture_guard(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._run_trunk(b, N, S, training)
            except Exception as exc:           # capture refused: keep training with eager launches (same kernels, same results)
                import warnings
                warnings.warn(f"hipGraph capture of the trunk forward failed ({exc}); falling back to eager launches")
                self.use_graph = False
                torch.cuda.synchronize()
                return self._run_trunk(b, N, S, training)
            self._graphs[key] = g
            g.replay()

    def _run_trunk(self, b: dict, N: int, S: int, training: bool) -> None:
        lib = L.load()
        stats = b["stats"] if training else None
        if training:
            b["stats"].zero_()
        rows = b["rows"]
        # stem: 7x7/2 on the zero-bordered NHWC4 image (window [7 x 8 x 4], no bounds checks), bn+relu+maxpool
        self._conv(self.stem, b["xin"], b["y0"], stats, N, S + 6, S + 6, cin=4, kw=8, pad=0)
        h = b["y0"].shape[1]
        a = self._bn_args(self.stem, stats, training)
        _check(lib.gic_bn_relu_maxpool(ptr(b["y0"]), *a, self._nrep[self.stem.name], float(rows[self.stem.name]), ptr(b["x0"]), self.dtype, N, h, h, 64, stream_ptr()),
               "gic_bn_relu_maxpool")
        x = b["x0"]
        pend = None     # a block output not yet materialised: (last step, raw output, shortcut, shortcut step | None, rows, out buffer)
        fuse_res = training and self.fuse_res and self.dtype != L.F32

        def flush():
            nonlocal pend
            if pend is not None:
                last, ylast, res, res_step, rows_, out = pend
                self._bn_act(last, ylast, out, stats, training, rows_, res=res, res_step=res_step)
                pend = None

        pend_b2b = None  # ... or one whose conv3 ran as a statistics-only pass: (c2, y2, c3, shortcut, shortcut step | None, rows, out buffer)
        for bi, (blk, e) in enumerate(zip(self.blocks, b["blocks"])):
            c1, c2, c3, ds = blk["c1"], blk["c2"], blk["c3"], blk["ds"]
            hin, ho = e["hin"], e["hout"]
            nxt = self.blocks[bi + 1] if bi + 1 < len(self.blocks) else None
            b2b = False
            if blk["kind"] == "basic":
                flush()
                self._conv(c1, x, e["y1"], stats, N, hin, hin)
                self._bn_relu_conv(c1, e["y1"], e["z1"], c2, e["y2"], stats, training, N, ho, ho, rows[c1.name])
                last, ylast = c2, e["y2"]
            else:
                # conv1: if the previous block's output is pending, it is formed on load (bn3 + shortcut + relu) and written by the
                # convolution's first N tile; otherwise a plain convolution on the materialised input
                if pend_b2b is not None:
                    self._conv_b2b(pend_b2b, c1, e["y1"], stats)
                    pend_b2b = None
                elif pend is None or not self._conv1_res_in(pend, c1, e["y1"], stats, N, hin, hin):
                    flush()
                    self._conv(c1, x, e["y1"], stats, N, hin, hin)
                pend = None
                # bn1 + ReLU ride into the 3x3 convolution where its input patch stays in LDS (stride 1: normalised once per chunk,
                # not once per tap; the library declines the stride-2 ones: bn_act + plain convolution), bn2 + ReLU into conv3
                self._bn_relu_conv(c1, e["y1"], e["z1"], c2, e["y2"], stats, training, N, hin, hin, rows[c1.name])
                b2b = self._b2b_ok(blk, nxt, rows[c3.name], training) and self._conv3_stats_only(blk, e["y2"], stats, rows[c2.name], rows[c3.name])
                blk["b2b"] = bool(b2b)                                   # (what this pass did: unstored_convs(), conv_shapes())
                if not b2b:
                    self._bn_relu_conv(c2, e["y2"], e["z2"], c3, e["y3"], stats, training, N, ho, ho, rows[c2.name])
                last, ylast = c3, e["y3"]
            if ds is not None:
                self._conv(ds, x, e["yd"], stats, N, hin, hin)        # reads the (by now materialised) block input
                res, res_step = e["yd"], ds
            else:
                res, res_step = x, None
            if blk["kind"] != "basic" and b2b:
                pend_b2b = (c2, e["y2"], c3, res, res_step, rows[c3.name], e["out"])
            elif fuse_res and blk["kind"] != "basic":
                pend = (last, ylast, res, res_step, rows[last.name], e["out"])
            else:
                self._bn_act(last, ylast, e["out"], stats, training, rows[last.name], res=res, res_step=res_step)
            x = e["out"]
        flush()                                                        # the last block's output feeds the average pool
        ho = x.shape[1]
        _check(lib.gic_avgpool(ptr(x), ptr(b["feat"]), self.dtype, N, ho * ho, x.shape[3], stream_ptr()), "gic_avgpool")
        if training:
            _check(lib.gic_bn_running_update(ptr(b["table"]), len(self.steps), stream_ptr()), "gic_bn_running_update")

    # ---------------------------------------------------------------- measurement helper for bench.py
    def replay(self, s: _ConvStep, xi, yo, stats, N: int, H: int, W: int, kw: dict, prev=None) -> None:
        """One launch of the step's convolution for layer `s` as the training forward issues it (measurement helper): the
        A-side-BatchNorm variant where the plan uses it (prev = (producer step, its raw output, its row count)), the residual-on-load
        variant (prev = ("res", pending tuple)), else gic_conv2d."""
        if prev is not None and prev[0] == "b2b":
            self._conv_b2b(prev[1], s, yo, stats)
        elif prev is not None and prev[0] == "b2bstats":
            if not self._conv3_stats_only(prev[1], prev[2], stats, prev[3], prev[4]):
                raise RuntimeError("statistics-only conv3 refused for " + s.name)
        elif prev is not None and prev[0] == "res":
            if s.fused_in and self._conv1_res_in(prev[1], s, yo, stats, N, H, W):
                return
            self._conv(s, xi, yo, stats, N, H, W, **kw)
        elif prev is not None and s.fused_in:
            p, yp, rows_p = prev
            self._bn_relu_conv(p, yp, None, s, yo, stats, True, N, H, W, rows_p)
        else:
            self._conv(s, xi, yo, stats, N, H, W, **kw)

    def conv_shapes(self, N: int, S: int):
        """[(step, input tensor, output tensor, H, W, conv kwargs, macs, prev)] in execution order; prev = (producer step, its raw
        output, its rows) for the layers whose input BatchNorm rides in the convolution, ("res", pending tuple) for a conv1 that
        forms the previous block's output on load."""
        b = self._buffers(N, S, self.stem.conv.weight.device)
        out = [(self.stem, b["xin"], b["y0"], S + 6, S + 6, dict(cin=4, kw=8, pad=0), b["y0"].shape[1] ** 2 * N * 64 * 147, None)]
        x = b["x0"]
        pend = None
        pend_b2b = None
        for blk, e in zip(self.blocks, b["blocks"]):
            c1, c2, c3, ds = blk["c1"], blk["c2"], blk["c3"], blk["ds"]
            hin, ho = e["hin"], e["hout"]
            first = ("b2b", pend_b2b) if pend_b2b is not None else (("res", pend) if (pend is not None and c1.fused_in) else None)
            seq = [(c1, x, e["y1"], hin, first),
                   (c2, e["z1"], e["y2"], ho if blk["kind"] == "basic" else hin, (c1, e["y1"], b["rows"][c1.name]))]
            b2b = c3 is not None and blk.get("b2b") is True                 # (as the last training pass decided it)
            if c3 is not None:
                seq.append((c3, e["z2"], e["y3"], ho, ("b2bstats", blk, e["y2"], b["rows"][c2.name], b["rows"][c3.name]) if b2b
                            else (c2, e["y2"], b["rows"][c2.name])))
            if ds is not None:
                seq.append((ds, x, e["yd"], hin, None))
            for s, xi, yo, hh, prev in seq:
                out.append((s, xi, yo, hh, hh, {}, yo.shape[0] * yo.shape[1] * yo.shape[2] * s.cout * s.cin * s.k * s.k, prev))
            pend = pend_b2b = None
            if blk["kind"] != "basic":
                last = c3
                if b2b:
                    pend_b2b = (c2, e["y2"], c3, e["yd"] if ds is not None else x, ds, b["rows"][c3.name], e["out"])
                else:
                    pend = (last, e["y3"], e["yd"] if ds is not None else x, ds, b["rows"][last.name], e["out"])
            x = e["out"]
        return out


# ------------------------------------------------------------------------------------------ encoder head
def head_fwd(dtype, feat, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, out=None):
    """Linear(feat -> E) + BatchNorm1d on [B,E] (generator.py:24).  feat: act or f32 [B,F].  Returns (out f32 [B,E], saved)."""
    engine.require_gpu(feat, weight, bias, gamma, beta)
    act = engine.TORCH_DTYPE[dtype]
    B, F = feat.shape
    E = weight.shape[0]
    dev = feat.device
    if feat.dtype != act:
        f2 = torch.empty(B, F, device=dev, dtype=act)
        engine.cast2d(feat.contiguous(), f2, B, F, F, F)
        feat = f2
    feat = feat.contiguous()
    wsh = weight
    if dtype != L.F32:
        wsh = torch.empty(E, F, device=dev, dtype=act)
        engine.cast2d(weight, wsh, E, F, F, F)
    y = torch.empty(B, E, device=dev, dtype=torch.float32)
    engine.gemm(feat, wsh, y, B, E, F, F, F, E, bias=bias)
    out = out if out is not None else torch.empty(B, E, device=dev, dtype=torch.float32)
    xhat = torch.empty(B, E, device=dev, dtype=torch.float32)
    invstd = torch.empty(E, device=dev, dtype=torch.float32)
    L.check(L.load().gic_bn1d_fwd(ptr(y), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), int(bool(training)), float(momentum),
                                  float(eps), ptr(out), ptr(xhat), ptr(invstd), B, E, stream_ptr()), "gic_bn1d_fwd")
    return out, (feat, xhat, invstd, bool(training))


def head_bwd(dtype, saved, weight, gamma, d_out, grads=None):
    """Returns (d_weight [E,F], d_bias [E], d_gamma [E], d_beta [E]); the trunk is frozen so no d_feat."""
    feat, xhat, invstd, training = saved
    act = engine.TORCH_DTYPE[dtype]
    B, F = feat.shape
    E = weight.shape[0]
    dev = feat.device
    d_out = d_out.contiguous().float()
    dw, db, dg, dbt = grads if grads is not None else (torch.empty(E, F, device=dev), torch.empty(E, device=dev),
                                                       torch.empty(E, device=dev), torch.empty(E, device=dev))
    dy = torch.empty(B, E, device=dev, dtype=torch.float32)
    L.check(L.load().gic_bn1d_bwd(ptr(d_out), ptr(xhat), ptr(invstd), ptr(gamma), int(training), ptr(dy), ptr(dg), ptr(dbt), B, E,
                                  stream_ptr()), "gic_bn1d_bwd")
    dya = dy
    if dtype != L.F32:
        dya = torch.empty(B, E, device=dev, dtype=act)
        engine.cast2d(dy, dya, B, E, E, E)
    engine.gemm(dya, feat, dw, E, F, B, E, F, F, a_kc=False, b_kc=False)          # dW = dy^T feat
    L.check(L.load().gic_colsum(ptr(dy), L.F32, E, B, E, ptr(db), 0, stream_ptr()), "gic_colsum")
    return dw, db, dg, dbt


def roofline_probe(encoder, args, event_time_ms, peak_tflops, pmc_traffic=None, in_step=None):
    """The dominant kernel of the step = the implicit-GEMM convolution kernel (every trunk convolution is a launch of it).

    ``in_step`` = {"in_step": {layer: [ms per launch, ...]}, "alone": {...}} measured by bench.py with HIP events around every
    convolution launch on its launch stream WHILE the real train step runs on the other streams (TrunkPlan.conv_trace), and the
    same brackets with the chip otherwise idle.  A bracket adds event + dispatch latency to the kernel's duration; per layer
    that overhead = bracket("alone") - the back-to-back replay of the layer (measured here), and it is subtracted from the in-step
    bracket.  `achieved` / `frac` = algorithmic FLOPs of all launches / the summed corrected in-step durations.  The isolated
    replay (hot caches, nothing else on the chip) is kept beside it under "isolated_replay"."""
    import sys
    plan = encoder.resnet._plan
    N, S = args.adv_train_batch_size, args.image_size
    stream = torch.cuda.current_stream()
    b = plan._buffers(N, S, encoder.linear.weight.device)
    seen = {}
    per_layer_step_ms = {}
    for s, xi, yo, H, W, kw, macs, prev in plan.conv_shapes(N, S):
        mode = 0
        if prev is not None and prev[0] == "b2b":
            mode = 3                   # conv3 recomputed + block output + this conv1 in one launch (gic_conv_b2b)
        elif prev is not None and prev[0] == "b2bstats":
            mode = 4                   # conv3's column sums only (the first pass of that pair)
        elif prev is not None and s.fused_in:
            mode = 2 if prev[0] == "res" else 1
        key = (s.cin, s.cout, s.k, s.stride, H, mode + (10 if (mode in (2, 3) and prev[1][4 if mode == 3 else 3] is not None) else 0))
        if key not in seen:
            seen[key] = [event_time_ms(lambda: plan.replay(s, xi, yo, b["stats"], N, H, W, kw, prev), 5, stream), 0, macs,
                         s.name + {0: "", 1: " [bn+relu on load]", 2: " [block output on load]", 12: " [block output on load, proj.]",
                                   3: " [conv3 + block output + conv1]", 13: " [conv3 + block output + conv1, proj.]", 4: " [statistics only]"}[key[5]], 0.0,
                         # the same layer as a plain convolution of an already normalised input (what the fused launches replaced)
                         event_time_ms(lambda: plan.replay(s, xi, yo, b["stats"], N, H, W, kw, None), 5, stream) if mode in (1, 2) else None]
        seen[key][1] += 1
        if in_step and s.name in in_step["in_step"]:
            v, a0 = in_step["in_step"][s.name], in_step["alone"].get(s.name)
            overhead = max(0.0, sum(a0) / len(a0) - seen[key][0]) if a0 else 0.0
            seen[key][4] += max(seen[key][0], sum(v) / len(v) - overhead)
    total_ms = total_flops = bound_us = total_bytes = step_ms = plain_ms = 0.0
    launches = 0
    layers = []
    for key, (ms, count, macs, name, ms_step, ms_plain) in seen.items():
        total_ms += ms * count
        plain_ms += (ms if ms_plain is None else ms_plain) * count
        step_ms += ms_step
        total_flops += 2.0 * macs * count
        launches += count
        tf = 2.0 * macs / (ms * 1e-3) / 1e12
        layers.append((tf, name, key, ms, count))
        Ho = (key[4] + 2 * (key[2] // 2) - key[2]) // key[3] + 1
        nbytes = 2.0 * (N * key[4] * key[4] * key[0] + N * Ho * Ho * key[1] + key[0] * key[1] * key[2] * key[2])
        if key[5] in (2, 12):          # block output formed on load: + the shortcut tensor read and the block output written back
            nbytes += 2.0 * 2 * N * key[4] * key[4] * key[0]
        elif key[5] in (3, 13):        # back to back: the "input" counted above is the shortcut; + the block output written, + y2 (a quarter of the channels) read
            nbytes += 2.0 * 1.25 * N * key[4] * key[4] * key[0]
        elif key[5] == 4:              # statistics only: nothing is written
            nbytes -= 2.0 * N * Ho * Ho * key[1]
        floor_us = max(2.0 * macs / (peak_tflops * 1e12), nbytes / 8e12) * 1e6
        bound_us += floor_us * count
        total_bytes += nbytes * count
        print(f"[conv] {name:30s} Cin={key[0]:5d} Cout={key[1]:5d} k={key[2]} s={key[3]} H={key[4]:4d} x{count}: isolated {ms * 1e3:7.1f} us {tf:7.1f} TFLOP/s "
              f"{nbytes / (ms * 1e-3) / 1e9:6.0f} GB/s | in-step {ms_step / count * 1e3 if ms_step else float('nan'):7.1f} us | floor {floor_us:6.1f} us",
              file=sys.stderr)
    print(f"[conv] all {launches} launches: isolated {total_ms * 1e3:.1f} us, in-step {step_ms * 1e3:.1f} us; per-layer max(MFMA, HBM) floor {bound_us:.1f} us",
          file=sys.stderr)
    layers.sort()
    plain = total_flops / (plain_ms * 1e-3) / 1e12
    iso = total_flops / (total_ms * 1e-3) / 1e12
    use_ms = step_ms if step_ms > 0 else total_ms
    achieved = total_flops / (use_ms * 1e-3) / 1e12
    fmt = lambda l: {"layer": l[1], "cin": l[2][0], "cout": l[2][1], "k": l[2][2], "stride": l[2][3], "tflops": round(l[0], 1), "us": round(l[3] * 1e3, 1)}
    return {"kernel": "conv kernels of the ResNet trunk (tile8_kernel<CONV, EPI_BNSTATS> implicit GEMM; conv3x3_patch_kernel for the 3x3 / stride-1 layers, "
                      f"input patch resident in LDS): bf16 16x16x32 MFMA, 8 waves, LDS-DMA ring, {launches} launches/step",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak_tflops, "unit": "TFLOP/s", "frac": round(achieved / peak_tflops, 4),
            "measured": ("HIP events around every convolution launch on its launch stream while the train step runs on the other streams, "
                         "minus the per-layer bracket overhead (bracket with the chip idle - back-to-back replay)"
                         if step_ms > 0 else "isolated replay of each layer shape under HIP events"),
            "traffic": pmc_traffic, "ms_per_launch": round(use_ms / launches, 5), "launches_per_step": launches,
            "ms_per_step": round(use_ms, 4), "algorithmic_gflop_per_step": round(total_flops / 1e9, 1),
            "algorithmic_bytes_per_launch": int(total_bytes / launches),
            "floor_ms_per_step": round(bound_us / 1e3, 4),      # sum over layers of max(flops / MFMA peak, bytes / 8 TB/s)
            "frac_of_floor": round(bound_us / 1e3 / use_ms, 4),
            "isolated_replay": {"ms_per_step": round(total_ms, 4), "achieved": round(iso, 2), "frac": round(iso / peak_tflops, 4)},
            # every layer as a plain convolution (input already normalised, block outputs materialised by bn_act): the kernels' own
            # quality, independent of how much BatchNorm / residual work rides in the launches
            "plain_conv_variant": {"ms_per_step": round(plain_ms, 4), "achieved": round(plain, 2), "frac": round(plain / peak_tflops, 4)},
            "slowest_layer": fmt(layers[0]), "fastest_layer": fmt(layers[-1])}
