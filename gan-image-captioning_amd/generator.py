"""Generator: Encoder / Decoder / Generator with the reference's module API
(src/generator.py:8-123) and state-dict key names, computing through libgicap.so.

Differences a caller can observe (all documented in DESIGN.md):
  * ``Decoder.sample`` returns its probabilities in the compute dtype (bf16 when
    ``--compute-dtype bf16``; float32 in parity mode), laid out [B, L, V] contiguous.
  * ``sample`` / ``Discriminator.forward`` take optional explicit noise (``noise_u``,
    ``keep_mask``) so a run can be replayed bit-for-bit against the CPU reference; without it
    noise comes from an on-device Philox stream seeded from ``torch.initial_seed()``.
  * ``Generator.forward`` reads ``args.conditional_gan`` (the reference reads a non-existent
    ``args.cgan``, generator.py:109, and is never called by training.py).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import engine
from .trunk import ResNetTrunk, encoder_head_bwd, encoder_head_fwd


def _compute_dtype(args) -> int:
    return engine.parse_dtype(getattr(args, "compute_dtype", "bf16"))


class _SeedStream:
    """Host-side counter that hands a fresh Philox seed to every stochastic kernel launch."""

    def __init__(self):
        self._n = 0

    rank = 0        # data-parallel rank (set by GANInstructor): replicas draw DIFFERENT Gumbel noise / dropout masks

    def reset(self, n: int = 0) -> None:
        """Restart the counter (tests: two runs that must draw the same device noise)."""
        self._n = int(n)

    def next(self) -> int:
        self._n += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._n * 0xD1B54A32D192ED03
                + self.rank * 0xA0761D6478BD642F) & (2 ** 64 - 1)


SEEDS = _SeedStream()


# ------------------------------------------------------------------------------------------ embedding
class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.vocab = weight.shape[0]
        return engine.embedding_fwd(weight.detach(), ids)

    @staticmethod
    def backward(ctx, d_out):
        (ids,) = ctx.saved_tensors
        return engine.embedding_bwd(d_out, ids, ctx.vocab), None


class Embedding(nn.Module):
    """nn.Embedding(V, E) stand-in (same parameter name / init) used as a callable (training.py:68,147)."""

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)

    def forward(self, ids):
        return _EmbeddingFn.apply(self.weight, ids)


class _LSTMParams(nn.Module):
    """Parameter container with nn.LSTM's names, shapes and default init (generator.py:32)."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        k = 1.0 / math.sqrt(hidden_size)
        for l in range(num_layers):
            din = input_size if l == 0 else hidden_size
            for name, shape in ((f"weight_ih_l{l}", (4 * hidden_size, din)), (f"weight_hh_l{l}", (4 * hidden_size, hidden_size)),
                                (f"bias_ih_l{l}", (4 * hidden_size,)), (f"bias_hh_l{l}", (4 * hidden_size,))):
                self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def layer_params(self, l: int) -> List[nn.Parameter]:
        return [getattr(self, f"{n}_l{l}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


class _LinearParams(nn.Module):
    """Parameter container with nn.Linear's names, shapes and default init."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        k = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-k, k))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-k, k)) if bias else None


# ------------------------------------------------------------------------------------------ decoder
class _SampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, temperature, pretrain, max_len, noise_u, seed, h0, c0, features, *params):
        dparams = [p.detach() for p in params]
        states = None if h0 is None else (h0.detach(), c0.detach())
        out, ids, st = eng.sample_fwd(dparams, features.detach().float(), max_len, temperature, pretrain, noise_u, seed, states=states)
        ctx.eng, ctx.temperature, ctx.pretrain = eng, temperature, pretrain
        ctx.st, ctx.dparams = st, dparams
        ctx.has_states = states is not None
        ctx.save_for_backward(out, ids)
        ctx.mark_non_differentiable(ids)
        return out, ids

    @staticmethod
    def backward(ctx, d_out, _d_ids):
        out, ids = ctx.saved_tensors
        ws = ctx.eng.alloc_bwd_ws(ids.shape[0], ids.shape[1], out.device)
        grads = ctx.eng.sample_bwd(ctx.dparams, ctx.st, out, ids, d_out, ctx.temperature, ctx.pretrain, ws=ws,
                                   phases=7 if ctx.has_states else 3)
        ctx.st = None
        d_h0 = d_c0 = None
        if ctx.has_states:        # gradients into the initial states: slot 0 of the recurrent input-gradient buffers
            d_h0, d_c0 = ctx.eng.state_grads(ws)
        return (None, None, None, None, None, None, d_h0, d_c0, grads[-1], *grads[:-1])


class _ForwardTfFn(torch.autograd.Function):
    """Decoder.forward with autograd (generator.py:39-53): gic_decoder_forward_tf / gic_decoder_forward_tf_bwd."""

    @staticmethod
    def forward(ctx, eng, temperature, pretrain, caps, lengths, noise_u, seed, features, *params):
        dparams = [p.detach() for p in params]
        pred, (h_n, c_n), saved = eng.forward_tf(dparams, features.detach().float(), caps, lengths, temperature, pretrain, noise_u, seed,
                                                 keep_state=True)
        ctx.eng, ctx.temperature, ctx.pretrain, ctx.saved, ctx.dparams = eng, temperature, pretrain, saved, dparams
        ctx.save_for_backward(pred)
        ctx.mark_non_differentiable(h_n, c_n)
        return pred, h_n, c_n

    @staticmethod
    def backward(ctx, d_pred, _d_h, _d_c):
        (pred,) = ctx.saved_tensors
        grads = ctx.eng.forward_tf_bwd(ctx.dparams, ctx.saved, pred, d_pred, ctx.temperature, ctx.pretrain)
        ctx.saved = None
        return (None, None, None, None, None, None, None, grads[-1], *grads[:-1])


class Decoder(nn.Module):
    """Embedding + LSTM + Linear caption decoder (generator.py:27-96)."""

    def __init__(self, args):
        super().__init__()
        self.embed = Embedding(args.vocab_size, args.gen_embed_dim)
        self.lstm = _LSTMParams(args.gen_embed_dim, args.gen_hidden_dim, args.gen_num_layers)
        self.linear = _LinearParams(args.gen_hidden_dim, args.vocab_size)
        self.max_seq_length = args.max_seq_len
        self.temperature = args.temperature          # mutated from outside (training.py:191)
        self.args = args
        self._engine: Optional[engine.DecoderEngine] = None

    def engine(self) -> engine.DecoderEngine:
        if self._engine is None:
            a = self.args
            self._engine = engine.DecoderEngine(a.vocab_size, a.gen_embed_dim, a.gen_hidden_dim, a.gen_num_layers, _compute_dtype(a))
        return self._engine

    def param_list(self) -> List[nn.Parameter]:
        ps = [self.embed.weight]
        for l in range(self.lstm.num_layers):
            ps += self.lstm.layer_params(l)
        return ps + [self.linear.weight, self.linear.bias]

    def sample(self, features, states=None, pretrain=False, max_caption_len=34, noise_u=None):
        """Greedy Gumbel-softmax roll-out (generator.py:55-81): returns (outputs [B,L,V], ids int64 [B,L]).
        Gradients flow through ``outputs`` to the decoder parameters and ``features``; never through ``ids``."""
        h0 = c0 = None
        if states is not None:          # (h0, c0), each [num_layers, B, H], as nn.LSTM takes them (generator.py:61)
            h0, c0 = states
        seed = 0 if noise_u is not None else SEEDS.next()
        return _SampleFn.apply(self.engine(), float(self.temperature), bool(pretrain), int(max_caption_len), noise_u, seed,
                               h0, c0, features, *self.param_list())

    def forward(self, features, caps, lengths, pretrain=False, noise_u=None):
        """Teacher-forced decode (generator.py:39-53): inputs [features ; embed(caps)] packed with ``lengths``; returns
        (pred [B, max(lengths), V], (h_n, c_n)) with pred = logits (pretrain) or softmax((logits + gumbel) * temperature).
        Dead on the reference's training path (training.py never calls it), kept for the module surface.  Gradients flow through
        ``pred`` to the decoder parameters and ``features`` (padded positions reach the projection's bias only, as
        pad_packed_sequence's zeros do); the returned hidden state is not differentiated.  ``noise_u`` [B, max(lengths), V]
        replaces the device draw (parity runs)."""
        seed = 0 if noise_u is not None else SEEDS.next()
        params = self.param_list()
        if not torch.is_grad_enabled() or not (features.requires_grad or any(p.requires_grad for p in params)):
            with torch.no_grad():
                return self.engine().forward_tf([p.detach() for p in params], features, caps, lengths, float(self.temperature),
                                                bool(pretrain), noise_u, seed)
        pred, h_n, c_n = _ForwardTfFn.apply(self.engine(), float(self.temperature), bool(pretrain), caps, lengths, noise_u, seed,
                                            features, *params)
        return pred, (h_n, c_n)

    def add_gumbel(self, o_t, eps=1e-10, gpu=0):
        """o_t + Gumbel(0,1) noise (generator.py:84-96); on the hot path this is fused into sample()."""
        u = torch.empty_like(o_t, dtype=torch.float32).uniform_(0, 1)
        return o_t + (-torch.log(-torch.log(u + eps) + eps))


# ------------------------------------------------------------------------------------------ visual-attention decoder
class _AttnSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, temperature, pretrain, max_len, noise_u, seed, states, fmap, features, *params):
        dparams = [p.detach() for p in params]
        out, ids, st = eng.sample_fwd(dparams, features.detach().float(), fmap.detach(), max_len, temperature, pretrain, noise_u, seed,
                                      states=states)
        ctx.eng, ctx.temperature, ctx.pretrain, ctx.st, ctx.dparams = eng, temperature, pretrain, st, dparams
        ctx.save_for_backward(out, ids)
        ctx.mark_non_differentiable(ids)
        return out, ids

    @staticmethod
    def backward(ctx, d_out, _d_ids):
        out, ids = ctx.saved_tensors
        grads = ctx.eng.sample_bwd(ctx.dparams, ctx.st, out, ids, d_out, ctx.temperature, ctx.pretrain)
        ctx.st = None
        return (None, None, None, None, None, None, None, None, grads[-1], *grads[:-1])


class _AttnParams(nn.Module):
    """Additive (Show-Attend-Tell) attention parameters: e_i = w_a . tanh(W_f a_i + b_f + W_h h)."""

    def __init__(self, feat_c: int, hidden: int, attn: int):
        super().__init__()
        k = 1.0 / math.sqrt(attn)
        self.w_f = nn.Parameter(torch.empty(attn, feat_c).uniform_(-k, k))
        self.b_f = nn.Parameter(torch.zeros(attn))
        self.w_h = nn.Parameter(torch.empty(attn, hidden).uniform_(-k, k))
        self.w_a = nn.Parameter(torch.empty(attn).uniform_(-k, k))


class AttnDecoder(nn.Module):
    """Caption decoder with soft visual attention over the trunk's feature map (``--decoder attention``, BASELINE config 4).  NO
    reference counterpart: the reference's Decoder (generator.py:27-96) with a context vector z_t = sum_i alpha_ti a_i concatenated
    to the LSTM input (oracle/cpu_attention.py).  One LSTM layer; same ``embed`` / ``lstm`` / ``linear`` state-dict keys as the
    reference's Decoder plus ``attn.*``; ``sample`` takes the feature map next to the start features."""

    def __init__(self, args, feat_c: int, positions: int):
        super().__init__()
        if args.gen_num_layers != 1:
            raise ValueError("--decoder attention supports one LSTM layer")
        if args.vocab_size % 4 or args.gen_embed_dim % 8 or args.gen_hidden_dim % 8 or feat_c % 8 or int(getattr(args, "attn_dim", 512)) % 8:
            # gic_attn_* (attention.hip check_attn_dims) would refuse these at the first step: say so at construction
            raise ValueError("--decoder attention needs vocab_size % 4 == 0 and gen_embed_dim / gen_hidden_dim / attn_dim % 8 == 0 "
                             f"(got V={args.vocab_size}, E={args.gen_embed_dim}, H={args.gen_hidden_dim}); main.py pads the vocabulary")
        self.embed = Embedding(args.vocab_size, args.gen_embed_dim)
        self.lstm = _LSTMParams(args.gen_embed_dim + feat_c, args.gen_hidden_dim, 1)
        self.linear = _LinearParams(args.gen_hidden_dim, args.vocab_size)
        self.attn = _AttnParams(feat_c, args.gen_hidden_dim, int(getattr(args, "attn_dim", 512)))
        self.max_seq_length = args.max_seq_len
        self.temperature = args.temperature
        self.args, self.feat_c, self.positions = args, feat_c, positions
        self._engine = None

    def engine(self):
        if self._engine is None:
            a = self.args
            self._engine = engine.AttnDecoderEngine(a.vocab_size, a.gen_embed_dim, a.gen_hidden_dim, self.feat_c, self.positions,
                                                    self.attn.w_a.numel(), _compute_dtype(a))
        return self._engine

    def param_list(self) -> List[nn.Parameter]:
        return [self.embed.weight] + self.lstm.layer_params(0) + [self.linear.weight, self.linear.bias, self.attn.w_f, self.attn.b_f,
                                                                   self.attn.w_h, self.attn.w_a]

    def sample(self, features, fmap=None, states=None, pretrain=False, max_caption_len=34, noise_u=None):
        """(outputs [B,L,V], ids [B,L]) as Decoder.sample; ``fmap`` [B, P, C]: the trunk's last feature map (no gradient into it)."""
        if fmap is None:
            raise ValueError("the attention decoder needs the trunk's feature map: sample(features, fmap)")
        if states is not None:          # (h0, c0), each [1, B, H] as nn.LSTM takes them (generator.py:55,61); constants of the backward pass
            states = tuple(t.detach() for t in states)
        seed = 0 if noise_u is not None else SEEDS.next()
        return _AttnSampleFn.apply(self.engine(), float(self.temperature), bool(pretrain), int(max_caption_len), noise_u, seed, states, fmap,
                                   features, *self.param_list())


# ------------------------------------------------------------------------------------------ encoder
class _EncoderHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dtype, training, momentum, eps, trunk_feat, weight, bias, gamma, beta, running_mean, running_var):
        out, saved = encoder_head_fwd(dtype, trunk_feat.detach(), weight.detach(), bias.detach(), gamma.detach(), beta.detach(),
                                      running_mean, running_var, training, momentum, eps)
        ctx.saved, ctx.dtype = saved, dtype
        ctx.w = weight.detach()
        ctx.gamma = gamma.detach()
        return out

    @staticmethod
    def backward(ctx, d_out):
        dw, db, dgamma, dbeta = encoder_head_bwd(ctx.dtype, ctx.saved, ctx.w, ctx.gamma, d_out.contiguous())
        return None, None, None, None, None, dw, db, dgamma, dbeta, None, None


class _BatchNorm1dParams(nn.Module):
    def __init__(self, num_features: int, momentum: float):
        super().__init__()
        self.num_features, self.momentum, self.eps = num_features, momentum, 1e-5
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Encoder(nn.Module):
    """ResNet trunk (frozen, forward only) -> Linear -> BatchNorm1d(momentum=0.01) (generator.py:8-25)."""

    def __init__(self, args):
        super().__init__()
        self.resnet = ResNetTrunk(getattr(args, "encoder_arch", "resnet18"))
        self.linear = _LinearParams(self.resnet.out_features, args.gen_embed_dim)
        self.bn = _BatchNorm1dParams(args.gen_embed_dim, momentum=0.01)
        self.args = args

    def forward(self, images, next_images=None):
        """generator.py:19-25.  ``next_images`` (optional, the next batch's images): their trunk forward is enqueued on the look-ahead
        stream now and picked up by the next call that is handed the same tensor."""
        with torch.no_grad():                                        # generator.py:21-22
            main = torch.cuda.current_stream(images.device)
            start = main.record_event()
            feats = self.take_trunk(images, self.training, main)
            if next_images is not None:
                self.prefetch_trunk(next_images, self.training, start)
        feats = feats.reshape(feats.size(0), -1)
        return _EncoderHeadFn.apply(_compute_dtype(self.args), self.training, self.bn.momentum, self.bn.eps, feats,
                                    self.linear.weight, self.linear.bias, self.bn.weight, self.bn.bias,
                                    self.bn.running_mean, self.bn.running_var)

    def forward_with_map(self, images, next_images=None):
        """(features [B,E] as forward(), feature map [B, P, C] = the trunk's last activation, detached) for the attention decoder.
        ``next_images``: as in forward() -- the look-ahead pass also keeps a copy of its feature map."""
        with torch.no_grad():
            main = torch.cuda.current_stream(images.device)
            start = main.record_event()
            pre, self._pre = getattr(self, "_pre", None), None
            fmap = None
            if pre is not None:
                main.wait_event(pre[3])             # the look-ahead pass (used or not: it shares the plan's buffers)
                if pre[0] is images and pre[1] == bool(self.training) and pre[4] is not None:
                    feats, fmap = pre[2], pre[4]
                    feats.record_stream(main)
                    fmap.record_stream(main)
            if fmap is None:
                busy = getattr(self, "_busy", None)
                if busy is not None:
                    main.wait_event(busy)
                feats = self.trunk_features(images, self.training).clone()      # a pass here: the map is the plan's live buffer
                fmap = self.resnet._plan.last_map(images.shape[0], images.shape[2]).clone()
                self._busy = main.record_event()
            if next_images is not None:
                self.prefetch_trunk(next_images, self.training, start, want_map=True)
        feats = feats.reshape(feats.size(0), -1)
        out = _EncoderHeadFn.apply(_compute_dtype(self.args), self.training, self.bn.momentum, self.bn.eps, feats,
                                   self.linear.weight, self.linear.bias, self.bn.weight, self.bn.bias,
                                   self.bn.running_mean, self.bn.running_var)
        return out, fmap.view(fmap.shape[0], -1, fmap.shape[-1])

    # ---- trunk look-ahead: the trunk is frozen (generator.py:21), so the pass for the NEXT batch depends on nothing the current
    # step updates; it runs on its own stream under the step's launch-bound phases and hands over a private copy of its output.
    def prefetch_trunk(self, images, training: bool, after, mark=None, want_map: bool = False) -> None:
        if getattr(self, "_s_pre", None) is None:
            self._s_pre = torch.cuda.Stream(device=images.device)
        s = self._s_pre
        with engine.on_stream(s):
            s.wait_event(after)                 # `images` is ready
            busy, self._busy = getattr(self, "_busy", None), None
            if busy is not None:
                # a synchronous pass on another stream (cold step, mispredicted look-ahead) is still using the plan's
                # shared buffers (packed image, statistics arena, activations): this pass starts behind it
                s.wait_event(busy)
            if mark is not None:
                mark("trunk prefetch start [s_pre]", s)
            feats = self.trunk_features(images, training).clone()
            fmap = self.resnet._plan.last_map(images.shape[0], images.shape[2]).clone() if want_map else None
            done = s.record_event()
            if mark is not None:
                mark("trunk prefetch done [s_pre]", s)
        images.record_stream(s)
        self._pre = (images, bool(training), feats, done, fmap)

    def take_trunk(self, images, training: bool, stream):
        """Trunk features of ``images`` on ``stream``: the prefetched copy if this very tensor was announced, else a pass now."""
        pre, self._pre = getattr(self, "_pre", None), None
        if pre is not None:
            stream.wait_event(pre[3])           # also orders a synchronous pass behind an unused look-ahead (shared buffers)
            if pre[0] is images and pre[1] == bool(training):
                pre[2].record_stream(stream)
                return pre[2]
        # synchronous pass: hand out a private copy (the plan's own output buffer is overwritten by the next pass, which may
        # run on the look-ahead stream) and remember where this pass ends for prefetch_trunk
        busy = getattr(self, "_busy", None)
        if busy is not None:
            stream.wait_event(busy)             # an earlier synchronous pass on some other stream
        feats = self.trunk_features(images, training).clone()
        self._busy = stream.record_event()
        return feats

    def take_trunk_with_map(self, images, training: bool, stream):
        """(trunk features, the trunk's last feature map [N, h, w, C]) of ``images`` on ``stream`` for the attention decoder: the
        look-ahead pass's private copies if this very tensor was announced (prefetch_trunk(want_map=True)), else a pass now."""
        pre, self._pre = getattr(self, "_pre", None), None
        if pre is not None:
            stream.wait_event(pre[3])           # also orders a synchronous pass behind an unused look-ahead (shared buffers)
            if pre[0] is images and pre[1] == bool(training) and pre[4] is not None:
                pre[2].record_stream(stream)
                pre[4].record_stream(stream)
                return pre[2], pre[4]
        busy = getattr(self, "_busy", None)
        if busy is not None:
            stream.wait_event(busy)
        feats = self.trunk_features(images, training).clone()
        fmap = self.resnet._plan.last_map(images.shape[0], images.shape[2]).clone()
        self._busy = stream.record_event()
        return feats, fmap

    # ---- direct (no autograd) forms used by the fused step driver
    def trunk_features(self, images, training: bool):
        """The frozen trunk alone (generator.py:20-22): pooled features in the compute dtype, in the plan's own buffer."""
        return self.resnet(images, _compute_dtype(self.args), training)

    def forward_fused(self, images, training: bool, trunk_feats=None):
        dt = _compute_dtype(self.args)
        feats = trunk_feats if trunk_feats is not None else self.resnet(images, dt, training)
        out, self._saved = encoder_head_fwd(dt, feats, self.linear.weight.detach(), self.linear.bias.detach(),
                                            self.bn.weight.detach(), self.bn.bias.detach(), self.bn.running_mean,
                                            self.bn.running_var, training, self.bn.momentum, self.bn.eps)
        return out

    def backward_fused(self, d_feat):
        """Gradients of the head into the .grad views of its four parameters (the trunk is frozen)."""
        grads = (self.linear.weight.grad, self.linear.bias.grad, self.bn.weight.grad, self.bn.bias.grad)
        encoder_head_bwd(_compute_dtype(self.args), self._saved, self.linear.weight.detach(), self.bn.weight.detach(), d_feat,
                         grads=grads)
        self._saved = None


class Generator(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.encoder = Encoder(args)
        if getattr(args, "decoder", "lstm") == "attention":
            if int(args.conditional_gan) != 1:
                raise ValueError("--decoder attention attends over image features: it needs --conditional-gan 1")
            side = int(getattr(args, "image_size", 224)) // 32
            self.decoder = AttnDecoder(args, self.encoder.resnet.out_features, side * side)
        else:
            self.decoder = Decoder(args)
        self.args = args
        self.init_params()

    def forward(self, images, caps, lengths, pretrain=False):
        if self.args.conditional_gan:
            features = self.encoder(images)
        else:
            features = self.decoder.embed(torch.ones(len(images), dtype=torch.long, device=images.device))
        return self.decoder(features, caps, lengths, pretrain)

    def init_params(self):
        """generator.py:116-123: every parameter with >= 1 dim (biases, BN affine and trunk convs included)."""
        for param in self.parameters():
            if param.requires_grad and len(param.shape) > 0:
                if self.args.gen_init == "uniform":
                    torch.nn.init.uniform_(param, a=-0.05, b=0.05)
                elif self.args.gen_init == "normal":
                    torch.nn.init.normal_(param, std=1 / math.sqrt(param.shape[0]))
