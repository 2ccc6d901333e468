"""Plain-torch CPU restatement of the reference's adversarial train step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style: weights are
dicts keyed by the reference's state-dict names, all randomness (Gumbel
uniforms ``u``, dropout keep-masks) is an explicit input, so the HIP path can be
compared on identical inputs.

Every function cites the reference lines it restates (paths relative to
/root/reference/).  Nothing here imports the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

GUMBEL_EPS = 1e-10          # src/generator.py:84 (eps default)
DROPOUT_P = 0.2             # src/discriminator.py:10 (dropout=0.2)
ADAM_BETAS = (0.9, 0.999)   # torch.optim.Adam defaults, src/training.py:24-26
ADAM_EPS = 1e-8


# --------------------------------------------------------------------------
# Generator: Decoder.sample (src/generator.py:55-96)
# --------------------------------------------------------------------------
def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor,
              b_ih: Tensor, b_hh: Tensor) -> Tuple[Tensor, Tensor]:
    """One nn.LSTM layer step, PyTorch gate order i,f,g,o (src/generator.py:32,61)."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    hid = h.shape[1]
    i = torch.sigmoid(gates[:, 0 * hid:1 * hid])
    f = torch.sigmoid(gates[:, 1 * hid:2 * hid])
    g = torch.tanh(gates[:, 2 * hid:3 * hid])
    o = torch.sigmoid(gates[:, 3 * hid:4 * hid])
    c_new = f * c + i * g
    h_new = o * torch.tanh(c_new)
    return h_new, c_new


def gumbel_from_uniform(u: Tensor, eps: float = GUMBEL_EPS) -> Tensor:
    """g = -log(-log(u+eps)+eps)  (src/generator.py:91)."""
    return -torch.log(-torch.log(u + eps) + eps)


def num_lstm_layers(gp: Params, prefix: str = "decoder.") -> int:
    n = 0
    while f"{prefix}lstm.weight_ih_l{n}" in gp:
        n += 1
    return n


def decoder_sample(gp: Params, features: Tensor, max_caption_len: int,
                   temperature: float, us: Optional[Sequence[Tensor]] = None,
                   pretrain: bool = False, prefix: str = "decoder.",
                   force_ids: Optional[Tensor] = None, states: Optional[Tuple[Tensor, Tensor]] = None,
                   force_len: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Decoder.sample (src/generator.py:55-81).

    ``us[t]`` is the U[0,1) draw of step t (shape [B,V]) that the reference
    takes from the global generator at src/generator.py:86-90.  Returns
    (outputs [B,L,V], ids int64 [B,L]).  In adversarial mode outputs are
    softmax((o+g)*T); in pretrain mode raw logits (src/generator.py:63-66).
    The next input is embed(argmax) with the index detached (:73-76).

    ``force_ids`` [B,L] (test aid, no reference counterpart): the trajectory to follow instead of the argmax -- the index is
    detached in the reference (:75), so the outputs' gradient on a GIVEN trajectory is the same function; used to compare
    gradients with a bf16 run whose argmax differs at a near-tie; ``force_len`` [B] restricts the forcing of row b to its first
    force_len[b] steps (prefix of a Monte-Carlo roll-out).  ``states`` = (h0, c0), each [num_layers, B, H] (:55,61).
    """
    nl = num_lstm_layers(gp, prefix)
    bsz = features.shape[0]
    hid = gp[f"{prefix}lstm.weight_hh_l0"].shape[1]
    if states is None:
        h = [features.new_zeros(bsz, hid) for _ in range(nl)]
        c = [features.new_zeros(bsz, hid) for _ in range(nl)]
    else:
        h = [states[0][l] for l in range(nl)]
        c = [states[1][l] for l in range(nl)]
    x = features
    outs: List[Tensor] = []
    ids: List[Tensor] = []
    for t in range(max_caption_len):
        inp = x
        for l in range(nl):
            h[l], c[l] = lstm_cell(inp, h[l], c[l],
                                   gp[f"{prefix}lstm.weight_ih_l{l}"], gp[f"{prefix}lstm.weight_hh_l{l}"],
                                   gp[f"{prefix}lstm.bias_ih_l{l}"], gp[f"{prefix}lstm.bias_hh_l{l}"])
            inp = h[l]
        o = inp @ gp[f"{prefix}linear.weight"].t() + gp[f"{prefix}linear.bias"]
        if pretrain:
            outs.append(o)
            pred = torch.softmax(o, dim=-1)
        else:
            g = gumbel_from_uniform(us[t])
            pred = torch.softmax((o + g) * temperature, dim=-1)
            outs.append(pred)
        idx = pred.max(1)[1]                       # first maximal index
        if force_ids is not None:
            idx = force_ids[:, t] if force_len is None else torch.where(force_len > t, force_ids[:, t], idx)
        ids.append(idx)
        x = gp[f"{prefix}embed.weight"][idx.detach()]
    return torch.stack(outs, 1), torch.stack(ids, 1)


def decoder_forward_tf(gp: Params, features: Tensor, caps: Tensor, lengths: Sequence[int], temperature: float,
                       pretrain: bool = False, u: Optional[Tensor] = None, prefix: str = "decoder."
                       ) -> Tuple[Tensor, Tuple[Tensor, Tensor]]:
    """Decoder.forward, the teacher-forced decode (src/generator.py:39-53).

    Inputs are [features ; embed(caps)] (L+1 steps, :41-42) packed with ``lengths`` (:43): a sequence takes part in step t
    only while t < its length; afterwards its state is frozen (the returned hidden is the state at ITS last step) and its
    padded outputs are zero (pad_packed_sequence, :45), so the projection sees zeros there (pred = bias, or its softmax).
    ``u`` [B, max(lengths), V]: the single uniform_(0,1) draw of add_gumbel over the whole logits tensor (:50, :86-90).
    Returns (pred [B, max(lengths), V], (h_n, c_n) each [num_layers, B, H])."""
    nl = num_lstm_layers(gp, prefix)
    bsz = features.shape[0]
    hid = gp[f"{prefix}lstm.weight_hh_l0"].shape[1]
    emb = gp[f"{prefix}embed.weight"][caps]                      # [B, L, E]
    xs = torch.cat((features.unsqueeze(1), emb), 1)              # [B, L+1, E]
    lens = torch.as_tensor(list(lengths), dtype=torch.long)
    tmax = int(lens.max())
    h = [features.new_zeros(bsz, hid) for _ in range(nl)]
    c = [features.new_zeros(bsz, hid) for _ in range(nl)]
    outs = []
    for t in range(tmax):
        live = (lens > t).unsqueeze(1)
        inp = xs[:, t]
        for l in range(nl):
            hn, cn = lstm_cell(inp, h[l], c[l], gp[f"{prefix}lstm.weight_ih_l{l}"], gp[f"{prefix}lstm.weight_hh_l{l}"],
                               gp[f"{prefix}lstm.bias_ih_l{l}"], gp[f"{prefix}lstm.bias_hh_l{l}"])
            h[l] = torch.where(live, hn, h[l])
            c[l] = torch.where(live, cn, c[l])
            inp = h[l]
        outs.append(torch.where(live, inp, torch.zeros_like(inp)))
    output = torch.stack(outs, 1)                                # [B, tmax, H]
    o = output @ gp[f"{prefix}linear.weight"].t() + gp[f"{prefix}linear.bias"]
    if pretrain:
        pred = o
    else:
        pred = torch.softmax((o + gumbel_from_uniform(u)) * temperature, dim=-1)
    return pred, (torch.stack(h, 0), torch.stack(c, 0))


def start_features(gp: Params, bsz: int, prefix: str = "decoder.") -> Tensor:
    """cgan=0 start feature = embed(<S>=1) broadcast (src/training.py:147)."""
    idx = torch.ones(bsz, dtype=torch.long)
    return gp[f"{prefix}embed.weight"][idx]


# --------------------------------------------------------------------------
# Encoder head: Linear + BatchNorm1d(momentum=0.01) (src/generator.py:15-16,24)
# --------------------------------------------------------------------------
def encoder_head(gp: Params, trunk_feat: Tensor, training: bool = True,
                 running: Optional[Dict[str, Tensor]] = None, eps: float = 1e-5,
                 momentum: float = 0.01) -> Tensor:
    y = trunk_feat @ gp["encoder.linear.weight"].t() + gp["encoder.linear.bias"]
    if training:
        mean = y.mean(0)
        var = y.var(0, unbiased=False)
        if running is not None:
            n = y.shape[0]
            with torch.no_grad():
                running["running_mean"].mul_(1 - momentum).add_(momentum * mean)
                running["running_var"].mul_(1 - momentum).add_(momentum * var * n / max(n - 1, 1))
    else:
        mean, var = running["running_mean"], running["running_var"]
    return (y - mean) / torch.sqrt(var + eps) * gp["encoder.bn.weight"] + gp["encoder.bn.bias"]


# --------------------------------------------------------------------------
# Discriminator.forward (src/discriminator.py:34-62)
# --------------------------------------------------------------------------
def disc_num_convs(dp: Params) -> int:
    n = 0
    while f"convs.{n}.weight" in dp:
        n += 1
    return n


def disc_forward(dp: Params, inp: Tensor, keep_mask: Optional[Tensor] = None,
                 num_rep: int = 64, return_stages: bool = False, dropout_p: float = DROPOUT_P):
    """inp [B,L,V] float -> logits [B*num_rep].

    keep_mask: 0/1 tensor [B*num_rep, F] (dropout keep mask, train mode,
    src/discriminator.py:30,58) or None for eval mode.  dropout_p: the constructor's
    ``dropout`` argument (src/discriminator.py:10; nn.Dropout scales kept values by 1/(1-p)).
    """
    bsz, seqlen, _ = inp.shape
    emb = inp @ dp["embeddings.weight"].t()                        # :40  [B,L,De]
    emb_dim = emb.shape[-1]
    s = emb_dim // num_rep                                         # emb_dim_single, :17
    emb_r = emb.reshape(bsz, seqlen, num_rep, s)
    pools = []
    for k in range(disc_num_convs(dp)):
        w = dp[f"convs.{k}.weight"]                                # [n,1,f,s]  :22-25
        b = dp[f"convs.{k}.bias"]
        f = w.shape[2]
        win = emb_r.unfold(1, f, 1)                                # [B, L-f+1, R, s, f]
        con = torch.einsum("btrej,cje->bctr", win, w[:, 0]) + b[None, :, None, None]
        con = torch.relu(con)                                      # :42
        pools.append(con.max(dim=2)[0])                            # :45  [B,n,R]
    pred = torch.cat(pools, 1)                                     # :49  [B,F,R]
    feat_dim = pred.shape[1]
    pooled = pred.permute(0, 2, 1).reshape(-1, feat_dim)           # :51  row = b*R + r
    hw = pooled @ dp["highway.weight"].t() + dp["highway.bias"]    # :53
    sig = torch.sigmoid(hw)
    hwout = sig * torch.relu(hw) + (1.0 - sig) * pooled            # :55
    dropped = hwout if keep_mask is None else hwout * (keep_mask / (1.0 - dropout_p))
    feat = dropped @ dp["feature2out.weight"].t() + dp["feature2out.bias"]   # :58
    logits = (feat @ dp["out2logits.weight"].t() + dp["out2logits.bias"]).squeeze(1)  # :60
    if return_stages:
        return logits, {"emb": emb, "pooled": pooled, "highway": hwout, "feat": feat}
    return logits


# --------------------------------------------------------------------------
# get_losses (src/utils.py:10-53)
# --------------------------------------------------------------------------
def _softplus(z: Tensor) -> Tensor:
    return torch.clamp(z, min=0) + torch.log1p(torch.exp(-z.abs()))


def bce_logits_mean(x: Tensor, target_one: bool) -> Tensor:
    """nn.BCEWithLogitsLoss() with an all-ones / all-zeros target, mean reduction."""
    return _softplus(-x).mean() if target_one else _softplus(x).mean()


def get_losses(d_out_real: Tensor, d_out_fake: Tensor, g_out: Tensor,
               loss_type: str = "JS") -> Tuple[Tensor, Tensor]:
    """Returns (g_loss, d_loss), argument/return order of src/utils.py:10,53.

    'hinge' and 'tv' raise TypeError in the reference (utils.py:36-37,43-44:
    nn.ReLU / nn.Tanh constructed with a tensor); they are restated here with
    the evident intent (relu / tanh applied elementwise) and have no reference
    vector.
    """
    if loss_type in ("standard", "JS", "KL"):
        d_loss = bce_logits_mean(d_out_real, True) + bce_logits_mean(d_out_fake, False)
        if loss_type == "standard":
            g_loss = bce_logits_mean(g_out, True)                  # :19
        elif loss_type == "JS":
            g_loss = -bce_logits_mean(g_out, False)                # :26
        else:
            g_loss = torch.mean(-g_out)                            # :33
    elif loss_type == "hinge":
        d_loss = torch.relu(1.0 - d_out_real).mean() + torch.relu(1.0 + d_out_fake).mean()
        g_loss = -g_out.mean()
    elif loss_type == "tv":
        d_loss = torch.mean(torch.tanh(d_out_fake) - torch.tanh(d_out_real))
        g_loss = torch.mean(-torch.tanh(g_out))
    elif loss_type == "rsgan":
        d_loss = bce_logits_mean(d_out_real - d_out_fake, True)    # :47
        g_loss = bce_logits_mean(d_out_fake - d_out_real, True)    # :48
    else:
        raise NotImplementedError("Divergence '%s' is not implemented" % loss_type)
    return g_loss, d_loss


def get_fixed_temperature(temper: float, i: float, N: float, adapt: str) -> float:
    """src/utils.py:55-76 (python/numpy float64 arithmetic)."""
    if adapt == "no":
        return 1.0
    if adapt == "lin":
        return 1 + i / (N - 1) * (temper - 1)
    if adapt == "exp":
        return temper ** (i / N)
    if adapt == "log":
        return 1 + (temper - 1) / math.log(N) * math.log(i + 1)
    if adapt == "sigmoid":
        return (temper - 1) * 1 / (1 + math.exp((N / 2 - i) * 20 / N)) + 1
    if adapt == "quad":
        return (temper - 1) / (N - 1) ** 2 * i ** 2 + 1
    if adapt == "sqrt":
        return (temper - 1) / math.sqrt(N - 1) * math.sqrt(i) + 1
    raise Exception("Unknown adapt type!")


# --------------------------------------------------------------------------
# optimize(): clip_grad_norm_ + Adam (src/training.py:194-199, :24-26)
# --------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tuple[Dict[str, Tensor], float]:
    """torch.nn.utils.clip_grad_norm_: global L2 norm, coef = max/(norm+1e-6) clamped to 1."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    coef = min(1.0, max_norm / (total + 1e-6))
    return {k: g * coef for k, g in grads.items()}, total


class AdamState:
    """Per-optimizer moment state; params with no gradient are skipped (never touched)."""

    def __init__(self, lr: float):
        self.lr = lr
        self.m: Dict[str, Tensor] = {}
        self.v: Dict[str, Tensor] = {}
        self.t: Dict[str, int] = {}

    def step(self, params: Params, grads: Dict[str, Tensor]) -> None:
        b1, b2 = ADAM_BETAS
        for k, g in grads.items():
            if k not in self.m:
                self.m[k] = torch.zeros_like(params[k])
                self.v[k] = torch.zeros_like(params[k])
                self.t[k] = 0
            self.t[k] += 1
            t = self.t[k]
            self.m[k] = b1 * self.m[k] + (1 - b1) * g
            self.v[k] = b2 * self.v[k] + (1 - b2) * g * g
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            denom = self.v[k].sqrt() / math.sqrt(bc2) + ADAM_EPS
            params[k] = params[k] - (self.lr / bc1) * self.m[k] / denom


# --------------------------------------------------------------------------
# The step (body of src/training.py:136-183)
# --------------------------------------------------------------------------
def adv_step(gp: Params, dp: Params, captions: Tensor, us: Sequence[Tensor],
             masks: Optional[Sequence[Tensor]], temperature: float,
             loss_type: str = "standard", clip_norm: float = 5.0,
             gen_opt: Optional[AdamState] = None, disc_opt: Optional[AdamState] = None,
             trunk_feat: Optional[Tensor] = None, num_rep: int = 64,
             bn_running: Optional[Dict[str, Tensor]] = None, train: bool = True,
             force_ids: Optional[Tensor] = None, dropout_p: float = DROPOUT_P) -> Dict[str, object]:
    """One adversarial G+D step with the fixed order: both backward passes on
    pre-update weights, then both optimizer steps (the literal order of
    src/training.py:168-169 cannot run on torch >= 1.5).

    gp/dp are updated IN PLACE (dict values replaced) when optimizers are given.
    masks = (mask_real, mask_fake, mask_gen) keep-masks, drawn in that order by
    the reference (src/training.py:162-164); None = eval mode.
    trunk_feat: [B,feat] ResNet trunk output for --conditional-gan 1, else None.
    """
    bsz, seqlen = captions.shape
    vocab = gp["decoder.linear.weight"].shape[0]
    g_names = [k for k in gp if k.startswith("decoder.") or
               (trunk_feat is not None and k.startswith(("encoder.linear.", "encoder.bn.")))]
    g_leaf = {k: gp[k].detach().clone().requires_grad_(train) for k in g_names}
    d_leaf = {k: v.detach().clone().requires_grad_(train) for k, v in dp.items()}

    if trunk_feat is not None:
        feats = encoder_head(g_leaf, trunk_feat, training=train, running=bn_running)
    else:
        feats = start_features(g_leaf, bsz)
    gen, ids = decoder_sample(g_leaf, feats, seqlen, temperature, us, force_ids=force_ids)
    real = torch.nn.functional.one_hot(captions, vocab).float()       # training.py:158
    m = masks if masks is not None else (None, None, None)
    d_real, st_real = disc_forward(d_leaf, real, m[0], num_rep, return_stages=True, dropout_p=dropout_p)
    d_fake, st_fake = disc_forward(d_leaf, gen.detach(), m[1], num_rep, return_stages=True, dropout_p=dropout_p)
    g_out, st_gen = disc_forward(d_leaf, gen, m[2], num_rep, return_stages=True, dropout_p=dropout_p)
    g_loss, d_loss = get_losses(d_real, d_fake, g_out, loss_type)
    out: Dict[str, object] = {
        "probs": gen.detach(), "ids": ids, "d_real": d_real.detach(), "d_fake": d_fake.detach(),
        "g_out": g_out.detach(), "g_loss": float(g_loss.detach()), "d_loss": float(d_loss.detach()),
        "stages": {"real": st_real, "fake": st_fake, "gen": st_gen},
    }
    if not train:
        return out
    d_grads = dict(zip(d_leaf, torch.autograd.grad(d_loss, list(d_leaf.values()), retain_graph=True)))
    g_grads_t = torch.autograd.grad(g_loss, list(g_leaf.values()), allow_unused=True)
    g_grads = {k: g for k, g in zip(g_leaf, g_grads_t) if g is not None}
    out["d_grads_raw"], out["g_grads_raw"] = d_grads, g_grads
    d_grads, d_norm = clip_grad_norm(d_grads, clip_norm)
    g_grads, g_norm = clip_grad_norm(g_grads, clip_norm)
    out.update(d_grads=d_grads, g_grads=g_grads, d_norm=d_norm, g_norm=g_norm)
    if disc_opt is not None:
        disc_opt.step(dp, d_grads)
    if gen_opt is not None:
        gen_opt.step(gp, g_grads)
    return out


# --------------------------------------------------------------------------
# MLE pre-train step (src/training.py:53-95; generator.py:63-66)
# --------------------------------------------------------------------------
def pretrain_step(gp: Params, captions: Tensor, clip_norm: float = 5.0,
                  opt: Optional[AdamState] = None, trunk_feat: Optional[Tensor] = None,
                  train: bool = True) -> Dict[str, object]:
    """Free-running greedy decode, CrossEntropyLoss over ALL B*L positions incl.
    PAD, no ignore_index (src/training.py:81-83)."""
    bsz, seqlen = captions.shape
    g_names = [k for k in gp if k.startswith("decoder.") or
               (trunk_feat is not None and k.startswith(("encoder.linear.", "encoder.bn.")))]
    g_leaf = {k: gp[k].detach().clone().requires_grad_(train) for k in g_names}
    feats = encoder_head(g_leaf, trunk_feat) if trunk_feat is not None else start_features(g_leaf, bsz)
    logits, ids = decoder_sample(g_leaf, feats, seqlen, 1.0, None, pretrain=True)
    flat = logits.reshape(-1, logits.shape[-1])
    lse = torch.logsumexp(flat, dim=1)
    nll = lse - flat.gather(1, captions.reshape(-1, 1)).squeeze(1)
    loss = nll.mean()
    out: Dict[str, object] = {"loss": float(loss.detach()), "ids": ids, "logits": logits.detach()}
    if not train:
        return out
    grads_t = torch.autograd.grad(loss, list(g_leaf.values()), allow_unused=True)
    grads = {k: g for k, g in zip(g_leaf, grads_t) if g is not None}
    out["g_grads_raw"] = grads
    grads, norm = clip_grad_norm(grads, clip_norm)
    out.update(g_grads=grads, g_norm=norm)
    if opt is not None:
        opt.step(gp, grads)
    return out


# --------------------------------------------------------------------------
# Synthetic inputs
# --------------------------------------------------------------------------
def init_uniform(shape, gen: torch.Generator, a: float = -0.05, b: float = 0.05) -> Tensor:
    return torch.empty(shape).uniform_(a, b, generator=gen)


def make_gen_params(vocab: int, embed: int, hidden: int, layers: int, gen: torch.Generator,
                    trunk_feat_dim: Optional[int] = None) -> Params:
    """U(-0.05,0.05) for every trainable tensor (generator.py:116-123, 'uniform')."""
    p: Params = {"decoder.embed.weight": init_uniform((vocab, embed), gen)}
    for l in range(layers):
        din = embed if l == 0 else hidden
        p[f"decoder.lstm.weight_ih_l{l}"] = init_uniform((4 * hidden, din), gen)
        p[f"decoder.lstm.weight_hh_l{l}"] = init_uniform((4 * hidden, hidden), gen)
        p[f"decoder.lstm.bias_ih_l{l}"] = init_uniform((4 * hidden,), gen)
        p[f"decoder.lstm.bias_hh_l{l}"] = init_uniform((4 * hidden,), gen)
    p["decoder.linear.weight"] = init_uniform((vocab, hidden), gen)
    p["decoder.linear.bias"] = init_uniform((vocab,), gen)
    if trunk_feat_dim is not None:
        p["encoder.linear.weight"] = init_uniform((embed, trunk_feat_dim), gen)
        p["encoder.linear.bias"] = init_uniform((embed,), gen)
        p["encoder.bn.weight"] = init_uniform((embed,), gen)
        p["encoder.bn.bias"] = init_uniform((embed,), gen)
    return p


def make_disc_params(vocab: int, gen: torch.Generator, embed_dim: int = 64, num_rep: int = 64,
                     filter_sizes=(3, 4, 5), num_filters=(300, 300, 300)) -> Params:
    """discriminator.py:20-29 shapes, U(-0.05,0.05) (discriminator.py:79-86)."""
    s = embed_dim // num_rep
    feat = sum(num_filters)
    p: Params = {"embeddings.weight": init_uniform((embed_dim, vocab), gen)}
    for k, (n, f) in enumerate(zip(num_filters, filter_sizes)):
        p[f"convs.{k}.weight"] = init_uniform((n, 1, f, s), gen)
        p[f"convs.{k}.bias"] = init_uniform((n,), gen)
    p["highway.weight"] = init_uniform((feat, feat), gen)
    p["highway.bias"] = init_uniform((feat,), gen)
    p["feature2out.weight"] = init_uniform((100, feat), gen)
    p["feature2out.bias"] = init_uniform((100,), gen)
    p["out2logits.weight"] = init_uniform((1, 100), gen)
    p["out2logits.bias"] = init_uniform((1,), gen)
    return p


def make_captions(bsz: int, seqlen: int, vocab: int, gen: torch.Generator) -> Tensor:
    """row = [1] + randint(4,V) + [2], full length (tasks.py:155 layout, no PAD)."""
    body = torch.randint(4, vocab, (bsz, seqlen - 2), generator=gen)
    return torch.cat([torch.ones(bsz, 1, dtype=torch.long), body,
                      torch.full((bsz, 1), 2, dtype=torch.long)], 1)


def make_noise(bsz: int, seqlen: int, vocab: int, feat: int, num_rep: int,
               gen: torch.Generator) -> Tuple[List[Tensor], List[Tensor]]:
    us = [torch.empty(bsz, vocab).uniform_(0, 1, generator=gen) for _ in range(seqlen)]
    masks = [torch.empty(bsz * num_rep, feat).bernoulli_(1 - DROPOUT_P, generator=gen) for _ in range(3)]
    return us, masks
