"""CPU oracle of the visual-attention caption decoder (BASELINE.json config 4).

TEST INFRASTRUCTURE (see oracle/__init__.py).  NO REFERENCE COUNTERPART: the reference's decoder sees the image only through the
pooled feature fed as the first LSTM input (src/generator.py:19-25, 55-81); its report (NLP_Report.pdf p.4, 4.2) describes
attention over grid features for a transformer that is not in the code drop.  This is the build's own definition: the reference's
roll-out loop (src/generator.py:55-81: Gumbel-softmax relaxation, greedy argmax feedback, temperature multiplied in) with a
Show-Attend-Tell soft attention (Xu et al., ICML 2015, eq. 4-6, 13) over the trunk's feature MAP inserted in front of the LSTM:

    a_i            the trunk's last feature map [B, P = h*w, C] (frozen, src/generator.py:21: no gradient into the trunk)
    fp_i = W_f a_i + b_f                                   [A]   (computed once per caption)
    e_ti = w_a . tanh(fp_i + W_h h_{t-1})                         additive attention energy
    alpha_t = softmax_i(e_t);   z_t = sum_i alpha_ti a_i   [C]   context vector
    (h_t, c_t) = LSTMCell([x_t ; z_t], (h_{t-1}, c_{t-1}))        x_0 = features (encoder head), x_t = embed(argmax_{t-1})
    o_t = W_out h_t + b_out;  p_t = softmax((o_t + gumbel(u_t)) * T)      exactly src/generator.py:68-76

Parameters added to the reference's decoder: decoder.attn.w_f [A,C], .b_f [A], .w_h [A,H], .w_a [A]; decoder.lstm.weight_ih_l0 is
[4H, E + C].  One LSTM layer.  Differentiable plain torch: autograd of this file is the gradient oracle.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import cpu_step as O

Tensor = torch.Tensor


def make_attn_params(vocab: int, embed: int, hidden: int, feat_c: int, attn: int, gen: torch.Generator) -> O.Params:
    u = O.init_uniform
    return {
        "decoder.embed.weight": u((vocab, embed), gen),
        "decoder.lstm.weight_ih_l0": u((4 * hidden, embed + feat_c), gen), "decoder.lstm.weight_hh_l0": u((4 * hidden, hidden), gen),
        "decoder.lstm.bias_ih_l0": u((4 * hidden,), gen), "decoder.lstm.bias_hh_l0": u((4 * hidden,), gen),
        "decoder.linear.weight": u((vocab, hidden), gen), "decoder.linear.bias": u((vocab,), gen),
        "decoder.attn.w_f": u((attn, feat_c), gen), "decoder.attn.b_f": u((attn,), gen),
        "decoder.attn.w_h": u((attn, hidden), gen), "decoder.attn.w_a": u((attn,), gen),
    }


def attention(gp: O.Params, fmap: Tensor, fproj: Tensor, h_prev: Tensor, prefix: str = "decoder.") -> Tuple[Tensor, Tensor]:
    """(context z [B,C], weights alpha [B,P]) for one step."""
    hp = h_prev @ gp[f"{prefix}attn.w_h"].t()                                   # [B,A]
    e = torch.tanh(fproj + hp.unsqueeze(1)) @ gp[f"{prefix}attn.w_a"]           # [B,P]
    alpha = torch.softmax(e, dim=1)
    z = (alpha.unsqueeze(2) * fmap).sum(1)
    return z, alpha


def attn_decoder_sample(gp: O.Params, features: Tensor, fmap: Tensor, max_caption_len: int, temperature: float,
                        us: Optional[Sequence[Tensor]] = None, pretrain: bool = False, prefix: str = "decoder.",
                        force_ids: Optional[Tensor] = None, states: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Returns (outputs [B,L,V], ids [B,L], alphas [B,L,P]); see the module docstring.  ``states`` = (h0, c0), each [1, B, H]
    (the `states` argument of the reference's Decoder.sample, src/generator.py:55,61); None = zeros."""
    bsz = features.shape[0]
    hid = gp[f"{prefix}lstm.weight_hh_l0"].shape[1]
    fproj = fmap @ gp[f"{prefix}attn.w_f"].t() + gp[f"{prefix}attn.b_f"]        # [B,P,A]
    h = features.new_zeros(bsz, hid) if states is None else states[0].reshape(bsz, hid)
    c = features.new_zeros(bsz, hid) if states is None else states[1].reshape(bsz, hid)
    x = features
    outs: List[Tensor] = []
    ids: List[Tensor] = []
    alphas: List[Tensor] = []
    for t in range(max_caption_len):
        z, alpha = attention(gp, fmap, fproj, h, prefix)
        h, c = O.lstm_cell(torch.cat([x, z], 1), h, c, gp[f"{prefix}lstm.weight_ih_l0"], gp[f"{prefix}lstm.weight_hh_l0"],
                           gp[f"{prefix}lstm.bias_ih_l0"], gp[f"{prefix}lstm.bias_hh_l0"])
        o = h @ gp[f"{prefix}linear.weight"].t() + gp[f"{prefix}linear.bias"]
        if pretrain:
            outs.append(o)
            pred = torch.softmax(o, dim=-1)
        else:
            pred = torch.softmax((o + O.gumbel_from_uniform(us[t])) * temperature, dim=-1)
            outs.append(pred)
        idx = pred.max(1)[1]
        if force_ids is not None:
            idx = force_ids[:, t]
        ids.append(idx)
        alphas.append(alpha)
        x = gp[f"{prefix}embed.weight"][idx.detach()]
    return torch.stack(outs, 1), torch.stack(ids, 1), torch.stack(alphas, 1)


def attn_adv_step(gp: O.Params, dp: O.Params, captions: Tensor, us: Sequence[Tensor], masks: Optional[Sequence[Tensor]], temperature: float,
                  trunk_feat: Tensor, fmap: Tensor, loss_type: str = "standard", num_rep: int = 64,
                  force_ids: Optional[Tensor] = None, clip_norm: float = 5.0, gen_opt: Optional[O.AdamState] = None,
                  disc_opt: Optional[O.AdamState] = None) -> dict:
    """The adversarial G+D step of oracle/cpu_step.adv_step (body of reference src/training.py:144-169 in the fixed order) with the
    attention decoder as the sampler: features = encoder head(trunk_feat) -> attn_decoder_sample over ``fmap`` [B,P,C] -> the three
    discriminator passes -> losses -> raw gradients of D (from d_loss) and of G's trainable tensors (decoder incl. attention, encoder
    head; from g_loss), then clip + Adam when optimizers are given (gp / dp updated in place, as cpu_step.adv_step).  ``force_ids``: as
    in cpu_step.decoder_sample."""
    vocab = gp["decoder.linear.weight"].shape[0]
    g_names = [k for k in gp if k.startswith(("decoder.", "encoder.linear.", "encoder.bn."))]
    g_leaf = {k: gp[k].detach().clone().requires_grad_(True) for k in g_names}
    d_leaf = {k: v.detach().clone().requires_grad_(True) for k, v in dp.items()}
    feats = O.encoder_head(g_leaf, trunk_feat)
    gen, ids, _ = attn_decoder_sample(g_leaf, feats, fmap, captions.shape[1], temperature, us, force_ids=force_ids)
    real = torch.nn.functional.one_hot(captions, vocab).float()
    m = masks if masks is not None else (None, None, None)
    d_real = O.disc_forward(d_leaf, real, m[0], num_rep)
    d_fake = O.disc_forward(d_leaf, gen.detach(), m[1], num_rep)
    g_out = O.disc_forward(d_leaf, gen, m[2], num_rep)
    g_loss, d_loss = O.get_losses(d_real, d_fake, g_out, loss_type)
    d_grads = dict(zip(d_leaf, torch.autograd.grad(d_loss, list(d_leaf.values()), retain_graph=True)))
    g_t = torch.autograd.grad(g_loss, list(g_leaf.values()), allow_unused=True)
    g_grads = {k: g for k, g in zip(g_leaf, g_t) if g is not None}
    if disc_opt is not None:
        disc_opt.step(dp, O.clip_grad_norm(d_grads, clip_norm)[0])
    if gen_opt is not None:
        gen_opt.step(gp, O.clip_grad_norm(g_grads, clip_norm)[0])
    return {"probs": gen.detach(), "ids": ids, "g_loss": float(g_loss.detach()), "d_loss": float(d_loss.detach()),
            "d_grads_raw": d_grads, "g_grads_raw": g_grads}
