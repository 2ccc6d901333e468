"""CPU oracle of the SeqGAN-style policy-gradient generator update with Monte-Carlo roll-outs (BASELINE.json config 5).

TEST INFRASTRUCTURE (see oracle/__init__.py).  NO REFERENCE COUNTERPART: the reference trains G only through the Gumbel-softmax
relaxation (src/training.py:144-169); SeqGAN appears in its report as related work.  This file is the build's own definition of
the step, restated in plain torch on top of the pinned pieces of oracle/cpu_step.py (the reference's LSTM decoder loop,
src/generator.py:55-81, and its discriminator, src/discriminator.py:34-62), so that the HIP path can be compared on identical
inputs.  The algorithm is Yu et al., "SeqGAN: Sequence Generative Adversarial Nets with Policy Gradient" (AAAI 2017), Alg. 1:

  1. Y ~ G: categorical sampling token by token, y_t ~ softmax(o_t), drawn as argmax(o_t + Gumbel(u_t)) (Gumbel-max), i.e. the
     reference's sample() at temperature 1 keeping only the ids.
  2. Reward Q(Y_{1:t}, y_t) for t < L: the mean over N Monte-Carlo roll-outs (complete Y_{1:t} with the same sampler) of D's
     score of the completed caption; for t = L: D's score of Y itself.  D's score of a caption = mean over its R representation
     logits of sigmoid(logit), D in eval mode (no dropout).
  3. Generator loss (REINFORCE): - mean over all B*L positions of reward[b,t] * log G(y_t | Y_{1:t-1})  (mean over ALL positions,
     like the reference's own MLE loss, src/training.py:81-83); gradient clipped (clip_norm 5.0) + Adam as src/training.py:194-199.
  4. Discriminator loss: BCE-with-logits, real captions -> 1, sampled captions Y -> 0 (src/utils.py:14-17 'standard' d_loss), D in
     train mode (dropout masks explicit); clip + Adam.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import cpu_step as O

Tensor = torch.Tensor


def seqgan_step(gp: O.Params, dp: O.Params, captions: Tensor, u_sample: Sequence[Tensor], u_mc: Tensor, n_rollouts: int,
                masks: Optional[Sequence[Tensor]], clip_norm: float = 5.0, gen_opt: Optional[O.AdamState] = None,
                disc_opt: Optional[O.AdamState] = None, trunk_feat: Optional[Tensor] = None, num_rep: int = 64,
                force_Y: Optional[Tensor] = None) -> Dict[str, object]:
    """u_sample: L draws [B,V] for Y.  u_mc: [L, (L-1)*N*B, V] draws of the roll-outs, row (t-1)*N*B + n*B + b = roll-out n of
    caption b from its prefix of length t.  masks: (mask_real, mask_fake) dropout keep masks of D's two training passes.
    ``force_Y`` [B,L] (test aid): the sampled captions to use instead of the oracle's own draw -- Y enters the losses only as data
    (REINFORCE differentiates log G(y_t | ...) at GIVEN y), so where a bf16 near-tie flipped a sample on the GPU the gradients are
    compared on the GPU's own Y."""
    bsz, seqlen = captions.shape
    vocab = gp["decoder.linear.weight"].shape[0]
    N = n_rollouts
    g_names = [k for k in gp if k.startswith("decoder.") or (trunk_feat is not None and k.startswith(("encoder.linear.", "encoder.bn.")))]
    g_leaf = {k: gp[k].detach().clone().requires_grad_(True) for k in g_names}
    d_leaf = {k: v.detach().clone().requires_grad_(True) for k, v in dp.items()}
    feats = O.encoder_head(g_leaf, trunk_feat) if trunk_feat is not None else O.start_features(g_leaf, bsz)
    with torch.no_grad():
        _, Y = O.decoder_sample(gp if trunk_feat is None else g_leaf, feats.detach(), seqlen, 1.0, u_sample)
        if force_Y is not None:
            Y = force_Y
        # Monte-Carlo roll-outs, one batch: rows ordered (prefix length t = 1..L-1, roll-out n, caption b)
        reps = (seqlen - 1) * N
        rewards = torch.empty(bsz, seqlen)
        mc_ids = None
        if reps > 0:
            f_big = feats.detach().repeat(reps, 1)
            force = Y.repeat(reps, 1)
            flen = torch.arange(1, seqlen).repeat_interleave(N * bsz)
            _, mc_ids = O.decoder_sample(gp if trunk_feat is None else g_leaf, f_big, seqlen, 1.0,
                                         [u_mc[t] for t in range(seqlen)], force_ids=force, force_len=flen)
            # D's reward evaluation in chunks of rows (the dense one-hot of all (L-1)*N*B roll-outs at BASELINE sizes is tens of GB;
            # captions are independent in D's eval-mode forward, so the result is the same)
            mc_logit = torch.cat([O.disc_forward(dp, torch.nn.functional.one_hot(mc_ids[r0:r0 + 608], vocab).float(), None, num_rep)
                                  for r0 in range(0, mc_ids.shape[0], 608)])
            score = torch.sigmoid(mc_logit).view(seqlen - 1, N, bsz, num_rep).mean(dim=(1, 3))          # [L-1, B]
            rewards[:, :seqlen - 1] = score.t()
        full = O.disc_forward(dp, torch.nn.functional.one_hot(Y, vocab).float(), None, num_rep)
        rewards[:, seqlen - 1] = torch.sigmoid(full).view(bsz, num_rep).mean(1)
    # REINFORCE loss on the sampled trajectory
    logits, _ = O.decoder_sample(g_leaf, feats, seqlen, 1.0, None, pretrain=True, force_ids=Y)
    flat = logits.reshape(-1, vocab)
    nll = torch.logsumexp(flat, 1) - flat.gather(1, Y.reshape(-1, 1)).squeeze(1)
    g_loss = (rewards.reshape(-1) * nll).mean()
    g_grads_t = torch.autograd.grad(g_loss, list(g_leaf.values()), allow_unused=True)
    g_grads = {k: g for k, g in zip(g_leaf, g_grads_t) if g is not None}
    # discriminator on ids: real -> 1, sampled -> 0
    m = masks if masks is not None else (None, None)
    d_real = O.disc_forward(d_leaf, torch.nn.functional.one_hot(captions, vocab).float(), m[0], num_rep)
    d_fake = O.disc_forward(d_leaf, torch.nn.functional.one_hot(Y, vocab).float(), m[1], num_rep)
    d_loss = O.bce_logits_mean(d_real, True) + O.bce_logits_mean(d_fake, False)
    d_grads = dict(zip(d_leaf, torch.autograd.grad(d_loss, list(d_leaf.values()))))
    out: Dict[str, object] = {"Y": Y, "mc_ids": mc_ids, "rewards": rewards, "g_loss": float(g_loss.detach()), "d_loss": float(d_loss.detach()),
                              "logits": logits.detach(), "g_grads_raw": g_grads, "d_grads_raw": d_grads, "d_real": d_real.detach(),
                              "d_fake": d_fake.detach()}
    d_c, d_norm = O.clip_grad_norm(d_grads, clip_norm)
    g_c, g_norm = O.clip_grad_norm(g_grads, clip_norm)
    out.update(d_norm=d_norm, g_norm=g_norm)
    if disc_opt is not None:
        disc_opt.step(dp, d_c)
    if gen_opt is not None:
        gen_opt.step(gp, g_c)
    return out
