"""SeqGAN-style adversarial step: policy-gradient generator update with Monte-Carlo roll-outs scored by the discriminator
(BASELINE.json config 5).  NO REFERENCE COUNTERPART -- the reference only trains G through the Gumbel-softmax relaxation
(src/training.py:144-169); this step reuses the reference's decoder loop (src/generator.py:55-81) as the sampler and its
discriminator (src/discriminator.py:34-62) on token ids as the reward model.  Definition and CPU oracle: oracle/cpu_seqgan.py
(Yu et al., SeqGAN, AAAI 2017, Algorithm 1).

Kernel sequence (all through libgicap.so; buffers from PyTorch's allocator):
  1. Y ~ G                       gic_decoder_sample_fwd, ids only (categorical sampling = Gumbel-max at temperature 1)
  2. (L-1)*N*B roll-outs         ONE gic_decoder_sample_fwd call: rows follow their prefix (force_ids / force_len), then sample;
                                 at this row count the per-step products are large MFMA GEMMs, and a row starts at its prefix
                                 length from the state of pass 4's forward (resume_from): half the row-steps
  3. rewards                     gic_disc_fwd on ids (eval mode) for all roll-outs and for Y, gic_rollout_rewards
  4. generator                   gic_decoder_sample_fwd along Y (logits, state kept) -> gic_xent with row weights = rewards (REINFORCE)
                                 -> gic_decoder_sample_bwd -> encoder head / start-token gradient -> clip + Adam
  5. discriminator               gic_disc_fwd / gic_disc_bwd on [real ; Y] ids in one batch, BCE (gic_gan_losses) -> clip + Adam
Data parallel: D's and G's flat gradient arenas are all-reduced separately on the reducer's side stream (G's first: D's backward runs
under it)."""
from __future__ import annotations

from typing import Optional

import torch

from . import engine
from .generator import SEEDS


class SeqGANStep:
    def __init__(self, gen, disc, gen_arena, disc_arena, args, reducer=None):
        self.gen, self.disc, self.args = gen, disc, args
        self.gen_arena, self.disc_arena, self.reducer = gen_arena, disc_arena, reducer
        self.cgan = int(args.conditional_gan) == 1
        self.dec = gen.decoder.engine()
        self.den = disc.engine()
        self.N = int(getattr(args, "mc_rollouts", 16))
        self._grads = None

    def bind_optimizers(self, gen_opt, disc_opt) -> "SeqGANStep":
        self.gen_opt, self.disc_opt = gen_opt, disc_opt
        return self

    def _grad_lists(self):
        if self._grads is None:
            by = {id(p): g for p, g in zip(self.gen_arena.params, self.gen_arena.grad_views())}
            gg = [by[id(p)] for p in self.gen.decoder.param_list()]
            by = {id(p): g for p, g in zip(self.disc_arena.params, self.disc_arena.grad_views())}
            self._grads = (gg, [by[id(p)] for p in self.disc.param_list()])
        return self._grads

    def __call__(self, images, captions, max_caption_len: int, train: bool = True, u_sample: Optional[torch.Tensor] = None,
                 u_mc: Optional[torch.Tensor] = None, keep_masks=None, opt_step: bool = True, next_images=None) -> dict:
        """One step.  ``u_sample`` [L,B,V] / ``u_mc`` [L,(L-1)*N*B,V] / ``keep_masks`` (2 x [B*R,F]: real, fake) make it deterministic
        (parity runs); otherwise noise is drawn on the device.  Returns device tensors: losses [g_loss, d_loss], ids (Y), rewards."""
        gen, disc = self.gen, self.disc
        B, L, N = captions.shape[0], int(max_caption_len), self.N
        dev = captions.device
        engine.require_gpu(captions, images)
        gparams = [p.detach() for p in gen.decoder.param_list()]
        dparams = [p.detach() for p in disc.param_list()]
        g_grads, d_grads = self._grad_lists()
        dec, den = self.dec, self.den
        R = den.R
        if self.cgan:
            main = torch.cuda.current_stream(dev)
            start = main.record_event()
            feats = gen.encoder.forward_fused(images, train, trunk_feats=gen.encoder.take_trunk(images, train, main))
            if next_images is not None:          # the frozen trunk's pass for the next batch runs under this step (generator.Encoder)
                gen.encoder.prefetch_trunk(next_images, train, start)
        else:
            ones = torch.ones(B, device=dev, dtype=torch.int64)
            feats = engine.embedding_fwd(gparams[0], ones)
        # 1. Y ~ G
        _, Y, _ = dec.sample_fwd(gparams, feats, L, 1.0, noise_u=u_sample, seed=0 if u_sample is not None else SEEDS.next(), ids_only=True)
        # 4a. the generator's own pass along Y (logits; its recurrent state is kept: the roll-outs resume from it, the backward uses it)
        logits, _, st = dec.sample_fwd(gparams, feats, L, 1.0, pretrain=True, force_ids=Y)
        # 2. + 3. roll-outs and rewards
        mc_logits = None
        reps = (L - 1) * N
        if reps > 0:
            f_big = feats.repeat(reps, 1)
            force = Y.repeat(reps, 1)
            flen = torch.arange(1, L, device=dev, dtype=torch.int32).repeat_interleave(N * B)
            resume = None
            if reps * B > dec.fused_rollout_rows():
                # generic-product path: a roll-out does not recompute its prefix, it starts at step t from pass 4a's state (rows are
                # sorted by prefix length: at step t the first t*N*B rows exist) -- half the row-steps of the full batch
                resume = (st, B, [min(t, L - 1) * N * B for t in range(L)])
            _, mc_ids, _ = dec.sample_fwd(gparams, f_big, L, 1.0, noise_u=u_mc, seed=0 if u_mc is not None else SEEDS.next(),
                                          ids_only=True, force_ids=force, force_len=flen, resume=resume)
            mc_logits, _ = den.fwd(dparams, None, mc_ids, False, forward_only=True)     # rewards: no backward follows
        full_logits, _ = den.fwd(dparams, None, Y, False, forward_only=True)
        rewards = engine.rollout_rewards(mc_logits, full_logits, B, L, N, R)
        out = {"ids": Y, "rewards": rewards}
        # 4b. REINFORCE
        g_loss, dlog = engine.xent(logits.view(B * L, dec.V), Y.reshape(-1), want_grad=train, row_weight=rewards.reshape(-1))
        # 5. D on [real ; Y]
        km = keep_masks if keep_masks is not None else (None, None)
        both = torch.cat([captions, Y], 0)
        mask = None if km[0] is None else torch.cat([km[0], km[1]], 0)
        d_logits, dst = den.fwd(dparams, None, both, train, mask, 0 if mask is not None else SEEDS.next())
        half = B * R
        losses, lgr = engine.gan_losses("standard", d_logits[:half], d_logits[half:], d_logits[half:], want_grads=train)
        out["losses"] = torch.stack([g_loss[0], losses[1]])
        out["logits"] = logits
        if not train:
            return out
        d_feat = torch.empty(B, dec.E, device=dev, dtype=torch.float32)
        dec.sample_bwd(gparams, st, logits, Y, dlog.view(B, L, dec.V), 1.0, True, grads=g_grads + [d_feat])
        if self.cgan:
            gen.encoder.backward_fused(d_feat)
        else:
            engine.embedding_bwd(d_feat, ones, dec.V, d_weight=g_grads[0], zero_first=False)
        if self.reducer is not None:
            self.reducer.start(self.gen_arena.grad)
        den.bwd(dparams, dst, None, both, train, lgr["dd_real_fake"], True, False, grads=d_grads)
        if self.reducer is not None:
            self.reducer.start(self.disc_arena.grad)
            self.reducer.wait_all()
        if opt_step:
            self.disc_opt.step()
            self.gen_opt.step()
        return out
