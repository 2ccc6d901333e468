"""GANInstructor: drop-in for the reference trainer (src/training.py:15-235).

Same public surface (``GANInstructor(args, train_dataset, dev_dataset)``, ``_run``,
``pretrain_generator``, ``genpretrain_loop``, ``adv_loop``, ``optimize``, ``update_temperature``),
same batch contract and logging, with these deliberate fixes:
  * step order: both backward passes on pre-update weights, then both optimizer steps -- the literal
    order (training.py:168-169) raises on every torch >= 1.5; losses are unaffected;
  * ``_run`` logs ``adv_epoch`` where the reference hits ``NameError: epoch`` (training.py:227);
  * the six per-batch host syncs (training.py:171-181) collapse into one;
  * data parallelism: one process per GPU, flat-gradient all-reduce over RCCL (parallel.py).
"""
from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader
from tqdm import tqdm

from . import engine, parallel
from .discriminator import Discriminator
from .fused_step import FusedAdvStep
from .seqgan import SeqGANStep
from .generator import Generator
from .optim import FusedClipAdam, ParamArena
from .tasks import collate_fn
from .utils import create_logger, get_fixed_temperature, get_losses


class _ScalarWriter:
    """SummaryWriter stand-in (tensorboard is optional): add_scalar -> <save_dir>/scalars.jsonl, buffered."""

    def __init__(self, logdir: Optional[str]):
        self._fh = None
        self._tb = None
        if logdir:
            try:
                from torch.utils.tensorboard import SummaryWriter      # noqa: WPS433
                self._tb = SummaryWriter(logdir)
            except Exception:
                os.makedirs(logdir, exist_ok=True)
                self._fh = open(os.path.join(logdir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        value = float(value)
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)
        elif self._fh is not None:
            self._fh.write(json.dumps({"tag": tag, "value": value, "step": int(step)}) + "\n")

    def flush(self):
        if self._fh is not None:
            self._fh.flush()


def _lookahead(loader, device):
    """(batch on device, next batch on device or None): the adversarial loop hands the next images to the step so that
    their trunk forward overlaps this step (FusedAdvStep._prefetch_trunk)."""
    def to_dev(b):
        return (b[0].to(device), b[1].to(device), b[2], b[3])
    it = iter(loader)
    try:
        cur = to_dev(next(it))
    except StopIteration:
        return
    for b in it:
        nxt = to_dev(b)
        yield cur, nxt
        cur = nxt
    yield cur, None


class GANInstructor:
    def __init__(self, args, train_dataset, dev_dataset):
        self.args = args
        self.dist = parallel.DistInfo.from_env()
        from .generator import SEEDS
        SEEDS.rank = self.dist.rank            # replicas share weights and the torch seed, not the device noise streams
        self.gen = Generator(args).to(args.device)                        # training.py:19
        self.disc = Discriminator(args).to(args.device)                   # training.py:20
        self.cgan = (args.conditional_gan == 1)
        self.attention = getattr(args, "decoder", "lstm") == "attention"
        if self.attention:
            if getattr(args, "adv_mode", "relgan") != "relgan":
                raise ValueError("--decoder attention is trained with the relaxation (--adv-mode relgan)")
        if self.dist.world_size > 1:
            parallel.broadcast_module(self.gen, self.dist)
            parallel.broadcast_module(self.disc, self.dist)
        rank0 = self.dist.rank == 0
        self.log = create_logger(__name__ + str(id(self)), silent=not rank0, to_disk=rank0 and bool(getattr(args, "log_file", None)),
                                 log_file=(args.log_file + ".txt") if getattr(args, "log_file", None) else None)

        # Trainable generator parameters: the decoder, plus the encoder head under --conditional-gan 1.  The
        # ResNet trunk runs under no_grad (generator.py:21) so Adam never touches it (grad is None there).
        g_params = list(self.gen.decoder.parameters())
        if self.cgan:
            g_params += list(self.gen.encoder.linear.parameters()) + list(self.gen.encoder.bn.parameters())
        self.gen_arena = ParamArena(g_params)
        self.disc_arena = ParamArena(self.disc.parameters())
        self.pretrain_opt = FusedClipAdam(self.gen_arena, args.pretrain_lr, args.clip_norm)   # training.py:24
        self.gen_opt = FusedClipAdam(self.gen_arena, args.gen_lr, args.clip_norm)             # training.py:25
        self.disc_opt = FusedClipAdam(self.disc_arena, args.disc_lr, args.clip_norm)          # training.py:26
        self.reducer = parallel.GradReducer(self.dist) if self.dist.world_size > 1 else None
        self.fused = FusedAdvStep(self.gen, self.disc, self.gen_arena, self.disc_arena, args, self.reducer
                                  ).bind_optimizers(self.gen_opt, self.disc_opt)
        # --adv-mode seqgan: policy gradient + Monte-Carlo roll-outs (no reference counterpart; seqgan.py)
        self.seqgan = SeqGANStep(self.gen, self.disc, self.gen_arena, self.disc_arena, args, self.reducer
                                 ).bind_optimizers(self.gen_opt, self.disc_opt)

        self.train_dataset, self.dev_dataset = train_dataset, dev_dataset
        nw = int(getattr(args, "num_workers", 4))
        dp = self.dist.world_size > 1
        # training.py:28-32 (shuffle=True for the two train loaders).  Under data parallelism the shuffling moves into the
        # DistributedSampler (the DataLoader itself must not shuffle when it is given a sampler); set_epoch() in the loops.
        mk = lambda ds, bs, shuffle: None if ds is None else DataLoader(   # noqa: E731
            ds, shuffle=shuffle and not dp, batch_size=bs, collate_fn=collate_fn, num_workers=nw,
            sampler=parallel.shard_sampler(ds, self.dist, shuffle) if dp else None)
        self.pre_train_loader = mk(train_dataset, args.pre_train_batch_size, True)
        self.pre_eval_loader = mk(dev_dataset, args.pre_eval_batch_size, False)
        self.adv_train_loader = mk(train_dataset, args.adv_train_batch_size, True)
        self.adv_eval_loader = mk(dev_dataset, args.adv_eval_batch_size, False)
        self._sampler_epoch = 0

        self.model_dir = getattr(args, "model_dir", None)
        self.writer = _ScalarWriter(getattr(args, "save_dir", None) if rank0 and getattr(args, "log_file", None) else None)
        self.pretrain_steps = 0
        self.gen_steps = 0
        self.disc_steps = 0
        self.adv_epoch = -1

    # ------------------------------------------------------------------ shared pieces
    def _features(self, images, batch, next_images=None):
        if self.attention:
            return self.gen.encoder.forward_with_map(images, next_images=next_images)      # (features, feature map)
        if self.cgan:
            return self.gen.encoder(images, next_images=next_images)       # training.py:66,145
        ones = torch.ones(batch, dtype=torch.long, device=self.args.device)
        return self.gen.decoder.embed(ones)                                # training.py:68,147

    def _reshuffle(self, loader, what) -> None:
        """Data parallel: a new permutation per pass over the training set (DistributedSampler.set_epoch), same on every rank."""
        if what == "train" and self.dist.world_size > 1 and hasattr(getattr(loader, "sampler", None), "set_epoch"):
            loader.sampler.set_epoch(self._sampler_epoch)
            self._sampler_epoch += 1

    def update_temperature(self, i, N):
        self.gen.decoder.temperature = get_fixed_temperature(self.args.temperature, i, N, self.args.temp_adpt)

    def optimize(self, opt, loss, model=None, retain_graph=False):
        """training.py:194-199; the clip_grad_norm_ of :198 is fused into ``opt.step()``."""
        opt.zero_grad()
        loss.backward(retain_graph=retain_graph)
        if self.reducer is not None:
            self.reducer.start(opt.arena.grad)
            self.reducer.wait_all()
        opt.step()

    # ------------------------------------------------------------------ MLE pre-training (training.py:48-126)
    def pretrain_step(self, images, captions, max_caption_len, train=True, next_images=None):
        feats = self._features(images, captions.shape[0], next_images)
        if self.attention:
            gen_captions, _ids = self.gen.decoder.sample(feats[0], fmap=feats[1], pretrain=True, max_caption_len=max_caption_len)
        else:
            gen_captions, _ids = self.gen.decoder.sample(feats, pretrain=True, max_caption_len=max_caption_len)
        flat = gen_captions.reshape(-1, gen_captions.size(-1))
        loss = _XentFn.apply(flat, captions.reshape(-1))                   # nn.CrossEntropyLoss(), training.py:81-83
        if train:
            self.optimize(self.pretrain_opt, loss, self.gen)
        return loss

    def genpretrain_loop(self, what):
        gen_loss = []
        loader = self.pre_train_loader if what == "train" else self.pre_eval_loader
        total = len(self.train_dataset) if what == "train" else len(self.dev_dataset)
        self._reshuffle(loader, what)
        with (torch.enable_grad() if what == "train" else torch.no_grad()), \
                tqdm(total=total, disable=self.dist.rank != 0) as progress:
            for (images, captions, lengths, max_caption_len), nxt in _lookahead(loader, self.args.device):
                loss = self.pretrain_step(images, captions, max_caption_len, train=(what == "train"),
                                          next_images=nxt[0] if nxt is not None and nxt[0].shape == images.shape else None)
                val = loss.item()
                if val != val:      # gic_xent poisons the loss when a target is outside [0, V) (nn.CrossEntropyLoss would raise)
                    raise ValueError("pre-train loss is NaN: a caption token is outside [0, vocab_size=%d) or the logits overflowed"
                                     % self.args.vocab_size)
                gen_loss.append(val)
                self.writer.add_scalar("GenPreTraining_train_loss" if what == "train" else "GenPreTraining_val_loss",
                                       val, self.pretrain_steps)
                progress.update(len(images) * self.dist.world_size)
                progress.set_postfix(loss=val)
        return gen_loss

    def pretrain_generator(self, epochs):
        self.log.info("Pretraining Generator")
        total_loss, best_loss = 0, None
        for epoch in range(self.args.pretrain_epochs):
            self.gen.train()
            train_epoch_loss = np.mean(self.genpretrain_loop("train"))
            total_loss += train_epoch_loss
            self.gen.eval()
            val_epoch_loss = np.mean(self.genpretrain_loop("val"))
            if best_loss is None or val_epoch_loss < best_loss:
                best_loss = val_epoch_loss
                self._save(self.gen.state_dict(), "pretrained_model.ckpt")                    # training.py:118
                self.log.info("Saving Best model [Gen Loss = {}] at Epoch {}".format(best_loss, epoch))
            if epoch % self.args.pre_log_step == 0:
                self.log.info("Epoch {}: \n \t Train: {} \n\t Val: {} ".format(epoch, train_epoch_loss, val_epoch_loss))
            self.pretrain_steps += 1
        return total_loss / epochs if epochs != 0 else 0

    # ------------------------------------------------------------------ adversarial step (training.py:136-183)
    def adv_step(self, images, captions, max_caption_len, train=True, noise_u=None, keep_masks=None, next_images=None):
        """One minibatch.  Returns (g_loss, d_loss) as a 2-element device tensor (one host sync to read).
        ``next_images`` (optional): the next batch's images on the device, for the trunk prefetch of the fused step."""
        impl = getattr(self.args, "step_impl", "fused")
        if getattr(self.args, "adv_mode", "relgan") == "seqgan":
            return self.seqgan(images, captions, max_caption_len, train, next_images=next_images)["losses"]
        if impl == "fused":
            return self.fused(images, captions, max_caption_len, train, noise_u, keep_masks, next_images=next_images)["losses"]
        return self._adv_step_autograd(images, captions, max_caption_len, train, noise_u, keep_masks, next_images=next_images)

    def _adv_step_autograd(self, images, captions, max_caption_len, train, noise_u=None, keep_masks=None, next_images=None):
        """The reference's flow through the module API + autograd, with the fixed order."""
        km = keep_masks if keep_masks is not None else (None, None, None)
        with (torch.enable_grad() if train else torch.no_grad()):
            features = self._features(images, captions.shape[0], next_images if self.cgan else None)
            if self.attention:
                gen_captions, _ids = self.gen.decoder.sample(features[0], fmap=features[1], max_caption_len=max_caption_len, noise_u=noise_u)
            else:
                gen_captions, _ids = self.gen.decoder.sample(features, max_caption_len=max_caption_len, noise_u=noise_u)
            fake_captions = gen_captions.detach()                                            # training.py:151
            if int(getattr(self.args, "real_as_ids", 1)):
                real = captions
            else:
                real = F.one_hot(captions, self.args.vocab_size).float()                     # training.py:158
            d_out_real = self.disc(real, keep_mask=km[0])                                    # training.py:162
            d_out_fake = self.disc(fake_captions, keep_mask=km[1])                           # training.py:163
            with self.disc.input_grad_only():
                g_out = self.disc(gen_captions, keep_mask=km[2])                             # training.py:164
            g_loss, d_loss = get_losses(d_out_real, d_out_fake, g_out, self.args.adv_loss_type, detach_d_for_g=True)
        if train:
            self.disc_opt.zero_grad()
            self.gen_opt.zero_grad()
            d_loss.backward()                       # reaches D's parameters only (fake is detached)
            if self.reducer is not None:
                self.reducer.start(self.disc_arena.grad)
            if g_loss.requires_grad:
                g_loss.backward()                   # reaches G through D's input gradient; D records no param grads here
            if self.reducer is not None:
                self.reducer.start(self.gen_arena.grad)
                self.reducer.wait_all()
            self.disc_opt.step()
            self.gen_opt.step()
        return torch.stack([g_loss.detach(), d_loss.detach()])

    def adv_loop(self, what):
        loader = self.adv_train_loader if what == "train" else self.adv_eval_loader
        total = len(self.train_dataset) if what == "train" else len(self.dev_dataset)
        self._reshuffle(loader, what)
        float_epoch = 0.0
        gen_loss, disc_loss = [], []
        with tqdm(total=total, disable=self.dist.rank != 0) as progress:
            for (images, captions, lengths, max_caption_len), nxt in _lookahead(loader, self.args.device):
                float_epoch += 1
                losses = self.adv_step(images, captions, max_caption_len, train=(what == "train"),
                                       next_images=nxt[0] if nxt is not None and nxt[0].shape == images.shape else None)
                g_val, d_val = losses.tolist()                                   # the step's single host sync
                self.writer.add_scalar("Discriminator_train_loss" if what == "train" else "Discriminator_val_loss", d_val, self.disc_steps)
                self.disc_steps += 1
                self.writer.add_scalar("Generator_train_loss" if what == "train" else "Generator_val_loss", g_val, self.gen_steps)
                self.gen_steps += 1
                gen_loss.append(g_val)
                disc_loss.append(d_val)
                progress.update(len(images) * self.dist.world_size)
                progress.set_postfix(disc_loss=d_val, gen_loss=g_val)
                self.update_temperature(self.adv_epoch + float_epoch / len(loader), self.args.adv_epochs)   # training.py:183
        return np.mean(gen_loss), np.mean(disc_loss)

    def load_checkpoint(self, path: str) -> str:
        """Load weights written by ``_run`` here or by the reference (training.py:118: the generator's state dict;
        training.py:225-226: {"generator", "discriminator"}).  Values are copied into the existing parameter storage (the flat
        arenas stay in place); optimizer moments start from zero, as after the reference's own cold start.  Returns the kind."""
        ckpt = torch.load(path, map_location=self.args.device)
        if isinstance(ckpt, dict) and set(ckpt) == {"generator", "discriminator"}:
            self.gen.load_state_dict(ckpt["generator"])
            self.disc.load_state_dict(ckpt["discriminator"])
            kind = "adversarial"
        else:
            self.gen.load_state_dict(ckpt)
            kind = "pretrained"
        from . import engine
        engine.bump_param_epoch()          # compute-dtype weight images are stale
        self.log.info("resumed %s weights from %s", kind, path)
        return kind

    def _save(self, obj, name):
        if self.model_dir and self.dist.rank == 0:
            torch.save(obj, os.path.join(self.model_dir, name))

    def _run(self):
        self.pretrain_generator(self.args.pretrain_epochs)
        self.log.info("Starting Adversarial Training...")
        best_loss = None
        for adv_epoch in range(self.args.adv_epochs):
            self.adv_epoch = adv_epoch
            self.disc.train()
            self.gen.train()
            train_g_loss, train_d_loss = self.adv_loop("train")
            self.disc.eval()
            self.gen.eval()
            val_g_loss, val_d_loss = self.adv_loop("val")
            if best_loss is None or val_g_loss < best_loss:
                best_loss = val_g_loss
                self._save({"generator": self.gen.state_dict(), "discriminator": self.disc.state_dict()}, "adv_model.ckpt")
                self.log.info("Saving Best model [Gen Loss = {}] at Epoch {}".format(best_loss, adv_epoch))
            if adv_epoch % self.args.adv_log_step == 0 or adv_epoch == self.args.adv_epochs - 1:
                self.log.info("[ADV] epoch %d (temperature: %.4f):\n\t g_loss: %.4f | %.4f \n\t d_loss: %.4f | %.4f" % (
                    adv_epoch, self.gen.decoder.temperature, train_g_loss, val_g_loss, train_d_loss, val_d_loss))
            self.writer.flush()


class _XentFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean over all rows, PAD included; training.py:81-83) via gic_xent."""

    @staticmethod
    def forward(ctx, logits, targets):
        loss, dlog = engine.xent(logits.detach().contiguous(), targets, want_grad=ctx.needs_input_grad[0])
        ctx.dlog = dlog
        return loss[0]

    @staticmethod
    def backward(ctx, d):
        return ctx.dlog * d, None
