"""The adversarial G+D train step as one direct kernel sequence (no autograd engine).

Same math and the same step order as ``GANInstructor._adv_step_autograd`` / SURVEY.md §8(c) -- body
of reference src/training.py:136-183 with both backward passes on pre-update weights, then both
optimizer steps -- but every buffer is allocated once per (batch, length) and gradients are written
straight into the flat arenas that the optimizer kernels and the RCCL all-reduce read.

Work that the reference executes and discards is skipped (results identical):
  * D's parameter gradients from g_loss (zeroed by the next zero_grad, training.py:195);
  * D(real) is evaluated on token ids (gather) rather than a dense one-hot (training.py:158), unless
    ``real_as_ids=0``;
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import engine
from .generator import SEEDS


class FusedAdvStep:
    def __init__(self, gen, disc, gen_arena, disc_arena, args, reducer=None):
        self.gen, self.disc, self.args = gen, disc, args
        self.gen_arena, self.disc_arena = gen_arena, disc_arena
        self.reducer = reducer
        self.cgan = int(args.conditional_gan) == 1
        self.attn = hasattr(gen.decoder, "attn")         # visual-attention decoder (cfg4): the roll-out also takes the trunk's feature map
        self.dec = gen.decoder.engine()
        self.den = disc.engine()
        self._buf: Dict[tuple, dict] = {}
        self._gen_grads = None
        self._disc_grads = None
        self.overlap = not os.environ.get("GIC_NO_STREAM_OVERLAP")
        self.trace = None
        self.use_graph = not (os.environ.get("GIC_NO_STEP_GRAPH") or os.environ.get("GIC_NO_GRAPH"))
        if self.attn and not os.environ.get("GIC_STEP_GRAPH_ATTN"):
            # the attention step replays as graphs too (host enqueue 2.10 -> 0.43 ms per step) but measured 3.87 ms per step against
            # 3.74 for eager launches at cfg4 (same box, back to back): eager is the default there, GIC_STEP_GRAPH_ATTN=1 opts in
            self.use_graph = False
        self._graphs: Dict[tuple, dict] = {}
        self._warm: set = set()

    # grads of the decoder parameters are views into the generator arena (order = Decoder.param_list())
    def _grad_lists(self):
        if self._gen_grads is None:
            by_param = {id(p): g for p, g in zip(self.gen_arena.params, self.gen_arena.grad_views())}
            self._gen_grads = [by_param[id(p)] for p in self.gen.decoder.param_list()]
            by_param = {id(p): g for p, g in zip(self.disc_arena.params, self.disc_arena.grad_views())}
            self._disc_grads = [by_param[id(p)] for p in self.disc.param_list()]
        return self._gen_grads, self._disc_grads

    def _buffers(self, B: int, L: int, dev) -> dict:
        key = (B, L)
        if key not in self._buf:
            dec, den = self.dec, self.den
            self._buf[key] = {
                "dec_state": dec.alloc_state(B, L, dev),
                "dec_ws": dec.alloc_bwd_ws(B, L, dev),
                "probs": torch.empty(B, L, dec.V, device=dev, dtype=dec.act),
                "ids": torch.empty(B, L, device=dev, dtype=torch.int64),
                "d_feat": torch.empty(B, dec.E, device=dev, dtype=torch.float32),
                "d_probs": torch.empty(B, L, dec.V, device=dev, dtype=dec.act),
                "st_rf": den.alloc_state(2 * B, L, dev),      # D(real) | D(fake) in the two halves: ONE backward over both
                "disc_ws": den.alloc_bwd_ws(2 * B, L, dev), "disc_ws_gen": den.alloc_bwd_ws(B, L, dev),
                "logits": torch.empty(3, B * den.R, device=dev, dtype=torch.float32),
                "ones": torch.ones(B, device=dev, dtype=torch.int64),
            }
            self._buf[key]["st_real"], self._buf[key]["st_fake"] = den.split_state(self._buf[key]["st_rf"])
            # D(gen) sees the same input as D(fake) (training.py:163-164): it shares everything up to the dropout draw
            self._buf[key]["st_gen"] = den.shared_state(self._buf[key]["st_fake"], B, L, dev)
        return self._buf[key]

    def _early_bucket(self):
        """Arena span of decoder.linear's weight and bias (None if they are not adjacent in the arena)."""
        if not hasattr(self, "_early"):
            lin = self.gen.decoder.linear
            self._early = self.gen_arena.span([lin.weight, lin.bias])
        return self._early

    def _mark(self, name: str, stream) -> None:
        """Phase marker for tools/step_timeline.py (``self.trace`` is a list while tracing, else None)."""
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.trace.append((name, ev))

    def _streams(self, dev):
        if getattr(self, "_s_real", None) is None:
            self._s_real = torch.cuda.Stream(device=dev)     # D(real) forward: independent of the generator
            self._s_gen = torch.cuda.Stream(device=dev)      # the generator's path: D(gen) forward / backward, decoder backward
        return self._s_real, self._s_gen

    # ---- trunk look-ahead (Encoder.prefetch_trunk / take_trunk): the ResNet trunk is frozen (generator.py:21), so the trunk forward
    # of batch k+1 depends on nothing that step k updates; it runs on its own stream under this step's launch-bound phases.
    def __call__(self, images, captions, max_caption_len: int, train: bool = True, noise_u=None, keep_masks=None,
                 opt_step: bool = True, next_images=None, next_train=None) -> dict:
        """One step.  Returns device tensors: losses [g_loss, d_loss], ids, probs, logits (real, fake, gen).
        ``noise_u`` [L,B,V] / ``keep_masks`` (3 x [B*R,F]) make the step deterministic for parity runs.
        ``next_images``: the NEXT batch's images (same shape); its trunk forward is enqueued on a side stream under this
        step and picked up by the next call when it is passed the same tensor (results are identical either way).

        Independent branches of the step's dependency graph run on side HIP streams behind events:
          D(real) forward          || encoder head + roll-out         (and the NEXT batch's trunk forward under all of it)
          G path (D(gen) input-gradient, decoder / encoder-head backward)  ||  D path (backward real + fake, D's Adam)
        D's weights are not updated before the G path has finished reading them; both paths join before G's Adam.

        A training step with device-drawn noise is replayed as six linear hipGraphs from its third call on (``_call_graph``): everything
        behind the trunk features -- ~110 launches on three streams -- with the per-step scalars (temperature, noise seeds) read from
        device memory (engine.StepScalarsBuffer); GIC_NO_STEP_GRAPH=1 (or GIC_NO_GRAPH=1) keeps the eager launches.  Returned tensors
        of a replayed step live in the graphs' memory pools and are overwritten by the next step."""
        B, L = captions.shape[0], int(max_caption_len)
        engine.require_gpu(captions, images)
        if (self.use_graph and train and opt_step and noise_u is None and keep_masks is None and self.reducer is None and self.overlap
                and self.trace is None):
            return self._call_graph(images, captions, B, L, next_images, next_train)
        dev = captions.device
        main = torch.cuda.current_stream(dev)
        km = keep_masks if keep_masks is not None else (None, None, None)
        seeds = [0 if km[i] is not None else SEEDS.next() for i in range(3)]
        seeds.append(0 if noise_u is not None else SEEDS.next())
        ev_start = main.record_event()
        self._mark("start", main)
        trunk = (lambda: self._take_and_prefetch(images, train, main, ev_start, next_images, next_train)) if self.cgan else None
        return self._body(self._buffers(B, L, dev), captions, trunk, images, float(self.gen.decoder.temperature), train, noise_u, km, seeds,
                          None, opt_step, main, ev_start)

    def _take_and_prefetch(self, images, train, main, ev_start, next_images, next_train):
        """This step's trunk features (the look-ahead pass of the previous step, or a pass now), then the NEXT batch's pass on its stream."""
        enc = self.gen.encoder
        trunk_feats = enc.take_trunk_with_map(images, train, main) if self.attn else enc.take_trunk(images, train, main)
        if next_images is not None:          # the prefetched output is a private copy: the next trunk pass may start now
            enc.prefetch_trunk(next_images, train if next_train is None else next_train, ev_start, mark=self._mark, want_map=self.attn)
        return trunk_feats

    def _body(self, buf, captions, trunk, images, T, train, noise_u, km, seeds, scal, opt_step, main, ev_start) -> dict:
        """Everything of the step on `main` and the two side streams.  ``trunk``: callable returning the trunk features (the eager
        path calls it after D(real) and the weight images are under way) or a tensor (graph path: the static input buffer).
        ``scal``: StepScalarsBuffer or None (temperature ``T`` / ``seeds`` by value)."""
        a = self.args
        gen, disc = self.gen, self.disc
        B, L = captions.shape
        dev = captions.device
        gparams = [p.detach() for p in gen.decoder.param_list()]
        dparams = [p.detach() for p in disc.param_list()]
        g_grads, d_grads = self._grad_lists()
        d_train = bool(train)            # dropout active in train mode only (disc.train()/eval(), training.py:215,219)
        overlap = self.overlap
        s_real, s_gen = self._streams(dev) if overlap else (main, main)
        lg = buf["logits"]

        # compute-dtype weight images are refreshed on the side streams, under the encoder
        with engine.on_stream(s_gen):
            s_gen.wait_event(ev_start)
            self.dec.prepare(gparams)
            ev_gprep = s_gen.record_event()

        # ---- D(real) (training.py:162), concurrently with the generator's forward
        with engine.on_stream(s_real):
            s_real.wait_event(ev_start)
            self.den.prepare(dparams)
            ev_dprep = s_real.record_event()
            if int(getattr(a, "real_as_ids", 1)):
                real_soft, real_ids = None, captions
            else:
                real_soft = self.den.soft_input(torch.nn.functional.one_hot(captions, self.den.V).float())
                real_ids = None
            self.den.fwd(dparams, real_soft, real_ids, d_train, km[0], seeds[0], state=buf["st_real"], logits=lg[0], dev_scalars=scal, seed_slot=0)
            ev_real = s_real.record_event()
            self._mark("D(real) fwd done [s_real]", s_real)

        # ---- features (training.py:144-147) and one roll-out (training.py:150)
        fmap = None
        if self.cgan:
            trunk_feats = trunk() if callable(trunk) else trunk
            if self.attn:
                trunk_feats, fmap = trunk_feats
                fmap = fmap.view(fmap.shape[0], -1, fmap.shape[-1])
            feats = gen.encoder.forward_fused(images, train, trunk_feats=trunk_feats)
        else:
            feats = engine.embedding_fwd(gparams[0], buf["ones"])
        self._mark("encoder done", main)
        main.wait_event(ev_gprep)
        # (measured: issuing the roll-out from a high-priority stream does not win it CU slots from the look-ahead trunk pass --
        # it finished only after the whole trunk pass -- so it stays on the main stream)
        if self.attn:
            probs, ids, dst = self.dec.sample_fwd(gparams, feats, fmap, L, T, False, noise_u, seeds[3], state=buf["dec_state"],
                                                  out=buf["probs"], ids=buf["ids"], dev_scalars=scal, seed_slot=3)
        else:
            probs, ids, dst = self.dec.sample_fwd(gparams, feats, L, T, False, noise_u, seeds[3], state=buf["dec_state"],
                                                  out=buf["probs"], ids=buf["ids"], dev_scalars=scal, seed_slot=3)
        self._mark("roll-out done", main)

        # ---- D(fake), D(gen) (training.py:163-164): one pass up to the highway layer, two dropout draws + heads
        main.wait_event(ev_dprep)
        self.den.fwd(dparams, probs, None, d_train, km[1], seeds[1], state=buf["st_fake"], logits=lg[1], dev_scalars=scal, seed_slot=1)
        self.den.fwd_redrop(dparams, buf["st_fake"], buf["st_gen"], d_train, km[2], seeds[2], logits=lg[2], dev_scalars=scal, seed_slot=2)
        self._mark("D(fake), D(gen) fwd done", main)
        main.wait_event(ev_real)
        losses, lgrads = engine.gan_losses(a.adv_loss_type, lg[0], lg[1], lg[2], want_grads=train)
        out = {"losses": losses, "ids": ids, "probs": probs, "logits": lg}
        if not train:
            return out
        ev_loss = main.record_event()
        self._mark("losses done", main)

        # ---- G path on its stream: g_loss -> D(gen) input grad -> decoder -> encoder head (training.py:169 minus the step)
        with engine.on_stream(s_gen):
            s_gen.wait_event(ev_loss)
            if overlap and scal is None:         # (a captured step keeps `lgrads` alive in the graph's own pool)
                lgrads["dg_out"].record_stream(s_gen)
            if a.adv_loss_type == "rsgan":
                self.gen_arena.grad.zero_()                        # utils.py:48: g_loss has no path to G
                ev_dgen = s_gen.record_event()
            else:
                self.den.bwd(dparams, buf["st_gen"], probs, None, d_train, lgrads["dg_out"], False, True,
                             ws=buf["disc_ws_gen"], d_inp=buf["d_probs"])
                ev_dgen = s_gen.record_event()                     # D's weights are free to change from here on
                self._mark("D(gen) input-grad done [s_gen]", s_gen)
                early = self._early_bucket() if (self.reducer is not None and not self.attn) else None
                if self.attn:
                    self.dec.sample_bwd(gparams, dst, probs, ids, buf["d_probs"], T, False, ws=buf["dec_ws"], grads=g_grads + [buf["d_feat"]],
                                        dev_scalars=scal)
                elif early is None:
                    self.dec.sample_bwd(gparams, dst, probs, ids, buf["d_probs"], T, False, ws=buf["dec_ws"],
                                        grads=g_grads + [buf["d_feat"]], dev_scalars=scal)
                else:
                    # data parallel: the vocabulary projection's gradient (40 % of G's arena) is complete before BPTT starts;
                    # its all-reduce runs under BPTT and the weight-gradient products instead of after them
                    self.dec.sample_bwd(gparams, dst, probs, ids, buf["d_probs"], T, False, ws=buf["dec_ws"],
                                        grads=g_grads + [buf["d_feat"]], phases=1, dev_scalars=scal)
                    self.reducer.start(self.gen_arena.grad[early[0]:early[1]])
                    self.dec.sample_bwd(gparams, dst, probs, ids, buf["d_probs"], T, False, ws=buf["dec_ws"],
                                        grads=g_grads + [buf["d_feat"]], phases=2, dev_scalars=scal)
                if self.cgan:
                    gen.encoder.backward_fused(buf["d_feat"])
                else:   # features = embed(<S>) broadcast: fold d_features into row 1 of the embedding gradient
                    engine.embedding_bwd(buf["d_feat"], buf["ones"], self.dec.V, d_weight=g_grads[0], zero_first=False)
            ev_g = s_gen.record_event()
            self._mark("G path done [s_gen]", s_gen)

        # ---- D path on the main stream: d_loss -> D parameters (training.py:168), then D's clip + Adam
        self.disc_arena.grad.zero_()          # ONE fill; both passes accumulate (instead of a fill per small gradient tensor)
        if real_ids is not None:                # real (ids) and fake (soft) passes differentiated as one batch of 2B captions
            self.den.bwd(dparams, buf["st_rf"], probs, real_ids, d_train, lgrads["dd_real_fake"], True, False,
                         grads=d_grads, accumulate=True, ws=buf["disc_ws"])
        else:                                   # --real-as-ids 0: two dense passes
            half_ws = {k: v[:v.shape[0] // 2] for k, v in buf["disc_ws"].items()}
            self.den.bwd(dparams, buf["st_real"], real_soft, None, d_train, lgrads["dd_real"], True, False,
                         grads=d_grads, accumulate=True, ws=half_ws)
            self.den.bwd(dparams, buf["st_fake"], probs, None, d_train, lgrads["dd_fake"], True, False,
                         grads=d_grads, accumulate=True, ws=half_ws)
        if self.reducer is not None:
            # collectives run one after the other on the reducer's stream, in the order they become ready:
            # G's early bucket (above), D's arena (now), the rest of G's arena (when the G path is through)
            ev_dred = self.reducer.start(self.disc_arena.grad)
            with engine.on_stream(s_gen):
                early = self._early_bucket() if (a.adv_loss_type != "rsgan" and not self.attn) else None
                if early is None:
                    self.reducer.start(self.gen_arena.grad)
                else:
                    if early[0] > 0:
                        self.reducer.start(self.gen_arena.grad[:early[0]])
                    if early[1] < self.gen_arena.numel:
                        self.reducer.start(self.gen_arena.grad[early[1]:])
        self._mark("D path done", main)
        main.wait_event(ev_dgen)                 # D's weights are no longer read by the G path
        if self.reducer is not None:
            self.reducer.wait(ev_dred)
        if opt_step:
            self.disc_opt.step()                 # D's clip + Adam runs under the rest of the G path
        main.wait_event(ev_g)
        if self.reducer is not None:
            self.reducer.wait_all()
        if opt_step:
            self.gen_opt.step()
        self._mark("optimizers done", main)
        return out

    # ---------------------------------------------------------------- the step as replayed hipGraphs
    def _graph_key(self, B: int, L: int) -> tuple:
        a = self.args
        return (B, L, self.cgan, a.adv_loss_type, int(getattr(a, "real_as_ids", 1)), self.gen_arena.flat.data_ptr(), self.disc_arena.flat.data_ptr(),
                id(self.gen_opt), id(self.disc_opt), self.gen_opt.lr, self.gen_opt.clip_norm, self.disc_opt.lr, self.disc_opt.clip_norm,
                self.den.drop_p, bool(self.gen.training), bool(self.disc.training))

    def _segments(self, buf, scal):
        """The step as LINEAR launch sequences, one per stretch between two cross-stream dependencies: each is captured as its own
        hipGraph and replayed on its stream, with the events between them recorded / waited on eagerly.  (The whole step as ONE
        multi-branch graph replays slower than eager launches on this runtime -- 3.49 against 2.67 ms per step, measured, round 3:
        fork / join nodes inside a graph cost more than the stream events they replace -- while a linear graph replays at the
        eager rate: the trunk pass has always been one.)  Returns {name: callable}; shared state travels through ``ctx``."""
        a = self.args
        gen, disc = self.gen, self.disc
        caps = buf["caps_in"]
        B, L = caps.shape
        lg = buf["logits"]
        ctx = {}
        g_grads, d_grads = self._grad_lists()
        gparams = [p.detach() for p in gen.decoder.param_list()]
        dparams = [p.detach() for p in disc.param_list()]
        real_ids = int(getattr(a, "real_as_ids", 1)) != 0

        def d_real():                     # s_real: D's weight images + D(real) forward (training.py:162)
            self.den.prepare(dparams)
            if real_ids:
                ctx["real_soft"], ctx["real_ids"] = None, caps
            else:
                ctx["real_soft"] = self.den.soft_input(torch.nn.functional.one_hot(caps, self.den.V).float())
                ctx["real_ids"] = None
            self.den.fwd(dparams, ctx["real_soft"], ctx["real_ids"], True, None, 0, state=buf["st_real"], logits=lg[0], dev_scalars=scal, seed_slot=0)

        def rollout():                    # main: G's weight images, features (training.py:144-147), one roll-out (training.py:150)
            self.dec.prepare(gparams)
            if self.cgan:
                ctx["feats"] = gen.encoder.forward_fused(None, True, trunk_feats=buf["trunk_in"])
            else:
                ctx["feats"] = engine.embedding_fwd(gparams[0], buf["ones"])
            if self.attn:
                ctx["probs"], ctx["ids"], ctx["dst"] = self.dec.sample_fwd(gparams, ctx["feats"], buf["fmap_in"], L, 0.0, False, None, 0,
                                                                          state=buf["dec_state"], out=buf["probs"], ids=buf["ids"],
                                                                          dev_scalars=scal, seed_slot=3)
            else:
                ctx["probs"], ctx["ids"], ctx["dst"] = self.dec.sample_fwd(gparams, ctx["feats"], L, 0.0, False, None, 0, state=buf["dec_state"],
                                                                          out=buf["probs"], ids=buf["ids"], dev_scalars=scal, seed_slot=3)

        def d_fake():                     # main, behind D(real): D(fake), D(gen) (training.py:163-164), losses
            self.den.fwd(dparams, ctx["probs"], None, True, None, 0, state=buf["st_fake"], logits=lg[1], dev_scalars=scal, seed_slot=1)
            self.den.fwd_redrop(dparams, buf["st_fake"], buf["st_gen"], True, None, 0, logits=lg[2], dev_scalars=scal, seed_slot=2)
            ctx["losses"], ctx["lgrads"] = engine.gan_losses(a.adv_loss_type, lg[0], lg[1], lg[2], want_grads=True)

        def g_dgen():                     # s_gen, behind the losses: g_loss -> D(gen) input gradient (reads D's weights)
            if a.adv_loss_type == "rsgan":
                self.gen_arena.grad.zero_()                        # utils.py:48: g_loss has no path to G
            else:
                self.den.bwd(dparams, buf["st_gen"], ctx["probs"], None, True, ctx["lgrads"]["dg_out"], False, True,
                             ws=buf["disc_ws_gen"], d_inp=buf["d_probs"])

        def g_bwd():                      # s_gen: decoder BPTT + weight gradients, encoder head (training.py:169 minus the step)
            if a.adv_loss_type == "rsgan":
                return
            if self.attn:
                self.dec.sample_bwd(gparams, ctx["dst"], ctx["probs"], ctx["ids"], buf["d_probs"], 0.0, False, ws=buf["dec_ws"],
                                    grads=g_grads + [buf["d_feat"]], dev_scalars=scal)
            else:
                self.dec.sample_bwd(gparams, ctx["dst"], ctx["probs"], ctx["ids"], buf["d_probs"], 0.0, False, ws=buf["dec_ws"],
                                    grads=g_grads + [buf["d_feat"]], dev_scalars=scal)
            if self.cgan:
                gen.encoder.backward_fused(buf["d_feat"])
            else:
                engine.embedding_bwd(buf["d_feat"], buf["ones"], self.dec.V, d_weight=g_grads[0], zero_first=False)

        def d_bwd():                      # main, behind the losses: d_loss -> D parameters (training.py:168)
            self.disc_arena.grad.zero_()
            if real_ids:
                self.den.bwd(dparams, buf["st_rf"], ctx["probs"], ctx["real_ids"], True, ctx["lgrads"]["dd_real_fake"], True, False,
                             grads=d_grads, accumulate=True, ws=buf["disc_ws"])
            else:
                half_ws = {k: v[:v.shape[0] // 2] for k, v in buf["disc_ws"].items()}
                self.den.bwd(dparams, buf["st_real"], ctx["real_soft"], None, True, ctx["lgrads"]["dd_real"], True, False,
                             grads=d_grads, accumulate=True, ws=half_ws)
                self.den.bwd(dparams, buf["st_fake"], ctx["probs"], None, True, ctx["lgrads"]["dd_fake"], True, False,
                             grads=d_grads, accumulate=True, ws=half_ws)

        return ctx, {"d_real": d_real, "rollout": rollout, "d_fake": d_fake, "g_dgen": g_dgen, "g_bwd": g_bwd, "d_bwd": d_bwd,
                     "d_opt": self.disc_opt.step, "g_opt": self.gen_opt.step}

    def _call_graph(self, images, captions, B: int, L: int, next_images, next_train) -> dict:
        """Call 1 with a given key runs the segments eagerly (lazy code-object loads, LDS grants and buffer allocation are not
        capturable), call 2 captures each of them as it goes, later calls replay them.  Outside the graphs: the trunk look-ahead
        hand-over (its own replayed graph on its own stream), the copies of this batch's captions / trunk features into the step's
        static input buffers, the one-thread launch that writes temperature and seeds into device memory, and the stream events."""
        dev = captions.device
        main = torch.cuda.current_stream(dev)
        buf = self._buffers(B, L, dev)
        if "caps_in" not in buf:
            buf["caps_in"] = torch.empty(B, L, device=dev, dtype=torch.int64)
            buf["scal"] = engine.StepScalarsBuffer(dev)
            if self.cgan:
                buf["trunk_in"] = torch.empty(B, self.gen.encoder.resnet.out_features, device=dev, dtype=self.dec.act)
            if self.attn:
                buf["fmap_in"] = torch.empty(B, self.dec.P, self.dec.C, device=dev, dtype=self.dec.act)
        scal = buf["scal"]
        key = self._graph_key(B, L)
        g = self._graphs.get(key)
        if g is None:
            mode = "capture" if key in self._warm else "eager"
            self._warm.add(key)
            if mode == "capture":
                self._graphs.clear()             # at most one live set of step graphs (stale pointers never replay); dies here, outside capture
                self.dec._shadow_key = self.den._shadow_key = None      # the weight-image refresh is part of every replay
            ctx, seg = self._segments(buf, scal)
            g = {"ctx": ctx, "seg": seg, "graphs": {}, "mode": mode}
            if mode == "capture":
                self._graphs[key] = g
        s_real, s_gen = self._streams(dev)

        def run(name, stream):
            """One segment on ``stream`` (already made current by the caller): replay, or capture-then-replay, or eager."""
            gr = g["graphs"].get(name)
            if gr is not None:
                gr.replay()
            elif g["mode"] == "capture":
                gr = torch.cuda.CUDAGraph()
                with engine.capture_guard(), torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    g["seg"][name]()
                g["graphs"][name] = gr
                gr.replay()
            else:
                g["seg"][name]()

        try:
            buf["caps_in"].copy_(captions)
            scal.set(float(self.gen.decoder.temperature), [SEEDS.next() for _ in range(4)])
            ev_start = main.record_event()
            with engine.on_stream(s_real):
                s_real.wait_event(ev_start)
                run("d_real", s_real)
                ev_real = s_real.record_event()
            if self.cgan:
                tf = self._take_and_prefetch(images, True, main, ev_start, next_images, next_train)
                if self.attn:
                    buf["fmap_in"].copy_(tf[1].view(B, self.dec.P, self.dec.C))
                    tf = tf[0]
                buf["trunk_in"].copy_(tf)
            run("rollout", main)
            main.wait_event(ev_real)
            run("d_fake", main)
            ev_loss = main.record_event()
            with engine.on_stream(s_gen):
                s_gen.wait_event(ev_loss)
                run("g_dgen", s_gen)
                ev_dgen = s_gen.record_event()       # D's weights are free to change from here on
                run("g_bwd", s_gen)
                ev_g = s_gen.record_event()
            run("d_bwd", main)
            main.wait_event(ev_dgen)
            g["seg"]["d_opt"]()                      # two launches each: eager (D's clip + Adam runs under the rest of the G path)
            main.wait_event(ev_g)
            g["seg"]["g_opt"]()
        except Exception as exc:
            if g["mode"] != "capture":
                raise
            import warnings                          # capture refused: eager launches from here on (same kernels, same results)
            warnings.warn(f"hipGraph capture of the train step failed ({exc}); falling back to eager launches")
            self.use_graph = False
            self._graphs.clear()
            torch.cuda.synchronize()
            self.dec._shadow_key = self.den._shadow_key = None
            return self(images, captions, L, True, next_images=next_images, next_train=next_train)
        g["mode"] = "replay" if g["mode"] == "capture" else g["mode"]
        ctx = g["ctx"]
        return {"losses": ctx["losses"], "ids": ctx["ids"], "probs": ctx["probs"], "logits": buf["logits"]}

    def bind_optimizers(self, gen_opt, disc_opt) -> "FusedAdvStep":
        self.gen_opt, self.disc_opt = gen_opt, disc_opt
        return self
