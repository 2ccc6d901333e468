"""Generate tests/golden/*.npz from the reference's OWN classes.

Run in the build container only:  python -m oracle.make_golden
(imports /root/reference/src in place through oracle/ref_stub.py; nothing of the
reference is copied - the fixtures hold tensors only).

For every case the reference ``Decoder`` / ``Discriminator`` / ``get_losses`` /
``get_fixed_temperature`` run the SURVEY.md §8(c) step with stock
``clip_grad_norm_`` + ``torch.optim.Adam``; the Gumbel uniforms and dropout masks
the reference drew from the global generator are recovered by replaying the
generator (and verified against a forward hook on the reference's dropout
module), so the fixtures carry them as explicit inputs.  The restatement in
oracle/cpu_step.py is cross-checked here too, but the binding check is
tests/test_oracle_golden.py.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

from . import cpu_step as O
from . import ref_stub

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SAMPLE_STRIDE = 257


def _np(t):
    return t.detach().cpu().numpy().copy()      # copy: parameters are mutated in place by later steps


def summarize(t: torch.Tensor) -> np.ndarray:
    """[sum, abs-sum, l2] + strided sample; for tensors too large to commit."""
    f = t.detach().double().reshape(-1)
    head = torch.stack([f.sum(), f.abs().sum(), f.pow(2).sum().sqrt()])
    return np.concatenate([_np(head), _np(f[::SAMPLE_STRIDE])])


def build_reference(case):
    rg, rd, ru = ref_stub.load()
    args = ref_stub.make_args(case["V"], case["E"], case["H"], case["NL"], temperature=case["T0"],
                              disc_embed_dim=case["De"], disc_num_rep=case["R"],
                              filter_sizes=case["fs"], num_filters=case["nf"])
    dec = rg.Decoder(args)
    disc = rd.Discriminator(args)
    return dec, disc, ru


def run_case(case: dict) -> dict:
    torch.manual_seed(1008)                              # src/main.py:14
    B, L, V, E, H, NL = (case[k] for k in ("B", "L", "V", "E", "H", "NL"))
    R, De = case["R"], case["De"]
    F = sum(case["nf"])
    full = case["full"]
    pg = torch.Generator().manual_seed(case["param_seed"])
    feat_dim = case.get("trunk_feat_dim")
    gp = O.make_gen_params(V, E, H, NL, pg, trunk_feat_dim=feat_dim)
    dp = O.make_disc_params(V, pg, embed_dim=De, num_rep=R, filter_sizes=case["fs"], num_filters=case["nf"])
    if case.get("g_scale", 1.0) != 1.0:
        gp = {k: v * case["g_scale"] for k, v in gp.items()}
    caps = O.make_captions(B, L, V, pg)
    trunk_feat = torch.randn(B, feat_dim, generator=pg) if feat_dim else None

    dec, disc, ru = build_reference(case)
    dec.load_state_dict({k[len("decoder."):]: v.clone() for k, v in gp.items() if k.startswith("decoder.")})
    disc.load_state_dict({k: v.clone() for k, v in dp.items()})
    head = None
    g_modules = [dec]
    if feat_dim:
        head_lin = torch.nn.Linear(feat_dim, E)           # generator.py:15
        head_bn = torch.nn.BatchNorm1d(E, momentum=0.01)  # generator.py:16
        with torch.no_grad():
            head_lin.weight.copy_(gp["encoder.linear.weight"]); head_lin.bias.copy_(gp["encoder.linear.bias"])
            head_bn.weight.copy_(gp["encoder.bn.weight"]); head_bn.bias.copy_(gp["encoder.bn.bias"])
        head = (head_lin, head_bn)
        g_modules += [head_lin, head_bn]
    g_params = [p for m in g_modules for p in m.parameters()]
    gen_opt = torch.optim.Adam(g_params, lr=case["gen_lr"])       # training.py:25
    disc_opt = torch.optim.Adam(disc.parameters(), lr=case["disc_lr"])  # training.py:26
    train = case.get("train", True)
    (disc.train(), dec.train()) if train else (disc.eval(), dec.eval())
    for m in g_modules[1:]:
        m.train(train)

    # restatement state, advanced in lockstep
    my_gp = {k: v.clone() for k, v in gp.items()}
    my_dp = {k: v.clone() for k, v in dp.items()}
    my_gopt, my_dopt = O.AdamState(case["gen_lr"]), O.AdamState(case["disc_lr"])
    bn_running = {"running_mean": torch.zeros(E), "running_var": torch.ones(E)} if feat_dim else None

    out = {"caps": _np(caps)}
    if full:
        for k, v in gp.items():
            out[f"gp0/{k}"] = _np(v)
        for k, v in dp.items():
            out[f"dp0/{k}"] = _np(v)
    else:
        for k, v in {**gp, **dp}.items():
            out[f"p0sum/{k}"] = summarize(v)
    if trunk_feat is not None:
        out["trunk_feat"] = _np(trunk_feat)

    temps = []
    captured = []
    hook = disc.dropout.register_forward_hook(lambda m, i, o: captured.append((i[0].detach().clone(), o.detach().clone())))
    for step in range(case["steps"]):
        T = dec.temperature
        temps.append(float(T))
        torch.manual_seed(case["noise_seed"] + step)
        captured.clear()
        # ---- the reference step ----
        with torch.enable_grad() if train else torch.no_grad():
            if head is not None:
                feats = head[1](head[0](trunk_feat))                       # generator.py:24
            else:
                feats = dec.embed(torch.ones(B, 1, dtype=torch.long).squeeze(1))   # training.py:147
            gen, ids = dec.sample(feats, max_caption_len=L)                # training.py:150
            real = torch.nn.functional.one_hot(caps, V).float()            # training.py:158
            d_r = disc(real); d_f = disc(gen.detach()); g_o = disc(gen)    # training.py:162-164
            g_loss, d_loss = ru.get_losses(d_r, d_f, g_o, case["loss"])    # training.py:165
        pre = f"s{step}/"
        out[pre + "probs"] = _np(gen); out[pre + "ids"] = _np(ids)
        out[pre + "d_real"] = _np(d_r); out[pre + "d_fake"] = _np(d_f); out[pre + "g_out"] = _np(g_o)
        out[pre + "g_loss"] = np.float64(g_loss.item()); out[pre + "d_loss"] = np.float64(d_loss.item())
        if train:
            disc_opt.zero_grad(); gen_opt.zero_grad()
            d_loss.backward(retain_graph=True)
            d_raw = {k: p.grad.clone() for k, p in disc.named_parameters()}
            d_norm = torch.nn.utils.clip_grad_norm_(disc.parameters(), case["clip"])   # training.py:198
            d_stash = [p.grad.clone() for p in disc.parameters()]
            g_loss.backward()
            g_raw = {}
            for k, p in dec.named_parameters():
                if p.grad is not None:          # rsgan's g_loss has no path to G (utils.py:48)
                    g_raw["decoder." + k] = p.grad.clone()
            if head is not None and head[0].weight.grad is not None:
                g_raw["encoder.linear.weight"] = head[0].weight.grad.clone(); g_raw["encoder.linear.bias"] = head[0].bias.grad.clone()
                g_raw["encoder.bn.weight"] = head[1].weight.grad.clone(); g_raw["encoder.bn.bias"] = head[1].bias.grad.clone()
            g_norm = torch.nn.utils.clip_grad_norm_(g_params, case["clip"]) if g_raw else 0.0
            for p, g in zip(disc.parameters(), d_stash):
                p.grad = g
            disc_opt.step(); gen_opt.step()
            out[pre + "d_norm"] = np.float64(float(d_norm)); out[pre + "g_norm"] = np.float64(float(g_norm))
            for k, g in {**d_raw, **g_raw}.items():
                out[pre + "grad/" + k] = _np(g) if full else summarize(g)
            post = {("decoder." + k): p for k, p in dec.named_parameters()}
            post.update({k: p for k, p in disc.named_parameters()})
            if head is not None:
                post.update({"encoder.linear.weight": head[0].weight, "encoder.linear.bias": head[0].bias,
                             "encoder.bn.weight": head[1].weight, "encoder.bn.bias": head[1].bias})
            for k, p in post.items():
                out[pre + "post/" + k] = _np(p) if full else summarize(p)
        # temperature update (training.py:183,190-191): adv_epoch + k/len(loader), adv_epochs
        dec.temperature = ru.get_fixed_temperature(case["T0"], case["adv_epoch"] + (step + 1) / case["n_batches"],
                                                   case["adv_epochs"], case["adapt"])

        # ---- recover the noise the reference drew, by replaying the generator ----
        torch.manual_seed(case["noise_seed"] + step)
        us = [torch.zeros(B, V).uniform_(0, 1) for _ in range(L)]          # generator.py:86-90
        masks = None
        if train:
            masks = [torch.empty(B * R, F).bernoulli_(1 - O.DROPOUT_P) for _ in range(3)]
            assert len(captured) == 3
            for (xin, xout), mk in zip(captured, masks):
                assert torch.equal(xin * (mk / (1 - O.DROPOUT_P)), xout), "dropout replay mismatch"
            out[pre + "masks"] = np.packbits(np.stack([_np(mk).astype(np.uint8) for mk in masks]), axis=-1)
        out[pre + "u"] = _np(torch.stack(us))

        # ---- restatement cross-check on the same inputs ----
        mine = O.adv_step(my_gp, my_dp, caps, us, masks, T, case["loss"], case["clip"],
                          my_gopt if train else None, my_dopt if train else None,
                          trunk_feat=trunk_feat, num_rep=R, bn_running=bn_running, train=train)
        assert torch.equal(mine["ids"], ids), f"{case['name']} step {step}: ids differ"
        torch.testing.assert_close(mine["probs"], gen.detach(), rtol=1e-4, atol=1e-6)
        assert abs(mine["g_loss"] - g_loss.item()) < 1e-5 and abs(mine["d_loss"] - d_loss.item()) < 1e-5
        if step == 0 and case.get("stages"):
            for call in ("real", "fake", "gen"):
                for nm, t in mine["stages"][call].items():
                    out[f"s0/stage/{call}/{nm}"] = _np(t)
        if train:
            for k, g in g_raw.items():
                torch.testing.assert_close(mine["g_grads_raw"][k], g, rtol=2e-3, atol=1e-7)
            for k, g in d_raw.items():
                torch.testing.assert_close(mine["d_grads_raw"][k], g, rtol=2e-3, atol=1e-7)
    hook.remove()
    if train and full:
        names_g = ["decoder." + k for k, _ in dec.named_parameters()]
        if head is not None:
            names_g += ["encoder.linear.weight", "encoder.linear.bias", "encoder.bn.weight", "encoder.bn.bias"]
        for nm, p in zip(names_g, g_params):
            if p in gen_opt.state:
                out["adam/m/" + nm] = _np(gen_opt.state[p]["exp_avg"]); out["adam/v/" + nm] = _np(gen_opt.state[p]["exp_avg_sq"])
        for nm, p in disc.named_parameters():
            out["adam/m/" + nm] = _np(disc_opt.state[p]["exp_avg"]); out["adam/v/" + nm] = _np(disc_opt.state[p]["exp_avg_sq"])
    if head is not None:
        out["bn_running_mean"] = _np(head[1].running_mean); out["bn_running_var"] = _np(head[1].running_var)
    meta = dict(case)
    meta["temperatures"] = temps
    out["meta"] = np.array(json.dumps(meta))
    return out


def run_pretrain_case(case: dict) -> dict:
    """MLE pre-train step: sample(pretrain=True) + CrossEntropyLoss (training.py:53-95)."""
    torch.manual_seed(1008)
    B, L, V, E, H, NL = (case[k] for k in ("B", "L", "V", "E", "H", "NL"))
    pg = torch.Generator().manual_seed(case["param_seed"])
    gp = O.make_gen_params(V, E, H, NL, pg)
    caps = O.make_captions(B, L, V, pg)
    caps[:, L - 2:] = 0                                   # some PAD positions (counted in the mean)
    rg, _, _ = ref_stub.load()
    args = ref_stub.make_args(V, E, H, NL)
    dec = rg.Decoder(args)
    dec.load_state_dict({k[len("decoder."):]: v.clone() for k, v in gp.items()})
    opt = torch.optim.Adam(dec.parameters(), lr=case["pretrain_lr"])       # training.py:24
    my_gp = {k: v.clone() for k, v in gp.items()}
    my_opt = O.AdamState(case["pretrain_lr"])
    out = {"caps": _np(caps)}
    for k, v in gp.items():
        out[f"gp0/{k}"] = _np(v)
    for step in range(case["steps"]):
        feats = dec.embed(torch.ones(B, 1, dtype=torch.long).squeeze(1))   # training.py:68
        logits, ids = dec.sample(feats, pretrain=True, max_caption_len=L)  # training.py:71
        loss = torch.nn.CrossEntropyLoss()(logits.view(-1, logits.size(-1)), caps.view(-1))   # training.py:81-83
        opt.zero_grad(); loss.backward()
        raw = {"decoder." + k: p.grad.clone() for k, p in dec.named_parameters()}
        norm = torch.nn.utils.clip_grad_norm_(dec.parameters(), case["clip"])
        opt.step()
        pre = f"s{step}/"
        out[pre + "logits"] = _np(logits); out[pre + "ids"] = _np(ids)
        out[pre + "loss"] = np.float64(loss.item()); out[pre + "g_norm"] = np.float64(float(norm))
        for k, g in raw.items():
            out[pre + "grad/" + k] = _np(g)
        for k, p in dec.named_parameters():
            out[pre + "post/decoder." + k] = _np(p)
        mine = O.pretrain_step(my_gp, caps, case["clip"], my_opt)
        assert torch.equal(mine["ids"], ids)
        assert abs(mine["loss"] - loss.item()) < 1e-5
    out["meta"] = np.array(json.dumps(case))
    return out


def run_forward_case(case: dict) -> dict:
    """Decoder.forward, the teacher-forced decode (generator.py:39-53), in both modes; dead on the training path but part of the
    module surface."""
    torch.manual_seed(1008)
    B, L, V, E, H, NL = (case[k] for k in ("B", "L", "V", "E", "H", "NL"))
    pg = torch.Generator().manual_seed(case["param_seed"])
    gp = O.make_gen_params(V, E, H, NL, pg)
    caps = O.make_captions(B, L, V, pg)
    feats = torch.randn(B, E, generator=pg) * 0.3
    lengths = case["lengths"]
    rg, _, _ = ref_stub.load()
    args = ref_stub.make_args(V, E, H, NL, temperature=case["T"])
    dec = rg.Decoder(args)
    dec.load_state_dict({k[len("decoder."):]: v.clone() for k, v in gp.items()})
    out = {"caps": _np(caps), "feats": _np(feats), "lengths": np.array(lengths, dtype=np.int32)}
    for k, v in gp.items():
        out[f"gp0/{k}"] = _np(v)
    with torch.no_grad():
        logits, (h_n, c_n) = dec(feats, caps, torch.tensor(lengths), pretrain=True)
        state = torch.get_rng_state()
        probs, (h_n2, c_n2) = dec(feats, caps, torch.tensor(lengths), pretrain=False)
        torch.set_rng_state(state)
        u = torch.zeros(probs.size()).uniform_(0, 1)               # the draw add_gumbel made (generator.py:86-90)
    assert torch.equal(h_n, h_n2) and torch.equal(c_n, c_n2)
    mine_l, (mh, mc) = O.decoder_forward_tf(gp, feats, caps, lengths, case["T"], pretrain=True)
    mine_p, _ = O.decoder_forward_tf(gp, feats, caps, lengths, case["T"], pretrain=False, u=u)
    assert torch.allclose(mine_l, logits, rtol=1e-5, atol=1e-6) and torch.allclose(mine_p, probs, rtol=1e-5, atol=1e-7)
    assert torch.allclose(mh, h_n, rtol=1e-5, atol=1e-6) and torch.allclose(mc, c_n, rtol=1e-5, atol=1e-6)
    out.update({"logits": _np(logits), "probs": _np(probs), "u": _np(u), "h_n": _np(h_n), "c_n": _np(c_n)})
    out["meta"] = np.array(json.dumps(case))
    return out


def run_api_case(case: dict) -> dict:
    """Module-API corners the trainer never exercises but the signatures offer:
    ``Decoder.sample(features, states=(h0, c0))`` with gradients into the states (generator.py:55,61);
    ``Discriminator(args, dropout=p)`` with p != 0.2 (discriminator.py:10,30);
    ``Decoder.forward`` differentiated (teacher forcing with packed sequences, generator.py:39-53)."""
    torch.manual_seed(1008)
    B, L, V, E, H, NL = (case[k] for k in ("B", "L", "V", "E", "H", "NL"))
    R, De, F = case["R"], case["De"], sum(case["nf"])
    pg = torch.Generator().manual_seed(case["param_seed"])
    gp = O.make_gen_params(V, E, H, NL, pg)
    dp = O.make_disc_params(V, pg, embed_dim=De, num_rep=R, filter_sizes=case["fs"], num_filters=case["nf"])
    gp = {k: v * case["g_scale"] for k, v in gp.items()}
    caps = O.make_captions(B, L, V, pg)
    feats = torch.randn(B, E, generator=pg) * 0.3
    h0 = torch.randn(NL, B, H, generator=pg) * 0.5
    c0 = torch.randn(NL, B, H, generator=pg) * 0.5
    rg, rd, _ = ref_stub.load()
    args = ref_stub.make_args(V, E, H, NL, temperature=case["T"], disc_embed_dim=De, disc_num_rep=R,
                              filter_sizes=case["fs"], num_filters=case["nf"])
    dec = rg.Decoder(args)
    dec.load_state_dict({k[len("decoder."):]: v.clone() for k, v in gp.items()})
    out = {"caps": _np(caps), "feats": _np(feats), "h0": _np(h0), "c0": _np(c0)}
    for k, v in {**gp, **dp}.items():
        out[f"p0/{k}"] = _np(v)

    # ---- sample(states=...)
    f_l, h_l, c_l = feats.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    torch.manual_seed(case["noise_seed"])
    probs, ids = dec.sample(f_l, states=(h_l, c_l), max_caption_len=L)
    torch.manual_seed(case["noise_seed"])
    us = [torch.zeros(B, V).uniform_(0, 1) for _ in range(L)]
    d_out = torch.randn(probs.shape, generator=pg)
    dec.zero_grad()
    (probs * d_out).sum().backward()
    out.update({"st/u": _np(torch.stack(us)), "st/probs": _np(probs), "st/ids": _np(ids), "st/d_out": _np(d_out),
                "st/d_feats": _np(f_l.grad), "st/d_h0": _np(h_l.grad), "st/d_c0": _np(c_l.grad)})
    for k, p_ in dec.named_parameters():
        out["st/grad/decoder." + k] = _np(p_.grad)
    mine_p, mine_i = O.decoder_sample(gp, feats, L, case["T"], us, states=(h0, c0))
    assert torch.equal(mine_i, ids) and torch.allclose(mine_p, probs, rtol=1e-4, atol=1e-7)

    # ---- Decoder.forward differentiated (both modes share the LSTM; pretrain=True keeps the check free of noise)
    lengths = case["lengths"]
    f_l = feats.clone().requires_grad_(True)
    dec.zero_grad()
    logits, (h_n, c_n) = dec(f_l, caps, torch.tensor(lengths), pretrain=True)
    d_log = torch.randn(logits.shape, generator=pg)
    (logits * d_log).sum().backward()
    out.update({"tf/lengths": np.array(lengths, dtype=np.int32), "tf/logits": _np(logits), "tf/d_logits": _np(d_log),
                "tf/d_feats": _np(f_l.grad)})
    for k, p_ in dec.named_parameters():
        out["tf/grad/decoder." + k] = _np(p_.grad)

    # ---- Discriminator(dropout=p)
    disc = rd.Discriminator(args, dropout=case["dropout"])
    disc.load_state_dict({k: v.clone() for k, v in dp.items()})
    disc.train()
    captured = []
    hook = disc.dropout.register_forward_hook(lambda m_, i, o: captured.append((i[0].detach().clone(), o.detach().clone())))
    soft = torch.softmax(torch.randn(B, L, V, generator=pg) * 2, -1).requires_grad_(True)
    torch.manual_seed(case["noise_seed"] + 1)
    logit = disc(soft)
    torch.manual_seed(case["noise_seed"] + 1)
    mask = torch.empty(B * R, F).bernoulli_(1 - case["dropout"])
    assert torch.equal(captured[0][0] * (mask / (1 - case["dropout"])), captured[0][1]), "dropout replay mismatch"
    hook.remove()
    d_logit = torch.randn(logit.shape, generator=pg)
    disc.zero_grad()
    (logit * d_logit).sum().backward()
    out.update({"dr/inp": _np(soft), "dr/mask": np.packbits(_np(mask).astype(np.uint8), axis=-1), "dr/logits": _np(logit),
                "dr/d_logits": _np(d_logit), "dr/d_inp": _np(soft.grad)})
    for k, p_ in disc.named_parameters():
        out["dr/grad/" + k] = _np(p_.grad)
    mine = O.disc_forward(dp, soft.detach(), mask, R, dropout_p=case["dropout"])
    assert torch.allclose(mine, logit, rtol=1e-4, atol=1e-6)
    out["meta"] = np.array(json.dumps(case))
    return out


def run_scalar_cases() -> dict:
    """get_losses for every functional loss type + the temperature schedules."""
    _, _, ru = ref_stub.load()
    g = torch.Generator().manual_seed(11)
    d_r, d_f, g_o = (torch.randn(96, generator=g) * 3 for _ in range(3))
    out = {"d_real": _np(d_r), "d_fake": _np(d_f), "g_out": _np(g_o)}
    for lt in ("standard", "JS", "KL", "rsgan"):
        gl, dl = ru.get_losses(d_r, d_f, g_o, lt)
        out[f"loss/{lt}"] = np.array([gl.item(), dl.item()], dtype=np.float64)
    broken = []
    for lt in ("hinge", "tv"):
        try:
            ru.get_losses(d_r, d_f, g_o, lt)
        except TypeError:
            broken.append(lt)
    out["broken_in_reference"] = np.array(json.dumps(broken))
    iters = [0.0, 0.02, 1.0, 7.5, 25.0, 29.98]
    for adapt in ("no", "lin", "exp", "log", "sigmoid", "quad", "sqrt"):
        out[f"temp/{adapt}"] = np.array([float(ru.get_fixed_temperature(100, i, 30, adapt)) for i in iters], dtype=np.float64)
    out["temp_iters"] = np.array(iters)
    return out


BASE = dict(R=64, De=64, fs=[3, 4, 5], loss="standard", clip=5.0, gen_lr=1e-4, disc_lr=1e-4, T0=100,
            adapt="exp", adv_epoch=0, adv_epochs=30, n_batches=50, param_seed=2024, noise_seed=7)

CASES = [
    dict(BASE, name="tiny", B=4, L=5, V=50, E=8, H=16, NL=2, nf=[20, 30, 10], steps=3, full=True, stages=True),
    dict(BASE, name="tiny_scaled", B=4, L=6, V=50, E=8, H=16, NL=2, nf=[20, 30, 10], steps=2, full=True, g_scale=20.0),
    dict(BASE, name="tiny_rep2", B=3, L=7, V=40, E=8, H=16, NL=1, nf=[12, 8, 16], De=128, steps=1, full=True),
    dict(BASE, name="tiny_rsgan", B=4, L=5, V=50, E=8, H=16, NL=1, nf=[20, 30, 10], steps=1, full=True, loss="rsgan"),
    dict(BASE, name="tiny_eval", B=4, L=5, V=50, E=8, H=16, NL=2, nf=[20, 30, 10], steps=1, full=True, train=False),
    dict(BASE, name="tiny_cgan_head", B=6, L=5, V=50, E=8, H=16, NL=1, nf=[20, 30, 10], steps=2, full=True, trunk_feat_dim=24),
    dict(BASE, name="cfg1", B=8, L=10, V=64, E=32, H=512, NL=1, nf=[300, 300, 300], steps=2, full=False),
]
FORWARD_TF = dict(name="forward_tf_tiny", B=4, L=6, V=50, E=8, H=16, NL=2, T=1.3, lengths=[7, 5, 3, 6], param_seed=2025)
API = dict(name="api_tiny", B=4, L=6, V=50, E=8, H=16, NL=2, R=64, De=64, fs=[3, 4, 5], nf=[20, 30, 10], T=1.3, g_scale=8.0,
           lengths=[7, 5, 3, 6], dropout=0.5, param_seed=2026, noise_seed=17)
PRETRAIN = dict(name="pretrain_tiny", B=4, L=6, V=50, E=8, H=16, NL=2, steps=2, clip=5.0, pretrain_lr=1e-2, param_seed=2024)


def main() -> int:
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for case in CASES:
        res = run_case(case)
        path = os.path.join(GOLDEN_DIR, case["name"] + ".npz")
        np.savez_compressed(path, **res)
        print(f"{case['name']}: {os.path.getsize(path) / 1024:.0f} KiB  g_loss={res['s0/g_loss']:.6f} d_loss={res['s0/d_loss']:.6f}")
    res = run_pretrain_case(PRETRAIN)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "pretrain_tiny.npz"), **res)
    print("pretrain_tiny: loss", res["s0/loss"])
    np.savez_compressed(os.path.join(GOLDEN_DIR, "forward_tf_tiny.npz"), **run_forward_case(FORWARD_TF))
    print("forward_tf_tiny written")
    np.savez_compressed(os.path.join(GOLDEN_DIR, "api_tiny.npz"), **run_api_case(API))
    print("api_tiny written")
    np.savez_compressed(os.path.join(GOLDEN_DIR, "scalars.npz"), **run_scalar_cases())
    return 0


if __name__ == "__main__":
    sys.exit(main())
