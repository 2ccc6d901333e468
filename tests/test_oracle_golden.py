"""The oracle (oracle/cpu_step.py) against the golden vectors generated from the
reference's own classes (oracle/make_golden.py).  CPU only."""
import math

import pytest
import torch

from oracle import cpu_step as O
from tests.golden_io import Golden, initial_params, summarize

ADV_CASES = ["tiny", "tiny_scaled", "tiny_rep2", "tiny_rsgan", "tiny_eval", "tiny_cgan_head", "cfg1"]


def grad_close(got, want, what):
    """grads: rtol 2e-3 plus an absolute floor of 1e-4 x the tensor's largest entry
    (fp32 reduction-order noise; at the reference init G's grads are ~1e-7)."""
    scale = float(want.abs().max()) if want.numel() else 0.0
    torch.testing.assert_close(got.double(), want.double(), rtol=2e-3, atol=1e-4 * scale + 1e-12,
                               msg=lambda s: f"{what}: {s}")


@pytest.mark.parametrize("name", ADV_CASES)
def test_adv_step_matches_reference(name):
    """Per step: forward outputs, losses, raw grads and norms on the reference's
    trajectory (params teacher-forced to the golden post-step weights when the
    fixture ships them), then clip+Adam driven by the GOLDEN grads so that the
    optimizer restatement is checked tightly (Adam divides by |g|, so feeding it
    our own grads would amplify fp32 noise on near-zero entries up to ~lr)."""
    g = Golden(name)
    m = g.meta
    gp, dp = initial_params(g)
    caps = g.t("caps")
    train = m.get("train", True)
    full = m["full"]
    gopt, dopt = O.AdamState(m["gen_lr"]), O.AdamState(m["disc_lr"])
    trunk = g.t("trunk_feat") if g.has("trunk_feat") else None
    running = {"running_mean": torch.zeros(m["E"]), "running_var": torch.ones(m["E"])} if trunk is not None else None
    for step in range(m["steps"]):
        T = m["temperatures"][step]
        if step > 0:   # get_fixed_temperature restated (utils.py:55-76)
            assert T == pytest.approx(O.get_fixed_temperature(m["T0"], m["adv_epoch"] + step / m["n_batches"], m["adv_epochs"], m["adapt"]), rel=1e-12)
        loose = (not full) and step > 0        # own trajectory: params differ by <= lr per entry
        out = O.adv_step(gp, dp, caps, g.us(step), g.masks(step) if train else None, T, m["loss"], m["clip"],
                         None, None, trunk_feat=trunk, num_rep=m["R"], bn_running=running, train=train)
        pre = f"s{step}/"
        assert torch.equal(out["ids"], g.t(pre + "ids")), f"step {step} ids"
        torch.testing.assert_close(out["probs"], g.t(pre + "probs"), rtol=1e-3 if loose else 1e-4, atol=1e-6)
        for k in ("d_real", "d_fake", "g_out"):
            torch.testing.assert_close(out[k], g.t(pre + k), rtol=1e-4, atol=1e-4 if loose else 1e-6)
        assert out["g_loss"] == pytest.approx(float(g.t(pre + "g_loss")), rel=1e-4 if loose else 1e-5)
        assert out["d_loss"] == pytest.approx(float(g.t(pre + "d_loss")), rel=1e-4 if loose else 1e-5)
        if not train:
            continue
        assert out["d_norm"] == pytest.approx(float(g.t(pre + "d_norm")), rel=1e-2 if loose else 1e-4)
        assert out["g_norm"] == pytest.approx(float(g.t(pre + "g_norm")), rel=1e-2 if loose else 1e-4, abs=1e-12)
        raw = {**out["d_grads_raw"], **out["g_grads_raw"]}
        want = g.group(pre + "grad/")
        assert set(raw) == set(want)
        if not loose:
            for k, w in want.items():
                grad_close(raw[k] if full else summarize(raw[k])[3:], w if full else w[3:], k)
        if full:   # optimizer restatement on the golden grads
            dg, dn = O.clip_grad_norm({k: v for k, v in want.items() if k in dp}, m["clip"])
            gg, gn = O.clip_grad_norm({k: v for k, v in want.items() if k in gp}, m["clip"])
            assert dn == pytest.approx(float(g.t(pre + "d_norm")), rel=1e-5)
            dopt.step(dp, dg)
            gopt.step(gp, gg)
            for k, w in g.group(pre + "post/").items():
                torch.testing.assert_close({**gp, **dp}[k], w, rtol=1e-6, atol=1e-9, msg=lambda s: f"post {k}: {s}")
            gp = {k: g.t(pre + "post/" + k).clone() for k in gp if g.has(pre + "post/" + k)} | {k: v for k, v in gp.items() if not g.has(pre + "post/" + k)}
            dp = {k: g.t(pre + "post/" + k).clone() for k in dp}
        else:
            dopt.step(dp, out["d_grads"])
            gopt.step(gp, out["g_grads"])
            for k, w in g.group(pre + "post/").items():
                got = summarize({**gp, **dp}[k])[3:]
                lr = m["gen_lr"] if k in gp else m["disc_lr"]
                torch.testing.assert_close(got, w[3:], rtol=0, atol=1.05 * lr * (step + 1), msg=lambda s: f"post {k}: {s}")
    if train and full:
        st = {**{k: (gopt.m[k], gopt.v[k]) for k in gopt.m}, **{k: (dopt.m[k], dopt.v[k]) for k in dopt.m}}
        for k, w in g.group("adam/m/").items():
            torch.testing.assert_close(st[k][0], w, rtol=1e-5, atol=1e-12)
        for k, w in g.group("adam/v/").items():
            torch.testing.assert_close(st[k][1], w, rtol=1e-5, atol=1e-16)
    if trunk is not None:
        torch.testing.assert_close(running["running_mean"], g.t("bn_running_mean"), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(running["running_var"], g.t("bn_running_var"), rtol=1e-5, atol=1e-7)


def test_stage_activations_tiny():
    g = Golden("tiny")
    gp, dp = initial_params(g)
    out = O.adv_step(gp, dp, g.t("caps"), g.us(0), g.masks(0), g.meta["temperatures"][0], train=False)
    # train=False skips the optimizers but masks were passed -> same activations as the train step
    for call in ("real", "fake", "gen"):
        for nm, t in out["stages"][call].items():
            torch.testing.assert_close(t.detach(), g.t(f"s0/stage/{call}/{nm}"), rtol=1e-4, atol=1e-6)


def test_known_answer_ids_from_survey():
    """SURVEY.md §8(c) known answer is regenerated in make_golden; here: ids of the tiny case are
    reproduced bit-exactly from the shipped u (replay contract of generator.py:86-90)."""
    g = Golden("tiny")
    gp, _ = initial_params(g)
    feats = O.start_features(gp, g.meta["B"])
    _, ids = O.decoder_sample(gp, feats, g.meta["L"], g.meta["temperatures"][0], g.us(0))
    assert torch.equal(ids, g.t("s0/ids"))


def test_pretrain_step_matches_reference():
    g = Golden("pretrain_tiny")
    m = g.meta
    gp = g.group("gp0/")
    opt = O.AdamState(m["pretrain_lr"])
    for step in range(m["steps"]):
        out = O.pretrain_step(gp, g.t("caps"), m["clip"], opt)
        pre = f"s{step}/"
        assert torch.equal(out["ids"], g.t(pre + "ids"))
        torch.testing.assert_close(out["logits"], g.t(pre + "logits"), rtol=1e-4, atol=1e-6)
        assert out["loss"] == pytest.approx(float(g.t(pre + "loss")), rel=1e-5)
        assert out["g_norm"] == pytest.approx(float(g.t(pre + "g_norm")), rel=1e-4)
        for k, w in g.group(pre + "grad/").items():
            torch.testing.assert_close(out["g_grads_raw"][k], w, rtol=2e-3, atol=2e-7)
        for k, w in g.group(pre + "post/").items():
            torch.testing.assert_close(gp[k], w, rtol=1e-4, atol=2e-5)


def test_teacher_forced_forward_matches_reference():
    """Decoder.forward (generator.py:39-53) of the reference, packed variable-length sequences, both modes."""
    g = Golden("forward_tf_tiny")
    m = g.meta
    gp = g.group("gp0/")
    lengths = [int(v) for v in g.t("lengths")]
    logits, (h_n, c_n) = O.decoder_forward_tf(gp, g.t("feats"), g.t("caps"), lengths, m["T"], pretrain=True)
    probs, _ = O.decoder_forward_tf(gp, g.t("feats"), g.t("caps"), lengths, m["T"], pretrain=False, u=g.t("u"))
    torch.testing.assert_close(logits, g.t("logits"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(probs, g.t("probs"), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(h_n, g.t("h_n"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(c_n, g.t("c_n"), rtol=1e-5, atol=1e-6)
    assert logits.shape[1] == max(lengths)
    # padded positions see a zero LSTM output: the projection returns its bias there
    b, t = 2, lengths[2]
    torch.testing.assert_close(logits[b, t], gp["decoder.linear.bias"], rtol=1e-6, atol=1e-7)


def test_losses_and_temperature_scalars():
    g = Golden("scalars")
    d_r, d_f, g_o = g.t("d_real"), g.t("d_fake"), g.t("g_out")
    for lt in ("standard", "JS", "KL", "rsgan"):
        gl, dl = O.get_losses(d_r, d_f, g_o, lt)
        want = g.t(f"loss/{lt}")
        assert float(gl) == pytest.approx(float(want[0]), rel=1e-5)
        assert float(dl) == pytest.approx(float(want[1]), rel=1e-5)
    with pytest.raises(NotImplementedError):
        O.get_losses(d_r, d_f, g_o, "nope")
    iters = g.t("temp_iters").tolist()
    for adapt in ("no", "lin", "exp", "log", "sigmoid", "quad", "sqrt"):
        want = g.t(f"temp/{adapt}").tolist()
        for i, w in zip(iters, want):
            assert O.get_fixed_temperature(100, i, 30, adapt) == pytest.approx(w, rel=1e-12)
    with pytest.raises(Exception):
        O.get_fixed_temperature(100, 1, 30, "bogus")
    assert math.isclose(O.get_fixed_temperature(100, 25, 50, "exp"), 10.0, rel_tol=1e-12)


def test_api_corners_match_reference():
    """sample(states=...), Decoder.forward differentiated, Discriminator(dropout=0.5): golden api_tiny.npz (reference's own classes)."""
    import numpy as np
    g = Golden("api_tiny")
    m = g.meta
    P = g.group("p0/")
    gp = {k: v for k, v in P.items() if k.startswith("decoder.")}
    dp = {k: v for k, v in P.items() if not k.startswith("decoder.")}
    leaf = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    f, h0, c0 = (g.t(k).clone().requires_grad_(True) for k in ("feats", "h0", "c0"))
    u = g.t("st/u")
    probs, ids = O.decoder_sample(leaf, f, m["L"], m["T"], [u[t] for t in range(m["L"])], states=(h0, c0))
    assert torch.equal(ids, g.t("st/ids"))
    torch.testing.assert_close(probs, g.t("st/probs"), rtol=1e-4, atol=1e-7)
    (probs * g.t("st/d_out")).sum().backward()
    for got, key in ((f.grad, "st/d_feats"), (h0.grad, "st/d_h0"), (c0.grad, "st/d_c0")):
        grad_close(got, g.t(key), key)
    for k, w in g.group("st/grad/").items():
        grad_close(leaf[k].grad, w, "st " + k)
    # teacher-forced forward, differentiated
    leaf = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    f = g.t("feats").clone().requires_grad_(True)
    logits, _ = O.decoder_forward_tf(leaf, f, g.t("caps"), g.t("tf/lengths").tolist(), m["T"], pretrain=True)
    torch.testing.assert_close(logits, g.t("tf/logits"), rtol=1e-4, atol=1e-6)
    (logits * g.t("tf/d_logits")).sum().backward()
    grad_close(f.grad, g.t("tf/d_feats"), "tf d_feats")
    for k, w in g.group("tf/grad/").items():
        grad_close(leaf[k].grad, w, "tf " + k)
    # discriminator with dropout p = 0.5
    F = sum(m["nf"])
    mask = torch.from_numpy(np.unpackbits(g.z["dr/mask"], axis=-1)[..., :F].astype(np.float32))
    dleaf = {k: v.clone().requires_grad_(True) for k, v in dp.items()}
    inp = g.t("dr/inp").clone().requires_grad_(True)
    logit = O.disc_forward(dleaf, inp, mask, m["R"], dropout_p=m["dropout"])
    torch.testing.assert_close(logit, g.t("dr/logits"), rtol=1e-4, atol=1e-6)
    (logit * g.t("dr/d_logits")).sum().backward()
    grad_close(inp.grad, g.t("dr/d_inp"), "dr d_inp")
    for k, w in g.group("dr/grad/").items():
        grad_close(dleaf[k].grad, w, "dr " + k)
