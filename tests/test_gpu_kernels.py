"""GPU parity tests of the HIP library, stage by stage, through the C ABI (ctypes binding in
gan_image_captioning_amd.engine).  Reference = oracle/cpu_step.py (pinned to the reference's own
classes by tests/test_oracle_golden.py) and plain fp64 torch for the generic GEMM.

Tolerances: fp32 mode -> outputs rtol 1e-4, grads rtol 2e-3 with an absolute floor of 1e-4 x the
tensor's largest entry (reduction order), token ids exact.  bf16 mode -> relative L2 error <= 2e-2.
"""
import pytest
import torch

from oracle import cpu_step as O
from tests.golden_io import Golden, initial_params
from tests.gpu_util import close, close_mostly, dec_params, disc_params, rel_l2, dec_param_names, disc_param_names

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def E():
    from gan_image_captioning_amd import engine
    return engine


# ------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [
    (64, 2048, 1024), (64, 10000, 512), (1280, 64, 10000), (4096, 900, 904), (128, 128, 64), (256, 384, 128),
    (130, 70, 50), (5, 3, 7), (100, 900, 4096), (33, 257, 24),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_gemm(E, dev, dtype, layout, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    a_kc = layout in ("NT", "NN")
    b_kc = layout == "NT"
    A = torch.randn((M, K) if a_kc else (K, M), generator=g).to(td)
    B = torch.randn((N, K) if b_kc else (K, N), generator=g).to(td)
    bias = torch.randn(N, generator=g)
    Am = A.double() if a_kc else A.double().t()
    Bm = B.double() if b_kc else B.double().t()
    want = 0.5 * (Am @ Bm.t()) + bias.double()
    for out_td in ([torch.float32] if dtype == "f32" else [torch.float32, torch.bfloat16]):
        C0 = torch.randn(M, N, generator=g).to(out_td)
        Cd = C0.to(dev).clone()
        E.gemm(A.to(dev), B.to(dev), Cd, M, N, K, A.shape[1], B.shape[1], N, a_kc=a_kc, b_kc=b_kc,
               bias=bias.to(dev), accumulate=True, alpha=0.5)
        torch.cuda.synchronize()
        ref = want + C0.double()
        tol = 2e-5 if out_td == torch.float32 else 8e-3
        err = float((Cd.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))
        assert err < tol * max(1.0, (K / 64) ** 0.5), f"{dtype}->{out_td} {layout} {shape}: rel max err {err}"


def test_gemm_strided_views(E, dev):
    """Operands that are column slices of wider buffers (the XH layout of the decoder)."""
    g = torch.Generator().manual_seed(5)
    M, Din, H, N = 64, 32, 64, 256
    XH = torch.randn(M, Din + H, generator=g)
    W = torch.randn(N, H, generator=g)
    Cd = torch.empty(M, N, device=dev)
    XHd = XH.to(dev)
    E.gemm(XHd[:, Din:], W.to(dev), Cd, M, N, H, Din + H, H, N)
    torch.cuda.synchronize()
    close(Cd, XH[:, Din:].double() @ W.double().t(), rtol=1e-4, what="strided A")


# ------------------------------------------------------------------------------------------ decoder
def _decoder(E, gp, dt):
    nl = O.num_lstm_layers(gp)
    V, Em = gp["decoder.embed.weight"].shape
    H = gp["decoder.lstm.weight_hh_l0"].shape[1]
    return E.DecoderEngine(V, Em, H, nl, dt)


@pytest.mark.parametrize("name", ["tiny", "tiny_scaled", "tiny_rep2", "cfg1"])
def test_decoder_sample_fwd_f32_matches_golden(E, dev, name):
    """fp32 parity mode against the reference's own outputs: ids exact, probabilities rtol 1e-4."""
    g = Golden(name)
    gp, _ = initial_params(g)
    m = g.meta
    eng = _decoder(E, gp, 0)
    params = dec_params(gp, dev)
    feats = O.start_features(gp, m["B"]).to(dev)
    u = g.t("s0/u").to(dev)
    out, ids, _ = eng.sample_fwd(params, feats, m["L"], m["temperatures"][0], noise_u=u)
    torch.cuda.synchronize()
    assert torch.equal(ids.cpu(), g.t("s0/ids")), "sampled token ids differ from the reference"
    close(out, g.t("s0/probs"), rtol=1e-4, atol_scale=1e-6, what="probs")


@pytest.mark.parametrize("name,temperature", [("tiny", 1.0031), ("tiny_scaled", 1.0031), ("tiny", 100.0), ("cfg1", 1.7)])
def test_decoder_sample_bwd_f32(E, dev, name, temperature):
    g = Golden(name)
    gp, _ = initial_params(g)
    m = g.meta
    B, Lc, V = m["B"], m["L"], m["V"]
    eng = _decoder(E, gp, 0)
    names = dec_param_names(m["NL"])
    rg = torch.Generator().manual_seed(99)
    feats = torch.randn(B, m["E"], generator=rg) * 0.1
    d_out = torch.randn(B, Lc, V, generator=rg)
    us = g.us(0)
    # oracle
    leaf = {k: gp[k].clone().requires_grad_(True) for k in names}
    f_leaf = feats.clone().requires_grad_(True)
    probs, ids_ref = O.decoder_sample(leaf, f_leaf, Lc, temperature, us)
    (probs * d_out).sum().backward()
    # HIP
    params = dec_params(gp, dev)
    out, ids, st = eng.sample_fwd(params, feats.to(dev), Lc, temperature, noise_u=torch.stack(us).to(dev))
    grads = eng.sample_bwd(params, st, out, ids, d_out.to(dev), temperature)
    torch.cuda.synchronize()
    assert torch.equal(ids.cpu(), ids_ref)
    close(out, probs, rtol=1e-4, atol_scale=1e-6, what="probs")
    for n, gt in zip(names, grads[:-1]):
        close(gt, leaf[n].grad, rtol=2e-3, atol_scale=1e-4, what=n)
    close(grads[-1], f_leaf.grad, rtol=2e-3, atol_scale=1e-4, what="d_features")


def test_decoder_pretrain_mode_f32(E, dev):
    g = Golden("pretrain_tiny")
    gp = g.group("gp0/")
    m = g.meta
    eng = _decoder(E, gp, 0)
    params = dec_params(gp, dev)
    feats = O.start_features(gp, m["B"]).to(dev)
    out, ids, st = eng.sample_fwd(params, feats, m["L"], 1.0, pretrain=True)
    torch.cuda.synchronize()
    assert torch.equal(ids.cpu(), g.t("s0/ids"))
    close(out, g.t("s0/logits"), rtol=1e-4, atol_scale=1e-6, what="logits")
    # loss + backward through the raw logits
    caps = g.t("caps").to(dev)
    loss, dlog = E.xent(out.view(-1, m["V"]), caps.view(-1))
    grads = eng.sample_bwd(params, st, out, ids, dlog.view_as(out), 1.0, pretrain=True)
    torch.cuda.synchronize()
    assert float(loss) == pytest.approx(float(g.t("s0/loss")), rel=1e-5)
    want = g.group("s0/grad/")
    grads[0][1] += grads[-1].sum(0)      # features = embed(<S>=1) (training.py:68): d_features folds into row 1
    for n, gt in zip(dec_param_names(m["NL"]), grads[:-1]):
        close(gt, want[n], rtol=2e-3, atol_scale=1e-4, what=n)


@pytest.mark.parametrize("dt", [0, 1])
def test_decoder_teacher_forced_forward_matches_reference(E, dev, dt):
    """Decoder.forward (generator.py:39-53) against the reference's own run (forward_tf_tiny.npz): packed variable lengths,
    frozen states, zero outputs past each length, one Gumbel draw over the whole tensor."""
    g = Golden("forward_tf_tiny")
    m = g.meta
    gp = g.group("gp0/")
    eng = _decoder(E, gp, dt)
    params = dec_params(gp, dev)
    lengths = [int(v) for v in g.t("lengths")]
    feats, caps = g.t("feats").to(dev), g.t("caps").to(dev)
    logits, (h_n, c_n) = eng.forward_tf(params, feats, caps, lengths, m["T"], pretrain=True)
    probs, (h2, c2) = eng.forward_tf(params, feats, caps, lengths, m["T"], pretrain=False, noise_u=g.t("u").to(dev))
    torch.cuda.synchronize()
    assert tuple(logits.shape) == (m["B"], max(lengths), m["V"])
    if dt == 0:
        close(logits, g.t("logits"), rtol=1e-4, atol_scale=1e-5, what="teacher-forced logits")
        close(probs, g.t("probs"), rtol=1e-4, atol_scale=1e-5, what="teacher-forced probs")
        close(h_n, g.t("h_n"), rtol=1e-4, atol_scale=1e-5, what="h_n")
        close(c_n, g.t("c_n"), rtol=1e-4, atol_scale=1e-5, what="c_n")
        assert torch.equal(h_n, h2) and torch.equal(c_n, c2)
    else:
        assert rel_l2(logits.float(), g.t("logits")) < 2e-2 and rel_l2(probs.float(), g.t("probs")) < 3e-2
        assert rel_l2(h_n, g.t("h_n")) < 2e-2 and rel_l2(c_n, g.t("c_n")) < 2e-2
    # the module surface: Decoder.forward through the host class
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.generator import Decoder
    args = default_args(vocab_size=m["V"], gen_embed_dim=m["E"], gen_hidden_dim=m["H"], gen_num_layers=m["NL"], temperature=1,
                        compute_dtype="fp32" if dt == 0 else "bf16", device="cuda")
    dec = Decoder(args).to(dev)
    with torch.no_grad():
        for n, p in zip(dec_param_names(m["NL"]), dec.param_list()):
            p.copy_(gp[n])
    dec.temperature = m["T"]
    pred, (hn3, cn3) = dec(feats, caps, torch.tensor(lengths), pretrain=True)
    torch.cuda.synchronize()
    assert torch.equal(pred, logits) and torch.equal(hn3, h_n)


def test_decoder_philox_noise_statistics(E, dev):
    """On-device Philox path: valid probabilities, ids in range, different seeds -> different samples."""
    g = Golden("cfg1")
    gp, _ = initial_params(g)
    m = g.meta
    eng = _decoder(E, gp, 0)
    params = dec_params(gp, dev)
    feats = O.start_features(gp, m["B"]).to(dev)
    out1, ids1, _ = eng.sample_fwd(params, feats, m["L"], 1.0, seed=1)
    out2, ids2, _ = eng.sample_fwd(params, feats, m["L"], 1.0, seed=2)
    out1b, ids1b, _ = eng.sample_fwd(params, feats, m["L"], 1.0, seed=1)
    torch.cuda.synchronize()
    assert torch.allclose(out1.sum(-1), torch.ones_like(out1.sum(-1)), atol=1e-4)
    assert int(ids1.min()) >= 0 and int(ids1.max()) < m["V"]
    assert torch.equal(ids1, ids1b) and torch.equal(out1, out1b), "same seed must reproduce"
    assert not torch.equal(ids1, ids2)
    # at init logits << gumbel scale, so ids ~ uniform over V: a crude uniformity check
    counts = torch.bincount(torch.cat([ids1.flatten(), ids2.flatten()]).cpu(), minlength=m["V"])
    assert counts.max() <= 12


# ------------------------------------------------------------------------------------------ discriminator
def _disc(E, m, dt):
    return E.DiscEngine(m["V"], m["De"], m["R"], m["fs"], m["nf"], dt)


@pytest.mark.parametrize("name", ["tiny", "tiny_rep2", "cfg1"])
def test_disc_fwd_f32_matches_golden(E, dev, name):
    g = Golden(name)
    _, dp = initial_params(g)
    m = g.meta
    eng = _disc(E, m, 0)
    params = disc_params(dp, dev)
    masks = g.masks(0)
    probs = g.t("s0/probs").to(dev)
    caps = g.t("caps").to(dev)
    lr, st_r = eng.fwd(params, None, caps, True, masks[0].to(dev))
    lf, st_f = eng.fwd(params, eng.soft_input(probs), None, True, masks[1].to(dev))
    onehot = torch.nn.functional.one_hot(caps, m["V"]).float()
    lr2, _ = eng.fwd(params, eng.soft_input(onehot), None, True, masks[0].to(dev))
    torch.cuda.synchronize()
    close(lr, g.t("s0/d_real"), rtol=1e-4, atol_scale=1e-5, what="d_real (ids path)")
    close(lr2, g.t("s0/d_real"), rtol=1e-4, atol_scale=1e-5, what="d_real (dense one-hot path)")
    close(lf, g.t("s0/d_fake"), rtol=1e-4, atol_scale=1e-5, what="d_fake")
    if g.has("s0/stage/fake/emb"):
        F = eng.F
        close(st_f["emb"].view(m["B"], m["L"], -1), g.t("s0/stage/fake/emb"), rtol=1e-4, what="emb")
        close(st_f["pooled"][:, :F], g.t("s0/stage/fake/pooled"), rtol=1e-4, what="pooled")
        close(st_f["feat"], g.t("s0/stage/fake/feat"), rtol=1e-4, what="feat")
        assert float(st_f["pooled"][:, F:].abs().max()) == 0.0 if eng.Fp > F else True


def test_disc_eval_mode_f32(E, dev):
    g = Golden("tiny_eval")
    _, dp = initial_params(g)
    m = g.meta
    eng = _disc(E, m, 0)
    params = disc_params(dp, dev)
    lf, _ = eng.fwd(params, eng.soft_input(g.t("s0/probs").to(dev)), None, False)
    torch.cuda.synchronize()
    close(lf, g.t("s0/d_fake"), rtol=1e-4, atol_scale=1e-5, what="eval d_fake")


@pytest.mark.parametrize("name", ["tiny", "tiny_rep2", "cfg1"])
def test_disc_bwd_f32(E, dev, name):
    g = Golden(name)
    _, dp = initial_params(g)
    m = g.meta
    eng = _disc(E, m, 0)
    names = disc_param_names(len(m["nf"]))
    params = disc_params(dp, dev)
    masks = g.masks(0)
    rg = torch.Generator().manual_seed(3)
    B, Lc, V = m["B"], m["L"], m["V"]
    soft = torch.softmax(torch.randn(B, Lc, V, generator=rg) * 2, -1)
    caps = g.t("caps")
    dl_r = torch.randn(B * m["R"], generator=rg)
    dl_f = torch.randn(B * m["R"], generator=rg)
    # oracle: loss = <D(real), dl_r> + <D(soft), dl_f>
    leaf = {k: dp[k].clone().requires_grad_(True) for k in names}
    s_leaf = soft.clone().requires_grad_(True)
    real = torch.nn.functional.one_hot(caps, V).float()
    o_r = O.disc_forward(leaf, real, masks[0], m["R"])
    o_f = O.disc_forward(leaf, s_leaf, masks[1], m["R"])
    ((o_r * dl_r).sum() + (o_f * dl_f).sum()).backward()
    # HIP: ids path (overwrite) then soft path (accumulate), input grad on the soft path
    sd = eng.soft_input(soft.to(dev))
    lr, st_r = eng.fwd(params, None, caps.to(dev), True, masks[0].to(dev))
    lf, st_f = eng.fwd(params, sd, None, True, masks[1].to(dev))
    grads, _ = eng.bwd(params, st_r, None, caps.to(dev), True, dl_r.to(dev), True, False)
    grads, d_inp = eng.bwd(params, st_f, sd, None, True, dl_f.to(dev), True, True, grads=grads, accumulate=True)
    torch.cuda.synchronize()
    close(lr, o_r, rtol=1e-4, atol_scale=1e-5, what="logits real")
    close(lf, o_f, rtol=1e-4, atol_scale=1e-5, what="logits soft")
    # The max-over-time index must equal the oracle's except at fp32 near-ties: with one-hot input two
    # windows of one caption can agree to ~1e-7 relative (measured: ~2 of 460k entries at cfg1 shapes),
    # and a differently-rounded sum then picks the other one.  Such a flip re-routes ONE contribution.
    flips = 0
    for st, inp in ((st_r, real), (st_f, soft)):
        emb = (inp @ dp["embeddings.weight"].t()).reshape(B, Lc, m["R"], -1)
        off = 0
        for k, f in enumerate(m["fs"]):
            w, bias = dp[f"convs.{k}.weight"], dp[f"convs.{k}.bias"]
            con = torch.relu(torch.einsum("btrej,cje->bctr", emb.unfold(1, f, 1), w[:, 0]) + bias[None, :, None, None])
            con = con.permute(0, 3, 1, 2).reshape(B * m["R"], w.shape[0], -1)              # [B*R, n, T]
            mine = st["argmax"][:, off:off + w.shape[0]].cpu().long()
            best = con.max(dim=2)[0]
            at_mine = con.gather(2, mine.unsqueeze(2)).squeeze(2)
            assert bool(((best - at_mine) <= 1e-6 * best.abs() + 1e-12).all()), f"conv {k}: argmax is not a (near-)maximum"
            flips += int(((con.max(dim=2)[1] != mine) & (best > 0)).sum())
            off += w.shape[0]
    print("max-over-time near-tie flips vs oracle:", flips)
    assert flips <= 8
    # grads upstream of the pooling tolerate those re-routed contributions; everything downstream is element-wise tight
    upstream = names[:1 + 2 * len(m["nf"])]
    report = {n: close_mostly(grads[names.index(n)], leaf[n].grad, 2e-3, 1e-4, n, 1e-2, 1.5e-2 if flips else 1e-5) for n in upstream}
    report["d_inp"] = close_mostly(d_inp, s_leaf.grad, 2e-3, 1e-4, "d_inp", 1e-2, 1.5e-2 if flips else 1e-5)
    print({k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})
    for n in names[1 + 2 * len(m["nf"]):]:
        close(grads[names.index(n)], leaf[n].grad, rtol=2e-3, atol_scale=1e-4, what=n)
    # input-grad only (generator path): no parameter grads requested
    _, d_inp2 = eng.bwd(params, st_f, sd, None, True, dl_f.to(dev), False, True)
    torch.cuda.synchronize()
    assert torch.equal(d_inp2, d_inp)


def test_disc_philox_dropout(E, dev):
    g = Golden("cfg1")
    _, dp = initial_params(g)
    m = g.meta
    eng = _disc(E, m, 0)
    params = disc_params(dp, dev)
    caps = g.t("caps").to(dev)
    l1, st1 = eng.fwd(params, None, caps, True, None, seed=11)
    l2, st2 = eng.fwd(params, None, caps, True, None, seed=12)
    l1b, _ = eng.fwd(params, None, caps, True, None, seed=11)
    torch.cuda.synchronize()
    keep = st1["keep"][:, :eng.F].float()
    assert abs(float(keep.mean()) - 0.8) < 0.01          # nn.Dropout(0.2)
    assert torch.equal(l1, l1b) and not torch.equal(l1, l2)
    # the mask that was written is the mask that was applied
    want = O.disc_forward(dp, torch.nn.functional.one_hot(g.t("caps"), m["V"]).float(), keep.cpu(), m["R"])
    close(l1, want, rtol=1e-4, atol_scale=1e-5, what="philox-dropout logits")


@pytest.mark.parametrize("dt", [0, 1])
def test_disc_second_forward_shares_the_first(E, dev, dt):
    """training.py:163-164 evaluate D twice on gen_captions; gic_disc_fwd_redrop reuses the first pass up to the highway
    pre-activation.  Same mask / same seed => the very logits of a full second forward; and its backward (on the aliased
    state) equals the backward of that full forward."""
    g = Golden("cfg1")
    _, dp = initial_params(g)
    m = g.meta
    eng = _disc(E, m, dt)
    params = disc_params(dp, dev)
    probs = eng.soft_input(g.t("s0/probs").to(dev))
    masks = g.masks(0)
    B, Lc = probs.shape[0], probs.shape[1]
    l_fake, st_fake = eng.fwd(params, probs, None, True, masks[1].to(dev))
    st_gen = eng.shared_state(st_fake, B, Lc, dev)
    F = eng.F
    # f32 mode is bit-reproducible; in bf16 mode the full second forward re-runs split-K products (atomic f32 sums)
    same = torch.equal if dt == 0 else (lambda x, y: bool(torch.allclose(x.float(), y.float(), rtol=2e-2, atol=1e-4)))
    l_gen, _ = eng.fwd_redrop(params, st_fake, st_gen, True, masks[2].to(dev))
    l_full, st_full = eng.fwd(params, probs, None, True, masks[2].to(dev))
    assert same(l_gen, l_full) and torch.equal(st_gen["keep"][:, :F], st_full["keep"][:, :F]) and same(st_gen["ydrop"], st_full["ydrop"])
    # device-drawn dropout: same (seed, row, column) -> same draw on both routes
    l_gen_p, _ = eng.fwd_redrop(params, st_fake, st_gen, True, None, seed=77)
    l_full_p, st_full_p = eng.fwd(params, probs, None, True, None, seed=77)
    assert same(l_gen_p, l_full_p) and torch.equal(st_gen["keep"][:, :F], st_full_p["keep"][:, :F])
    # eval mode: no dropout, both routes give the plain forward
    l_gen_e, _ = eng.fwd_redrop(params, st_fake, st_gen, False)
    l_full_e, _ = eng.fwd(params, probs, None, False)
    assert same(l_gen_e, l_full_e)
    # backward through the aliased state
    l_gen, _ = eng.fwd_redrop(params, st_fake, st_gen, True, masks[2].to(dev))
    d_logits = torch.linspace(-1, 1, l_gen.numel(), device=dev)
    _, d_a = eng.bwd(params, st_gen, probs, None, True, d_logits, False, True)
    _, d_b = eng.bwd(params, st_full, probs, None, True, d_logits, False, True)
    torch.cuda.synchronize()
    if dt == 0:
        assert torch.equal(d_a, d_b)
    else:
        assert rel_l2(d_a.float(), d_b.float()) < 2e-2
    assert float(st_gen["ydrop"][:, F:].abs().max() if eng.Fp > F else 0.0) == 0.0


# ------------------------------------------------------------------------------------------ losses / optimizer
def test_gan_losses_all_types(E, dev):
    g = Golden("scalars")
    d_r, d_f, g_o = g.t("d_real"), g.t("d_fake"), g.t("g_out")
    for lt in ("standard", "JS", "KL", "rsgan", "hinge", "tv"):
        losses, grads = E.gan_losses(lt, d_r.to(dev), d_f.to(dev), g_o.to(dev))
        torch.cuda.synchronize()
        leaf = [t.clone().requires_grad_(True) for t in (d_r, d_f, g_o)]
        gl, dl = O.get_losses(*leaf, lt)
        if g.has(f"loss/{lt}"):                  # the reference's own numbers
            assert float(losses[0]) == pytest.approx(float(g.t(f"loss/{lt}")[0]), rel=1e-5)
            assert float(losses[1]) == pytest.approx(float(g.t(f"loss/{lt}")[1]), rel=1e-5)
        assert float(losses[0]) == pytest.approx(float(gl), rel=1e-5)
        assert float(losses[1]) == pytest.approx(float(dl), rel=1e-5)
        dd = torch.autograd.grad(dl, leaf[:2], allow_unused=True)
        close(grads["dd_real"], dd[0], rtol=1e-4, what=f"{lt} dd_real")
        close(grads["dd_fake"], dd[1], rtol=1e-4, what=f"{lt} dd_fake")
        dg = torch.autograd.grad(gl, leaf, allow_unused=True)
        zero = torch.zeros_like(d_r)
        close(grads["dg_out"], dg[2] if dg[2] is not None else zero, rtol=1e-4, what=f"{lt} dg_out")
        close(grads["dg_real"], dg[0] if dg[0] is not None else zero, rtol=1e-4, what=f"{lt} dg_real")
        close(grads["dg_fake"], dg[1] if dg[1] is not None else zero, rtol=1e-4, what=f"{lt} dg_fake")
    with pytest.raises(NotImplementedError):
        E.gan_losses("nope", d_r.to(dev), d_f.to(dev), g_o.to(dev))


def test_clip_adam_matches_reference_trajectory(E, dev):
    """clip_grad_norm_ + Adam on the GOLDEN grads of the tiny case, 3 steps, vs the reference's post-step weights."""
    g = Golden("tiny")
    m = g.meta
    for names, lr in ((disc_param_names(len(m["nf"])), m["disc_lr"]), (dec_param_names(m["NL"]), m["gen_lr"])):
        src = g.group("gp0/") if names[0].startswith("decoder") else g.group("dp0/")
        flat = torch.cat([src[n].reshape(-1) for n in names]).to(dev)
        n = flat.numel()
        mom, var = torch.zeros_like(flat), torch.zeros_like(flat)
        step = torch.zeros(1, dtype=torch.int64, device=dev)
        norm = torch.zeros(1, device=dev)
        parts = torch.zeros(E.clip_adam_partials(n), device=dev)
        for s in range(m["steps"]):
            gr = g.group(f"s{s}/grad/")
            gflat = torch.cat([gr[k].reshape(-1) for k in names]).to(dev)
            E.clip_adam(flat, gflat, mom, var, lr, 0.9, 0.999, 1e-8, m["clip"], step, norm, parts)
            torch.cuda.synchronize()
            key = "d_norm" if names[0] == "embeddings.weight" else "g_norm"
            assert float(norm) == pytest.approx(float(g.t(f"s{s}/{key}")), rel=1e-4, abs=1e-12)
            want = torch.cat([g.t(f"s{s}/post/{k}").reshape(-1) for k in names])
            torch.testing.assert_close(flat.cpu(), want, rtol=1e-6, atol=2e-9)
        assert int(step) == m["steps"]
        torch.testing.assert_close(mom.cpu(), torch.cat([g.t("adam/m/" + k).reshape(-1) for k in names]), rtol=1e-5, atol=1e-12)
        torch.testing.assert_close(var.cpu(), torch.cat([g.t("adam/v/" + k).reshape(-1) for k in names]), rtol=1e-5, atol=1e-16)


def test_clip_adam_clips(E, dev):
    rg = torch.Generator().manual_seed(1)
    n = 100003
    p = torch.randn(n, generator=rg)
    gr = torch.randn(n, generator=rg) * 3
    params = {"p": p.clone()}
    gcl, norm = O.clip_grad_norm({"p": gr}, 5.0)
    st = O.AdamState(1e-3)
    st.step(params, gcl)
    flat, mom, var = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    nrm = torch.zeros(1, device=dev)
    E.clip_adam(flat, gr.to(dev), mom, var, 1e-3, 0.9, 0.999, 1e-8, 5.0, step, nrm, torch.zeros(E.clip_adam_partials(n), device=dev))
    torch.cuda.synchronize()
    assert float(nrm) == pytest.approx(norm, rel=1e-5)
    torch.testing.assert_close(flat.cpu(), params["p"], rtol=1e-5, atol=1e-7)


def test_embedding_fwd_bwd(E, dev):
    rg = torch.Generator().manual_seed(2)
    w = torch.randn(50, 8, generator=rg)
    ids = torch.randint(0, 50, (4, 7), generator=rg)
    out = E.embedding_fwd(w.to(dev), ids.to(dev))
    d_out = torch.randn(4, 7, 8, generator=rg)
    dw = E.embedding_bwd(d_out.to(dev), ids.to(dev), 50)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), w[ids])
    want = torch.zeros(50, 8).index_add_(0, ids.reshape(-1), d_out.reshape(-1, 8))
    close(dw, want, rtol=1e-5, what="embedding grad")


# ------------------------------------------------------------------------------------------ bf16 compute mode
def test_bf16_decoder_and_disc_close_to_f32_oracle(E, dev):
    """bf16 perf mode on cfg1 shapes: relative L2 error of outputs/grads vs the fp32 oracle <= 2e-2
    (ids are reported, not asserted: argmax feedback amplifies rounding by design)."""
    g = Golden("cfg1")
    gp, dp = initial_params(g)
    m = g.meta
    B, Lc, V = m["B"], m["L"], m["V"]
    T = 1.5
    us = g.us(0)
    masks = g.masks(0)
    rg = torch.Generator().manual_seed(4)
    d_out = torch.randn(B, Lc, V, generator=rg)
    gnames, dnames = dec_param_names(m["NL"]), disc_param_names(len(m["nf"]))
    gleaf = {k: gp[k].clone().requires_grad_(True) for k in gnames}
    feats = O.start_features(gp, B)
    probs, ids_ref = O.decoder_sample(gleaf, feats, Lc, T, us)
    (probs * d_out).sum().backward()
    deng, geng = _disc(E, m, 1), _decoder(E, gp, 1)
    gparams, dparams = dec_params(gp, dev), disc_params(dp, dev)
    out, ids, st = geng.sample_fwd(gparams, feats.to(dev), Lc, T, noise_u=torch.stack(us).to(dev))
    ggr = geng.sample_bwd(gparams, st, out, ids, d_out.to(dev).bfloat16(), T)
    torch.cuda.synchronize()
    match = float((ids.cpu() == ids_ref).float().mean())
    print(f"bf16 id match-rate vs fp32 oracle: {match:.3f}")
    assert match > 0.9
    if match < 1.0:      # a bf16 near-tie flipped an argmax: the oracle follows the GPU's trajectory (the index carries no gradient)
        gleaf = {k: gp[k].clone().requires_grad_(True) for k in gnames}
        probs, _ = O.decoder_sample(gleaf, feats, Lc, T, us, force_ids=ids.cpu())
        (probs * d_out).sum().backward()
    gerrs = {n: rel_l2(gt, gleaf[n].grad) for n, gt in zip(gnames, ggr[:-1])}
    gerrs["probs"] = rel_l2(out.float(), probs)
    print("bf16 decoder rel-L2 errors:", {k: f"{v:.1e}" for k, v in gerrs.items()})
    assert max(gerrs.values()) < 3e-2, gerrs
    # discriminator on the oracle's probabilities
    dleaf = {k: dp[k].clone().requires_grad_(True) for k in dnames}
    p_leaf = probs.detach().clone().requires_grad_(True)
    o = O.disc_forward(dleaf, p_leaf, masks[1], m["R"])
    dl = torch.randn(o.shape, generator=rg)
    (o * dl).sum().backward()
    sd = deng.soft_input(probs.detach().to(dev))
    lg, dst = deng.fwd(dparams, sd, None, True, masks[1].to(dev))
    dgr, d_inp = deng.bwd(dparams, dst, sd, None, True, dl.to(dev), True, True)
    torch.cuda.synchronize()
    errs = {n: rel_l2(gt, dleaf[n].grad) for n, gt in zip(dnames, dgr)}
    errs["logits"] = rel_l2(lg, o)
    errs["d_inp"] = rel_l2(d_inp.float(), p_leaf.grad)
    print("bf16 discriminator rel-L2 errors:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert errs["logits"] < 2e-2
    assert max(errs.values()) < 6e-2, errs      # conv/emb grads also carry re-routed max-pool near-ties


def test_bf16_discriminator_at_bench_batch(E, dev):
    """B = 32 rows x 64 representations: the highway product (forward epilogue with dropout, input gradient on the transposed
    weight image, accumulating weight gradients with split-K) runs on the 8-wave kernel, as in the benchmark; checked against
    the fp32 oracle on its own inputs."""
    g = Golden("cfg1")
    _, dp = initial_params(g)
    m = g.meta
    B, Lc, V = 32, 12, m["V"]
    rg = torch.Generator().manual_seed(21)
    probs = torch.softmax(torch.randn(B, Lc, V, generator=rg) * 3, -1)
    mask = (torch.rand(B * m["R"], 900, generator=rg) >= 0.2).to(torch.uint8)
    dnames = disc_param_names(len(m["nf"]))
    dleaf = {k: dp[k].clone().requires_grad_(True) for k in dnames}
    p_leaf = probs.clone().requires_grad_(True)
    o = O.disc_forward(dleaf, p_leaf, mask, m["R"])
    dl = torch.randn(o.shape, generator=rg)
    (o * dl).sum().backward()
    deng = _disc(E, m, 1)
    dparams = disc_params(dp, dev)
    sd = deng.soft_input(probs.to(dev))
    lg, dst = deng.fwd(dparams, sd, None, True, mask.to(dev))
    dgr, d_inp = deng.bwd(dparams, dst, sd, None, True, dl.to(dev), True, True)
    # second pass accumulates on top of the first (the step's real + fake passes): gradients double
    dgr2, _ = deng.bwd(dparams, dst, sd, None, True, dl.to(dev), True, False, grads=[t.clone() for t in dgr], accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(dst["keep"][:, :900].cpu(), mask)
    errs = {n: rel_l2(gt, dleaf[n].grad) for n, gt in zip(dnames, dgr)}
    errs["logits"] = rel_l2(lg, o)
    errs["d_inp"] = rel_l2(d_inp.float(), p_leaf.grad)
    print("bf16 discriminator (B=32) rel-L2 errors:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert errs["logits"] < 2e-2, errs
    for n, e in errs.items():        # conv / embedding / input gradients also carry re-routed max-pool near-ties (bf16 pooled values)
        assert e < (8e-2 if n.startswith(("convs", "embeddings", "d_inp")) else 3e-2), (n, errs)
    for n, g1, g2 in zip(dnames, dgr, dgr2):
        assert rel_l2(g2, 2 * g1) < 2e-2, n
    # device-drawn dropout on the wide kernel: reproducible per seed, rate 0.2, and the same (seed, row, column) -> draw map as the
    # element-wise re-draw kernel that serves D(gen) in the step
    l1, st1 = deng.fwd(dparams, sd, None, True, None, seed=123)
    l2, st2 = deng.fwd(dparams, sd, None, True, None, seed=123)
    st3 = deng.shared_state(st1, B, Lc, dev)
    deng.fwd_redrop(dparams, st1, st3, True, None, seed=123)
    torch.cuda.synchronize()
    k1 = st1["keep"][:, :900]
    assert torch.equal(k1, st2["keep"][:, :900]) and torch.equal(k1, st3["keep"][:, :900])
    assert abs(float(k1.float().mean()) - 0.8) < 0.01
    assert float(st1["ydrop"][:, 900:].abs().max()) == 0.0            # pad columns of the dropped output stay zero
    assert torch.equal(st1["ydrop"], st3["ydrop"])


def test_forward_only_discriminator_convolution_in_bf16_matches_the_fp32_kernels(E, dev):
    """The reward-evaluation form of Discriminator.forward (eval mode, nothing kept for a backward pass) runs its convolution + ReLU +
    max-over-time with bf16 products on the 32x32x16 MFMA (disc.hip); the same forward WITH a saved state runs the fp32-product kernel
    (discriminator.py:40-47 as the reference computes it).  Same parameters and ids, 70 captions x 64 representations (a partial
    32-pair workgroup at the end), odd and even window counts: pooled features agree to bf16 rounding of the embedding and the
    filter weights, logits to the bf16 budget of the rest of the reward path, and both stay close to the fp32 oracle."""
    g = Golden("cfg1")
    _, dp = initial_params(g)
    m = g.meta
    B, Lc, V = 70, 13, m["V"]
    rg = torch.Generator().manual_seed(33)
    ids = torch.randint(0, V, (B, Lc), generator=rg)
    deng = _disc(E, m, 1)
    dparams = disc_params(dp, dev)
    lg_keep, st_keep = deng.fwd(dparams, None, ids.to(dev), False)
    lg_fwd, st_fwd = deng.fwd(dparams, None, ids.to(dev), False, forward_only=True)
    torch.cuda.synchronize()
    F = sum(m["nf"])
    assert st_fwd["argmax"] is None and st_keep["argmax"] is not None
    pk, pf = st_keep["pooled"][:, :F].float(), st_fwd["pooled"][:, :F].float()
    assert float(st_fwd["pooled"][:, F:].abs().max() if deng.Fp > F else 0.0) == 0.0
    assert rel_l2(pf, pk) < 1e-2, rel_l2(pf, pk)
    assert float((pf - pk).abs().max()) <= 4e-2 * float(pk.abs().max())
    want = O.disc_forward(dp, torch.nn.functional.one_hot(ids, V).float(), None, m["R"])
    assert rel_l2(lg_keep, want) < 2e-2 and rel_l2(lg_fwd, want) < 3e-2, (rel_l2(lg_keep, want), rel_l2(lg_fwd, want))


def test_no_cpu_fallback(E):
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    from gan_image_captioning_amd._lib import GicError
    with pytest.raises(GicError):
        E.gemm(torch.zeros(4, 4), torch.zeros(4, 4), torch.zeros(4, 4), 4, 4, 4, 4, 4, 4)


# ------------------------------------------------------------------------------------------ fused roll-out step kernels
FUSED_SHAPES = [  # (B, L, V, E, H, NL)
    (5, 6, 52, 8, 24, 2), (70, 4, 132, 16, 40, 1), (8, 10, 64, 32, 512, 1), (64, 3, 10000, 64, 128, 1), (3, 5, 4, 8, 8, 3),
]


def _rand_decoder(V, Em, H, NL, seed, scale=6.0):
    g = torch.Generator().manual_seed(seed)
    gp = O.make_gen_params(V, Em, H, NL, g)
    return {k: v * scale for k, v in gp.items()}, g


@pytest.mark.parametrize("shape", FUSED_SHAPES)
def test_fused_rollout_f32_matches_oracle(E, dev, shape):
    """lstm_step / vocab_step / sample_finish (decoder_step.hip) in fp32 parity mode against the oracle (pinned to the reference's
    Decoder.sample by the goldens): ids exact, probabilities rtol 1e-4, and -- through the backward pass that consumes the saved
    state (x rows, hidden states, gates, cell states) -- every gradient.  Initial states (generator.py:55,61) on the first two."""
    from gan_image_captioning_amd import _lib
    B, Lc, V, Em, H, NL = shape
    gp, g = _rand_decoder(V, Em, H, NL, sum(shape))
    names = dec_param_names(NL)
    feats = torch.randn(B, Em, generator=g) * 0.3
    us = [torch.empty(B, V).uniform_(0, 1, generator=g) for _ in range(Lc)]
    d_out = torch.randn(B, Lc, V, generator=g)
    states = None
    if shape in FUSED_SHAPES[:2]:
        states = (torch.randn(NL, B, H, generator=g) * 0.5, torch.randn(NL, B, H, generator=g) * 0.5)
    T = 1.3
    leaf = {k: gp[k].clone().requires_grad_(True) for k in names}
    f_leaf = feats.clone().requires_grad_(True)
    st_leaf = None if states is None else tuple(t.clone().requires_grad_(True) for t in states)
    probs, ids_ref = O.decoder_sample(leaf, f_leaf, Lc, T, us, states=st_leaf)
    (probs * d_out).sum().backward()
    eng = E.DecoderEngine(V, Em, H, NL, 0)
    assert bool(_lib.load())            # the library is loaded: no other implementation exists
    params = dec_params(gp, dev)
    dstates = None if states is None else tuple(t.to(dev) for t in states)
    out, ids, st = eng.sample_fwd(params, feats.to(dev), Lc, T, noise_u=torch.stack(us).to(dev), states=dstates)
    assert st["part"] is not None
    ws = eng.alloc_bwd_ws(B, Lc, dev)
    grads = eng.sample_bwd(params, st, out, ids, d_out.to(dev), T, ws=ws, phases=7)
    torch.cuda.synchronize()
    assert torch.equal(ids.cpu(), ids_ref), "token ids differ from the oracle"
    close(out, probs, rtol=1e-4, atol_scale=1e-6, what="probs")
    for n, gt in zip(names, grads[:-1]):
        close(gt, leaf[n].grad, rtol=2e-3, atol_scale=1e-4, what=n)
    close(grads[-1], f_leaf.grad, rtol=2e-3, atol_scale=1e-4, what="d_features")
    if states is not None:
        d_h0, d_c0 = eng.state_grads(ws)
        close(d_h0, st_leaf[0].grad, rtol=2e-3, atol_scale=1e-4, what="d_h0")
        close(d_c0, st_leaf[1].grad, rtol=2e-3, atol_scale=1e-4, what="d_c0")
    # the same call through the unfused launches (no partials scratch): identical ids, probabilities to rounding
    st2 = eng.alloc_state(B, Lc, dev)
    st2["part"] = None
    out2, ids2, _ = eng.sample_fwd(params, feats.to(dev), Lc, T, noise_u=torch.stack(us).to(dev), states=dstates, state=st2)
    torch.cuda.synchronize()
    assert torch.equal(ids2, ids)
    close(out2, out, rtol=1e-4, atol_scale=1e-6, what="fused vs unfused probs")
    # pretrain mode (generator.py:63-66): raw logits out, greedy feedback
    logits, ids_p = O.decoder_sample(gp, feats, Lc, 1.0, None, pretrain=True, states=states)
    outp, idsp, _ = eng.sample_fwd(params, feats.to(dev), Lc, 1.0, pretrain=True, states=dstates)
    torch.cuda.synchronize()
    assert torch.equal(idsp.cpu(), ids_p)
    close(outp, logits, rtol=1e-4, atol_scale=1e-5, what="pretrain logits")


def test_fused_rollout_forced_prefix_and_ids_only(E, dev):
    """force_ids / force_len: rows follow a given prefix and continue on their own argmax (the Monte-Carlo roll-out primitive);
    ids_only: the same ids with nothing else written."""
    B, Lc, V, Em, H, NL = 9, 7, 48, 8, 32, 2
    gp, g = _rand_decoder(V, Em, H, NL, 77)
    feats = torch.randn(B, Em, generator=g) * 0.3
    us = [torch.empty(B, V).uniform_(0, 1, generator=g) for _ in range(Lc)]
    forced = torch.randint(0, V, (B, Lc), generator=g)
    flen = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 3], dtype=torch.int32)
    T = 2.0
    # oracle: per row, follow the prefix for flen steps, then the argmax -- emulate with a two-pass force (ids of the free pass differ per row)
    want_ids = torch.empty(B, Lc, dtype=torch.long)
    want_probs = torch.empty(B, Lc, V)
    for b in range(B):
        cur = forced[b:b + 1].clone()
        for t in range(int(flen[b]), Lc):          # extend the trajectory one free step at a time
            p, _ = O.decoder_sample(gp, feats[b:b + 1], Lc, T, [u[b:b + 1] for u in us], force_ids=cur)
            cur[0, t] = p[0, t].argmax()
        p, _ = O.decoder_sample(gp, feats[b:b + 1], Lc, T, [u[b:b + 1] for u in us], force_ids=cur)
        want_ids[b], want_probs[b] = cur[0], p[0]
    eng = E.DecoderEngine(V, Em, H, NL, 0)
    params = dec_params(gp, dev)
    u = torch.stack(us).to(dev)
    out, ids, _ = eng.sample_fwd(params, feats.to(dev), Lc, T, noise_u=u, force_ids=forced.to(dev), force_len=flen.to(dev))
    _, ids2, _ = eng.sample_fwd(params, feats.to(dev), Lc, T, noise_u=u, force_ids=forced.to(dev), force_len=flen.to(dev), ids_only=True)
    torch.cuda.synchronize()
    assert torch.equal(ids.cpu(), want_ids)
    assert torch.equal(ids2, ids)
    close(out, want_probs, rtol=1e-4, atol_scale=1e-6, what="probs on the forced trajectory")


def test_fused_rollout_bf16_close_to_oracle(E, dev):
    """bf16 fused step kernels at the benchmark's decoder shapes (B=64, V=10000, E=H=512), 6 steps: ids match-rate >= 0.95 against
    the fp32 oracle; probabilities compared on the GPU's own trajectory (relative L2 <= 3e-2)."""
    B, Lc, V, Em, H, NL = 64, 6, 10000, 512, 512, 1
    g = torch.Generator().manual_seed(5)
    gp = O.make_gen_params(V, Em, H, NL, g)
    feats = torch.randn(B, Em, generator=g) * 0.3
    us = [torch.empty(B, V).uniform_(0, 1, generator=g) for _ in range(Lc)]
    T = 1.7
    eng = E.DecoderEngine(V, Em, H, NL, 1)
    out, ids, _ = eng.sample_fwd(dec_params(gp, dev), feats.to(dev), Lc, T, noise_u=torch.stack(us).to(dev))
    torch.cuda.synchronize()
    _, ids_ref = O.decoder_sample(gp, feats, Lc, T, us)
    match = float((ids.cpu() == ids_ref).float().mean())
    assert match >= 0.95, match
    probs, _ = O.decoder_sample(gp, feats, Lc, T, us, force_ids=ids.cpu())
    assert rel_l2(out.float(), probs) < 3e-2
    assert float((out.float().sum(-1) - 1).abs().max()) < 2e-2       # rows are normalised


def test_api_corners_match_reference_golden(E, dev):
    """Module-API corners against the reference's own outputs (golden api_tiny.npz): Decoder.sample(states=...) with gradients into
    the states (generator.py:55,61) and Discriminator(dropout=0.5) (discriminator.py:10,30)."""
    import numpy as np
    g = Golden("api_tiny")
    m = g.meta
    P = g.group("p0/")
    gp = {k: v for k, v in P.items() if k.startswith("decoder.")}
    dp = {k: v for k, v in P.items() if not k.startswith("decoder.")}
    eng = _decoder(E, gp, 0)
    params = dec_params(gp, dev)
    B, Lc = m["B"], m["L"]
    out, ids, st = eng.sample_fwd(params, g.t("feats").to(dev), Lc, m["T"], noise_u=g.t("st/u").to(dev),
                                  states=(g.t("h0").to(dev), g.t("c0").to(dev)))
    ws = eng.alloc_bwd_ws(B, Lc, dev)
    grads = eng.sample_bwd(params, st, out, ids, g.t("st/d_out").to(dev), m["T"], ws=ws, phases=7)
    d_h0, d_c0 = eng.state_grads(ws)
    torch.cuda.synchronize()
    assert torch.equal(ids.cpu(), g.t("st/ids"))
    close(out, g.t("st/probs"), rtol=1e-4, atol_scale=1e-6, what="probs")
    want = g.group("st/grad/")
    for n, gt in zip(dec_param_names(m["NL"]), grads[:-1]):
        close(gt, want[n], rtol=2e-3, atol_scale=1e-4, what=n)
    close(grads[-1], g.t("st/d_feats"), rtol=2e-3, atol_scale=1e-4, what="d_features")
    close(d_h0, g.t("st/d_h0"), rtol=2e-3, atol_scale=1e-4, what="d_h0")
    close(d_c0, g.t("st/d_c0"), rtol=2e-3, atol_scale=1e-4, what="d_c0")
    # discriminator with dropout p = 0.5
    F = sum(m["nf"])
    mask = torch.from_numpy(np.unpackbits(g.z["dr/mask"], axis=-1)[..., :F].astype(np.float32))
    den = E.DiscEngine(m["V"], m["De"], m["R"], m["fs"], m["nf"], 0, dropout=m["dropout"])
    dparams = disc_params(dp, dev)
    soft = den.soft_input(g.t("dr/inp").to(dev))
    lg, dst = den.fwd(dparams, soft, None, True, mask.to(dev))
    dgr, d_inp = den.bwd(dparams, dst, soft, None, True, g.t("dr/d_logits").to(dev), True, True)
    torch.cuda.synchronize()
    close(lg, g.t("dr/logits"), rtol=1e-4, atol_scale=1e-5, what="logits (dropout 0.5)")
    close_mostly(d_inp, g.t("dr/d_inp"), 2e-3, 1e-4, "d_inp", 1e-2, 1.5e-2)
    wantd = g.group("dr/grad/")
    up = set(disc_param_names(len(m["nf"]))[:1 + 2 * len(m["nf"])])
    for n, gt in zip(disc_param_names(len(m["nf"])), dgr):
        if n in up:
            close_mostly(gt, wantd[n], 2e-3, 1e-4, n, 1e-2, 1.5e-2)
        else:
            close(gt, wantd[n], rtol=2e-3, atol_scale=1e-4, what=n, atol_abs=1e-6)


def _decoder_module(gp, m, dt, dev, temperature):
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.generator import Decoder
    args = default_args(vocab_size=m["V"], gen_embed_dim=m["E"], gen_hidden_dim=m["H"], gen_num_layers=m["NL"], temperature=temperature,
                        compute_dtype="fp32" if dt == 0 else "bf16", device="cuda")
    dec = Decoder(args).to(dev)
    with torch.no_grad():
        for n, p in zip(dec_param_names(m["NL"]), dec.param_list()):
            p.copy_(gp[n])
    return dec


@pytest.mark.parametrize("dt", [0, 1])
def test_teacher_forced_decode_gradients_match_reference_golden(E, dev, dt):
    """Autograd through Decoder.forward (generator.py:39-53) against the reference's own backward (golden api_tiny.npz, tf/ group:
    ragged lengths, pretrain mode): logits, d features and every parameter gradient, through the module surface."""
    g = Golden("api_tiny")
    m = g.meta
    gp = {k: v for k, v in g.group("p0/").items() if k.startswith("decoder.")}
    dec = _decoder_module(gp, m, dt, dev, m["T"])
    feats = g.t("feats").to(dev).requires_grad_(True)
    lengths = [int(v) for v in g.t("tf/lengths")]
    logits, (h_n, c_n) = dec(feats, g.t("caps").to(dev), lengths, pretrain=True)
    assert logits.requires_grad and not h_n.requires_grad and not c_n.requires_grad
    (logits.float() * g.t("tf/d_logits").to(dev)).sum().backward()
    torch.cuda.synchronize()
    want = g.group("tf/grad/")
    if dt == 0:
        close(logits, g.t("tf/logits"), rtol=1e-4, atol_scale=1e-5, what="teacher-forced logits")
        close(feats.grad, g.t("tf/d_feats"), rtol=2e-3, atol_scale=1e-4, what="d_features")
        for n, p in zip(dec_param_names(m["NL"]), dec.param_list()):
            close(p.grad, want[n], rtol=2e-3, atol_scale=1e-4, what=n)
    else:
        assert rel_l2(logits.float(), g.t("tf/logits")) < 2e-2
        assert rel_l2(feats.grad, g.t("tf/d_feats")) < 6e-2
        for n, p in zip(dec_param_names(m["NL"]), dec.param_list()):
            assert rel_l2(p.grad, want[n]) < 6e-2, (n, rel_l2(p.grad, want[n]))
    # nothing to differentiate: the forward-only path (no saved state), same values
    with torch.no_grad():
        again, _ = dec(feats.detach(), g.t("caps").to(dev), lengths, pretrain=True)
    assert not again.requires_grad and torch.equal(again, logits.detach())


@pytest.mark.parametrize("shape", [(5, 7, 64, 16, 32, 2, [8, 3, 1, 6, 5]), (3, 4, 50, 8, 16, 1, [2, 4, 3])])
def test_teacher_forced_decode_gradients_adversarial_mode_match_oracle(E, dev, shape):
    """The softmax((logits + gumbel) * temperature) mode of Decoder.forward differentiated (generator.py:50-51), explicit noise, ragged
    lengths (one row of length 1, max(lengths) below and at L + 1), against autograd through the oracle's restatement; the first shape
    takes the fused BPTT step kernels (E % 8, H % 8 == 0), the second the generic products."""
    B, Lc, V, Em, H, NL, lengths = shape
    gen = torch.Generator().manual_seed(77)
    gp = {k: v * 3 for k, v in O.make_gen_params(V, Em, H, NL, gen).items()}
    m = dict(V=V, E=Em, H=H, NL=NL)
    T = 1.7
    dec = _decoder_module(gp, m, 0, dev, T)
    feats = torch.randn(B, Em, generator=gen) * 0.5
    caps = torch.randint(0, V, (B, Lc), generator=gen)
    tmax = max(lengths)
    u = torch.empty(B, tmax, V).uniform_(0, 1, generator=gen)
    d_out = torch.randn(B, tmax, V, generator=gen)
    leaf = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    f_ref = feats.clone().requires_grad_(True)
    want, _ = O.decoder_forward_tf(leaf, f_ref, caps, lengths, T, pretrain=False, u=u)
    (want * d_out).sum().backward()
    f_dev = feats.to(dev).requires_grad_(True)
    got, _ = dec(f_dev, caps.to(dev), lengths, pretrain=False, noise_u=u.to(dev))
    (got * d_out.to(dev)).sum().backward()
    torch.cuda.synchronize()
    close(got, want.detach(), rtol=1e-4, atol_scale=1e-5, what="teacher-forced probabilities")
    close(f_dev.grad, f_ref.grad, rtol=2e-3, atol_scale=1e-4, what="d_features")
    for n, p in zip(dec_param_names(NL), dec.param_list()):
        close(p.grad, leaf[n].grad, rtol=2e-3, atol_scale=1e-4, what=n)


@pytest.mark.parametrize("explicit_noise", [False, True])
def test_ids_only_rollout_with_the_gumbel_max_in_the_vocabulary_product(E, dev, explicit_noise):
    """An ids-only roll-out of many rows in the bf16 mode (the Monte-Carlo roll-outs of the SeqGAN-style step) runs its vocabulary
    product with the Gumbel-max in the epilogue (gemm.hip EPI_GUMBELMAX: swapped operands, one argmax key per row, no [rows, V] logits
    in memory).  Same weights, inputs and noise (device Philox stream or explicit uniforms) through the probability-returning path
    (separate product + Gumbel-softmax kernel) must give the same tokens (generator.py:68-73: argmax of softmax((o + g) T) = argmax
    of o + g), up to bf16-level near-ties between the two products' summation orders."""
    from gan_image_captioning_amd import _lib
    B, L, V, Em, H = 640, 4, 10000, 64, 128                       # 79 x 5 = 395 tiles: the fused product engages
    g = torch.Generator().manual_seed(31)
    gp = {k: v * 4 for k, v in O.make_gen_params(V, Em, H, 1, g).items()}
    eng = E.DecoderEngine(V, Em, H, 1, _lib.BF16)
    assert eng.fused_rollout_rows() == 512 and B > 512             # beyond the fused step kernels: the generic-product path
    params = dec_params(gp, dev)
    feats = (torch.randn(B, Em, generator=g) * 0.5).to(dev)
    u = torch.empty(L, B, V).uniform_(0, 1, generator=g).to(dev) if explicit_noise else None
    _, ids_a, _ = eng.sample_fwd(params, feats, L, 1.0, noise_u=u, seed=99, ids_only=True)
    probs, ids_b, _ = eng.sample_fwd(params, feats, L, 1.0, noise_u=u, seed=99)
    torch.cuda.synchronize()
    assert int(ids_a.min()) >= 0 and int(ids_a.max()) < V
    agree0 = float((ids_a[:, 0] == ids_b[:, 0]).float().mean())     # step 0: identical inputs to both paths
    agree = float((ids_a == ids_b).float().mean())                   # later steps inherit a flipped token's different input
    assert agree0 >= 0.995 and agree >= 0.98, (agree0, agree)
    assert len(torch.unique(ids_a)) > B                               # the noise differs from row to row and from step to step
    # the token the ids-only path picked is (one of) the most probable under the other path's probabilities
    p_pick = probs.float().gather(2, ids_a.unsqueeze(-1)).squeeze(-1)[:, 0]
    assert float((p_pick >= probs.float()[:, 0].max(-1)[0] * 0.98).float().mean()) >= 0.995
