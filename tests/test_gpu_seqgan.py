"""SeqGAN-style step (policy gradient + Monte-Carlo roll-outs, BASELINE config 5) on the GPU against the build-owned CPU oracle
(oracle/cpu_seqgan.py; NO reference counterpart -- the pieces it is built from, the reference's decoder loop and discriminator, are
pinned by the goldens).  fp32 parity mode, explicit noise: sampled captions and every roll-out exact, rewards / losses / gradients
within the tolerances of the adversarial step's tests."""
import pytest
import torch

from oracle import cpu_seqgan as S
from oracle import cpu_step as O
from tests.gpu_util import close, close_mostly, dec_param_names, disc_param_names

pytestmark = pytest.mark.gpu


def _problem(B, L, V, E, H, NL, N, nf, seed):
    g = torch.Generator().manual_seed(seed)
    gp = {k: v * 6 for k, v in O.make_gen_params(V, E, H, NL, g).items()}
    dp = {k: v * 8 for k, v in O.make_disc_params(V, g, num_filters=nf).items()}       # scaled: an informative reward model
    caps = O.make_captions(B, L, V, g)
    us = [torch.empty(B, V).uniform_(0, 1, generator=g) for _ in range(L)]
    umc = torch.empty(L, (L - 1) * N * B, V).uniform_(0, 1, generator=g)
    masks = [torch.empty(B * 64, sum(nf)).bernoulli_(0.8, generator=g) for _ in range(2)]
    return gp, dp, caps, us, umc, masks


@pytest.mark.parametrize("shape", [(4, 5, 52, 8, 16, 1, 3), (6, 6, 64, 16, 32, 2, 2), (8, 6, 64, 16, 32, 1, 14)])   # the last one: 560 roll-outs, resumed
def test_seqgan_step_f32_matches_oracle(shape):
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    B, L, V, E, H, NL, N = shape
    nf = [20, 30, 10]
    gp, dp, caps, us, umc, masks = _problem(B, L, V, E, H, NL, N, nf, sum(shape))
    ref = S.seqgan_step(dict(gp), dict(dp), caps, us, umc, N, masks)
    assert float(ref["rewards"].std()) > 1e-3                # the rewards actually differ between positions
    args = default_args(vocab_size=V, gen_embed_dim=E, gen_hidden_dim=H, gen_num_layers=NL, disc_num_filters=nf, adv_mode="seqgan",
                        mc_rollouts=N, compute_dtype="fp32", device="cuda", log_file=None, model_dir=None, save_dir=None)
    inst = GANInstructor(args, None, None)
    dev = args.device
    with torch.no_grad():
        for n, p in zip(dec_param_names(NL), inst.gen.decoder.param_list()):
            p.copy_(gp[n])
        for n, p in zip(disc_param_names(3), inst.disc.param_list()):
            p.copy_(dp[n])
    inst.gen.train(); inst.disc.train()
    out = inst.seqgan(None, caps.to(dev), L, True, torch.stack(us).to(dev), umc.to(dev), [m.to(dev) for m in masks], opt_step=False)
    torch.cuda.synchronize()
    assert torch.equal(out["ids"].cpu(), ref["Y"]), "sampled captions differ"
    close(out["rewards"], ref["rewards"], rtol=1e-4, atol_scale=1e-5, what="rewards")
    close(out["logits"], ref["logits"], rtol=1e-4, atol_scale=1e-5, what="logits along Y")
    assert float(out["losses"][0]) == pytest.approx(ref["g_loss"], rel=1e-5)
    assert float(out["losses"][1]) == pytest.approx(ref["d_loss"], rel=1e-5)
    ggot = {n: p.grad for n, p in zip(dec_param_names(NL), inst.gen.decoder.param_list())}
    for n, w in ref["g_grads_raw"].items():
        close(ggot[n], w, rtol=2e-3, atol_scale=1e-4, what=n)
    dgot = {n: p.grad for n, p in zip(disc_param_names(3), inst.disc.param_list())}
    up = set(disc_param_names(3)[:7])
    for n, w in ref["d_grads_raw"].items():
        if n in up:
            close_mostly(dgot[n], w, 2e-3, 1e-4, n, 1e-2, 1.5e-2)                 # may carry a re-routed max-pool near-tie
        else:
            close(dgot[n], w, rtol=2e-3, atol_scale=1e-4, what=n, atol_abs=1e-6)
    # the optimizers move both models
    before = inst.gen_arena.flat.clone(), inst.disc_arena.flat.clone()
    inst.disc_opt.step(); inst.gen_opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(before[0], inst.gen_arena.flat) and not torch.equal(before[1], inst.disc_arena.flat)


def test_seqgan_rollouts_keep_their_prefix_and_run_at_scale():
    """Size-independent property at a larger shape in bf16 with device noise (B=16, L=12, V=2000, N=4 -> 704 roll-outs, past the
    fused kernels' row limit: the generic-product path): every roll-out row starts with its caption's prefix; losses are finite; a
    step through adv_step (--adv-mode seqgan) updates both models."""
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    B, L, V, N = 16, 12, 2000, 4
    args = default_args(vocab_size=V, gen_embed_dim=64, gen_hidden_dim=128, adv_mode="seqgan", mc_rollouts=N, compute_dtype="bf16",
                        device="cuda", log_file=None, model_dir=None, save_dir=None)
    inst = GANInstructor(args, None, None)
    dev = args.device
    g = torch.Generator().manual_seed(1)
    caps = O.make_captions(B, L, V, g).to(dev)
    inst.gen.train(); inst.disc.train()
    dec = inst.gen.decoder.engine()
    gparams = [p.detach() for p in inst.gen.decoder.param_list()]
    feats = torch.randn(B, 64, device=dev) * 0.3
    _, Y, _ = dec.sample_fwd(gparams, feats, L, 1.0, seed=11, ids_only=True)
    reps = (L - 1) * N
    flen = torch.arange(1, L, device=dev, dtype=torch.int32).repeat_interleave(N * B)
    _, mc, _ = dec.sample_fwd(gparams, feats.repeat(reps, 1), L, 1.0, seed=12, ids_only=True, force_ids=Y.repeat(reps, 1), force_len=flen)
    torch.cuda.synchronize()
    mc = mc.view(L - 1, N, B, L).cpu()
    Yc = Y.cpu()
    for t in range(1, L):
        assert torch.equal(mc[t - 1, :, :, :t], Yc[None, :, :t].expand(N, B, t)), f"prefix of length {t} not kept"
    assert len(torch.unique(mc[0].reshape(-1, L), dim=0)) > N * B // 2          # the completions actually differ
    # resumed roll-outs (rows start at their prefix length from the state of a teacher-forced pass along Y) are the same roll-outs:
    # same device noise (seed, row, step), so the completions agree except where bf16 rounding of the two routes to the prefix state
    # (64-row fused kernels / 704-row products) flips a sample
    _, _, st_y = dec.sample_fwd(gparams, feats, L, 1.0, pretrain=True, force_ids=Y)
    _, mc2, _ = dec.sample_fwd(gparams, feats.repeat(reps, 1), L, 1.0, seed=12, ids_only=True, force_ids=Y.repeat(reps, 1), force_len=flen,
                               resume=(st_y, B, [min(t, L - 1) * N * B for t in range(L)]))
    torch.cuda.synchronize()
    mc2 = mc2.view(L - 1, N, B, L).cpu()
    for t in range(1, L):
        assert torch.equal(mc2[t - 1, :, :, :t], Yc[None, :, :t].expand(N, B, t)), f"resumed: prefix of length {t} not kept"
    agree = float((mc2 == mc).float().mean())
    assert agree > 0.9, agree
    before = inst.gen_arena.flat.clone(), inst.disc_arena.flat.clone()
    losses = inst.adv_step(None, caps, L, train=True)
    torch.cuda.synchronize()
    assert torch.isfinite(losses).all()
    assert not torch.equal(before[0], inst.gen_arena.flat) and not torch.equal(before[1], inst.disc_arena.flat)
