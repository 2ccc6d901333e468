"""End-to-end parity of the adversarial train step on the GPU against the golden vectors produced by
the reference's own classes (tests/golden/, oracle/make_golden.py): GANInstructor through BOTH step
drivers (``fused`` kernel sequence and ``autograd`` module API), fp32 parity mode, explicit noise.

Per step (parameters teacher-forced to the reference's trajectory, like tests/test_oracle_golden.py):
token ids exact; probabilities / logits rtol 1e-4; losses rel 1e-5; gradient norms rel 1e-4; raw
gradients rtol 2e-3 (+1e-4 x max entry); optimizer: clip+Adam from the golden gradients reproduces the
reference's post-step weights to 1e-6 (tests/test_gpu_kernels.py), and so does the end-to-end step (post/ weights of the `full` goldens).
"""
import pytest
import torch

from tests.golden_io import Golden, initial_params
from tests.gpu_util import close, close_mostly, dec_param_names, disc_param_names

pytestmark = pytest.mark.gpu


def make_instructor(m, impl, dtype="fp32", real_as_ids=1):
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    args = default_args(vocab_size=m["V"], gen_embed_dim=m["E"], gen_hidden_dim=m["H"], gen_num_layers=m["NL"],
                        disc_embed_dim=m["De"], disc_num_rep=m["R"], disc_filter_sizes=m["fs"], disc_num_filters=m["nf"],
                        adv_loss_type=m["loss"], clip_norm=m["clip"], gen_lr=m["gen_lr"], disc_lr=m["disc_lr"],
                        temperature=m["T0"], temp_adpt=m["adapt"], adv_epochs=m["adv_epochs"], compute_dtype=dtype,
                        step_impl=impl, real_as_ids=real_as_ids, device="cuda", log_file=None, model_dir=None, save_dir=None)
    return GANInstructor(args, None, None), args


def load_params(inst, gp, dp):
    with torch.no_grad():
        for n, p in zip(dec_param_names(inst.args.gen_num_layers), inst.gen.decoder.param_list()):
            p.copy_(gp[n])
        for n, p in zip(disc_param_names(len(inst.args.disc_num_filters)), inst.disc.param_list()):
            p.copy_(dp[n])


@pytest.mark.parametrize("impl", ["fused", "autograd"])
@pytest.mark.parametrize("name", ["tiny", "tiny_scaled", "tiny_rep2", "tiny_rsgan", "cfg1"])
def test_adv_step_matches_reference(name, impl):
    g = Golden(name)
    m = g.meta
    inst, args = make_instructor(m, impl)
    dev = args.device
    gp, dp = initial_params(g)
    caps = g.t("caps").to(dev)
    gnames, dnames = dec_param_names(m["NL"]), disc_param_names(len(m["nf"]))
    inst.gen.train()
    inst.disc.train()
    for step in range(m["steps"]):
        load_params(inst, gp, dp)                      # teacher-forced trajectory
        inst.gen.decoder.temperature = m["temperatures"][step]
        u = g.t(f"s{step}/u").to(dev)
        masks = [mk.to(dev) for mk in g.masks(step)]
        pre = f"s{step}/"
        if impl == "fused":
            out = inst.fused(None, caps, m["L"], True, u, masks, opt_step=False)
            losses, ids, probs, lg = out["losses"], out["ids"], out["probs"], out["logits"]
            torch.cuda.synchronize()
            assert torch.equal(ids.cpu(), g.t(pre + "ids")), "token ids differ from the reference"
            close(probs, g.t(pre + "probs"), rtol=1e-4, atol_scale=1e-6, what="probs")
            for i, k in enumerate(("d_real", "d_fake", "g_out")):
                close(lg[i], g.t(pre + k), rtol=1e-4, atol_scale=1e-5, what=k)
        else:
            # module-API flow; stop before the optimizer by running the pieces of _adv_step_autograd
            losses = _autograd_backward_only(inst, caps, m["L"], u, masks)
            torch.cuda.synchronize()
        assert float(losses[0]) == pytest.approx(float(g.t(pre + "g_loss")), rel=1e-5)
        assert float(losses[1]) == pytest.approx(float(g.t(pre + "d_loss")), rel=1e-5)
        # raw gradients in the flat arenas
        full = m["full"]
        want = g.group(pre + "grad/")
        got = {n: p.grad for n, p in zip(gnames, inst.gen.decoder.param_list())}
        got.update({n: p.grad for n, p in zip(dnames, inst.disc.param_list())})
        dn = float(torch.sqrt(sum((got[n].double() ** 2).sum() for n in dnames)))
        gn = float(torch.sqrt(sum((got[n].double() ** 2).sum() for n in gnames)))
        assert dn == pytest.approx(float(g.t(pre + "d_norm")), rel=1e-4)
        # At T=100 the softmax is saturated and G's gradient is numerically nil (norm ~1e-11, made of
        # e^-100-sized terms): only its insignificance is comparable across exp/log implementations.
        g_nil = float(g.t(pre + "g_norm")) < 1e-8
        if g_nil:
            assert gn < 1e-8
        else:
            assert gn == pytest.approx(float(g.t(pre + "g_norm")), rel=2e-3)
        if full:
            upstream_of_pool = set(dnames[:1 + 2 * len(m["nf"])])
            for n, w in want.items():
                if n in gnames and g_nil:
                    continue
                if n in upstream_of_pool or n in gnames:      # may carry a re-routed max-pool near-tie
                    close_mostly(got[n], w, 2e-3, 1e-4, n, 1e-2, 1.5e-2)
                else:
                    close(got[n], w, rtol=2e-3, atol_scale=1e-4, what=n, atol_abs=1e-6)   # rsgan: exact cancellations (sum of +g and -g)
            if m["loss"] == "rsgan":
                assert set(want) == set(dnames)               # no generator gradient at all (utils.py:48)
                assert gn == 0.0
        # optimizer step from our own gradients stays within lr of the reference's post-step weights
        inst.disc_opt.step()
        inst.gen_opt.step()
        torch.cuda.synchronize()
        if full:
            post = g.group(pre + "post/")
            # the reference's own post-step weights: Adam's update is lr * m_hat / (sqrt(v_hat) + 1e-8), i.e. +-lr wherever |g| >> 1e-8,
            # so 5 % of lr pins the sign of every gradient entry, the clip scale and the moments' arithmetic (entries with |g| ~ 1e-7,
            # e.g. feature2out.bias under rsgan, turn a 3 % gradient difference into 3 % of lr: measured 3.4e-6 at lr 1e-4); an entry that
            # carries a re-routed max-pool near-tie may differ more (counted: <= 0.1 % of a tensor, never > 2.1 lr); an optimizer that
            # does nothing fails (the weights must have MOVED by ~lr from the pre-step values)
            pre_w = {**gp, **dp}
            for n, p in list(zip(gnames, inst.gen.decoder.param_list())) + list(zip(dnames, inst.disc.param_list())):
                lr = m["gen_lr"] if n in gnames else m["disc_lr"]
                got_w = p.detach().cpu()
                err = (got_w - post[n]).abs()
                moved_ref = float((post[n] - pre_w[n]).abs().max())
                # entries whose reference gradient is rounding noise are excluded: there Adam maps the noise to anything in
                # [-lr, lr] (out2logits.bias under rsgan cancels exactly in d_real - d_fake: analytically zero gradient)
                live = want[n].abs() > 1e-6 if n in want else torch.zeros_like(err, dtype=torch.bool)
                bad = float((err[live] > 0.05 * lr).float().mean()) if bool(live.any()) else 0.0
                assert bad <= 1e-3 and float(err.max()) <= 2.1 * lr, (n, bad, float(err.max()))
                if moved_ref > 0.5 * lr:
                    assert float((got_w - pre_w[n]).abs().max()) > 0.5 * lr, f"{n}: the optimizer did not move the weights"
            gp = {n: post[n] for n in gnames}
            dp = {n: post[n] for n in dnames}
        else:
            gp = {n: p.detach().cpu().clone() for n, p in zip(gnames, inst.gen.decoder.param_list())}
            dp = {n: p.detach().cpu().clone() for n, p in zip(dnames, inst.disc.param_list())}
            if step > 0:
                break        # own trajectory beyond step 0 is only loosely comparable (see test_oracle_golden)


def _autograd_backward_only(inst, caps, L, u, masks):
    from gan_image_captioning_amd.utils import get_losses
    features = inst._features(None, caps.shape[0])
    gen_captions, _ = inst.gen.decoder.sample(features, max_caption_len=L, noise_u=u)
    d_real = inst.disc(caps, keep_mask=masks[0])
    d_fake = inst.disc(gen_captions.detach(), keep_mask=masks[1])
    with inst.disc.input_grad_only():
        g_out = inst.disc(gen_captions, keep_mask=masks[2])
    g_loss, d_loss = get_losses(d_real, d_fake, g_out, inst.args.adv_loss_type, detach_d_for_g=True)
    inst.disc_opt.zero_grad()
    inst.gen_opt.zero_grad()
    d_loss.backward()
    if g_loss.requires_grad:
        g_loss.backward()
    return torch.stack([g_loss.detach(), d_loss.detach()])


def test_dense_one_hot_real_path_equals_ids_path():
    g = Golden("tiny")
    m = g.meta
    gp, dp = initial_params(g)
    res = []
    for real_as_ids in (1, 0):
        inst, args = make_instructor(m, "fused", real_as_ids=real_as_ids)
        load_params(inst, gp, dp)
        inst.gen.decoder.temperature = m["temperatures"][0]
        out = inst.fused(None, g.t("caps").to(args.device), m["L"], True, g.t("s0/u").to(args.device),
                         [mk.to(args.device) for mk in g.masks(0)], opt_step=False)
        torch.cuda.synchronize()
        res.append((out["losses"].cpu(), inst.disc_arena.grad.cpu().clone()))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-6, atol=0)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-8)


def test_eval_mode_step_matches_reference():
    g = Golden("tiny_eval")
    m = g.meta
    inst, args = make_instructor(m, "fused")
    gp, dp = initial_params(g)
    load_params(inst, gp, dp)
    inst.gen.eval()
    inst.disc.eval()
    inst.gen.decoder.temperature = m["temperatures"][0]
    out = inst.fused(None, g.t("caps").to(args.device), m["L"], False, g.t("s0/u").to(args.device), None)
    torch.cuda.synchronize()
    assert torch.equal(out["ids"].cpu(), g.t("s0/ids"))
    assert float(out["losses"][0]) == pytest.approx(float(g.t("s0/g_loss")), rel=1e-5)
    assert float(out["losses"][1]) == pytest.approx(float(g.t("s0/d_loss")), rel=1e-5)


def test_bf16_step_close_to_reference():
    """bf16 compute mode on the cfg1 fixture: loss rel error <= 2e-2; id match-rate reported."""
    g = Golden("cfg1")
    m = g.meta
    inst, args = make_instructor(m, "fused", dtype="bf16")
    gp, dp = initial_params(g)
    load_params(inst, gp, dp)
    inst.gen.decoder.temperature = m["temperatures"][0]
    out = inst.fused(None, g.t("caps").to(args.device), m["L"], True, g.t("s0/u").to(args.device),
                     [mk.to(args.device) for mk in g.masks(0)])
    torch.cuda.synchronize()
    match = float((out["ids"].cpu() == g.t("s0/ids")).float().mean())
    print(f"bf16 id match-rate vs reference: {match:.3f}")
    assert match >= 0.95
    assert float(out["losses"][0]) == pytest.approx(float(g.t("s0/g_loss")), rel=2e-2)
    assert float(out["losses"][1]) == pytest.approx(float(g.t("s0/d_loss")), rel=2e-2)


def test_multi_step_training_runs_and_updates(tmp_path):
    """A few free-running steps with on-device Philox noise through adv_step (the path bench.py times)."""
    g = Golden("cfg1")
    m = g.meta
    inst, args = make_instructor(m, "fused", dtype="bf16")
    caps = g.t("caps").to(args.device)
    before = inst.gen_arena.flat.clone()
    vals = []
    for k in range(4):
        losses = inst.adv_step(None, caps, m["L"], train=True)
        inst.update_temperature(0 + (k + 1) / 50, 30)
        vals.append(losses.tolist())
    torch.cuda.synchronize()
    assert all(torch.isfinite(torch.tensor(v)).all() for v in vals)
    assert int(inst.gen_opt.step_count) == 4 and int(inst.disc_opt.step_count) == 4
    assert not torch.equal(before, inst.gen_arena.flat)


def test_bf16_step_at_bench_scale_close_to_oracle():
    """The benchmark's own shapes (B=64, L=20, V=10000, E=H=512, R=64, F=900; --conditional-gan 0 so that the CPU oracle finishes in
    seconds): the kernels that only engage at this scale (8-wave highway products, split-K gradients, shared D forward, stream
    overlap) against one fp32 oracle step on the same weights, noise and dropout masks."""
    from oracle import cpu_step as O
    from tests.gpu_util import dec_param_names, disc_param_names
    m = dict(B=64, L=20, V=10000, E=512, H=512, NL=1, De=64, R=64, fs=[3, 4, 5], nf=[300, 300, 300], loss="standard", clip=5.0,
             gen_lr=1e-4, disc_lr=1e-4, T0=100, adapt="exp", adv_epochs=30)
    g = torch.Generator().manual_seed(99)
    gp = O.make_gen_params(m["V"], m["E"], m["H"], m["NL"], g)
    dp = O.make_disc_params(m["V"], g)
    caps = O.make_captions(m["B"], m["L"], m["V"], g)
    us, masks = O.make_noise(m["B"], m["L"], m["V"], 900, m["R"], g)
    T = 1.7
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = O.adv_step(dict(gp), dict(dp), caps, us, masks, T, "standard", 5.0, None, None)
    inst, args = make_instructor(m, "fused", dtype="bf16")
    dev = args.device
    load_params(inst, gp, dp)
    inst.gen.train(); inst.disc.train()
    inst.gen.decoder.temperature = T
    out = inst.fused(None, caps.to(dev), m["L"], True, torch.stack(us).to(dev), [k.to(dev) for k in masks], opt_step=False)
    torch.cuda.synchronize()
    match = float((out["ids"].cpu() == ref["ids"]).float().mean())
    print(f"bench-scale bf16 id match-rate vs fp32 oracle: {match:.3f}")
    assert match >= 0.9
    assert float(out["losses"][1]) == pytest.approx(ref["d_loss"], rel=2e-2)
    assert float(out["losses"][0]) == pytest.approx(ref["g_loss"], rel=3e-2)
    from tests.gpu_util import rel_l2
    dgot = {n: p.grad for n, p in zip(disc_param_names(3), inst.disc.param_list())}
    for n in ("highway.weight", "feature2out.weight", "out2logits.weight", "embeddings.weight"):
        err = rel_l2(dgot[n], ref["d_grads_raw"][n])
        assert err < 8e-2, f"{n}: rel L2 {err}"
    # G's gradient flows through the sampled trajectory.  The index is detached (generator.py:75), so on a GIVEN trajectory the
    # gradient is the same function of the weights: where a bf16 near-tie flipped an argmax, the oracle is re-run on the GPU's own
    # trajectory (force_ids) and the comparison is made unconditionally.
    if match < 1.0:
        ref = O.adv_step(dict(gp), dict(dp), caps, us, masks, T, "standard", 5.0, None, None, force_ids=out["ids"].cpu())
    ggot = {n: p.grad for n, p in zip(dec_param_names(1), inst.gen.decoder.param_list())}
    for n in ("decoder.linear.weight", "decoder.lstm.weight_hh_l0", "decoder.lstm.weight_ih_l0", "decoder.embed.weight"):
        err = rel_l2(ggot[n], ref["g_grads_raw"][n])
        assert err < 8e-2, f"{n}: rel L2 {err}"


def test_cfg2_composed_step_bf16_vs_oracle():
    """BASELINE configs[1] itself, composed: --conditional-gan 1, ResNet-50-shaped trunk at 224x224, B=64, L=20, V=10000, E=H=512,
    bf16 compute, ONE step through FusedAdvStep with explicit Gumbel uniforms / dropout masks, against the fp32 CPU oracle
    (oracle.cpu_step.adv_step on oracle.cpu_encoder.trunk_forward's features; body of reference src/training.py:144-169).
    Tolerances per tensor are written below; the per-stage trunk error (the budget behind the 5e-2 of the pooled features) is
    asserted stage by stage and recorded in gpurun_out/cfg2_parity.json."""
    import json
    import os
    from oracle import cpu_encoder as OE
    from oracle import cpu_step as O
    from tests.gpu_util import dec_param_names, disc_param_names, rel_l2
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    B, L, V, E, H = 64, 20, 10000, 512, 512
    g = torch.Generator().manual_seed(4242)
    gp = O.make_gen_params(V, E, H, 1, g, trunk_feat_dim=OE.out_features("resnet50"))
    dp = O.make_disc_params(V, g)
    tp = OE.make_trunk_params("resnet50", g)
    caps = O.make_captions(B, L, V, g)
    images = torch.randn(B, 3, 224, 224, generator=g)
    us, masks = O.make_noise(B, L, V, 900, 64, g)
    T = 1.7
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    taps, taps16 = {}, {}
    with torch.no_grad():
        feat = OE.trunk_forward(tp, images, "resnet50", taps=taps)
        # bf16 STORAGE emulated on the CPU (round 3); conv3 of these blocks is normalised from its f32 accumulators (gic_conv_b2b)
        unstored = {"4.0.conv3", "4.1.conv3", "4.2.conv3", "5.0.conv3", "5.1.conv3", "5.2.conv3", "5.3.conv3"}
        feat16 = OE.trunk_forward(tp, images, "resnet50", taps=taps16, emulate_bf16=True, unstored=unstored)
    ref = O.adv_step(dict(gp), dict(dp), caps, us, masks, T, "standard", 5.0, None, None, trunk_feat=feat)

    args = default_args(vocab_size=V, gen_embed_dim=E, gen_hidden_dim=H, conditional_gan=1, encoder_arch="resnet50", compute_dtype="bf16",
                        step_impl="fused", adv_train_batch_size=B, image_size=224, device="cuda", log_file=None, model_dir=None, save_dir=None)
    inst = GANInstructor(args, None, None)
    dev = args.device
    with torch.no_grad():
        for n, p in zip(dec_param_names(1), inst.gen.decoder.param_list()):
            p.copy_(gp[n])
        for n, p in zip(disc_param_names(3), inst.disc.param_list()):
            p.copy_(dp[n])
        inst.gen.encoder.resnet.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
        enc = inst.gen.encoder
        for n in ("linear.weight", "linear.bias", "bn.weight", "bn.bias"):
            mod, attr = n.split(".")
            getattr(getattr(enc, mod), attr).copy_(gp["encoder." + n])
    inst.gen.train(); inst.disc.train()
    inst.gen.decoder.temperature = T
    out = inst.fused(images.to(dev), caps.to(dev), L, True, torch.stack(us).to(dev), [k.to(dev) for k in masks], opt_step=False)
    torch.cuda.synchronize()
    report = {}

    # ---- trunk, stage by stage (the B=64 plan: the ring variants the benchmark runs).  Budget: each stage adds bf16 rounding of
    # 3-4 BatchNorm-normalised convolution outputs per block; the pooled feature averages 49 positions.
    plan = enc.resnet._plan
    pb = plan._bufs[(B, 224)]
    stage_last = []
    i = -1
    for stage in enc.resnet.stages():
        i += len(stage)
        stage_last.append(i)
    # measured (gpurun_out/cfg2_parity.json, committed as profiles/r02_cfg2_parity.json): stem 2.5e-3, stage0 1.1e-2, stage1 2.4e-2,
    # stage2 5.5e-2, stage3 9.8e-2, pooled 1.7e-2.  The error roughly doubles per stage: under the synthetic U(-0.05, 0.05) init a
    # BatchNorm'd convolution output has |mean| comparable to its spread, so the mean subtraction amplifies the bf16 rounding of its
    # input; the 49-position average of the pooled feature cancels most of it again.  fp32 mode matches the same oracle to 1e-3.
    budget = {"stem": 6e-3, "stage0": 2.5e-2, "stage1": 5e-2, "stage2": 9e-2, "stage3": 1.4e-1, "pooled": 3e-2}
    got_taps = {"stem": pb["x0"]}
    for si, bi in enumerate(stage_last):
        got_taps[f"stage{si}"] = pb["blocks"][bi]["out"]
    for k, t in got_taps.items():
        report["trunk_rel_l2/" + k] = rel_l2(t.float().permute(0, 3, 1, 2), taps[k])
    # the plan's own feature buffer still holds this step's pooled output (no later pass ran)
    report["trunk_rel_l2/pooled"] = rel_l2(pb["feat"].float(), feat)
    failures = [f"trunk {k}: rel L2 {report['trunk_rel_l2/' + k]:.3e} over its budget {lim}" for k, lim in budget.items()
                if not report["trunk_rel_l2/" + k] < lim]
    # the same stages against the oracle that rounds to bf16 what the bf16 mode stores (cpu_encoder.trunk_forward(emulate_bf16=True)): the
    # budget above is reproduced by storage alone, and the kernels sit about twice closer to that oracle (exactly on it at the stem);
    # limits as in tests/test_gpu_encoder.py::test_trunk_forward_bf16_is_explained_by_bf16_storage
    storage_limit = {"stem": 1e-4, "stage0": 5e-3, "stage1": 2.5e-2, "stage2": 6e-2, "stage3": 1e-1, "pooled": 2e-2}
    assert inst.gen.encoder.resnet._plan.unstored_convs() == unstored, inst.gen.encoder.resnet._plan.unstored_convs()
    for k, t in got_taps.items():
        report["trunk_vs_bf16_storage/" + k] = rel_l2(t.float().permute(0, 3, 1, 2), taps16[k])
        report["trunk_storage_vs_fp32/" + k] = rel_l2(taps16[k], taps[k])
    report["trunk_vs_bf16_storage/pooled"] = rel_l2(pb["feat"].float(), feat16)
    report["trunk_storage_vs_fp32/pooled"] = rel_l2(feat16, feat)
    failures += [f"trunk {k} vs the bf16-storage oracle: rel L2 {report['trunk_vs_bf16_storage/' + k]:.3e} over {lim}" for k, lim in storage_limit.items()
                 if not report["trunk_vs_bf16_storage/" + k] < lim]

    # ---- roll-out: token ids (bf16 near-ties may flip an argmax; everything downstream then follows the GPU's trajectory)
    ids = out["ids"].cpu()
    match = float((ids == ref["ids"]).float().mean())
    report["id_match_rate"] = match
    if match < 1.0:
        ref = O.adv_step(dict(gp), dict(dp), caps, us, masks, T, "standard", 5.0, None, None, trunk_feat=feat, force_ids=ids)
    report["probs_rel_l2"] = rel_l2(out["probs"].float(), ref["probs"])
    # ---- losses: rel 2e-2 (d_loss) / 3e-2 (g_loss), SURVEY §8(c)
    gl, dl = (float(v) for v in out["losses"])
    report["g_loss"], report["g_loss_ref"], report["d_loss"], report["d_loss_ref"] = gl, ref["g_loss"], dl, ref["d_loss"]
    # ---- gradients: relative L2 per tensor, D 8e-2, decoder 1e-1.  The encoder head's own gradients are compared twice: composed
    # (against the oracle's trunk: loose, 3e-1 -- under the synthetic init the pooled features of different random images are nearly
    # identical, BatchNorm1d removes their common part, and the 1.7e-2 feature error is ~2e-1 of what is left), and with the oracle
    # fed the GPU's own trunk features (everything downstream of the trunk alone: 1e-1)
    dgot = {n: p.grad for n, p in zip(disc_param_names(3), inst.disc.param_list())}
    ggot = {n: p.grad for n, p in zip(dec_param_names(1), inst.gen.decoder.param_list())}
    ggot.update({"encoder.linear.weight": enc.linear.weight.grad, "encoder.linear.bias": enc.linear.bias.grad,
                 "encoder.bn.weight": enc.bn.weight.grad, "encoder.bn.bias": enc.bn.bias.grad})
    for n in ("highway.weight", "feature2out.weight", "out2logits.weight", "embeddings.weight", "convs.0.weight", "convs.2.weight"):
        report["d_grad_rel_l2/" + n] = rel_l2(dgot[n], ref["d_grads_raw"][n])
        if not report["d_grad_rel_l2/" + n] < 8e-2:
            failures.append(f"{n}: rel L2 {report['d_grad_rel_l2/' + n]}")
    centered = lambda t: t - t.mean(0, keepdim=True)
    report["trunk_rel_l2/pooled_centered"] = rel_l2(centered(pb["feat"].float().cpu()), centered(feat))
    ref_own = O.adv_step(dict(gp), dict(dp), caps, us, masks, T, "standard", 5.0, None, None, trunk_feat=pb["feat"].float().cpu(), force_ids=ids)
    for n in ("encoder.linear.weight", "encoder.bn.weight", "encoder.bn.bias", "decoder.lstm.weight_ih_l0", "decoder.linear.weight"):
        report["g_grad_rel_l2_own_trunk/" + n] = rel_l2(ggot[n], ref_own["g_grads_raw"][n])
        if not report["g_grad_rel_l2_own_trunk/" + n] < 1e-1:
            failures.append(f"{n} (oracle on the GPU's trunk features): rel L2 {report['g_grad_rel_l2_own_trunk/' + n]}")
    for n, lim in (("decoder.linear.weight", 1e-1), ("decoder.linear.bias", 1e-1), ("decoder.lstm.weight_hh_l0", 1e-1),
                   ("decoder.lstm.weight_ih_l0", 1e-1), ("decoder.embed.weight", 1e-1), ("encoder.linear.weight", 3e-1), ("encoder.bn.weight", 3e-1),
                   ("encoder.bn.bias", 1.5e-1)):
        report["g_grad_rel_l2/" + n] = rel_l2(ggot[n], ref["g_grads_raw"][n])
        if not report["g_grad_rel_l2/" + n] < lim:
            failures.append(f"{n}: rel L2 {report['g_grad_rel_l2/' + n]} (limit {lim})")
    print("cfg2 composed parity:", json.dumps(report))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "cfg2_parity.json"), "w") as fh:
            json.dump(report, fh, indent=1)
    except OSError:
        pass
    assert match >= 0.9
    assert report["probs_rel_l2"] < 5e-2
    assert dl == pytest.approx(ref["d_loss"], rel=2e-2)
    assert gl == pytest.approx(ref["g_loss"], rel=3e-2)
    assert not failures, failures


def test_step_graph_replay_equals_eager_launches(monkeypatch):
    """The train step replayed as six linear hipGraphs (FusedAdvStep._call_graph: temperature and noise seeds read from device memory,
    gicap.h gic_step_scalars) against the same step as eager launches (GIC_NO_STEP_GRAPH=1): fp32, device-drawn noise from the same
    seed sequence, five steps (eager, capture, three replays) with the temperature changing every step (training.py:183).  Step 0's
    losses and token ids are exact; later steps carry the f32-atomic summation order of D's weight gradients (not reproducible run to
    run in either mode), so weights are compared at the tolerance of the data-parallel tests.  reference: src/training.py:136-183."""
    from gan_image_captioning_amd.generator import SEEDS
    from gan_image_captioning_amd.utils import get_fixed_temperature
    g = Golden("cfg1")                  # B=8, L=10, V=64, E=32, H=512: the shapes at which the fused roll-out / BPTT kernels engage
    m = g.meta
    gp, dp = initial_params(g)
    res = {}
    for mode in ("eager", "graph"):
        if mode == "eager":
            monkeypatch.setenv("GIC_NO_STEP_GRAPH", "1")
        else:
            monkeypatch.delenv("GIC_NO_STEP_GRAPH", raising=False)
        inst, args = make_instructor(m, "fused")
        assert inst.fused.use_graph == (mode == "graph")
        dev = args.device
        load_params(inst, gp, dp)
        inst.gen.train(); inst.disc.train()
        caps = g.t("caps").to(dev)
        SEEDS.reset(1234)
        losses, ids = [], []
        for k in range(5):
            inst.gen.decoder.temperature = get_fixed_temperature(m["T0"], k + 1, 7, "exp")     # changes every step
            out = inst.fused(None, caps, m["L"], True)
            losses.append(out["losses"].clone()); ids.append(out["ids"].clone())
        torch.cuda.synchronize()
        if mode == "graph":
            assert len(inst.fused._graphs) == 1 and len(next(iter(inst.fused._graphs.values()))["graphs"]) == 6, "the step was not captured"
        res[mode] = {"losses": torch.stack(losses).cpu(), "ids": torch.stack(ids).cpu(), "gen": inst.gen_arena.flat.cpu().clone(),
                     "disc": inst.disc_arena.flat.cpu().clone(), "steps": int(inst.gen_opt.step_count)}
    e, r = res["eager"], res["graph"]
    assert e["steps"] == r["steps"] == 5
    assert torch.equal(e["losses"][0], r["losses"][0]) and torch.equal(e["ids"][0], r["ids"][0])
    torch.testing.assert_close(e["losses"], r["losses"], rtol=1e-4, atol=1e-6)
    assert float((e["ids"] == r["ids"]).float().mean()) >= 0.98
    torch.testing.assert_close(e["disc"], r["disc"], rtol=1e-4, atol=5e-6)
    torch.testing.assert_close(e["gen"], r["gen"], rtol=1e-4, atol=5e-6)
    assert not torch.equal(e["losses"][1], e["losses"][3])               # the steps differ from each other (weights, temperature, noise)


def test_attention_step_graph_replay_equals_eager_launches(monkeypatch):
    """The same check for the attention decoder's step (cfg4), whose graphs are opt-in (GIC_STEP_GRAPH_ATTN=1: measured slightly slower
    than eager launches there): ResNet-18 trunk at 64x64, fp32, device noise from the same seed sequence, four steps."""
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.generator import SEEDS
    from gan_image_captioning_amd.training import GANInstructor
    from oracle import cpu_step as O
    B, L, V = 8, 6, 64
    g = torch.Generator().manual_seed(77)
    images = torch.randn(B, 3, 64, 64, generator=g)
    caps = O.make_captions(B, L, V, g)
    res = {}
    state = None
    for mode in ("eager", "graph"):
        if mode == "graph":
            monkeypatch.setenv("GIC_STEP_GRAPH_ATTN", "1")
        else:
            monkeypatch.delenv("GIC_STEP_GRAPH_ATTN", raising=False)
        torch.manual_seed(5)
        args = default_args(vocab_size=V, gen_embed_dim=16, gen_hidden_dim=32, conditional_gan=1, encoder_arch="resnet18", decoder="attention",
                            attn_dim=24, compute_dtype="fp32", image_size=64, adv_train_batch_size=B, device="cuda", log_file=None,
                            model_dir=None, save_dir=None)
        inst = GANInstructor(args, None, None)
        assert inst.fused.use_graph == (mode == "graph")
        if state is None:
            state = ({k: v.clone() for k, v in inst.gen.state_dict().items()}, {k: v.clone() for k, v in inst.disc.state_dict().items()})
        else:
            inst.gen.load_state_dict(state[0]); inst.disc.load_state_dict(state[1])
            from gan_image_captioning_amd import engine
            engine.bump_param_epoch()
        inst.gen.train(); inst.disc.train()
        dev = args.device
        im, cp = images.to(dev), caps.to(dev)
        SEEDS.reset(4321)
        losses = []
        for k in range(4):
            inst.gen.decoder.temperature = 1.0 + 0.3 * k
            losses.append(inst.fused(im, cp, L, True)["losses"].clone())
        torch.cuda.synchronize()
        if mode == "graph":
            assert len(inst.fused._graphs) == 1
        res[mode] = (torch.stack(losses).cpu(), inst.gen_arena.flat.cpu().clone(), inst.disc_arena.flat.cpu().clone())
    # (the trunk's BatchNorm sums are f32 atomics: the features differ in the last bits from run to run)
    torch.testing.assert_close(res["eager"][0], res["graph"][0], rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(res["eager"][2], res["graph"][2], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(res["eager"][1], res["graph"][1], rtol=1e-3, atol=2e-5)
