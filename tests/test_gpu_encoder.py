"""GPU parity of the image encoder (trunk convolutions on MFMA, BatchNorm on batch statistics, pooling, the
Linear+BatchNorm1d head) against the CPU oracle (oracle/cpu_encoder.py; PARITY UNPINNED w.r.t. the reference
because torchvision is absent - see its header) and, for the head, against the golden vectors produced with
the reference's own nn.Linear / nn.BatchNorm1d(momentum=0.01) composition (tiny_cgan_head).

fp32 mode: activations rtol 1e-3 of the tensor's max (53 stacked BatchNorms amplify fp32 reduction-order noise);
bf16 mode: relative L2 error reported, <= 5e-2.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_encoder as OE
from oracle import cpu_step as O
from tests.golden_io import Golden
from tests.gpu_util import close, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _lib():
    from gan_image_captioning_amd import _lib as L
    return L


def conv_hip(x_nchw, w, stride, pad, dtype, dev, want_stats=True):
    """x [N,C,H,W] f32 CPU, w [Co,Ci,k,k] -> (y NCHW f32 cpu, stats)"""
    from gan_image_captioning_amd import engine
    L = _lib()
    act = engine.TORCH_DTYPE[dtype]
    N, C, H, W = x_nchw.shape
    Co, _, k, _ = w.shape
    xh = x_nchw.permute(0, 2, 3, 1).contiguous().to(dev).to(act)
    wp = torch.empty(Co, k, k, C, device=dev, dtype=act)
    L.check(L.load().gic_repack_conv_weight(w.to(dev).contiguous().data_ptr(), wp.data_ptr(), dtype, Co, C, k, k, C, k, engine.stream_ptr()), "repack")
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(N, Ho, Wo, Co, device=dev, dtype=act)
    stats = torch.zeros(4, 2 * Co, device=dev)
    L.check(L.load().gic_conv2d(xh.data_ptr(), wp.data_ptr(), y.data_ptr(), stats.data_ptr() if want_stats else None, 4, dtype,
                                N, H, W, C, Co, k, k, stride, pad, engine.stream_ptr()), "conv2d")
    torch.cuda.synchronize()
    return y.float().cpu().permute(0, 3, 1, 2), stats.sum(0).cpu()


CONV_CASES = [  # (N, Cin, H, Cout, k, stride, pad)
    (2, 64, 16, 64, 3, 1, 1), (2, 64, 16, 128, 3, 2, 1), (3, 64, 8, 256, 1, 1, 0), (2, 128, 8, 64, 1, 2, 0),
    (1, 256, 7, 512, 3, 1, 1), (4, 64, 56, 64, 3, 1, 1), (2, 512, 4, 2048, 1, 1, 0), (2, 8, 10, 24, 3, 1, 1),
    # grids past one / two workgroups per CU: the 2-stage ring and the ring-less shallow-K variant of the 8-wave kernel
    (16, 64, 56, 128, 1, 1, 0), (32, 64, 56, 64, 1, 1, 0), (12, 128, 28, 256, 3, 1, 1),
    # the patch-resident 3x3 kernel (conv3x3.hip): eight 64-channel chunks with tiles crossing several 7x7 images, three chunks on an
    # odd width with a ragged output-channel tile, two chunks / 64 output channels, per-image tiling with a short last tile
    (8, 512, 7, 512, 3, 1, 1), (2, 192, 12, 96, 3, 1, 1), (5, 128, 14, 64, 3, 1, 1), (3, 64, 56, 128, 3, 1, 1), (20, 256, 14, 256, 3, 1, 1),
    # the streaming shallow-K 1x1 kernel (conv1x1_stream.hip; >= 1024 output tiles): K = 64 with a ragged 64-wide channel tile and a
    # row tail, K = 64 / 256 output channels, K = 128 / 512 output channels
    (41, 64, 56, 96, 1, 1, 0), (40, 64, 56, 256, 1, 1, 0), (44, 128, 28, 512, 1, 1, 0),
    # the A-panel-resident 1x1 kernel (conv1x1_panel.hip; K = 256, >= 512 output channels): a row tail, groups of channel tiles
    # ... and a short last row tile whose workgroups walk four channel tiles each
    (5, 256, 14, 1024, 1, 1, 0), (64, 256, 14, 512, 1, 1, 0), (33, 256, 14, 1024, 1, 1, 0),
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(dev, case, dtype):
    N, Ci, H, Co, k, st, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * 0.1
    if dtype == 1:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    want = F.conv2d(x.double(), w.double(), None, st, pad)
    y, stats = conv_hip(x, w, st, pad, dtype, dev)
    tol = 2e-5 if dtype == 0 else 8e-3
    err = float((y.double() - want).abs().max() / want.abs().max())
    assert err < tol * max(1.0, (Ci * k * k / 64) ** 0.5), f"conv {case} dtype {dtype}: rel max err {err}"
    rows = want.shape[0] * want.shape[2] * want.shape[3]
    s1 = want.sum((0, 2, 3))
    s2 = (want ** 2).sum((0, 2, 3))
    close(stats[:Co], s1, rtol=1e-3, atol_scale=1e-4 * rows ** 0.5, what="bn sum")
    close(stats[Co:], s2, rtol=1e-3, what="bn sumsq")


@pytest.mark.parametrize("case", [(3, 64, 10, 14, 64), (2, 128, 5, 33, 72), (7, 64, 3, 3, 128), (1, 320, 21, 8, 64), (9, 64, 57, 6, 64),
                                  (2, 64, 1, 40, 64), (5, 192, 13, 13, 200),
                                  # round 3: full-size grids (14x14 maps of 64 images, 196 workgroups), an odd map with a ragged channel
                                  # tile on such a grid, and a few-tile several-chunk case on 7x7 maps
                                  (64, 128, 14, 14, 256), (70, 128, 13, 15, 200), (16, 128, 7, 7, 128)])
def test_patch_resident_conv3x3_on_odd_shapes(dev, case):
    """conv3x3.hip away from the ResNet shapes: non-square maps, widths around the 64-pixel piece stride, single rows, tiles that
    cross many small images, a ragged output-channel tile, five 64-channel chunks -- against torch (float64) on bf16-rounded operands,
    with and without BatchNorm + ReLU on load (the latter against the normalised tensor convolved the same way)."""
    from gan_image_captioning_amd import engine
    L = _lib()
    lib = L.load()
    N, Ci, H, W, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = (torch.randn(N, Ci, H, W, generator=g)).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1).bfloat16().float()
    want = F.conv2d(x.double(), w.double(), None, 1, 1)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
    wp = w.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
    y = torch.empty(N, H, W, Co, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(4, 2 * Co, device=dev)
    s = engine.stream_ptr()
    L.check(lib.gic_conv2d(xh.data_ptr(), wp.data_ptr(), y.data_ptr(), stats.data_ptr(), 4, 1, N, H, W, Ci, Co, 3, 3, 1, 1, s), "conv2d")
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 8e-3 * max(1.0, (Ci * 9 / 64) ** 0.5), f"{case}: rel max err {err}"
    close(stats.sum(0)[:Co].cpu(), want.sum((0, 2, 3)), rtol=2e-3, atol_scale=1e-3 * (N * H * W) ** 0.5, what="bn sum")
    # BatchNorm + ReLU on load: statistics of x itself, random affine
    rows = N * H * W
    xf = xh.float().reshape(rows, Ci)
    in_stats = torch.zeros(3, 2 * Ci, device=dev)
    in_stats[0, :Ci] = xf.sum(0)
    in_stats[2, Ci:] = (xf * xf).sum(0)
    gamma = (torch.rand(Ci, generator=g) + 0.5).to(dev)
    beta = (torch.randn(Ci, generator=g) * 0.2).to(dev)
    mean = xf.mean(0)
    var = ((xf * xf).mean(0) - mean * mean).clamp_min(0)
    sc = gamma * torch.rsqrt(var + 1e-5)
    z = torch.relu(xf * sc + (beta - mean * sc)).bfloat16().float().reshape(N, H, W, Ci).permute(0, 3, 1, 2).cpu()
    want2 = F.conv2d(z.double(), w.double(), None, 1, 1)
    y2 = torch.empty_like(y)
    st2 = torch.zeros_like(stats)
    status = lib.gic_conv2d_bn_in(xh.data_ptr(), in_stats.data_ptr(), 3, gamma.data_ptr(), beta.data_ptr(), float(rows), wp.data_ptr(),
                                  y2.data_ptr(), st2.data_ptr(), 4, 1, N, H, W, Ci, Co, 3, 3, 1, 1, s)
    torch.cuda.synchronize()
    L.check(status, "conv2d_bn_in")
    got2 = y2.float().cpu().permute(0, 3, 1, 2).double()
    err2 = float((got2 - want2).abs().max() / want2.abs().max())
    assert err2 < 1.2e-2 * max(1.0, (Ci * 9 / 64) ** 0.5), f"{case} with bn on load: rel max err {err2}"


@pytest.mark.parametrize("case", [(3, 70), (2, 230), (5, 38), (70, 134)])
def test_streaming_stem_conv_matches_torch(dev, case):
    """conv_stem.hip: window 7 x 8 over a pre-padded NHWC4 bf16 image, stride 2, 64 channels -- against torch (float64) on the same
    bf16-rounded operands (all 8 x 4 window entries random: the kernel does not rely on the zero tap column / channel), output and
    BatchNorm sums; image counts that give one and several row ranges per image, and a last range that is short."""
    from gan_image_captioning_amd import engine
    L = _lib()
    lib = L.load()
    N, S = case
    g = torch.Generator().manual_seed(N * 1000 + S)
    x = torch.randn(N, 4, S, S, generator=g).bfloat16().float()
    w = (torch.randn(64, 4, 7, 8, generator=g) * 0.1).bfloat16().float()
    want = F.conv2d(x.double(), w.double(), None, 2, 0)
    Ho, Wo = want.shape[2], want.shape[3]
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()
    wp = w.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()                       # [64, 7, 8, 4]
    y = torch.empty(N, Ho, Wo, 64, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(8, 128, device=dev)
    L.check(lib.gic_conv2d(xh.data_ptr(), wp.data_ptr(), y.data_ptr(), stats.data_ptr(), 8, 1, N, S, S, 4, 64, 7, 8, 2, 0, engine.stream_ptr()), "conv2d")
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1.5e-2, f"{case}: rel max err {err}"
    rows = N * Ho * Wo
    close(stats.sum(0)[:64].cpu(), want.sum((0, 2, 3)), rtol=2e-3, atol_scale=1e-3 * rows ** 0.5, what="bn sum")
    close(stats.sum(0)[64:].cpu(), (want ** 2).sum((0, 2, 3)), rtol=2e-3, what="bn sumsq")


def test_trunk_forward_bf16_at_bench_resolution(dev, monkeypatch):
    """ResNet-50 at 224x224, 16 images: the grids of the benchmark's layers (1-, 2- and 4-stage rings, ring-less shallow-K launches,
    BatchNorm on load in every conv3, per-layer replica counts) against the fp32 CPU restatement; graph replay equals eager launches;
    the running statistics move the same way."""
    from gan_image_captioning_amd.trunk import ResNetTrunk
    g = torch.Generator().manual_seed(18)
    tp = OE.make_trunk_params("resnet50", g)
    images = torch.randn(16, 3, 224, 224, generator=g)
    running = {}                                                               # fresh buffers: mean 0, variance 1
    for _n, _ci, co, _k, _s, _p in OE.layer_specs("resnet50"):
        b = "encoder.resnet." + OE.bn_name(_n)
        running[b + ".running_mean"] = torch.zeros(co)
        running[b + ".running_var"] = torch.ones(co)
    want = OE.trunk_forward(tp, images, "resnet50", training=True, running=running)
    trunk = ResNetTrunk("resnet50")
    trunk.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
    trunk = trunk.to(dev).train()
    x = images.to(dev)
    f1 = trunk(x, 1).float().clone()          # eager (first call)
    f2 = trunk(x, 1).float().clone()          # captures the hipGraph and replays it
    f3 = trunk(x, 1).float().clone()          # replay
    torch.cuda.synchronize()
    err = rel_l2(f1, want)
    print(f"bf16 resnet50 @224 trunk rel-L2 error vs fp32 oracle: {err:.3e}")
    assert err < 5e-2
    assert rel_l2(f2, f1) < 2e-2 and rel_l2(f3, f1) < 2e-2        # BatchNorm sums are f32 atomics: not bit-reproducible
    plan = trunk._plan
    assert sum(1 for blk in plan.blocks if blk["c3"].fused_in) == 16 and len(plan._graphs) == 1
    sd = trunk.state_dict()
    assert int(sd["1.num_batches_tracked"]) == 3
    # three identical batches with momentum 0.1 from (mean 0, var 1): running = (1 - 0.9^3) * batch statistic (+ 0.9^3 for the variance)
    got_m, got_v = sd["4.0.bn1.running_mean"].float().cpu(), sd["4.0.bn1.running_var"].float().cpu()
    bm = running["encoder.resnet.4.0.bn1.running_mean"] / 0.1                      # the oracle updated fresh buffers once
    bv = (running["encoder.resnet.4.0.bn1.running_var"] - 0.9) / 0.1
    assert rel_l2(got_m, bm * (1 - 0.9 ** 3)) < 2e-2
    assert rel_l2(got_v, bv * (1 - 0.9 ** 3) + 0.9 ** 3) < 2e-2


def test_trunk_forward_bf16_is_explained_by_bf16_storage(dev):
    """What the per-stage bf16 budgets of test_cfg2_composed_step_bf16_vs_oracle (up to 1.4e-1 at stage 3 against the fp32 oracle)
    consist of.  ResNet-50 at 224x224, 16 images, two initialisations: the reference's U(-0.05, 0.05) of every tensor
    (generator.py:116-123) and the published ResNet init (Kaiming convolutions, gamma 1, beta 0: what torchvision hands the reference
    before init_params overwrites it).  The HIP trunk is compared with (a) the fp32 oracle and (b) the SAME oracle with every tensor
    that the bf16 mode stores rounded to bf16 where it is stored (oracle/cpu_encoder.trunk_forward(emulate_bf16=True): f32 products,
    accumulation, statistics and affine maps).  Finding (round 3): the deviation from fp32 is a property of bf16 STORAGE of this
    untrained network, not of the kernels -- it is larger, not smaller, under the Kaiming init (the residual stream of an untrained
    ResNet-50 with gamma 1 grows a per-channel mean many times its spread, and subtracting that mean in the next BatchNorm costs the
    same factor in relative precision: 5e-1 at stage 3, measured) -- while against the storage-emulating oracle the kernels agree
    to within the chaotic amplification of summation-order differences.  Asserted: the emulated oracle explains the deviation (GPU
    vs emulation well below GPU vs fp32 wherever the latter is large) and stays within the written limits at every stage."""
    from gan_image_captioning_amd.trunk import ResNetTrunk
    N, S = 16, 224
    report = {}
    for init in ("kaiming", "uniform"):
        g = torch.Generator().manual_seed(2024)
        tp = OE.make_trunk_params("resnet50", g, init=init)
        images = torch.randn(N, 3, S, S, generator=g)
        taps, taps16 = {}, {}
        want = OE.trunk_forward(tp, images, "resnet50", taps=taps)
        want16 = OE.trunk_forward(tp, images, "resnet50", taps=taps16, emulate_bf16=True)
        trunk = ResNetTrunk("resnet50")
        trunk.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
        trunk = trunk.to(dev).train()
        feat = trunk(images.to(dev), 1).float().clone()
        torch.cuda.synchronize()
        pb = trunk._plan._bufs[(N, S)]
        last, i = [], -1
        for stage in trunk.stages():
            i += len(stage)
            last.append(i)
        got = {"stem": pb["x0"].float().permute(0, 3, 1, 2)}
        for si, bi in enumerate(last):
            got[f"stage{si}"] = pb["blocks"][bi]["out"].float().permute(0, 3, 1, 2)
        errs = {k: {"vs_fp32": rel_l2(v, taps[k]), "vs_bf16_storage": rel_l2(v, taps16[k]), "storage_vs_fp32": rel_l2(taps16[k], taps[k])}
                for k, v in got.items()}
        errs["pooled"] = {"vs_fp32": rel_l2(feat, want), "vs_bf16_storage": rel_l2(feat, want16), "storage_vs_fp32": rel_l2(want16, want)}
        report[init] = errs
    import json
    print("bf16 trunk rel-L2 per stage:", json.dumps(report))
    try:
        import os
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "trunk_bf16_storage.json"), "w") as fh:
            json.dump(report, fh, indent=1)
    except OSError:
        pass
    for init, errs in report.items():
        for k, e in errs.items():
            # the storage-emulating oracle reproduces the size of the deviation from fp32 ...
            assert 0.4 * e["vs_fp32"] <= e["storage_vs_fp32"] <= 2.5 * e["vs_fp32"] or e["vs_fp32"] < 5e-3, (init, k, e)
            # ... and the kernels stay close to it
            assert e["vs_bf16_storage"] < LIMIT_VS_STORAGE[init][k], (init, k, e)


# GPU trunk vs the bf16-storage-emulating oracle, relative L2 per stage (see the test above)
# measured (profiles/r03_trunk_bf16_storage.json): uniform 0 / 2.4e-3 / 1.2e-2 / 3.1e-2 / 5.6e-2, pooled 9.7e-3; kaiming 0 / 5.0e-3 / 2.4e-2 /
# 1.1e-1 / 2.7e-1, pooled 5.5e-2 -- the stem agrees bit for bit; deeper stages carry the amplification of f32 summation-order differences
LIMIT_VS_STORAGE = {"uniform": {"stem": 1e-4, "stage0": 5e-3, "stage1": 2.5e-2, "stage2": 6e-2, "stage3": 1e-1, "pooled": 2e-2},
                    "kaiming": {"stem": 1e-4, "stage0": 1e-2, "stage1": 5e-2, "stage2": 2e-1, "stage3": 4.5e-1, "pooled": 1e-1}}


@pytest.mark.parametrize("arch,S,N", [("resnet18", 64, 4), ("resnet50", 64, 2), ("resnet18", 96, 2)])
def test_trunk_forward_f32_matches_oracle(dev, arch, S, N):
    from gan_image_captioning_amd.trunk import ResNetTrunk
    g = torch.Generator().manual_seed(7)
    tp = OE.make_trunk_params(arch, g)
    images = torch.randn(N, 3, S, S, generator=g)
    running = {}
    for name, _ci, co, *_ in OE.layer_specs(arch):
        b = "encoder.resnet." + OE.bn_name(name)
        running[b + ".running_mean"] = torch.zeros(co)
        running[b + ".running_var"] = torch.ones(co)
    taps = {}
    want = OE.trunk_forward(tp, images, arch, training=True, running=running, taps=taps)
    trunk = ResNetTrunk(arch)
    trunk.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
    trunk = trunk.to(dev).train()
    feat = trunk(images.to(dev), 0)
    torch.cuda.synchronize()
    close(feat, want, rtol=2e-3, atol_scale=2e-3, what=f"{arch} features")
    # intermediate activation after the stem and the first stage
    plan = trunk._plan
    b = plan._bufs[(N, S)]
    close(b["x0"].permute(0, 3, 1, 2), taps["stem"], rtol=1e-3, atol_scale=1e-4, what="stem output")
    nb0 = len(trunk.stages()[0])
    close(b["blocks"][nb0 - 1]["out"].permute(0, 3, 1, 2), taps["stage0"], rtol=1e-3, atol_scale=5e-4, what="stage0 output")
    # running statistics (side effect of train-mode BatchNorm)
    sd = trunk.state_dict()
    close(sd["1.running_mean"], running["encoder.resnet.1.running_mean"], rtol=1e-3, atol_scale=1e-4, what="stem running_mean")
    close(sd["1.running_var"], running["encoder.resnet.1.running_var"], rtol=1e-3, atol_scale=1e-4, what="stem running_var")
    assert int(sd["1.num_batches_tracked"]) == 1
    # eval mode uses the running statistics
    trunk.eval()
    want_eval = OE.trunk_forward(tp, images, arch, training=False, running=running)
    feat_eval = trunk(images.to(dev), 0)
    torch.cuda.synchronize()
    close(feat_eval, want_eval, rtol=2e-3, atol_scale=2e-3, what="eval features")


def test_trunk_forward_bf16_close(dev):
    from gan_image_captioning_amd.trunk import ResNetTrunk
    g = torch.Generator().manual_seed(8)
    tp = OE.make_trunk_params("resnet50", g)
    images = torch.randn(4, 3, 64, 64, generator=g)
    want = OE.trunk_forward(tp, images, "resnet50")
    trunk = ResNetTrunk("resnet50")
    trunk.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
    trunk = trunk.to(dev).train()
    feat = trunk(images.to(dev), 1)
    torch.cuda.synchronize()
    err = rel_l2(feat.float(), want)
    print(f"bf16 resnet50 trunk rel-L2 error vs fp32 oracle: {err:.3e}")
    assert err < 8e-2      # 53 stacked BatchNorms over only N*2*2 = 16 samples in the last stage amplify bf16 rounding


def test_encoder_head_matches_reference_composition(dev):
    """Linear + BatchNorm1d(momentum=0.01) forward/backward vs the golden vectors (reference's own nn modules)."""
    from gan_image_captioning_amd import encoder_engine
    g = Golden("tiny_cgan_head")
    gp = g.group("gp0/")
    tf = g.t("trunk_feat")
    leaf = {k: gp[k].clone().requires_grad_(True) for k in ("encoder.linear.weight", "encoder.linear.bias", "encoder.bn.weight", "encoder.bn.bias")}
    run = {"running_mean": torch.zeros(g.meta["E"]), "running_var": torch.ones(g.meta["E"])}
    want = O.encoder_head(leaf, tf, training=True, running=run)
    rg = torch.Generator().manual_seed(1)
    d_out = torch.randn(want.shape, generator=rg)
    (want * d_out).sum().backward()
    rm, rv = torch.zeros(g.meta["E"], device=dev), torch.ones(g.meta["E"], device=dev)
    out, saved = encoder_engine.head_fwd(0, tf.to(dev), gp["encoder.linear.weight"].to(dev), gp["encoder.linear.bias"].to(dev),
                                         gp["encoder.bn.weight"].to(dev), gp["encoder.bn.bias"].to(dev), rm, rv, True, 0.01, 1e-5)
    dw, db, dg, dbt = encoder_engine.head_bwd(0, saved, gp["encoder.linear.weight"].to(dev), gp["encoder.bn.weight"].to(dev), d_out.to(dev))
    torch.cuda.synchronize()
    close(out, want, rtol=1e-4, atol_scale=1e-5, what="head out")
    close(rm, run["running_mean"], rtol=1e-4, atol_scale=1e-6, what="running_mean")
    close(rv, run["running_var"], rtol=1e-4, atol_scale=1e-6, what="running_var")
    close(dw, leaf["encoder.linear.weight"].grad, rtol=2e-3, atol_scale=1e-4, what="d linear.weight")
    close(db, leaf["encoder.linear.bias"].grad, rtol=2e-3, atol_scale=1e-4, what="d linear.bias", atol_abs=1e-6)
    close(dg, leaf["encoder.bn.weight"].grad, rtol=2e-3, atol_scale=1e-4, what="d bn.weight")
    close(dbt, leaf["encoder.bn.bias"].grad, rtol=2e-3, atol_scale=1e-4, what="d bn.bias")


@pytest.mark.parametrize("impl", ["fused", "autograd"])
def test_conditional_step_f32_matches_oracle(dev, impl):
    """--conditional-gan 1, ResNet-18 @64x64, batch 8 (SURVEY cfg1): the whole step against the CPU oracle."""
    from tests.test_gpu_step import make_instructor
    from tests.gpu_util import dec_param_names, disc_param_names
    m = dict(B=8, L=10, V=64, E=32, H=512, NL=1, De=64, R=64, fs=[3, 4, 5], nf=[300, 300, 300], loss="standard", clip=5.0,
             gen_lr=1e-4, disc_lr=1e-4, T0=100, adapt="exp", adv_epochs=30)
    g = torch.Generator().manual_seed(2024)
    gp = O.make_gen_params(m["V"], m["E"], m["H"], m["NL"], g, trunk_feat_dim=512)
    dp = O.make_disc_params(m["V"], g)
    tp = OE.make_trunk_params("resnet18", g)
    caps = O.make_captions(m["B"], m["L"], m["V"], g)
    images = torch.randn(m["B"], 3, 64, 64, generator=g)
    us, masks = O.make_noise(m["B"], m["L"], m["V"], 900, 64, g)
    T = 1.2
    feat = OE.trunk_forward(tp, images, "resnet18")
    ref = O.adv_step(dict(gp), dict(dp), caps, us, masks, T, "standard", 5.0, None, None, trunk_feat=feat)
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    args = default_args(vocab_size=m["V"], gen_embed_dim=m["E"], gen_hidden_dim=m["H"], conditional_gan=1, encoder_arch="resnet18",
                        compute_dtype="fp32", step_impl=impl, device="cuda", log_file=None, model_dir=None, save_dir=None, image_size=64)
    inst = GANInstructor(args, None, None)
    with torch.no_grad():
        for n, p in zip(dec_param_names(1), inst.gen.decoder.param_list()):
            p.copy_(gp[n])
        for n, p in zip(disc_param_names(3), inst.disc.param_list()):
            p.copy_(dp[n])
        inst.gen.encoder.resnet.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
        for n in ("linear.weight", "linear.bias", "bn.weight", "bn.bias"):
            mod, attr = n.split(".")
            getattr(getattr(inst.gen.encoder, mod), attr).copy_(gp["encoder." + n])
    inst.gen.train(); inst.disc.train()
    inst.gen.decoder.temperature = T
    u = torch.stack(us).to(dev)
    km = [k.to(dev) for k in masks]
    if impl == "fused":
        out = inst.fused(images.to(dev), caps.to(dev), m["L"], True, u, km, opt_step=False)
        losses = out["losses"]
        torch.cuda.synchronize()
        assert torch.equal(out["ids"].cpu(), ref["ids"])
    else:
        from gan_image_captioning_amd.utils import get_losses
        feats = inst._features(images.to(dev), m["B"])
        gen_caps, ids = inst.gen.decoder.sample(feats, max_caption_len=m["L"], noise_u=u)
        d_real = inst.disc(caps.to(dev), keep_mask=km[0])
        d_fake = inst.disc(gen_caps.detach(), keep_mask=km[1])
        with inst.disc.input_grad_only():
            g_out = inst.disc(gen_caps, keep_mask=km[2])
        g_loss, d_loss = get_losses(d_real, d_fake, g_out, "standard", detach_d_for_g=True)
        inst.disc_opt.zero_grad(); inst.gen_opt.zero_grad()
        d_loss.backward(); g_loss.backward()
        losses = torch.stack([g_loss.detach(), d_loss.detach()])
        torch.cuda.synchronize()
        assert torch.equal(ids.cpu(), ref["ids"])
    assert float(losses[0]) == pytest.approx(ref["g_loss"], rel=1e-4)
    assert float(losses[1]) == pytest.approx(ref["d_loss"], rel=1e-4)
    want = {**ref["g_grads_raw"], **ref["d_grads_raw"]}
    enc = inst.gen.encoder
    got = {"encoder.linear.weight": enc.linear.weight.grad, "encoder.linear.bias": enc.linear.bias.grad,
           "encoder.bn.weight": enc.bn.weight.grad, "encoder.bn.bias": enc.bn.bias.grad}
    got.update({n: p.grad for n, p in zip(dec_param_names(1), inst.gen.decoder.param_list())})
    for n in ("encoder.linear.weight", "encoder.bn.weight", "encoder.bn.bias", "decoder.lstm.weight_ih_l0", "decoder.linear.weight"):
        err = rel_l2(got[n], want[n])
        assert err < 2e-2, f"{n}: rel L2 {err}"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_trunk_prefetch_is_equivalent(dev, dtype):
    """The next batch's trunk forward enqueued under the current step (FusedAdvStep next_images=) changes nothing:
    three steps on three different image batches, with and without the look-ahead, incl. a mispredicted look-ahead."""
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    kw = dict(vocab_size=64, gen_embed_dim=32, gen_hidden_dim=64, conditional_gan=1, encoder_arch="resnet18", compute_dtype=dtype,
              device="cuda", log_file=None, model_dir=None, save_dir=None, image_size=32)
    a = GANInstructor(default_args(**kw), None, None)
    b = GANInstructor(default_args(**kw), None, None)
    b.gen.load_state_dict(a.gen.state_dict())
    b.disc.load_state_dict(a.disc.state_dict())
    g = torch.Generator().manual_seed(5)
    B, L, V = 8, 6, 64
    imgs = [torch.randn(B, 3, 32, 32, generator=g).to(dev) for _ in range(3)]
    caps = O.make_captions(B, L, V, g).to(dev)
    us, masks = O.make_noise(B, L, V, 900, 64, g)
    u = torch.stack(us).to(dev)
    km = [k.to(dev) for k in masks]
    stray = torch.randn(B, 3, 32, 32, generator=g).to(dev)
    outs = {}
    for name, inst, nxt in (("plain", a, [None, None, None]), ("prefetch", b, [imgs[1], stray, None])):   # step 1 mispredicts
        inst.gen.train(); inst.disc.train()
        rows = []
        for k in range(3):
            o = inst.fused(imgs[k], caps, L, True, u, km, next_images=nxt[k])
            rows.append((o["losses"].clone(), o["ids"].clone()))
        torch.cuda.synchronize()
        outs[name] = (rows, inst.gen_arena.flat.clone(), inst.disc_arena.flat.clone(),
                      int(inst.gen.encoder.resnet.state_dict()["1.num_batches_tracked"]))
    tol = dict(rtol=2e-4, atol=1e-6) if dtype == "fp32" else dict(rtol=3e-2, atol=1e-4)
    for (l0, i0), (l1, i1) in zip(outs["plain"][0], outs["prefetch"][0]):
        if dtype == "fp32":
            assert torch.equal(i0, i1)
        torch.testing.assert_close(l0, l1, **tol)
    # trunk BatchNorm running statistics saw the same three batches in the same order (+ the stray one, once)
    assert outs["plain"][3] == 3 and outs["prefetch"][3] == 4
    if dtype == "fp32":
        torch.testing.assert_close(outs["plain"][2], outs["prefetch"][2], rtol=1e-3, atol=1e-5)


def test_block_output_formed_on_load_equals_the_separate_pass(dev):
    """gic_conv1x1_res_in (the next block's conv1 forms relu(bn3(y3) + shortcut) on load and writes the block output from its first
    N tile) against the separate gic_bn_act pass + plain convolution: ResNet-50 trunk, bf16, every block output, the pooled feature
    and the running statistics agree to bf16 rounding; 15 of the 16 block outputs ride in a convolution."""
    from gan_image_captioning_amd.trunk import ResNetTrunk
    g = torch.Generator().manual_seed(21)
    tp = OE.make_trunk_params("resnet50", g)
    images = torch.randn(8, 3, 128, 128, generator=g).to(dev)
    outs = {}
    for fused in (True, False):
        trunk = ResNetTrunk("resnet50")
        trunk.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
        trunk = trunk.to(dev).train()
        trunk(images, 1)                                   # builds the plan
        plan = trunk._plan
        plan.fuse_res = fused
        plan.res_min_rows = 0                              # every eligible block (the production plan fuses the large grids with a
        plan.res_max_cout = 1 << 20                        # single output-channel tile only)
        plan.use_graph = False
        for s in plan.steps:
            s.fused_in = None                              # re-probe
        feat = trunk(images, 1).float().clone()
        torch.cuda.synchronize()
        blocks = [e["out"].float().clone() for e in plan._bufs[(8, 128)]["blocks"]]
        nfused = sum(1 for blk in plan.blocks if blk["c1"].fused_in)
        outs[fused] = (feat, blocks, nfused, trunk.state_dict()["7.2.bn3.running_var"].float().clone())
    assert outs[True][2] == 15 and outs[False][2] == 0, (outs[True][2], outs[False][2])
    # both routes round the same quantities to bf16, but every BatchNorm sum is an f32 atomic accumulation (order varies) and the
    # stack amplifies such differences stage by stage (see test_cfg2_composed_step_bf16_vs_oracle; here the last stage normalises over
    # 128 rows only): measured over repeated runs 5e-5..5e-3 on the first three blocks and 5.1e-2..6.1e-2 on the last one
    errs = [rel_l2(a, b_) for a, b_ in zip(outs[True][1], outs[False][1])]
    report = " ".join(f"{e:.2e}" for e in errs) + f" | pooled {rel_l2(outs[True][0], outs[False][0]):.2e} running_var {rel_l2(outs[True][3], outs[False][3]):.2e}"
    print("block outputs rel L2:", report)
    for i, e in enumerate(errs):
        assert e < (1e-2 if i < 3 else 1e-1), f"block {i} output: {report}"
    assert rel_l2(outs[True][0], outs[False][0]) < 3e-2, report
    assert rel_l2(outs[True][3], outs[False][3]) < 5e-2, report


def test_conv3_recomputed_inside_the_next_conv1_equals_the_separate_launches(dev):
    """gic_conv1x1_bn_in_stats + gic_conv_b2b (conv3 as a statistics-only pass, then recomputed inside the launch that forms the block
    output and runs the next block's conv1: its output never reaches memory) against the separate launches (conv3 writes y3, the next
    conv1 forms the block output on load): ResNet-50 at the BASELINE shape (64 images, 224 x 224, bf16), seven block boundaries on the
    56 x 56 and 28 x 28 maps take the fused form; every block output, the pooled feature and the running statistics agree to bf16
    rounding (the fused form never rounds y3 to bf16: it is the more accurate of the two)."""
    from gan_image_captioning_amd.trunk import ResNetTrunk
    g = torch.Generator().manual_seed(23)
    tp = OE.make_trunk_params("resnet50", g)
    images = torch.randn(64, 3, 224, 224, generator=g).to(dev)
    outs = {}
    for fused in (True, False):
        trunk = ResNetTrunk("resnet50")
        trunk.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
        trunk = trunk.to(dev).train()
        trunk(images, 1)                                   # builds the plan
        plan = trunk._plan
        plan.fuse_b2b = fused
        plan.use_graph = False
        for blk in plan.blocks:
            blk.pop("b2b", None); blk.pop("b2b_refused", None)   # re-probe
        feat = trunk(images, 1).float().clone()
        torch.cuda.synchronize()
        blocks = [e["out"].float().clone() for e in plan._bufs[(64, 224)]["blocks"]]
        y1_first = plan._bufs[(64, 224)]["blocks"][1]["y1"].float().clone()     # conv1 of the second block: the first fused launch's other output
        y1_last = plan._bufs[(64, 224)]["blocks"][7]["y1"].float().clone()      # conv1 of block 6.0: the last one's (C2 = 128 into 256 channels)
        nb2b = sum(1 for blk in plan.blocks if blk.get("b2b") is True)
        sd = trunk.state_dict()
        outs[fused] = (feat, blocks, nb2b, sd["5.1.bn1.running_var"].float().clone(), sd["7.2.bn3.running_var"].float().clone(), y1_first, y1_last)
    assert outs[True][2] == 7 and outs[False][2] == 0, (outs[True][2], outs[False][2])
    errs = [rel_l2(a, b_) for a, b_ in zip(outs[True][1], outs[False][1])]
    report = " ".join(f"{e:.2e}" for e in errs) + (f" | pooled {rel_l2(outs[True][0], outs[False][0]):.2e} running_var "
                                                    f"{rel_l2(outs[True][3], outs[False][3]):.2e} {rel_l2(outs[True][4], outs[False][4]):.2e}")
    print("block outputs rel L2:", report)
    for i, e in enumerate(errs):
        assert e < (2e-2 if i < 7 else 1e-1), f"block {i} output: {report}"
    # ... and element by element: a 16-byte store whose data registers are overwritten one instruction later can put a stray value into
    # two bytes of a few rows -- invisible in an L2 norm (seen in conv1x1_pix.hip; conv_b2b_kernel<128, 256> had two such sites, which
    # a measurement build -DGIC_STORE_SOFF brings back: no stray element was observed there, the check stays as the guard)
    for what, k, a, b_ in (("first fused launch's block output", 0.25, outs[True][1][0], outs[False][1][0]),
                           ("first fused launch's conv1 output", 0.25, outs[True][5], outs[False][5]),
                           ("last fused launch's conv1 output", 0.5, outs[True][6], outs[False][6])):
        scale = float(b_.abs().mean())
        stray = int(((a - b_).abs() > k * scale + 0.05 * b_.abs()).sum())
        assert stray == 0, f"{stray} stray elements in the {what} (max |diff| {float((a - b_).abs().max()):.3f}, mean |value| {scale:.3f})"
    assert rel_l2(outs[True][0], outs[False][0]) < 3e-2, report
    assert rel_l2(outs[True][3], outs[False][3]) < 1e-2 and rel_l2(outs[True][4], outs[False][4]) < 5e-2, report


@pytest.mark.parametrize("N,expect", [(17, 0), (20, 0), (24, 3)])
def test_back_to_back_launches_engage_only_where_their_kernels_take_the_shapes(dev, N, expect):
    """Batch sizes around the fused form's conditions (rows a multiple of 128, enough row tiles for the statistics-only pass, >= 50 000
    rows): 17 images (53 312 rows at 56 x 56: not a multiple of 128) and 20 (62 720 rows: 980 tiles, below the streaming kernel's
    minimum) stay on the separate launches, 24 fuses the three 56 x 56 boundaries only (18 816 rows at 28 x 28); every plan runs and
    agrees with the unfused plan."""
    from gan_image_captioning_amd.trunk import ResNetTrunk
    g = torch.Generator().manual_seed(29)
    tp = OE.make_trunk_params("resnet50", g)
    images = torch.randn(N, 3, 224, 224, generator=g).to(dev)
    feats = {}
    for fused in (True, False):
        trunk = ResNetTrunk("resnet50")
        trunk.load_state_dict({k[len("encoder.resnet."):]: v for k, v in tp.items()}, strict=False)
        trunk = trunk.to(dev).train()
        trunk(images, 1)
        plan = trunk._plan
        plan.fuse_b2b = fused
        plan.use_graph = False
        for blk in plan.blocks:
            blk.pop("b2b", None); blk.pop("b2b_refused", None)
        feats[fused] = trunk(images, 1).float().clone()
        torch.cuda.synchronize()
        if fused:
            assert len(plan.unstored_convs()) == expect, (N, plan.unstored_convs())
    assert torch.isfinite(feats[True]).all()
    assert rel_l2(feats[True], feats[False]) < 3e-2


def test_cold_and_mispredicted_trunk_passes_do_not_race_the_lookahead(dev):
    """A synchronous trunk pass (the first batch of a loop, or the pass after a mispredicted look-ahead) shares the plan's buffers
    (packed image, statistics arena, activations, pooled output) with the look-ahead pass that is enqueued right behind it on
    another stream: the look-ahead must start after the synchronous pass has finished, and the synchronous pass hands out a
    private copy of its features.  ResNet-50 at 224x224, 32 images: a pass takes ~1 ms, far longer than enqueueing the next one."""
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.generator import Generator
    args = default_args(vocab_size=64, gen_embed_dim=32, gen_hidden_dim=64, conditional_gan=1, encoder_arch="resnet50", compute_dtype="bf16",
                        device="cuda", log_file=None, model_dir=None, save_dir=None, image_size=224)
    enc = Generator(args).to(dev).encoder.train()
    g = torch.Generator().manual_seed(31)
    A, Bm, Cm = (torch.randn(32, 3, 224, 224, generator=g).to(dev) for _ in range(3))
    ref = {}
    for _ in range(3):                                   # warm-up: eager pass, graph capture, replay (host enqueue then takes microseconds)
        for name, x in (("A", A), ("B", Bm)):
            ref[name] = enc.trunk_features(x, True).float().clone()
            torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)
    for rep in range(3):
        start = main.record_event()
        tA = enc.take_trunk(A, True, main)               # cold: nothing was announced -> synchronous pass on main
        enc.prefetch_trunk(Bm, True, start)              # the step announces the next batch immediately
        tB = enc.take_trunk(Bm, True, main)              # ... and the next step picks it up
        start = main.record_event()
        enc.prefetch_trunk(Cm, True, start)              # mispredicted look-ahead
        tA2 = enc.take_trunk(A, True, main)              # synchronous pass behind the unused look-ahead
        enc.prefetch_trunk(Bm, True, start)              # and a new look-ahead right behind it
        tB2 = enc.take_trunk(Bm, True, main)
        torch.cuda.synchronize()
        for got, want in ((tA, "A"), (tB, "B"), (tA2, "A"), (tB2, "B")):
            err = rel_l2(got.float(), ref[want])
            assert err < 2e-2, f"rep {rep}: features of batch {want} corrupted (rel L2 {err:.3e})"     # f32-atomic BatchNorm sums: not bit-exact
        assert tA.data_ptr() != tA2.data_ptr() and tA.data_ptr() != enc.resnet._plan._bufs[(32, 224)]["feat"].data_ptr()


@pytest.mark.parametrize("N,H,Ci,Co", [(64, 14, 256, 1024), (33, 14, 256, 1024), (5, 14, 256, 1024), (64, 7, 512, 2048), (9, 7, 512, 2048)])
def test_pixel_resident_conv3_every_element(dev, N, H, Ci, Co):
    """conv1x1_pix.hip (K = 256 into 1024 channels on the 14 x 14 map, K = 512 into 2048 on the 7 x 7 one; BatchNorm + ReLU of the
    input on load, pixels in registers, 32-byte stores from the accumulators) against a float matmul, EVERY element to bf16 rounding,
    and its column sums (through the turning scratch / the DPP butterfly): full row tiles, ragged last tiles.  The first version passed every norm-based check with 1e-4 of its elements replaced by
    stray values (a VALU instruction overwrote a store's data registers one instruction behind it: DESIGN.md section 4)."""
    from gan_image_captioning_amd import engine
    L = _lib()
    lib = L.load()
    rows = N * H * H
    g = torch.Generator().manual_seed(N + H)
    y = (torch.randn(rows, Ci, generator=g) * 1.5 + 0.3).to(dev).bfloat16()
    w = (torch.randn(Co, Ci, generator=g) * 0.05).to(dev).bfloat16()
    gamma = (torch.rand(Ci, generator=g) + 0.5).to(dev)
    beta = (torch.randn(Ci, generator=g) * 0.2).to(dev)
    yf = y.float()
    nrep = 4
    in_stats = torch.zeros(nrep, 2 * Ci, device=dev)
    in_stats[3, :Ci] = yf.sum(0)
    in_stats[0, Ci:] = (yf * yf).sum(0)
    mean = yf.mean(0)
    var = ((yf * yf).mean(0) - mean * mean).clamp_min(0)
    sc = gamma * torch.rsqrt(var + 1e-5)
    z = torch.relu(yf * sc + (beta - mean * sc)).bfloat16().float()       # the operand the MFMAs see
    ref = z @ w.float().t()
    out = torch.full((rows, Co), 7.0, device=dev, dtype=torch.bfloat16)
    st = torch.zeros(nrep, 2 * Co, device=dev)
    L.check(lib.gic_conv2d_bn_in(y.data_ptr(), in_stats.data_ptr(), nrep, gamma.data_ptr(), beta.data_ptr(), float(rows), w.data_ptr(),
                                 out.data_ptr(), st.data_ptr(), nrep, 1, N, H, H, Ci, Co, 1, 1, 1, 0, engine.stream_ptr()), "conv2d_bn_in")
    torch.cuda.synchronize()
    o = out.float()
    bad = (o - ref).abs() > 2e-2 + 1.5e-2 * ref.abs()                     # bf16 rounding of the output + of a few normalised inputs
    assert int(bad.sum()) == 0, (int(bad.sum()), bad.any(1).nonzero().flatten()[:8].tolist(), bad.any(0).nonzero().flatten()[:8].tolist())
    torch.testing.assert_close(st.sum(0)[:Co], ref.sum(0), rtol=2e-3, atol=2e-2 * rows ** 0.5)
    torch.testing.assert_close(st.sum(0)[Co:], (ref * ref).sum(0), rtol=2e-3, atol=2e-2 * rows ** 0.5)


@pytest.mark.parametrize("case", [(8, 28, 128, 512, 1, 1, 0), (4, 56, 64, 256, 1, 1, 0), (16, 14, 256, 1024, 1, 1, 0), (2, 7, 512, 2048, 1, 1, 0),
                                  (3, 9, 72, 40, 1, 1, 0), (4, 28, 128, 128, 3, 1, 1), (4, 28, 128, 128, 3, 2, 1), (8, 56, 64, 64, 3, 1, 1),
                                  (16, 14, 256, 256, 3, 1, 1), (8, 7, 512, 512, 3, 1, 1), (2, 12, 192, 96, 3, 1, 1),
                                  (40, 56, 64, 256, 1, 1, 0), (44, 28, 128, 512, 1, 1, 0), (41, 56, 64, 96, 1, 1, 0),
                                  (5, 14, 256, 1024, 1, 1, 0), (64, 14, 256, 1024, 1, 1, 0), (33, 14, 256, 1024, 1, 1, 0)])
def test_conv_with_input_batchnorm_equals_bn_act_then_conv(dev, case):
    """gic_conv2d_bn_in (bn + ReLU applied to the A tiles in LDS, padding taps left at zero) against gic_bn_act followed by
    gic_conv2d on the same raw tensor and statistics: same bf16 input to the MFMAs, so outputs and column sums agree to rounding."""
    from gan_image_captioning_amd import engine
    L = _lib()
    lib = L.load()
    N, H, Ci, Co, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    rows = N * H * H
    Ho = (H + 2 * pad - k) // stride + 1
    rows_out = N * Ho * Ho
    y_prev = (torch.randn(rows, Ci, generator=g) * 1.5 + 0.3).to(dev).bfloat16()
    gamma = (torch.rand(Ci, generator=g) + 0.5).to(dev)
    beta = (torch.randn(Ci, generator=g) * 0.2).to(dev)
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.05).to(dev).bfloat16()          # packed [Cout, KH, KW, Cin]
    nrep = 4
    yf = y_prev.float()
    in_stats = torch.zeros(nrep, 2 * Ci, device=dev)
    in_stats[1, :Ci] = yf.sum(0)
    in_stats[2, Ci:] = (yf * yf).sum(0)                 # replicas are summed by the readers
    s = engine.stream_ptr()
    if Ci & (Ci - 1) == 0:                              # reference route: bn_act -> conv2d
        z = torch.empty_like(y_prev)
        L.check(lib.gic_bn_act(y_prev.data_ptr(), in_stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, None, None, None, None,
                               None, None, nrep, float(rows), 1, z.data_ptr(), 1, rows, Ci, s), "bn_act")
    else:                                               # bn_act wants a power-of-two C: normalise on the host for the odd case
        mean = in_stats.sum(0)[:Ci] / rows
        var = (in_stats.sum(0)[Ci:] / rows - mean * mean).clamp_min(0)
        sc = gamma * torch.rsqrt(var + 1e-5)
        z = torch.relu(yf * sc + (beta - mean * sc)).bfloat16()
    out_ref = torch.empty(rows_out, Co, device=dev, dtype=torch.bfloat16)
    st_ref = torch.zeros(nrep, 2 * Co, device=dev)
    L.check(lib.gic_conv2d(z.data_ptr(), w.data_ptr(), out_ref.data_ptr(), st_ref.data_ptr(), nrep, 1, N, H, H, Ci, Co, k, k, stride, pad, s),
            "conv2d")
    out = torch.empty_like(out_ref)
    st = torch.zeros_like(st_ref)
    args = (y_prev.data_ptr(), in_stats.data_ptr(), nrep, gamma.data_ptr(), beta.data_ptr(), float(rows), w.data_ptr(), out.data_ptr(),
            st.data_ptr(), nrep)
    status = lib.gic_conv2d_bn_in(*args, 1, N, H, H, Ci, Co, k, k, stride, pad, s)
    torch.cuda.synchronize()
    if (k == 1 and rows_out < 128) or (k > 1 and stride != 1):
        # below one tile of rows the 8-wave kernel declines; windows other than 1x1 are normalised on load by the patch-resident
        # kernel only (stride 1): the plan falls back to bn_act + convolution
        assert status == L.ERR_UNSUPPORTED
        return
    L.check(status, "conv2d_bn_in")
    err = float((out.float() - out_ref.float()).abs().max() / out_ref.float().abs().max())
    assert err < 1e-2, err
    torch.testing.assert_close(st.sum(0), st_ref.sum(0), rtol=2e-3, atol=2e-2 * rows_out ** 0.5)
    # f32 mode has no fused variant
    assert lib.gic_conv2d_bn_in(*args, 0, N, H, H, Ci, Co, k, k, stride, pad, s) == L.ERR_UNSUPPORTED
