#!/usr/bin/env python3
"""Headline benchmark: captions/sec of the full adversarial G+D train step (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]            # N=1 directly;
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W   # N>1: one rank per GPU over RCCL

A step = one pass of the hot path over one synthetic minibatch that is already resident
in HBM: (encoder fwd) + Decoder.sample + 3x Discriminator.forward + losses + D backward + G backward +
2x (clip + Adam) + temperature update.  Workload = BASELINE.json configs[1] ("cfg2"): batch 64 per GPU,
224x224 images, caption length 20, V=10000, E=H=512, 1 LSTM layer, bf16 MFMA operands with f32
accumulation / master weights / optimizer state.  Weak scaling: every rank processes its own 64 captions.

Prints ONE JSON line (rank 0) with the driver's contract plus:
  "roofline":     the dominant kernel replayed under HIP events on its launch stream
  "cpu_baseline": the CPU oracle (oracle/cpu_step.py, a port pinned to the reference's golden vectors)
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(B=64, S=224, L=20, V=10000, E=512, H=512, NL=1)
MFMA_BF16_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s HBM3E


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=None, help="captions per GPU per step (default: 64 for cfg2, 32 for cfg4 / cfg5)")
    p.add_argument("--cgan", type=int, default=None, help="1: image-conditional (encoder in the step); default: 1 if the encoder is built")
    p.add_argument("--encoder", default="resnet50", choices=["resnet18", "resnet50"])
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--step-impl", default="fused", choices=["fused", "autograd"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true", help="skip the roofline probe (profiling runs: only the timed steps' kernels)")
    p.add_argument("--cpu-steps", type=int, default=20, help="timed CPU-oracle steps (after one warm-up; BASELINE.md section 2: median of >= 20)")
    p.add_argument("--no-prefetch", action="store_true", help="do not hand the next batch's images to the step (no trunk prefetch)")
    p.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "cfg5"],
                   help="cfg2 (default, the headline: BASELINE configs[1..2], 64 captions/GPU); cfg4: visual-attention decoder, 32 captions/GPU; "
                        "cfg5: SeqGAN policy gradient + 16 Monte-Carlo roll-outs per prefix, 32 captions/GPU (BASELINE configs[3], [4]: "
                        "global batch 256 on 8 GPUs)")
    p.add_argument("--mc-rollouts", type=int, default=16)
    a = p.parse_args()
    if a.batch is None:                        # BASELINE configs[1..2]: 64 per GPU; configs[3], [4]: global batch 256 over 8 GPUs
        a.batch = CFG2["B"] if a.workload == "cfg2" else 32
    return a


def encoder_available() -> bool:
    try:
        from gan_image_captioning_amd import encoder_engine
        return bool(getattr(encoder_engine, "AVAILABLE", False))
    except Exception:
        return False


def build_instructor(a, cgan):
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    extra = {}
    if a.workload == "cfg4":
        extra = dict(decoder="attention", attn_dim=512)
    elif a.workload == "cfg5":
        extra = dict(adv_mode="seqgan", mc_rollouts=a.mc_rollouts)
    args = default_args(vocab_size=CFG2["V"], gen_embed_dim=CFG2["E"], gen_hidden_dim=CFG2["H"], gen_num_layers=CFG2["NL"],
                        conditional_gan=cgan, encoder_arch=a.encoder, compute_dtype=a.dtype, step_impl=a.step_impl,
                        adv_train_batch_size=a.batch, image_size=CFG2["S"], device="cuda", log_file=None, model_dir=None,
                        save_dir=None, **extra)
    torch.manual_seed(1008)                      # src/main.py:14
    inst = GANInstructor(args, None, None)
    inst.gen.train()
    inst.disc.train()
    return inst, args


def event_time_ms(fn, iters, stream):
    """Average duration of fn() over `iters` back-to-back launches, HIP events on the launch stream."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    start.record(stream)
    for _ in range(iters):
        fn()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / iters


def conv_times_in_step(inst, step, n_steps=4):
    """HIP events around EVERY trunk convolution launch, on its launch stream, while real train steps run: the trunk pass of the
    next batch is issued as eager launches (no hipGraph: graph nodes cannot carry per-kernel events) on the look-ahead stream
    under the step, as in the timed region.  Returns {layer name: [ms, ...]}."""
    plan = inst.gen.encoder.resnet._plan
    was = plan.use_graph
    plan.use_graph = False

    def collect(fn, n):
        for k in range(2):
            fn(k)
        torch.cuda.synchronize()
        plan.conv_trace = []
        for k in range(n):
            fn(k)
        torch.cuda.synchronize()
        trace, plan.conv_trace = plan.conv_trace, None
        res = {}
        for name, a, b in trace:
            res.setdefault(name, []).append(a.elapsed_time(b))
        return res

    in_step = collect(step, n_steps)
    # calibration: the same event brackets around the same launches with the chip otherwise idle.  bracket - back-to-back replay of
    # that layer (roofline_probe measures it) = what one bracket adds (event + dispatch latency); it is subtracted per layer there.
    enc = inst.gen.encoder
    N, S = inst.args.adv_train_batch_size, inst.args.image_size
    imgs = torch.randn(N, 3, S, S, device=inst.args.device)
    alone = collect(lambda k: enc.trunk_features(imgs, True), n_steps)
    plan.use_graph = was
    return {"in_step": in_step, "alone": alone}


def roofline_probe(inst, args, cgan, step=None):
    """The kernel family that dominates the step (rocprofv3 --stats, profiles/): in-step HIP-event timing + an isolated replay."""
    from gan_image_captioning_amd import engine
    stream = torch.cuda.current_stream()
    dev = args.device
    if cgan:
        from gan_image_captioning_amd import encoder_engine
        traffic = None
        name = None
        try:     # HBM bytes per launch: NOT measured by this run -- read from the committed rocprofv3 --pmc passes of this command
            # (the committed passes are of the default workload: cfg2, batch 64, ResNet-50; other workloads report no traffic figure)
            if args.adv_train_batch_size == CFG2["B"] and args.encoder_arch == "resnet50" and getattr(args, "decoder", "lstm") == "lstm" \
                    and getattr(args, "adv_mode", "relgan") == "relgan":
                name = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))[-1]      # the latest round's
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    traffic = json.load(fh)["conv_bnstats"]["hbm_bytes_per_launch"]
        except Exception:
            name = None
        in_step = conv_times_in_step(inst, step) if step is not None else None
        r = encoder_engine.roofline_probe(inst.gen.encoder, args, event_time_ms, MFMA_BF16_PEAK_TFLOPS, traffic, in_step)
        r["traffic_source"] = (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; "
                               "not re-measured by this run") if traffic is not None else None
        return r
    # discriminator highway GEMM: [B*64, 904] x [900, 904]^T, bf16 MFMA, fused gate+dropout epilogue is separate;
    # the plain GEMM of the same shape is the dominant launch without the encoder.
    den = inst.disc.engine()
    M, N, K = args.adv_train_batch_size * den.R, den.F, den.Fp
    A = torch.randn(M, K, device=dev).to(den.act)
    Bm = torch.randn(N, K, device=dev).to(den.act)
    C = torch.empty(M, N, device=dev, dtype=torch.float32)
    ms = event_time_ms(lambda: engine.gemm(A, Bm, C, M, N, K, K, K, N), 50, stream)
    flops = 2.0 * M * N * K
    achieved = flops / (ms * 1e-3) / 1e12
    return {"kernel": "gemm_kernel<bf16,f32,NT,64x64|128x128> highway [%d,%d,%d]" % (M, N, K), "bound": "mfma",
            "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "ms_per_launch": round(ms, 5)}


def usable_cores() -> int:
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:               # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
        else:
            n = min(n, 16)                                       # no visible quota: the documented share of a one-GPU box
    except (OSError, ValueError):
        n = min(n, 16)
    return n


def cpu_baseline(a, cgan):
    """The CPU oracle on a bounded sample of the same workload: `cpu_steps` full steps at the same batch, median; with the
    encoder in the step (the headline workload) and without it (the part the reference itself owns: torchvision's trunk is not in
    the reference tree).  cfg4 / cfg5 (no reference counterpart): the build-owned oracles of the same steps, fewer
    timed steps (they are not the driver's default; one cfg5 step with 16 roll-outs per prefix is ~2 minutes of CPU work)."""
    from oracle import cpu_step as O
    # BASELINE.md section 2 asks for all host cores; what this process may actually USE is the smaller of its affinity mask and its
    # cgroup CPU quota (a one-GPU box hands out a 16-core share of a many-core host: 128 threads on that share ran 100x slower)
    cores = usable_cores()
    torch.set_num_threads(cores)
    import statistics
    g = torch.Generator().manual_seed(1008)
    B, L, V = a.batch, CFG2["L"], CFG2["V"]
    feat_dim = None
    trunk_feat = None
    if cgan:
        from oracle import cpu_encoder as OE
        feat_dim = OE.out_features(a.encoder)
    dp = O.make_disc_params(V, g)
    caps = O.make_captions(B, L, V, g)
    gopt, dopt = O.AdamState(1e-4), O.AdamState(1e-4)
    if cgan:
        tp = OE.make_trunk_params(a.encoder, g)
        images = torch.randn(B, 3, CFG2["S"], CFG2["S"], generator=g)
    n_timed = a.cpu_steps
    if a.workload == "cfg4":
        from oracle import cpu_attention as OA
        n_timed = min(a.cpu_steps, 5)
        gp = OA.make_attn_params(V, CFG2["E"], CFG2["H"], feat_dim, 512, g)
        gp.update({k: v for k, v in O.make_gen_params(8, CFG2["E"], 8, 1, g, trunk_feat_dim=feat_dim).items() if k.startswith("encoder.")})
        us, masks = O.make_noise(B, L, V, 900, 64, g)
    elif a.workload == "cfg5":
        from oracle import cpu_seqgan as OS
        n_timed = 1
        gp = O.make_gen_params(V, CFG2["E"], CFG2["H"], CFG2["NL"], g, trunk_feat_dim=feat_dim)
        us = [torch.empty(B, V).uniform_(0, 1, generator=g) for _ in range(L)]
        masks = [torch.empty(B * 64, 900).bernoulli_(0.8, generator=g) for _ in range(2)]
    else:
        gp = O.make_gen_params(V, CFG2["E"], CFG2["H"], CFG2["NL"], g, trunk_feat_dim=feat_dim)
        us, masks = O.make_noise(B, L, V, 900, 64, g)
    times, times_noenc = [], []
    for i in range((0 if a.workload == "cfg5" else 1) + n_timed):
        log(f"cpu_baseline step {i}")
        t0 = time.perf_counter()
        taps = {}
        if cgan:
            with torch.no_grad():
                trunk_feat = OE.trunk_forward(tp, images, a.encoder, taps=taps)
        t1 = time.perf_counter()
        if a.workload == "cfg4":
            fmap = taps["stage3"].permute(0, 2, 3, 1).reshape(B, -1, feat_dim)
            OA.attn_adv_step(gp, dp, caps, us, masks, 1.5, trunk_feat, fmap, gen_opt=gopt, disc_opt=dopt)
        elif a.workload == "cfg5":
            umc = torch.empty(L, (L - 1) * a.mc_rollouts * B, V).uniform_(0, 1, generator=g)      # (drawing it is not timed)
            t0 += time.perf_counter() - t1
            t1 = time.perf_counter()
            OS.seqgan_step(gp, dp, caps, us, umc, a.mc_rollouts, masks, 5.0, gopt, dopt, trunk_feat=trunk_feat)
        else:
            O.adv_step(gp, dp, caps, us, masks, 1.5, "standard", 5.0, gopt, dopt, trunk_feat=trunk_feat)
        t2 = time.perf_counter()
        times.append(t2 - t0)
        times_noenc.append(t2 - t1)
    first = 0 if a.workload == "cfg5" else 1
    t = statistics.median(times[first:])
    tn = statistics.median(times_noenc[first:])
    oracle_name = {"cfg2": "oracle/cpu_step.py", "cfg4": "oracle/cpu_attention.py (attn_adv_step; no reference counterpart)",
                   "cfg5": "oracle/cpu_seqgan.py (no reference counterpart)"}[a.workload]
    out = {"value": round(B / t, 2), "unit": "captions/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"median of {n_timed} full step(s)" + (" (after 1 warm-up)" if first else " (no warm-up)")
                     + f" of the same batch={B} workload, fp32, {oracle_name}" + (" + oracle/cpu_encoder.py trunk" if cgan else ""),
           "ms_per_step": round(t * 1e3, 1)}
    if cgan:
        out["without_encoder"] = {"value": round(B / tn, 2), "ms_per_step": round(tn * 1e3, 1),
                                  "note": "the same steps minus the trunk forward: generator + discriminator + optimizers only "
                                          "(the part of the path the reference's own files define)"}
    return out


def self_launch(a) -> int:
    """`python bench.py --gpus N` outside torchrun: start one rank per GPU as a FRESH child process tree
    (`python -m torch.distributed.run ...`, rendezvous on 127.0.0.1) before this process touches the GPU, relay rank 0's JSON
    line and exit with the child's code.  No exec, nothing after GPU initialisation."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this host driver (RCCL across processes)
    log("launching " + " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus} (or without torchrun)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    import importlib
    importlib.import_module("gan_image_captioning_amd.build").build()
    cgan = a.cgan if a.cgan is not None else (1 if encoder_available() else 0)
    inst, args = build_instructor(a, cgan)
    dev = args.device
    from gan_image_captioning_amd.tasks import synthetic_batch
    images, captions, _lengths, L = synthetic_batch(a.batch, CFG2["V"], CFG2["S"], CFG2["L"], seed=1008 + rank, device=dev,
                                                    with_images=bool(cgan))
    n_batches, adv_epochs = 50, args.adv_epochs

    def step(k):
        inst.adv_step(images, captions, L, train=True, next_images=images if (cgan and not a.no_prefetch) else None)
        inst.update_temperature(0 + (k + 1) / n_batches, adv_epochs)      # training.py:183

    log(f"instructor ready (cgan={cgan}, dtype={a.dtype}); warm-up {a.warmup} steps")
    for k in range(a.warmup):
        step(k)
    torch.cuda.synchronize()
    log(f"timing {a.steps} steps")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k)
    t_enq = time.perf_counter() - t0           # host time to enqueue the K steps (the GPU runs behind it)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    log(f"host enqueue {t_enq / a.steps * 1e3:.3f} ms/step of {elapsed / a.steps * 1e3:.3f} ms/step")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    if rank != 0:
        return
    value = world * a.batch * a.steps / elapsed
    out = {
        "metric": "captions/sec (G+D train step)", "value": round(value, 2), "unit": "captions/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": "%s: adversarial G+D step%s, batch %d/GPU, 224x224 images, caption len 20, V=10000, E=H=512, "
                               "1-layer LSTM G, CNN-seq D (R=64,F=900), %s" % (
                                   a.workload, {"cfg2": "", "cfg4": " with the visual-attention decoder (7x7 map, A=512)",
                                                "cfg5": " SeqGAN-style (policy gradient, %d Monte-Carlo roll-outs per prefix)" % a.mc_rollouts}[a.workload],
                                   a.batch, ("ResNet-50-shaped" if a.encoder == "resnet50" else "ResNet-18-shaped") + " encoder, --conditional-gan 1"
                                   if cgan else "--conditional-gan 0 (no encoder in the step)"),
                   "global_batch": world * a.batch, "caption_len": CFG2["L"], "image_size": CFG2["S"], "vocab": CFG2["V"],
                   "conditional_gan": cgan, "encoder": a.encoder if cgan else None, "step_impl": a.step_impl,
                   "parallelism": "dp%d" % world},
    }
    log(f"timed region done: {elapsed / a.steps * 1e3:.3f} ms/step; roofline probe")
    if not a.no_roofline:
        # the in-step measurement runs train steps: only without data parallelism (the other ranks have left; a step would wait for
        # them in its all-reduce) -- with N > 1 the line carries the isolated replay of rank 0
        out["roofline"] = roofline_probe(inst, args, cgan, step if (a.workload != "cfg4" and world == 1) else None)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a, cgan)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
