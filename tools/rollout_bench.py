"""Roll-out alone (tools only): Decoder.sample forward / backward at the benchmark's shapes under HIP events, fused step kernels vs the
generic launches.  python tools/rollout_bench.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd import engine as E
B, L, V, Em, H = 64, 20, 10000, 512, 512
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
eng = E.DecoderEngine(V, Em, H, 1, 1)
P = [torch.empty(V, Em).uniform_(-0.05, 0.05, generator=g), torch.empty(4 * H, Em).uniform_(-0.05, 0.05, generator=g),
     torch.empty(4 * H, H).uniform_(-0.05, 0.05, generator=g), torch.zeros(4 * H), torch.zeros(4 * H),
     torch.empty(V, H).uniform_(-0.05, 0.05, generator=g), torch.zeros(V)]
P = [p.to(dev) for p in P]
feats = torch.randn(B, Em, device=dev) * 0.3
d_out = (torch.randn(B, L, V, device=dev) * 1e-3).bfloat16()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


from gan_image_captioning_amd import _lib
# ablation bits (gic_debug_decoder_step): 1 = lstm_step's weight loads, 2 = lstm_step's activation staging and vocab_step's MFMA loop,
# 4 = vocab_step's staging (weights and h) removed
for name, part, dbg in (("fused", True, 0), ("fused-noW", True, 1), ("fused-noA", True, 2), ("fused-noWA", True, 3), ("fused-noStage", True, 4),
                        ("fused-none", True, 7), ("generic", False, 0)):
    _lib.load().gic_debug_decoder_step(dbg)
    st = eng.alloc_state(B, L, dev)
    if not part:
        st["part"] = None
    out = torch.empty(B, L, V, device=dev, dtype=torch.bfloat16)
    ids = torch.empty(B, L, device=dev, dtype=torch.int64)
    ws = eng.alloc_bwd_ws(B, L, dev)
    grads = eng.alloc_grads(P, B)
    f = lambda: eng.sample_fwd(P, feats, L, 1.5, seed=7, state=st, out=out, ids=ids)
    print(f"{name:8s} forward  {timed(f):8.1f} us")
    if dbg == 0:
        b = lambda: eng.sample_bwd(P, st, out, ids, d_out, 1.5, ws=ws, grads=grads)
        print(f"{name:8s} backward {timed(b):8.1f} us")
_lib.load().gic_debug_decoder_step(0)
