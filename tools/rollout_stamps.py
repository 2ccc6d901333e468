"""Phase stamps of the last lstm_step / vocab_step launch of a roll-out, and the roll-out replayed as one hipGraph (tools only; needs a
build with GIC_EXTRA_FLAGS=-DGIC_STAMPS).  python tools/rollout_stamps.py"""
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd import engine as E, _lib
B, L, V, Em, H = 64, 20, 10000, 512, 512
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
eng = E.DecoderEngine(V, Em, H, 1, 1)
P = [torch.empty(V, Em).uniform_(-0.05, 0.05, generator=g), torch.empty(4 * H, Em).uniform_(-0.05, 0.05, generator=g),
     torch.empty(4 * H, H).uniform_(-0.05, 0.05, generator=g), torch.zeros(4 * H), torch.zeros(4 * H),
     torch.empty(V, H).uniform_(-0.05, 0.05, generator=g), torch.zeros(V)]
P = [p.to(dev) for p in P]
feats = torch.randn(B, Em, device=dev) * 0.3
st = eng.alloc_state(B, L, dev)
out = torch.empty(B, L, V, device=dev, dtype=torch.bfloat16)
ids = torch.empty(B, L, device=dev, dtype=torch.int64)
f = lambda: eng.sample_fwd(P, feats, L, 1.5, seed=7, state=st, out=out, ids=ids)
lib = _lib.load()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"eager launches : {timed(f):7.1f} us per roll-out")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    f(); f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        f()
    torch.cuda.synchronize()
    print(f"one hipGraph   : {timed(gr.replay):7.1f} us per roll-out")
f(); torch.cuda.synchronize()
if hasattr(lib, "gic_debug_decoder_stamps"):
    buf = (ctypes.c_ulonglong * 64)()
    lib.gic_debug_decoder_stamps(buf)
    v = lambda k, blk, ph, c: buf[((k * 2 + blk) * 8 + ph) * 2 + c]
    for k, name, nph, labels in ((0, "lstm_step", 7, ["ids", "issue+LDS writes", "barrier", "MFMA", "reduce", "cell"]),
                                 (1, "vocab_step", 8, ["stage", "barrier", "MFMA", "exchange", "argmax", "exp", "partials"])):
        for blk, bn in ((0, "first"), (1, "last")):
            rt = [v(k, blk, p, 1) for p in range(nph)]
            cy = [v(k, blk, p, 0) for p in range(nph)]
            print(f"{name:10s} {bn:5s} block: " + "  ".join(f"{labels[i]} {(rt[i + 1] - rt[i]) / 100:.2f}us/{cy[i + 1] - cy[i]}cyc" for i in range(nph - 1))
                  + f"  | total {(rt[-1] - rt[0]) / 100:.2f} us")
    # the last step's pair: lstm_step end -> vocab_step start (kernel boundary), on the 100 MHz clock
    print(f"lstm_step(last block end) -> vocab_step(first block start): {(v(1, 0, 0, 1) - v(0, 1, 6, 1)) / 100:.2f} us;"
          f" lstm first-block start -> last-block end {(v(0, 1, 6, 1) - v(0, 0, 0, 1)) / 100:.2f} us;"
          f" vocab first-block start -> last-block end {(v(1, 1, 7, 1) - v(1, 0, 0, 1)) / 100:.2f} us")
