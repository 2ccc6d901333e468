"""Per-phase cycle stamps of one trunk convolution launch, first and last workgroup (needs a build with GIC_EXTRA_FLAGS=-DGIC_STAMPS;
tools only).  python tools/conv_stamps.py N,H,Cin,Cout,k,stride,pad[,bn]   (bn = 1: BatchNorm + ReLU on load)"""
import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd import engine as E, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
names = ["prologue", "first tile", "K loop", "stats+C tile", "store"]
for a in sys.argv[1:]:
    v = [int(x) for x in a.split(",")]
    N, H, Ci, Co, k, st, pad = v[:7]
    bn = len(v) > 7 and v[7]
    Ho = (H + 2 * pad - k) // st + 1
    x = torch.randn(N, H, H, Ci, device=dev).bfloat16()
    w = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).bfloat16()
    y = torch.empty(N, Ho, Ho, Co, device=dev, dtype=torch.bfloat16)
    nrep = 8
    stats = torch.zeros(nrep, 2 * Co, device=dev)
    in_stats = torch.zeros(nrep, 2 * Ci, device=dev)
    in_stats[:, Ci:] = float(N * H * H) / nrep
    gamma, beta = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        if bn:
            _lib.check(lib.gic_conv2d_bn_in(p(x), p(in_stats), nrep, p(gamma), p(beta), ctypes.c_float(N * H * H), p(w), p(y), p(stats), nrep, 1,
                                            N, H, H, Ci, Co, k, k, st, pad, s), "conv_bn_in")
        else:
            _lib.check(lib.gic_conv2d(p(x), p(w), p(y), p(stats), nrep, 1, N, H, H, Ci, Co, k, k, st, pad, s), "conv")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); e1.synchronize()
    print(f"{a}: {e0.elapsed_time(e1) * 100:.1f} us per launch")
    if not hasattr(lib, "gic_debug_stamps"):
        continue
    if k == 3 and st == 1 and Ci % 64 == 0 and not os.environ.get("GIC_NO_CONV3X3_PATCH"):
        cb = (ctypes.c_ulonglong * 32)()
        lib.gic_debug_conv3x3_stamps(cb)
        ph = ["index math", "issue+coefficients", "patch landed", "normalise", "K loop", "epilogue"]
        for blk, o in (("first", 0), ("last", 16)):
            rt = [cb[o + 2 * i + 1] for i in range(7)]; cy = [cb[o + 2 * i] for i in range(7)]
            if not bn:
                rt[3] = rt[2]; cy[3] = cy[2]
            print(f"   {blk:5s} workgroup: " + "  ".join(f"{ph[i]} {(rt[i + 1] - rt[i]) / 100:.2f}us/{cy[i + 1] - cy[i]}cyc" for i in range(6))
                  + f" | total {(rt[6] - rt[0]) / 100:.2f} us, started {(rt[0] - cb[1]) / 100:.2f} us after the first")
        continue
    buf = (ctypes.c_ulonglong * 32)()
    lib.gic_debug_stamps(buf)
    for blk, o in (("first", 0), ("last", 16)):
        cyc = [buf[o + 2 * i] for i in range(6)]; rt = [buf[o + 2 * i + 1] for i in range(6)]
        print(f"   {blk:5s} workgroup: " + "  ".join(f"{names[i]} {(rt[i + 1] - rt[i]) / 100:.2f}us/{cyc[i + 1] - cyc[i]}cyc" for i in range(5))
              + f" | total {(rt[5] - rt[0]) / 100:.2f} us, started {(rt[0] - buf[1]) / 100:.2f} us after the first")
