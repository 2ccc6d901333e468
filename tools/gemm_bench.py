"""Micro-benchmark of the MFMA GEMM kernel (tools only; not part of the product path)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd import engine as E
dev = torch.device("cuda:0")
shapes = [(4096, 4096, 4096), (8192, 8192, 1024), (12544, 256, 2304), (200704, 64, 576), (200704, 256, 64), (4096, 900, 904), (64, 2048, 1024), (64, 10000, 512), (1280, 512, 10000), (10000, 512, 1280)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    for dt, odt in ((torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)):
        A = torch.randn(M, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt)
        C = torch.empty(M, N, device=dev, dtype=odt)
        f = lambda: E.gemm(A, B, C, M, N, K, K, K, N)
        for _ in range(3): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); e.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"NT {M:7d}x{N:6d}x{K:6d} out={str(odt)[6:]:9s} {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
