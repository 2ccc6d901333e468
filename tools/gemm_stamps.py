"""Per-phase cycle stamps of one GEMM / conv block (needs a build with GIC_EXTRA_FLAGS=-DGIC_STAMPS; tools only)."""
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd import engine as E, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for a in sys.argv[1:]:
    M, N, K = (int(v) for v in a.split("x"))
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): E.gemm(A, B, C, M, N, K, K, K, N)
    torch.cuda.synchronize()
    E.gemm(A, B, C, M, N, K, K, K, N)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    rc = lib.gic_debug_stamps(buf)
    for blk, o in (("first", 0), ("last", 16)):
        cyc = [buf[o + 2 * i] for i in range(6)]; rt = [buf[o + 2 * i + 1] for i in range(6)]
        print(f"{a} {blk}: cycles d01={cyc[1]-cyc[0]} d12={cyc[2]-cyc[1]} d23(loop)={cyc[3]-cyc[2]} d34={cyc[4]-cyc[3]} d45={cyc[5]-cyc[4]} total={cyc[5]-cyc[0]}  realtime total={(rt[5]-rt[0])/100:.2f} us start={(rt[0]-buf[1])/100:.2f} us")
