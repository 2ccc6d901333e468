"""dW_hw-shaped gradient GEMM alone (tools only): C[960, 960] += A[8192, 960]^T B[8192, 960], bf16 operands, f32 accumulate in place.
python tools/gemm_hw_bench.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd import engine as E
dev = torch.device("cuda:0")
torch.manual_seed(0)
K, M, N = 8192, 960, 960
A = torch.randn(K, M, device=dev).bfloat16()
B = torch.randn(K, N, device=dev).bfloat16()
C = torch.zeros(M, N, device=dev)
run = lambda: E.gemm(A, B, C, M, N, K, M, N, N, a_kc=False, b_kc=False, accumulate=True)
run(); torch.cuda.synchronize()
ref = A.float().t() @ B.float()
print("rel err", float((C - ref).abs().max() / ref.abs().max()))
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): run()
b.record(); b.synchronize()
print(f"{a.elapsed_time(b) / 50 * 1e3:.1f} us per launch")
