"""One conv3-shaped launch (1x1, BatchNorm + ReLU of the input on load) against a float matmul, element by element, and its launch rate:
   python tools/pix_probe.py <images> <H> <Cin> <Cout>     (GIC_NO_CONV1X1_PIX=1: the panel kernel; GIC_PIX_WG=n: workgroup target)
Prints which rows / columns differ -- how the store-data hazard of conv1x1_pix.hip / conv_b2b.hip was found."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd import _lib as L, engine
lib = L.load()
dev = torch.device("cuda:0")
N, H, Ci, Co = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (1, 16, 256, 512)
rows = N * H * H
g = torch.Generator().manual_seed(1)
y = torch.rand(rows, Ci, generator=g).to(dev).bfloat16()
w = (torch.randn(Co, Ci, generator=g) * 0.05).to(dev).bfloat16()
gamma = torch.ones(Ci, device=dev); beta = torch.zeros(Ci, device=dev)
in_stats = torch.zeros(1, 2 * Ci, device=dev); in_stats[0, Ci:] = rows * (1 - 1e-5)
out = torch.full((rows, Co), 7.0, device=dev, dtype=torch.bfloat16)
st = torch.zeros(1, 2 * Co, device=dev)
s = engine.stream_ptr()
status = lib.gic_conv2d_bn_in(y.data_ptr(), in_stats.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), float(rows), w.data_ptr(), out.data_ptr(),
                              st.data_ptr(), 1, 1, N, H, H, Ci, Co, 1, 1, 1, 0, s)
torch.cuda.synchronize()
print("status", status)
ref = y.float() @ w.float().t()
o = out.float()
bad = ~torch.isclose(o, ref, rtol=2e-2, atol=2e-2)
print("bad fraction", bad.float().mean().item(), "finite", torch.isfinite(o).all().item())
print("bad rows", bad.any(1).nonzero().flatten()[:40].tolist())
print("bad cols", bad.any(0).nonzero().flatten()[:80].tolist())
print("out[0,:8]", o[0, :8].tolist(), "ref", ref[0, :8].tolist())
print("stats err", (st[0, :Co] - ref.sum(0)).abs().max().item(), (st[0, Co:] - (ref * ref).sum(0)).abs().max().item())
idx = bad.nonzero()[:6]
for r, c in idx.tolist():
    print(r, c, o[r, c].item(), ref[r, c].item())
import os
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
args = (y.data_ptr(), in_stats.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), float(rows), w.data_ptr(), out.data_ptr(), st.data_ptr(), 1, 1, N, H, H, Ci, Co, 1, 1, 1, 0, s)
for _ in range(20): lib.gic_conv2d_bn_in(*args)
torch.cuda.synchronize(); e0.record()
for _ in range(200): lib.gic_conv2d_bn_in(*args)
e1.record(); torch.cuda.synchronize()
print("WG", os.environ.get("GIC_PIX_WG"), f"us/launch {e0.elapsed_time(e1) * 5:.2f}")
