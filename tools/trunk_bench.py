"""Trunk forward alone (tools only): ResNet-50 @224, batch 64, bf16, graph replay under HIP events.  python tools/trunk_bench.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from gan_image_captioning_amd.trunk import ResNetTrunk
dev = torch.device("cuda:0")
torch.manual_seed(0)
trunk = ResNetTrunk("resnet50").to(dev).train()
with torch.no_grad():
    for p in trunk.parameters():
        p.uniform_(-0.05, 0.05)
x = torch.randn(64, 3, 224, 224, device=dev)
for _ in range(4):
    trunk(x, 1)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    trunk(x, 1)
b.record(); b.synchronize()
print(f"trunk forward alone: {a.elapsed_time(b) / 20 * 1e3:.1f} us")
plan = trunk._plan
print("conv2 with bn1 on load:", sum(1 for blk in plan.blocks if blk["c2"].fused_in), "of", len(plan.blocks),
      "| conv3 with bn2 on load:", sum(1 for blk in plan.blocks if blk["c3"] is not None and blk["c3"].fused_in))
