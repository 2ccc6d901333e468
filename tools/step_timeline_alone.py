import sys, torch
sys.path.insert(0, "/root/repo")
import bench
a = bench.parse()
inst, args = bench.build_instructor(a, 1)
from gan_image_captioning_amd.tasks import synthetic_batch
images, captions, _l, L = synthetic_batch(a.batch, bench.CFG2["V"], bench.CFG2["S"], bench.CFG2["L"], seed=1008, device=args.device, with_images=True)
for k in range(6):
    inst.adv_step(images, captions, L, train=True, next_images=images)
torch.cuda.synchronize()
acc = {}
for k in range(5):
    inst.adv_step(images, captions, L, train=True, next_images=images)      # leaves a prefetched trunk pass behind
    torch.cuda.synchronize()
    inst.fused.trace = []
    inst.adv_step(images, captions, L, train=True, next_images=None)        # consumes it, launches no trunk pass: the chain alone
    torch.cuda.synchronize()
    t0 = inst.fused.trace[0][1]
    for name, ev in inst.fused.trace:
        acc.setdefault(name, []).append(t0.elapsed_time(ev) * 1e3)
    inst.fused.trace = None
for name, v in acc.items():
    print(f"{sum(v) / len(v):9.1f} us  {name}")
