"""MLE pre-train step at cfg2 shapes: ms per step, HIP-synchronised wall clock (tools only)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
a = bench.parse()
inst, args = bench.build_instructor(a, 1 if a.cgan is None else a.cgan)
from gan_image_captioning_amd.tasks import synthetic_batch
images, captions, _l, L = synthetic_batch(a.batch, bench.CFG2["V"], bench.CFG2["S"], bench.CFG2["L"], seed=1008, device=args.device, with_images=True)
for _ in range(5):
    inst.pretrain_step(images, captions, L, train=True, next_images=None if a.no_prefetch else images)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    inst.pretrain_step(images, captions, L, train=True, next_images=None if a.no_prefetch else images)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"pretrain step (cgan={int(inst.cgan)}, B={a.batch}, L={L}, V={bench.CFG2['V']}, {a.dtype}): {ms:.3f} ms/step = {a.batch / ms * 1e3:.0f} captions/s")
