"""Sanity of the replayed step at bench shapes (tools only): token diversity, losses, temperature actually used."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
a = bench.parse()
inst, args = bench.build_instructor(a, 1)
from gan_image_captioning_amd.tasks import synthetic_batch
images, captions, _l, L = synthetic_batch(a.batch, bench.CFG2["V"], bench.CFG2["S"], bench.CFG2["L"], seed=1008, device=args.device, with_images=True)
for k in range(8):
    out = inst.fused(images, captions, L, True, next_images=images)
    inst.update_temperature((k + 1) / 50, args.adv_epochs)
    torch.cuda.synchronize()
    ids = out["ids"]
    print(k, "graphs", len(inst.fused._graphs), "unique ids", int(torch.unique(ids).numel()), "of", ids.numel(), "losses", [round(float(v), 5) for v in out["losses"]],
          "T", round(float(inst.gen.decoder.temperature), 4), "probs max mean", round(float(out["probs"].float().max(-1)[0].mean()), 4))
