"""Worker of tests/test_gpu_dp.py: one rank of a data-parallel fused step (launched by torch.distributed.run)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cpu_step as O                                        # noqa: E402  (problem generator only)

B, L, V, E, H, R = 8, 6, 64, 16, 32, 64


def problem():
    g = torch.Generator().manual_seed(77)
    caps = O.make_captions(B, L, V, g)
    us, masks = O.make_noise(B, L, V, 900, R, g)
    return caps, torch.stack(us), masks


def make_instructor(batch):
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.training import GANInstructor
    torch.manual_seed(1008)
    args = default_args(vocab_size=V, gen_embed_dim=E, gen_hidden_dim=H, conditional_gan=0, compute_dtype="fp32", clip_norm=0.05,
                        adv_train_batch_size=batch, device="cuda", log_file=None, model_dir=None, save_dir=None)
    inst = GANInstructor(args, None, None)
    inst.gen.train(); inst.disc.train()
    return inst, args


def run(out_path):
    from gan_image_captioning_amd import parallel
    inst, args = make_instructor(B // int(os.environ.get("WORLD_SIZE", "1")))
    info = inst.dist
    caps, us, masks = problem()
    dev = args.device
    caps_r = parallel.shard_rows(caps, info).to(dev)
    us_r = parallel.shard_rows(us, info, dim=1).contiguous().to(dev)
    masks_r = [parallel.shard_rows(m.view(B, R, -1), info).reshape(-1, m.shape[1]).contiguous().to(dev) for m in masks]
    losses = []
    for _ in range(2):
        out = inst.fused(None, caps_r, L, True, us_r, masks_r)
        losses.append(out["losses"].clone())
    torch.cuda.synchronize()
    result = {"gen": inst.gen_arena.flat.cpu(), "disc": inst.disc_arena.flat.cpu(), "losses": torch.stack(losses).cpu(),
              "d_norm": float(inst.disc_opt.grad_norm), "world": info.world_size}
    # one more step with DEVICE-drawn noise on identical inputs: the replicas must not draw the same Gumbel noise (seed mixes the rank)
    same = caps[:B // info.world_size].to(dev)
    result["free_ids"] = inst.fused(None, same, L, True, None, None, opt_step=False)["ids"].cpu()
    torch.save(result, out_path % info.rank)
    if info.world_size > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
