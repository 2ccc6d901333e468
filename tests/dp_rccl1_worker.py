"""Worker of tests/test_gpu_dp.py::test_one_rank_rccl_communicator_drives_the_device_branch: ONE rank, backend "nccl" (RCCL)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cpu_step as O                                        # noqa: E402  (problem generator only)

B, L, V, E, H, R = 8, 6, 64, 16, 32, 64


def make(reducer_for=None):
    from gan_image_captioning_amd.args import default_args
    from gan_image_captioning_amd.fused_step import FusedAdvStep
    from gan_image_captioning_amd.training import GANInstructor
    torch.manual_seed(1008)
    args = default_args(vocab_size=V, gen_embed_dim=E, gen_hidden_dim=H, conditional_gan=0, compute_dtype="fp32", clip_norm=0.05,
                        adv_train_batch_size=B, device="cuda", log_file=None, model_dir=None, save_dir=None)
    inst = GANInstructor(args, None, None)
    inst.gen.train(); inst.disc.train()
    step = inst.fused
    if reducer_for is not None:
        step = FusedAdvStep(inst.gen, inst.disc, inst.gen_arena, inst.disc_arena, args, reducer_for).bind_optimizers(inst.gen_opt, inst.disc_opt)
    return inst, args, step


def run(out_path):
    from gan_image_captioning_amd import parallel
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    info = parallel.DistInfo(0, 0, 1)
    red = parallel.GradReducer(info, force=True)
    count = {"n": 0}
    real_start = red.start

    def counting_start(flat):
        ev = real_start(flat)
        assert ev is not None, "the device branch did not run"
        count["n"] += 1
        return ev

    red.start = counting_start
    # the branch itself on a raw buffer: start -> wait (one), start x2 -> wait_all
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    ev = red.start(x)
    red.wait(ev)
    red.start(x[: 1 << 10]); red.start(x[1 << 10:])
    red.wait_all()
    torch.cuda.synchronize()
    raw_identity = bool(torch.equal(x, ref))
    pending = len(red._pending)

    g = torch.Generator().manual_seed(77)
    caps = O.make_captions(B, L, V, g).cuda()
    us, masks = O.make_noise(B, L, V, 900, R, g)
    us = torch.stack(us).cuda()
    masks = [m.cuda() for m in masks]
    res = {}
    for name, r in (("plain", None), ("dp", red)):
        inst, args, step = make(r)
        losses = []
        for _ in range(2):
            out = step(None, caps, L, True, us, masks)
            losses.append(out["losses"].clone())
        torch.cuda.synchronize()
        res[name] = {"gen": inst.gen_arena.flat.cpu(), "disc": inst.disc_arena.flat.cpu(), "losses": torch.stack(losses).cpu(),
                     "ids": out["ids"].cpu()}
    torch.save({"backend": dist.get_backend(), "collectives": count["n"], "raw_identity": raw_identity, "pending_after_wait_all": pending,
                **res}, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
