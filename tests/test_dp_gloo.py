"""Data-parallel path rehearsed on the CPU: world_size 2, gloo backend.

``gan_image_captioning_amd/parallel.py`` is compute-agnostic (it shards batches and mean-all-reduces FLAT gradient
buffers), so here each rank's compute is the CPU oracle (tests may use it; the product path never does).  Checked:
  * shard_rows partitions the batch / the explicit noise / the dropout masks the way SURVEY.md §8(e) states;
  * mean-all-reduce of the two flat gradient arenas reproduces the single-process global-batch gradient;
  * clipping AFTER the all-reduce and Adam on every replica give identical weights on all ranks, equal to the
    single-process result (||mean g|| != mean ||g||: clipping per shard would differ - also asserted);
  * broadcast_module makes replicas identical.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_step as O

B, L, V, E, H, NL, R = 8, 6, 40, 8, 16, 1, 64
NF, FS = [12, 8, 16], [3, 4, 5]
CLIP = 0.005         # small enough that clipping is active (||d grad|| ~ 0.016 here), so its placement matters


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    g = torch.Generator().manual_seed(5)
    gp = O.make_gen_params(V, E, H, NL, g)
    dp = O.make_disc_params(V, g, num_filters=NF, filter_sizes=FS)
    caps = O.make_captions(B, L, V, g)
    us, masks = O.make_noise(B, L, V, sum(NF), R, g)
    return gp, dp, caps, torch.stack(us), masks


def _flat(d, names):
    return torch.cat([d[n].reshape(-1) for n in names])


def _unflat(flat, like, names):
    out, o = {}, 0
    for n in names:
        k = like[n].numel()
        out[n] = flat[o:o + k].view_as(like[n]).clone()
        o += k
    return out


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gan_image_captioning_amd import parallel
    info = parallel.DistInfo.from_env()
    assert dist.get_backend() == "gloo" and info.world_size == world
    gp, dp, caps, us, masks = _problem()
    # replicas start different on purpose; broadcast_module must fix that
    lin = torch.nn.Linear(4, 4)
    torch.manual_seed(100 + rank)
    torch.nn.init.normal_(lin.weight)
    parallel.broadcast_module(lin, info)
    gathered = [torch.zeros_like(lin.weight) for _ in range(world)]
    dist.all_gather(gathered, lin.weight.data)
    assert all(torch.equal(gathered[0], t) for t in gathered)

    # shard the batch, the per-step noise [L,B,V] along B, the masks [B*R,F] along rows b*R+r
    caps_r = parallel.shard_rows(caps, info)
    us_r = parallel.shard_rows(us, info, dim=1)
    masks_r = [parallel.shard_rows(m.view(B, R, -1), info).reshape(-1, m.shape[1]) for m in masks]
    out = O.adv_step(dict(gp), dict(dp), caps_r, list(us_r), masks_r, 1.3, "standard", 1e9, None, None, num_rep=R)
    gnames, dnames = sorted(out["g_grads_raw"]), sorted(out["d_grads_raw"])
    gflat, dflat = _flat(out["g_grads_raw"], gnames), _flat(out["d_grads_raw"], dnames)
    local_d_norm = float(dflat.norm())
    red = parallel.GradReducer(info)
    red.start(dflat)               # D's all-reduce is issued first (overlaps G's backward on the GPU)
    red.start(gflat)
    red.wait_all()
    # clip AFTER the all-reduce, then Adam, identically on every replica
    gopt, dopt = O.AdamState(1e-3), O.AdamState(1e-3)
    dg, dnorm = O.clip_grad_norm(_unflat(dflat, out["d_grads_raw"], dnames), CLIP)
    gg, gnorm = O.clip_grad_norm(_unflat(gflat, out["g_grads_raw"], gnames), CLIP)
    dopt.step(dp, dg)
    gopt.step(gp, gg)
    ret[rank] = {"gflat": gflat, "dflat": dflat, "dnorm": dnorm, "gnorm": gnorm, "local_d_norm": local_d_norm,
                 "dp": {k: v.clone() for k, v in dp.items()}, "gp": {k: v.clone() for k, v in gp.items()},
                 "ids": out["ids"], "losses": (out["g_loss"], out["d_loss"])}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    gp, dp, caps, us, masks = _problem()
    ref = O.adv_step(dict(gp), dict(dp), caps, list(us), masks, 1.3, "standard", 1e9, None, None, num_rep=R)
    gnames, dnames = sorted(ref["g_grads_raw"]), sorted(ref["d_grads_raw"])
    r0, r1 = ret[0], ret[1]
    # per-caption work is independent: concatenated shards == global batch
    assert torch.equal(torch.cat([r0["ids"], r1["ids"]]), ref["ids"])
    assert 0.5 * (r0["losses"][1] + r1["losses"][1]) == pytest.approx(ref["d_loss"], rel=1e-6)
    # mean-all-reduced flat gradients == global-batch gradients, identical on both ranks
    torch.testing.assert_close(r0["dflat"], _flat(ref["d_grads_raw"], dnames), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(r0["gflat"], _flat(ref["g_grads_raw"], gnames), rtol=1e-4, atol=1e-9)
    assert torch.equal(r0["dflat"], r1["dflat"]) and torch.equal(r0["gflat"], r1["gflat"])
    # the clip threshold bites, and the norm of the mean is NOT the mean of the shard norms
    assert r0["dnorm"] > CLIP
    assert abs(0.5 * (r0["local_d_norm"] + r1["local_d_norm"]) - r0["dnorm"]) > 1e-3 * r0["dnorm"]
    # weights after clip + Adam: identical replicas == single process
    gopt, dopt = O.AdamState(1e-3), O.AdamState(1e-3)
    dg, dn = O.clip_grad_norm(ref["d_grads_raw"], CLIP)
    gg, gn = O.clip_grad_norm(ref["g_grads_raw"], CLIP)
    dopt.step(dp, dg)
    gopt.step(gp, gg)
    assert r0["dnorm"] == pytest.approx(dn, rel=1e-5)
    for k in dp:
        assert torch.equal(r0["dp"][k], r1["dp"][k])
        torch.testing.assert_close(r0["dp"][k], dp[k], rtol=1e-5, atol=1e-6)
    for k in gp:
        assert torch.equal(r0["gp"][k], r1["gp"][k])


def test_shard_rows_rejects_indivisible_batch():
    from gan_image_captioning_amd import parallel
    with pytest.raises(ValueError):
        parallel.shard_rows(torch.zeros(7, 3), parallel.DistInfo(0, 0, 2))
    x = torch.arange(8).view(8, 1)
    assert parallel.shard_rows(x, parallel.DistInfo(1, 1, 4)).flatten().tolist() == [2, 3]
