"""Data parallelism on the GPU path, rehearsed on ONE card: two ranks share cuda:0 and all-reduce through gloo
(RCCL refuses two ranks on one device; GIC_DIST_BACKEND=gloo swaps the backend, everything else is the production path:
DistInfo.from_env, shard_rows, broadcast of the replicas, GradReducer inside FusedAdvStep, clip after the all-reduce).
Two steps of 2 x 4 captions must reproduce the single-process run on the 8 captions."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> str:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def test_two_ranks_reproduce_the_single_process_step(tmp_path):
    worker = os.path.join(ROOT, "tests", "dp_gpu_worker.py")
    env = dict(os.environ, GIC_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    single = str(tmp_path / "single_%d.pt")
    r = subprocess.run([sys.executable, worker, single], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    multi = str(tmp_path / "multi_%d.pt")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", free_port(), worker, multi], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    one = torch.load(single % 0)
    r0, r1 = torch.load(multi % 0), torch.load(multi % 1)
    assert one["world"] == 1 and r0["world"] == 2
    # replicas stay identical
    assert torch.equal(r0["gen"], r1["gen"]) and torch.equal(r0["disc"], r1["disc"])
    # the mean of the two shard losses is the global-batch loss; weights after two clip+Adam steps match the single process
    torch.testing.assert_close((r0["losses"] + r1["losses"]) / 2, one["losses"], rtol=1e-5, atol=1e-7)
    assert r0["d_norm"] == pytest.approx(one["d_norm"], rel=1e-4)      # norm of the AVERAGED gradient, not of a shard's
    torch.testing.assert_close(r0["disc"], one["disc"], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(r0["gen"], one["gen"], rtol=1e-4, atol=2e-6)
    # device-drawn noise differs between the replicas (same weights, same inputs, same torch seed)
    assert not torch.equal(r0["free_ids"], r1["free_ids"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_ranks_over_rccl_reproduce_the_single_process_step(tmp_path):
    """The production collective path itself (GradReducer's device branch: RCCL ReduceOp.AVG on the side stream, early bucket of
    G's arena, D's Adam gated on its own collective) on two GPUs; runs wherever >= 2 devices are visible."""
    worker = os.path.join(ROOT, "tests", "dp_gpu_worker.py")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GIC_DIST_BACKEND"):
        env.pop(k, None)
    single = str(tmp_path / "single_%d.pt")
    r = subprocess.run([sys.executable, worker, single], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    multi = str(tmp_path / "multi_%d.pt")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", free_port(), worker, multi], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    one = torch.load(single % 0)
    r0, r1 = torch.load(multi % 0), torch.load(multi % 1)
    assert torch.equal(r0["gen"], r1["gen"]) and torch.equal(r0["disc"], r1["disc"])
    torch.testing.assert_close((r0["losses"] + r1["losses"]) / 2, one["losses"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(r0["disc"], one["disc"], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(r0["gen"], one["gen"], rtol=1e-4, atol=2e-6)


def test_one_rank_rccl_communicator_drives_the_device_branch(tmp_path):
    """SURVEY 8(e) on a one-GPU box: a 1-rank "nccl" (= RCCL) process group and GradReducer(force=True) execute the device branch --
    side-stream ReduceOp.AVG behind an event, start / wait / wait_all -- and FusedAdvStep with that reducer attached (early bucket of
    G's arena under BPTT, D's Adam gated on its own collective, the rest of G's arena) reproduces the reducer-less fused step in fp32 over two
    optimizer steps (an average over one rank is the identity: the raw buffers come back bit for bit)."""
    worker = os.path.join(ROOT, "tests", "dp_rccl1_worker.py")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port())
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GIC_DIST_BACKEND"):
        env.pop(k, None)
    out = str(tmp_path / "rccl1.pt")
    r = subprocess.run([sys.executable, worker, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    res = torch.load(out)
    assert res["backend"] == "nccl" and res["collectives"] >= 2 * 3 + 3, res["collectives"]      # per step: early bucket, D, 1-2 G spans
    assert res["raw_identity"] and res["pending_after_wait_all"] == 0
    # (the step is not bit-reproducible run to run: D's weight gradients, the column sums and the embedding scatter add with f32
    # atomics; the tolerances are those of the two-rank test above, the first step's losses -- no atomics upstream -- are exact)
    assert torch.equal(res["plain"]["losses"][0], res["dp"]["losses"][0])
    torch.testing.assert_close(res["plain"]["losses"], res["dp"]["losses"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(res["plain"]["disc"], res["dp"]["disc"], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(res["plain"]["gen"], res["dp"]["gen"], rtol=1e-4, atol=2e-6)
    assert float((res["plain"]["ids"] == res["dp"]["ids"]).float().mean()) >= 0.98
