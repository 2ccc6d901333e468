"""Data parallelism for the train step: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU rehearsal tests).

The reference is single-device (src/args.py:275-278); everything here is new capability
(SURVEY.md §8(e)).  The minibatch shards along the batch axis; parameters are replicated; the only
exchange on the data path is one sum-all-reduce per optimizer of that model's FLAT gradient arena
(D: 1.5 M floats, G: 12-13 M floats), issued on a side HIP stream as soon as the corresponding
backward has been enqueued so that D's all-reduce overlaps G's backward.  Averaging (1/W) happens
in the same pass; clipping happens after it (||mean g|| != mean ||g||, training.py:198).

This module is compute-agnostic (it only sees flat tensors), which is what lets the gloo tests
drive it with the CPU oracle as the per-rank compute.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist


@dataclass
class DistInfo:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1

    @staticmethod
    def from_env(init: bool = True) -> "DistInfo":
        ws = int(os.environ.get("WORLD_SIZE", "1"))
        info = DistInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), ws)
        if ws > 1 and init and not dist.is_initialized():
            use_gpu = torch.cuda.is_available()
            if use_gpu:
                # one GPU per rank; the modulo only matters for the single-GPU rehearsal (GIC_DIST_BACKEND=gloo), where
                # several ranks share a card that RCCL would refuse
                torch.cuda.set_device(info.local_rank % torch.cuda.device_count())
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("GIC_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
            dist.init_process_group(backend=backend, rank=info.rank, world_size=ws)
        return info


def shard_rows(t: torch.Tensor, info: DistInfo, dim: int = 0) -> torch.Tensor:
    """Replica r of W takes rows [r*B/W, (r+1)*B/W) (SURVEY.md §8(e) partitioning)."""
    n = t.shape[dim]
    if n % info.world_size:
        raise ValueError(f"global batch {n} is not divisible by world size {info.world_size}")
    per = n // info.world_size
    return t.narrow(dim, info.rank * per, per)


def broadcast_module(module: torch.nn.Module, info: DistInfo, src: int = 0) -> None:
    if info.world_size <= 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def shard_sampler(dataset, info: DistInfo, shuffle: bool):
    from torch.utils.data.distributed import DistributedSampler
    return DistributedSampler(dataset, num_replicas=info.world_size, rank=info.rank, shuffle=shuffle)


class GradReducer:
    """Asynchronous mean-all-reduce of flat gradient buffers.

    GPU tensors: the collective is enqueued on a dedicated side stream behind an event recorded on the
    producer (current) stream; ``wait_all`` makes the current stream wait for every outstanding
    collective.  CPU tensors (gloo rehearsal): synchronous.
    """

    def __init__(self, info: DistInfo, force: bool = False):
        """``force``: run the collectives even with world_size 1 (a 1-rank RCCL communicator: the identity, bit for bit) -- lets a
        one-GPU box execute the device branch and the fused step's reducer schedule (tests/test_gpu_dp.py)."""
        self.info = info
        self.force = bool(force)
        self._side: Optional[torch.cuda.Stream] = None
        self._pending: List[torch.cuda.Event] = []

    def start(self, flat: torch.Tensor):
        """Returns the collective's completion event (None when it ran synchronously) for ``wait``."""
        if self.info.world_size <= 1 and not self.force:
            return None
        if not flat.is_cuda or dist.get_backend() == "gloo":      # gloo has no AVG; on device tensors it stages through the host
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / self.info.world_size)
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=flat.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            done = torch.cuda.Event()
            done.record(self._side)
        flat.record_stream(self._side)
        self._pending.append(done)
        return done

    def wait(self, done) -> None:
        """Make the current stream wait for ONE collective (the others stay pending)."""
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
            if done in self._pending:
                self._pending.remove(done)

    def wait_all(self) -> None:
        for ev in self._pending:
            torch.cuda.current_stream().wait_event(ev)
        self._pending.clear()
