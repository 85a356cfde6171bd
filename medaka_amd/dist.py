"""One-process-per-GPU harness: shard, barrier, time, max over ranks.

The consensus path shards by window/region with NO data-path collective; the only
communication is a barrier before/after the timed region and a MAX all-reduce of the elapsed
time.  Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
import os
import time


class Ranks:
    """Process-group context read from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE)."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self._pg = False
        if self.world > 1:
            import torch
            import torch.distributed as dist
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.backend = backend
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
                self._pg = True

    def _device(self):
        import torch
        return torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.world > 1:
            import torch
            import torch.distributed as dist
            # an all-reduce is a barrier that works identically on nccl and gloo
            t = torch.zeros(1, device=self._device())
            dist.all_reduce(t)
            if self.backend == "nccl":
                torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self._pg:
            import torch.distributed as dist
            dist.destroy_process_group()
            self._pg = False


def timed_steps(ranks, step_fn, sync_fn, steps, warmup):
    """W untimed + exactly K timed calls of step_fn, bracketed by barrier + device sync on both
    sides; returns (max elapsed seconds over ranks, this rank's elapsed)."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    mine = time.perf_counter() - t0
    ranks.barrier()
    return ranks.max_over_ranks(mine), mine
