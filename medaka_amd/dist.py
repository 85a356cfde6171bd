"""One-process-per-GPU harness: shard, barrier, time, max over ranks.

The consensus path shards by window/region with NO data-path collective; the only
communication is a barrier before/after the timed region and a MAX all-reduce of the elapsed
time.  Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
import os
import sys
import time


class Ranks:
    """Process-group context read from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE).

    The group only carries the barrier and the MAX / SUM of a few scalars around the timed region (the data path has no
    collective), so the backend is a convenience, not a requirement: RCCL ("nccl") is tried where every rank has a GPU
    of its own, and on ANY failure -- a rank without its device, an init error, a first all-reduce that raises or times
    out -- ALL ranks fall back to gloo together (they agree through a TCPStore of their own, which needs no backend).
    `backend` = "gloo" skips the attempt; `barrier_backend` says what is in use, `fallback_reason` why."""

    def __init__(self, backend=None, nccl_timeout_s=90):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self.fallback_reason = None
        self._pg = False
        self.device_index = self.local_rank          # which visible HIP device this rank computes on (see below)
        if self.world > 1:
            import datetime
            import torch
            import torch.distributed as dist
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend not in ("nccl", "gloo"):
                raise ValueError(f"medaka_amd.dist.Ranks: backend {backend!r} is not supported (\"nccl\" = RCCL, \"gloo\", or None to choose)")
            if torch.cuda.is_available() and torch.cuda.device_count() > 0:
                # a launcher that isolates the ranks (HIP_VISIBLE_DEVICES per rank) shows each of them ONE device, index 0
                self.device_index = self.local_rank if self.local_rank < torch.cuda.device_count() else 0
            if dist.is_initialized():
                # somebody else's group: use it as it is -- on the device this rank can really see
                self.backend = dist.get_backend()
                if self.backend == "nccl" and torch.cuda.is_available():
                    torch.cuda.set_device(self.device_index)
                return
            # the ranks' own store: agreement on the backend must not depend on the backend.  Its port: MEDAKA_AMD_STORE_PORT, else
            # MASTER_PORT + 17 folded back into the unprivileged range (a MASTER_PORT near 65535 must not produce an invalid port,
            # on which every rank would wait out the 300 s store time-out)
            port = os.environ.get("MEDAKA_AMD_STORE_PORT")
            port = int(port) if port else 1024 + (int(os.environ["MASTER_PORT"]) + 17 - 1024) % (65536 - 1024)
            store = dist.TCPStore(os.environ["MASTER_ADDR"], port, self.world, is_master=(self.rank == 0),
                                  timeout=datetime.timedelta(seconds=300), wait_for_workers=True)
            if backend == "nccl":
                err = None
                try:
                    if not torch.cuda.is_available() or torch.cuda.device_count() == 0:
                        raise RuntimeError(f"rank {self.rank}: no HIP device {self.local_rank} (device_count = {torch.cuda.device_count()})")
                    torch.cuda.set_device(self.device_index)
                except Exception as exc:               # noqa: BLE001 -- whatever it is, this rank cannot take part in RCCL
                    err = f"{type(exc).__name__}: {exc}"
                # phase A: does every rank have its device?  (nobody enters a collective that another rank cannot join)
                store.set(f"mdk/dev/{self.rank}", err or "ok")
                devs = [store.get(f"mdk/dev/{r}").decode() for r in range(self.world)]
                bad = [d for d in devs if d != "ok"]
                if not bad:
                    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")      # a timeout raises instead of aborting the process
                    try:
                        dist.init_process_group(backend="nccl", store=dist.PrefixStore("mdk/nccl", store), rank=self.rank,
                                                world_size=self.world, timeout=datetime.timedelta(seconds=nccl_timeout_s))
                        t = torch.zeros(1, device=torch.device("cuda", self.device_index))
                        dist.all_reduce(t)                    # the communicator is only built here
                        torch.cuda.synchronize()
                    except Exception as exc:                  # noqa: BLE001
                        err = f"{type(exc).__name__}: {str(exc)[:300]}"
                    # phase B: did it work everywhere?
                    store.set(f"mdk/nccl/{self.rank}", err or "ok")
                    res = [store.get(f"mdk/nccl/{r}").decode() for r in range(self.world)]
                    bad = [d for d in res if d != "ok"]
                if bad:
                    self.fallback_reason = bad[0]
                    if dist.is_initialized():
                        try:
                            dist.destroy_process_group()
                        except Exception:                      # noqa: BLE001
                            pass
                    if self.rank == 0:
                        print(f"[medaka_amd.dist] RCCL process group not usable ({bad[0]}): barrier and reductions over gloo "
                              "(the data path has no collective)", file=sys.stderr, flush=True)
                    backend = "gloo"
                else:
                    self._pg = True
            if backend == "gloo":
                dist.init_process_group(backend="gloo", store=dist.PrefixStore("mdk/gloo", store), rank=self.rank, world_size=self.world,
                                        timeout=datetime.timedelta(seconds=300))
                self._pg = True
            self.backend = backend
            self._store = store

    @property
    def barrier_backend(self):
        return self.backend if self.world > 1 else "none (one rank)"

    def ranks_seen(self):
        """How many ranks answer a SUM of ones: the world size, if the group works."""
        return int(round(self.sum_over_ranks(1.0)))

    def _device(self):
        import torch
        return torch.device("cuda", self.device_index) if self.backend == "nccl" else torch.device("cpu")

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
