"""Multi-process plumbing for bench.py (one process per GPU, torch.distributed: backend "nccl" is
RCCL on ROCm, "gloo" on CPU).  Pure communication logic -- no field arithmetic -- so it is covered
by world_size-2 gloo tests on CPU.  The hot path itself never imports torch."""
import os


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


class Group:
    """thin wrapper: barrier + device sync, max-over-ranks of a scalar, gather of small byte strings"""

    def __init__(self, backend=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.rank, self.local_rank, self.world = env_world()
        # The control plane (the barriers around the timed region, max-over-ranks, the digest gather) is a handful of
        # 8..32-byte collectives: it runs on gloo / CPU tensors by default, so that several ranks may share one GPU
        # (rehearsal on a 1-GPU box) and the bench does not depend on a collective library for work that has no bulk
        # exchange (the worker axis: DESIGN.md 6).  COZK_DIST_BACKEND=nccl puts it on RCCL (needs one GPU per rank);
        # the bulk Rep3 ring is RCCL inside libcozk either way (cozk_ring_*).
        backend = backend or os.environ.get("COZK_DIST_BACKEND") or "gloo"
        if backend == "nccl" and device is not None and self.world > torch.cuda.device_count():
            backend = "gloo"  # RCCL refuses two ranks of one communicator on the same GPU
        self.backend = backend
        self.device = device
        self.cuda = device is not None and backend != "gloo"
        self.sync_device = device
        self._cpu_group = None
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def _dev(self):
        return self.torch.device("cuda", self.device) if self.cuda else self.torch.device("cpu")

    def cpu_group(self):
        """a gloo group over all ranks for host-side byte exchanges (None = the default group when that is gloo already)"""
        if self.world == 1 or self.backend == "gloo":
            return None
        if self._cpu_group is None:
            self._cpu_group = self.dist.new_group(backend="gloo")
        return self._cpu_group

    def barrier(self):
        if self.sync_device is not None:
            self.torch.cuda.synchronize(self.sync_device)
        if self.world > 1:
            # an all_reduce on the rank's own device doubles as the barrier (works for nccl and gloo)
            t = self.torch.zeros(1, device=self._dev())
            self.dist.all_reduce(t)
            if self.cuda:
                self.torch.cuda.synchronize(self.device)

    def max_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def all_gather_bytes(self, b, width=32):
        """every rank contributes exactly `width` bytes (e.g. a proof digest); returns the list"""
        assert len(b) == width
        if self.world == 1:
            return [bytes(b)]
        t = self.torch.tensor(list(b), dtype=self.torch.uint8, device=self._dev())
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [bytes(x.cpu().tolist()) for x in out]

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()


def shard_seed(base_seed, rank):
    """each rank proves its own trace segment: segment r is seeded base_seed + 7919 * r"""
    return base_seed + 7919 * rank
