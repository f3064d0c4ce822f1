"""Multi-GPU plumbing: the hot path shards over independent filters/frames, so the
only collectives are the timing barrier / max and one end-of-run gather of per-rank summaries.
One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU)."""
import os


def shard(total, world, rank):
    """Static block partition frame_id -> rank = frame_id // ceil(total / world).  Returns (start, count)."""
    per = (total + world - 1) // world
    start = min(rank * per, total)
    return start, max(0, min(per, total - start))


class Group:
    """Thin wrapper: works for world == 1 without importing torch."""

    def __init__(self, backend=None, force_init=False):
        """force_init: create the process group even at world size 1 (exercises the collective backend itself)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = None
        self.backend = None
        if self.world > 1 or force_init:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL)
            import torch
            import torch.distributed as dist
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group("nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend)
            self.dist = dist
            self.torch = torch
            self.backend = backend

    def shard(self, total):
        """[lo, hi) of the frame ids this rank owns (static block partition)."""
        start, count = shard(total, self.world, self.rank)
        return start, start + count

    def gather_scalars(self, x):
        """One float per rank -> list over ranks (per-rank ms_per_step: hidden host serialisation would show here)."""
        return [float(r[0]) for r in self.gather_summaries([float(x)])]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([float(x)], device=self.device, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([float(x)], device=self.device, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_summaries(self, vec):
        """All-gather of a small per-rank float vector (pose / diag(P) / accepted counts ...): [world, len]."""
        import numpy as np
        v = np.asarray(vec, dtype=np.float64).reshape(-1)
        if self.dist is None:
            return v[None, :]
        t = self.torch.from_numpy(v.copy()).to(self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
