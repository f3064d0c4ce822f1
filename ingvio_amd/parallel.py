"""Multi-GPU plumbing: the hot path shards over independent filters/frames, so the
only collectives are the timing barrier / max and one end-of-run gather of per-rank summaries.
One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU).
Optional second mode (stress only): ONE filter whose features are dealt to the ranks, with exactly one exchange
step per frame - the sum of the ranks' [A | b] (sharded_frame_update)."""
import os


def shard(total, world, rank):
    """Static block partition frame_id -> rank = frame_id // ceil(total / world).  Returns (start, count)."""
    per = (total + world - 1) // world
    start = min(rank * per, total)
    return start, max(0, min(per, total - start))


class Group:
    """Thin wrapper: works for world == 1 without importing torch."""

    def __init__(self, backend=None, force_init=False):
        """force_init: create the process group even at world size 1 (exercises the collective backend itself).
        Environment: INGVIO_DIST_BACKEND = nccl | gloo overrides the backend choice; INGVIO_DEVICE = ordinal puts every rank's
        context on that device instead of LOCAL_RANK (both for exercising the N > 1 code path of bench.py / the sharded filter
        on a box with ONE GPU: RCCL refuses two ranks on one device, gloo carries the barrier / max / gather instead)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device_index = int(os.environ["INGVIO_DEVICE"]) if os.environ.get("INGVIO_DEVICE") else self.local_rank
        backend = backend or os.environ.get("INGVIO_DIST_BACKEND") or None
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
                torch.cuda.set_device(self.device_index)
                self.device = torch.device("cuda", self.device_index)
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

    def sum_device(self, ptr, count, device_index=None):
        """In-place all-reduce(sum) of `count` doubles at DEVICE pointer `ptr` (a buffer of libingvio_hip.so): torch sees the
        memory through __cuda_array_interface__ and, with backend "nccl", RCCL reduces it in place over xGMI - no host staging.
        With the gloo backend (the two-ranks-on-one-GPU test) the view is copied to the host, reduced and copied back.  The
        caller must have synchronised the stream that produced the buffer; on return the collective has completed."""
        if self.dist is None:
            return
        idx = self.device_index if device_index is None else int(device_index)
        t = self.torch.as_tensor(_DeviceBuffer(ptr, count), device=self.torch.device("cuda", idx))
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            t.copy_(h)
        self.torch.cuda.synchronize(idx)


    def sum_array(self, a):
        """All-reduce(sum) of a float64 host array (the host-staged variant of the feature-sharded filter's exchange step)."""
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self.dist is None:
            return a
        t = self.torch.from_numpy(a.copy()).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()


class _DeviceBuffer:
    """Zero-copy view of a device buffer of `count` doubles for torch (``torch.as_tensor(_DeviceBuffer(...), device=...)``)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = dict(shape=(int(count),), typestr="<f8", data=(int(ptr), False), version=2, strides=None)


def shard_features(frame, world, rank):
    """The frame dict restricted to the features j = rank (mod world): every rank sees the same window and poses."""
    import numpy as np
    keep = np.arange(rank, len(frame["pf"]), world)
    out = dict(frame)
    for k in ("pf", "anchor", "obs_mask", "uv", "dof"):
        out[k] = np.asarray(frame[k])[keep]
    return out, keep


def sharded_frame_update(ctx, grp, b, step, frame, sigma, enable_gnss, sigma_cb, sigma_rw, restore_prior=False, device_exchange=True):
    """One frame of ONE filter (replicated prior on every rank, filter b of each rank's context) with its features dealt to the
    ranks: local propagate + clone + gate + Gram, ONE all-reduce of [A | b] (n x (n+1) doubles), then the identical solve + apply +
    marginalisation everywhere.  Returns (dx, accepted feature ids of this rank, rows).  `ctx` must hold only this filter (batch 1)
    or the call applies the staged frames of the whole batch."""
    import numpy as np
    local, keep = shard_features(frame, grp.world, grp.rank)
    ctx.frame_stage(b, [step], [local], sigma, enable_gnss, sigma_cb, sigma_rw, max_accept=0, compress_rule=1)
    ctx.frame_run_phase(1, restore_prior=restore_prior)
    _, acc, _ = ctx.frame_fetch(b, 1)
    if device_exchange:
        # the ONE exchange step, on the device: [A | b | n_accepted] is summed over the chunk partials into one buffer of the
        # library, all-reduced in place through a zero-copy torch view (RCCL / xGMI with backend "nccl") and committed
        ptr, count, n = ctx.info_reduce(b)
        grp.sum_device(ptr, count, getattr(ctx, "device", None))
        ctx.info_commit(b)
    else:                                                       # host-staged variant (round 2), kept as the cross-check
        A, bvec = ctx.debug_msckf_info(b)
        packed = np.concatenate([np.column_stack([A, bvec]).reshape(-1), [float(acc[0, :len(keep)].sum())]])
        tot = grp.sum_array(packed)
        n = A.shape[0]
        ctx.info_set(b, tot[:-1].reshape(n, n + 1), int(round(tot[-1])))
    ctx.frame_run_phase(2)
    dx, _, rows = ctx.frame_fetch(b, 1)
    return dx[0], keep[acc[0, :len(keep)] != 0], int(rows[0])
