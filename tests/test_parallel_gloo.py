"""N > 1 path on CPU: world_size-2 gloo run of the same sharding / barrier / max / gather plumbing
bench.py uses with RCCL, with the oracle standing in for the per-rank GPU work."""
import os
import socket
import sys

import numpy as np

from conftest import ROOT, frame_from_golden, load_golden


def test_shard_partition_is_exact():
    from ingvio_amd.parallel import shard
    for total in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = shard(total, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(total))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ingvio_amd.parallel import Group, shard
    from oracle import oracle as orc
    g = Group(backend="gloo")
    z = load_golden("msckf_small")
    names = ["stereo_ragged", "mono_ragged", "stereo_cap", "selected_q10", "keyframe_like"]
    s, c = shard(len(names), world, rank)
    g.barrier()
    summ = np.zeros(len(names))
    for i in range(s, s + c):
        fr = frame_from_golden(z, names[i] + "_")
        kw = dict(zip(("max_accept", "compress_rule", "selected_variant"), [int(x) for x in z[names[i] + "_kw"]]))
        cov = orc.Cov(z[names[i] + "_P"])
        cov.msckf_update(fr, **kw)
        summ[i] = np.trace(cov.P)
    g.barrier()
    tmax = g.max_over_ranks(1.0 + rank)
    tot = g.sum_over_ranks(c)
    allv = g.gather_summaries(summ)
    per_rank = g.gather_scalars(0.5 + rank)          # bench.py's per-rank ms_per_step
    lo, hi = g.shard(len(names))
    assert (lo, hi) == (s, s + c) and per_rank == [0.5, 1.5]
    q.put((rank, tmax, tot, allv))
    g.close()


def test_two_rank_gloo_matches_single_process():
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = load_golden("msckf_small")
    names = ["stereo_ragged", "mono_ragged", "stereo_cap", "selected_q10", "keyframe_like"]
    expect = np.array([np.trace(z[n + "_Pn"]) for n in names])
    for rank, tmax, tot, allv in res:
        assert tmax == 2.0 and tot == len(names)
        got = allv.sum(axis=0)                      # every frame is computed by exactly one rank
        assert allv.shape == (2, len(names)) and np.allclose(got, expect, rtol=1e-9)
        assert (np.count_nonzero(allv, axis=0) == 1).all()
