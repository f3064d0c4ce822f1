"""bench.py's N > 1 path on the ONE GPU a test box has: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`), both contexts on device 0 (INGVIO_DEVICE=0) and
gloo carrying the barrier / max / gather (RCCL refuses two ranks on one device; INGVIO_DIST_BACKEND=gloo).  Checks the contract
of the line, not a scaling figure: one JSON line from rank 0 only, n_gpus = 2, one per-rank time each, value = all ranks'
updates / the slowest rank's time, and the in-run oracle cross-check of BOTH ranks' first filter."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, port):
    env = dict(os.environ, INGVIO_DIST_BACKEND="gloo", INGVIO_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu", "--no-aux", "--steps", "3", "--warmup", "1"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0 only): %r" % lines           # rank 1 prints nothing
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("extra,port", [(["--batch", "64"], 29571), (["--config", "5", "--batch", "4", "--state", "literal"], 29572)])
def test_bench_two_ranks_on_one_gpu(extra, port):
    d = run_bench(extra, port)
    B = int(extra[extra.index("--batch") + 1])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["metric"] == "ekf_updates_per_sec"
    assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert d["config"]["filters_per_gpu"] == B and "2 rank(s)" in d["config"]["parallelism"]
    # value = the units ALL ranks processed / the slowest rank's time: ms_per_step is the max over ranks, value follows from it
    assert d["ms_per_step"] >= max(d["per_rank_ms_per_step"]) * (1 - 1e-4)
    assert abs(d["value"] - 2 * B / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-4
    assert d["results_finite"] is True
    par = d["parity_vs_oracle"]
    assert par["sample"] == 2 and len(par["per_rank_rel_cov_err"]) == 2            # filter 0 of EACH rank against the oracle
    assert par["accept_mask_equal"] is True and par["max_rel_cov_err"] < 1e-6
    assert d["cpu_baseline"] is None                                              # timed on rank 0 at N = 1 only
