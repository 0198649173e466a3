"""-m gpu: the multi-rank path of bench.py on a one-GPU box.  Two ranks
(torch.distributed.run, one process each) share device 0 and talk over gloo
(BENCH_SHARE_GPU=1; on a multi-GPU node every rank has its own GPU and the
backend is RCCL): restarts sharded over the ranks, no data-path collective,
barrier + max-over-ranks timing, allreduce(MIN) of the incumbent."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_share_the_gpu():
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--cpu-seconds", "0", "--exchange-every", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 6
    # every rank swept its 64 instances x 1000 rows in each of the 6 steps
    assert rec["config"]["row_updates"] == 2 * 6 * 64 * 1000
    assert rec["value"] > 0 and rec["roofline"]["frac"] > 0
    assert rec["config"]["best_objective"] is None or rec["config"]["best_objective"] > 0
    # the incumbent went round (allreduce(min) + broadcast of the winner's x) every
    # second step inside the timed region and once at the end
    assert rec["config"]["incumbent_exchanges"] == 4 and rec["config"]["world_size"] == 2


def test_two_ranks_on_the_telebus_shape():
    """BASELINE configs[4] (inequalities-Z, 256 restarts per GPU) over two ranks on
    the one GPU of the box: restarts sharded, the packed-key exchange inside the timed
    region."""
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c5", "--batch", "256", "--steps", "4",
           "--warmup", "1", "--cpu-seconds", "0", "--exchange-every", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["steps"] == 4
    assert rec["config"]["restarts_per_gpu"] == 256 and rec["config"]["world_size"] == 2
    assert rec["config"]["row_updates"] > 0 and rec["value"] > 0
    assert rec["config"]["incumbent_exchanges"] == 3
