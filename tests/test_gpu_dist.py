"""The N > 1 path of the real resampling classes on the GPU: two gloo ranks (sharing
the one device of the test box) shard the permutation and bootstrap resamples of a
task and a behaviour PLS, exchange, and must end with what one rank computes."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_match_one(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dist_gpu_worker as worker
    single = worker.run(True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "two_ranks.npz"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(out)]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    two = np.load(out)
    for key, want in single.items():
        np.testing.assert_allclose(two[key], want, rtol=1e-10, atol=1e-12, err_msg=key)


def test_bench_two_ranks_verified():
    """bench.py's N > 1 path (the driver's scaling run) as two gloo ranks on the test box's
    one device: the step that overlaps the bootstrap's collectives with the permutation kernel
    must hand rank 0 the same numbers as a single-rank pass over the whole job (--verify)."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PLSR_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--verify"]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert "[verify] 2 rank(s)" in proc.stderr
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["roofline"]["bound"] == "mfma" and line["cpu_baseline"] is None
