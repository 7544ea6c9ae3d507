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
