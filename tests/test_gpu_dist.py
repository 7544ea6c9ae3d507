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


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_match_one(tmp_path, world):
    """Permutation + bootstrap of mct / rb / mb and the split-half tests of mct / mb / rb, sharded over
    `world` gloo ranks (three ranks: ragged shards and, for a single split, a rank without items)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dist_gpu_worker as worker
    single = worker.run(True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "ranks.npz"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(out)]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    many = np.load(out)
    assert any(key.startswith("sh_mb_1_") for key in single)
    for key, want in single.items():
        np.testing.assert_allclose(many[key], want, rtol=1e-9, atol=1e-11, err_msg=key)


def _bench_line(proc):
    import json
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert "[verify] 2 rank(s), strong job" in proc.stderr and "[verify] 2 rank(s), weak job" in proc.stderr
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["backend"] == "gloo"
    assert line["scaling"] == "strong" and line["value"] == line["strong"]["value"] > 0
    assert line["strong"]["job"].startswith("1000 perm + 1000 boot in total, 500 + 500")
    assert line["weak"]["job"].startswith("2000 perm + 2000 boot in total, 1000 + 1000")
    assert line["roofline"]["bound"] == "mfma" and line["roofline"]["resamples_per_launch"] == 500
    assert line["cpu_baseline"] is None
    return line


def test_bench_self_launch_two_ranks_verified():
    """`python bench.py --gpus 2` exactly as the driver types it for N = 1 -- no launcher, no
    RANK / WORLD_SIZE in the environment: the script starts its own two ranks (gloo here, on the
    test box's one device; RCCL on a multi-GPU node).  Both jobs (strong: the fixed 1000 + 1000
    sharded; weak: 1000 + 1000 per rank) must hand rank 0 the same numbers as a single-rank pass
    over the whole job (--verify), with the bootstrap's collectives overlapping the permutation
    kernel."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                            "MASTER_PORT")}
    env["PLSR_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu", "--verify"]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    _bench_line(proc)


def test_bench_under_launcher_weak_value():
    """The other start the contract names: under torch.distributed.run; --scaling weak makes the
    per-GPU-fixed job the headline value."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PLSR_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--scaling", "weak"]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    import json
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "weak" and line["value"] == line["weak"]["value"] > 0
    assert line["roofline"]["resamples_per_launch"] == 1000
