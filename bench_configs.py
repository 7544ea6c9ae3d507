#!/usr/bin/env python3
"""Secondary benchmark: the other BASELINE.json configurations through the public
``plspy_amd.PLS`` / resample seams (not the driver's bench -- that is bench.py).

  --config 3   rb  X=(120x200000) Y=(120x8)  groups [20,20] x 3, perm + boot
  --config 4   mb  same data, bscan=[1,2], split-half (splits/s)
  --config 5   mct X=(240x500000) groups [20]x4 x 3, perm + boot
  --config 2   mct X=(60x200000) through PLS() end to end (host draws, observed SVD included)
  --config 6   mb  config 4's data, permutation + bootstrap (not in BASELINE.json)
  --config 7   mct X=(128x200000) groups [8]x8 x 2 (k = 16: LDS-fed bootstrap kernel, period 4)
  --config 8   mct X=(64x200000)  groups [16,16] x 2 (sixteen k-steps: register-resident kernels at nk = 16)

``--gpus N`` (N > 1): one rank per GPU over RCCL, resamples / splits sharded by plspy_amd/dist.py exactly as
BASELINE.json's configs 4 ("sharded 4xMI355X") and 5 ("sharded 8xMI355X") ask.  Without a launcher's RANK /
WORLD_SIZE in the environment the script starts its own ranks through torch.distributed.run BEFORE anything
touches the GPU (the parent never imports torch), relays rank 0's line and exits with the child's code; rank 0
prints the line, which records ``backend`` and ``world_size``.

``--count`` sets the number of resamples per loop (BASELINE's full counts are
2000/2000, 1000 splits, 5000/5000); rates are per second of the resampling
phase, host index generation and operator construction included, from the second
of two runs of each phase (the first, cold run is reported beside it).  Prints one
JSON object per run."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def data(n, p, b=0):
    X = np.random.RandomState(0).randn(n, p)
    Y = np.random.RandomState(1).randn(n, b) if b else None
    return X, Y


COLD = {}          # label -> seconds of the first (cold) run of a timed phase


def timed(fn, label=None):
    """Runs fn twice and returns the second run's result and time: the first run of a phase in
    a fresh process also pays the first-use device allocations of its scratch (hipMalloc and
    first touch of several GiB: 0.2-0.7 s on some boxes of the pool, 0 on others), which is a
    property of the box, not of the path.  The cold time is reported beside it (COLD)."""
    import torch
    import torch.distributed as td
    multi = td.is_available() and td.is_initialized()
    for rep in range(2 if label else 1):
        torch.cuda.synchronize()
        if multi:
            td.barrier()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        if multi:                      # a phase ends when its last rank does
            td.barrier()
        dt = time.perf_counter() - t0
        if label and rep == 0:
            COLD[label] = dt
    return out, dt


def self_launch(args):
    """N fresh rank processes through torch.distributed.run (as bench.py does); this process has not imported
    torch and never will."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, cwd=os.path.dirname(os.path.abspath(__file__)), env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True)
    ap.add_argument("--count", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = None
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        import torch.distributed as td
        # PLSR_DIST_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks (ranks then share devices)
        backend = os.environ.get("PLSR_DIST_BACKEND", "nccl")
        dev_id = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(dev_id)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_id}"))
        else:
            td.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    import plspy_amd
    from plspy_amd import split_half_resampling as sh
    from plspy_amd.bootstrap_permutation import ResampleTest
    from plspy_amd.engine import ProjectionEngine

    np.random.seed(1234)
    out = {"config": args.config, "device": torch.cuda.get_device_name(0), "n_gpus": world, "world_size": world,
           "backend": backend}
    if args.config in (2, 5, 7, 8):
        # 7 / 8: mct shapes that used to run spilling instances -- k = 16, n = 128 (LDS-fed bootstrap,
        # period 4) and n = 64 (the sixteen-step register-resident kernels)
        nc = 3
        if args.config == 2:
            n, p, groups = 60, 200_000, [10, 10]
        elif args.config == 5:
            n, p, groups = 240, 500_000, [20] * 4
        elif args.config == 7:
            n, p, groups, nc = 128, 200_000, [8] * 8, 2
        else:
            n, p, groups, nc = 64, 200_000, [16, 16], 2
        R = args.count or (5000 if args.config == 5 else 1000)
        X, _ = data(n, p)
        res, t_all = timed(lambda: plspy_amd.PLS(X, groups, nc, num_perm=R, num_boot=R, pls_method="mct"), "pls_call")
        out.update(workload=f"mct X={n}x{p}, groups {groups} x {nc}, {R} perm + {R} boot via PLS()", seconds_total=t_all,
                   resamples_per_s_end_to_end=2 * R / t_all, s=res.s.tolist())
        # resampling phases alone (observed decomposition excluded)
        eng = ProjectionEngine(X)
        U, s, V = res.V, res.s.copy(), res.U
        co = np.array([[g] * nc for g in groups])
        rt, t_rs = timed(lambda: ResampleTest._create("mct", X, None, U, s.copy(), V, co, 0, nperm=R, nboot=R,
                                                      Tvsc_orig=np.zeros((len(s), len(s))), engine=eng), "resampling")
        out.update(seconds_resampling=t_rs, resamples_per_s=2 * R / t_rs)
    elif args.config == 3:
        R = args.count or 2000
        X, Y = data(120, 200_000, 8)
        res, t_obs = timed(lambda: plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=0, num_boot=0, pls_method="rb"), "observed")
        U, s, V = res.V, res.s.copy(), res.U
        co = np.array([[20] * 3, [20] * 3])
        eng = ProjectionEngine(X)
        rt, t_perm = timed(lambda: ResampleTest._create("rb", X, Y, U, s.copy(), V, co, None, nperm=R, nboot=0,
                                                        engine=eng), "perm")
        rt, t_boot = timed(lambda: ResampleTest._create("rb", X, Y, U, s.copy(), V, co, None, nperm=0, nboot=R,
                                                        lvcorrs_orig=res.lvcorrs, engine=eng), "boot")
        out.update(workload=f"rb X=120x200000 Y=120x8 (k=48), {R} perm + {R} boot", seconds_observed=t_obs,
                   seconds_perm=t_perm, seconds_boot=t_boot, perms_per_s=R / t_perm, boots_per_s=R / t_boot,
                   resamples_per_s=2 * R / (t_perm + t_boot))
    elif args.config == 4:
        S = args.count or 1000
        X, Y = data(120, 200_000, 8)
        co = np.array([[20] * 3, [20] * 3])
        eng = ProjectionEngine(X)
        kw = dict(mctype=0, bscan=[1, 2], engine=eng)
        _, t_tt = timed(lambda: sh.split_half_test_train("mb", X, Y, co, S, **kw), "test_train")
        _, t_sh = timed(lambda: sh.split_half("mb", X, Y, co, S, lv=2, CI=0.95, **kw), "split_half")
        out.update(workload=f"mb X=120x200000 Y=120x8 bscan=[1,2] (k=38), {S} splits (tt + sh, real + null)",
                   seconds_test_train=t_tt, seconds_split_half=t_sh, splits_per_s=S / (t_tt + t_sh))
    elif args.config == 6:
        # not a BASELINE configuration: the multiblock permutation / bootstrap
        # on config 4's data, which BASELINE only exercises through split-half
        R = args.count or 500
        X, Y = data(120, 200_000, 8)
        mk = lambda nperm, nboot: plspy_amd.PLS(X, [20, 20], 3, Y=Y, num_perm=nperm, num_boot=nboot,
                                                pls_method="mb", bscan=[1, 2])
        _, t_obs = timed(lambda: mk(0, 0), "observed")
        _, t_p = timed(lambda: mk(R, 0), "perm")
        _, t_b = timed(lambda: mk(0, R), "boot")
        t_perm, t_boot = t_p - t_obs, t_b - t_obs                # the observed decomposition is in both
        out.update(workload=f"mb X=120x200000 Y=120x8 bscan=[1,2] (k=38), {R} perm + {R} boot",
                   seconds_observed=t_obs, seconds_perm=t_perm, seconds_boot=t_boot, perms_per_s=R / t_perm,
                   boots_per_s=R / t_boot)
    out["seconds_cold_first_run"] = COLD
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
