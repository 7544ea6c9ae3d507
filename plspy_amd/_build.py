"""Builds the HIP C-ABI library in-tree (plspy_amd/csrc/libplsr_hip.so).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
build container; the resulting .so travels to the GPU box with the repo
snapshot.  There is one code path and one target: no fallbacks."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libplsr_hip.so")
SOURCES = ["plsr_abi.hip", "plsr_rng.cpp"]
HEADERS = ["plsr_project.hip.h", os.path.join("..", "..", "include", "plsr.h")]  # + every *.h in csrc (see _stale)
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile libplsr_hip.so if missing or older than its sources."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
