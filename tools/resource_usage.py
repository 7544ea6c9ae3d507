"""Parse hipcc's -Rpass-analysis=kernel-resource-usage remarks into one row per kernel instance.

Usage:  python tools/resource_usage.py [remarks.txt]      (no argument: compiles plsr_abi.hip for gfx950
with the remarks on and parses them -- about a minute).  Prints the instances that use scratch and
returns rows of (demangled name, VGPRs, AGPRs, scratch bytes / lane, waves / SIMD, LDS bytes)."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plspy_amd", "csrc")


def compile_remarks(out_path="/tmp/plsr_remarks.txt"):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/plsr_remarks.so", "plsr_abi.hip", "plsr_rng.cpp"]
    with open(out_path, "w") as f:
        subprocess.run(cmd, cwd=CSRC, stderr=f, check=True)
    return out_path


def parse(path):
    txt = open(path).read()
    blocks = re.split(r"remark: Function Name: ", txt)[1:]
    names = [b.split(" [-Rpass")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    rows = []
    for name, b in zip(dem, blocks):
        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         waves=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    rows = parse(sys.argv[1] if len(sys.argv) > 1 else compile_remarks())
    bad = [r for r in rows if r["scratch"] > 0]
    for r in sorted(bad, key=lambda r: -r["scratch"]):
        print(f'{r["scratch"]:5d} B/lane  v{r["vgpr"]:3d} a{r["agpr"]:3d} occ {r["waves"]}  {r["name"][:110]}')
    print(f"{len(rows)} kernel instances, {len(bad)} with scratch")
