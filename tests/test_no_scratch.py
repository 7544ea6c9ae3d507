"""No shipped kernel instance may use scratch memory (register spills): a spilling instance is a
silent performance cliff next to the tuned shapes (round 1 had 28 of them, e.g. every LDS-fed mct
bootstrap with k in {5, 16, 20, 24, ...} and n > 64).  The check compiles the library for gfx950 with
-Rpass-analysis=kernel-resource-usage (hipcc cross-compiles without a GPU, about a minute) and
parses the remarks (tools/resource_usage.py)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_every_kernel_instance_is_free_of_scratch(tmp_path):
    import resource_usage
    rows = resource_usage.parse(resource_usage.compile_remarks(str(tmp_path / "remarks.txt")))
    assert len(rows) > 100, "remarks not parsed"
    names = " ".join(r["name"] for r in rows)
    for must in ("project_boot_reg_kernel<15, false, 2>", "project_kernel<4, 1, 4>", "project_kernel<6, 1, 0>",
                 "gram_kernel<6, 1, true, 3, 3>", "gram_kernel<5, 1, true, 2, 3>", "item_beh_kernel<5, 6, 8, true>", "item_fused2_kernel", "latent_kernel<3, 1, 1, 8>"):
        assert must in names, f"{must} not among the compiled instances"
    bad = [(r["name"], r["scratch"]) for r in rows if r["scratch"] != 0]
    assert not bad, f"kernel instances with scratch: {bad}"
    # the headline instances keep their occupancy: two waves per SIMD
    head = [r for r in rows if "project_boot_reg_kernel<15, false, 2>" in r["name"] or
            "project_perm_reg_kernel<15>" in r["name"]]
    assert head and all(r["waves"] >= 2 for r in head), head
