"""CPU: the register / scratch / LDS budget of the hot kernels, read from the code objects inside the BUILT library
(tools/kernel_resources.py: amdhsa.kernels metadata, no compilation), against the committed baseline
profiles/kernel_resources.json.  A hot kernel that gains scratch or register spills, loses register-limited occupancy or
grows its LDS block fails here instead of showing up as an unexplained slowdown on the GPU (rank-32 Newton-Schulz
had silently dropped from two waves per SIMD to one).  After an intended change: python tools/kernel_resources.py --update."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import kernel_resources as kr  # noqa: E402


@pytest.fixture(scope="module")
def now():
    if not os.path.exists(kr.LIB):
        # a CPU-only checkout without hipcc has nothing to inspect: the driver's own order is build() first, then the tests
        pytest.skip(f"{kr.LIB} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return {r["kernel"]: r for r in kr.resources()}


def test_every_kernel_is_read_and_gfx950_only(now):
    assert len(now) > 200 and all(r["vgpr_count"] > 0 for r in now.values())
    assert any(k.startswith("k_sweep<") for k in now) and any(k.startswith("k_slab_unimodal_v4<") for k in now)


def test_hot_kernels_keep_their_budget(now):
    base = json.load(open(kr.BASELINE))
    worse = []
    for k in base["hot"]:
        b = base["kernels"][k]
        if k not in now:
            worse.append(f"{k}: in the baseline but not in the library (renamed? run tools/kernel_resources.py --update)")
            continue
        n = now[k]
        if n["scratch_bytes"] > b["scratch_bytes"] or n["vgpr_spill"] > b["vgpr_spill"]:
            worse.append(f"{k}: scratch {b['scratch_bytes']} -> {n['scratch_bytes']} B/lane, spilled VGPRs {b['vgpr_spill']} -> {n['vgpr_spill']}")
        if n["occupancy"] < b["occupancy"]:
            worse.append(f"{k}: {b['vgpr_count']} -> {n['vgpr_count']} VGPRs, register-limited occupancy {b['occupancy']} -> {n['occupancy']}")
        if n["lds_bytes"] > b["lds_bytes"]:
            worse.append(f"{k}: LDS {b['lds_bytes']} -> {n['lds_bytes']} B per workgroup")
    assert not worse, "\n".join(worse)
    new_hot = sorted(k for k in now if kr.is_hot(k) and k not in base["kernels"])
    assert not new_hot, f"hot kernels without a baseline entry (tools/kernel_resources.py --update): {new_hot}"


def test_the_kernels_of_the_baseline_configurations_fit_their_design_points(now):
    # config 3's one-pass sweep: one wave per SIMD by design (344 of 512 registers), no scratch
    sweep = now["k_sweep<1, 1, 1, 2, 4, false, false, false, false>"]
    assert sweep["scratch_bytes"] == 0 and sweep["vgpr_spill"] == 0 and sweep["occupancy"] == 1
    # rank-32 Newton-Schulz (config 5): TWO waves per SIMD - one wave issues an fp64 MFMA every ~143 cycles, the pipe takes one per 64
    assert now["k_pf2_algebra_ns<2, true>"]["occupancy"] >= 2
    # unimodal regressions, throughput forms: two waves per SIMD (the 20 KB LDS ring allows no more)
    # (the default form, <3, 4>: four independent waves per workgroup with their rings in 80 KB of DYNAMIC LDS - two workgroups per CU)
    for form in ("0, 1", "3, 1", "3, 4"):
        u = now[f"k_slab_unimodal_v4<{form}>"]
        assert u["occupancy"] >= 2 and u["scratch_bytes"] == 0 and u["lds_bytes"] <= 20480
    # rank <= 16 Newton-Schulz (config 4): four slabs per workgroup, one per SIMD - at most 256 registers, no scratch
    ns = now["k_pf2_algebra_ns<1, true>"]
    assert ns["occupancy"] >= 2 and ns["scratch_bytes"] == 0
