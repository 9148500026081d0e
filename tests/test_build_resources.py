"""Register / scratch budget of the compiled step kernels, read from the resource remarks the build leaves next to every object
(deepmimic_amd/csrc/build/k_f32_<family>.o.res, written by `make`: -Rpass-analysis=kernel-resource-usage).  A production kernel that
starts spilling is a silent 1.7x (dog3d: one divergent store in the level buffer of the tree factor cost 389 spilled VGPRs and nobody
saw it until the next profile) -- this holds the budgets the measurements in DESIGN.md section 6 / 7 were taken at.  Skipped when the
library was not built in this checkout."""
import os
import re

import pytest

BUILD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepmimic_amd", "csrc", "build")


def _res(family, prec="f32"):
    path = os.path.join(BUILD, "k_%s_%d.o.res" % (prec, family))
    if not os.path.exists(path):
        pytest.skip("no resource remarks (libdm_hip.so not built here)")
    txt = open(path).read()
    out = {}
    for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
        vals = [int(v) for v in re.findall(pat, txt)]
        assert vals, (path, key)
        out[key] = max(vals) if key != "occupancy" else min(vals)
    return out


# family ids: deepmimic_amd/csrc/dm_kernels.cpp
@pytest.mark.parametrize("family,what,occupancy,lds_max", [
    (0, "two characters per wavefront, plain (the headline kernel)", 2, 20480),
    (1, "two characters per wavefront, AMP / goal / perturbation instantiation", 2, 20480),
    (7, "ClsLarge, AMP", 2, 20480),
    (12, "ClsLargeTree (dog3d on its compiled topology), plain", 2, 20480),
    (13, "ClsLargeTree, AMP (dog3d imitate_amp)", 2, 20480),
])
def test_production_kernels_do_not_spill_vector_registers(family, what, occupancy, lds_max):
    r = _res(family)
    assert r["spill"] == 0, (what, r)
    assert r["occupancy"] == occupancy and r["lds"] <= lds_max, (what, r)          # 160 KB LDS per CU: 8 (16) waves need <= 20480 (10240) B each


@pytest.mark.parametrize("family,what,spill_max", [
    (3, "ClsBiped one per wavefront, plain: 20 spilled VGPRs in the prologue / epilogue (DESIGN.md section 6)", 24),
    (6, "ClsLarge (dense dog3d, DM_TREE=0), plain: 4 (DESIGN.md section 6)", 8),
    (4, "ClsBiped AMP (odd batches of the task scenes)", 16),
    (9, "ClsBipedObj (dribble_amp)", 16),
    (18, "ClsBiped, DM-physics v2", 32),
])
def test_secondary_kernels_stay_inside_their_measured_spill_budget(family, what, spill_max):
    r = _res(family)
    assert r["spill"] <= spill_max, (what, r)
