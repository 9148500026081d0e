"""Design studies under tools/ that carry numbers quoted in DESIGN.md: they must keep running and keep saying what the document says."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tree_factor_design_study():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "proto_tree_factor.py")], cwd=ROOT)
    hum, dog = json.loads(out)
    assert (hum["dofs"], hum["nnz_L"], hum["elimination_levels"], hum["multiply_adds_sparse"]) == (34, 310, 13, 1432)
    assert (dog["dofs"], dog["nnz_L"], dog["elimination_levels"], dog["multiply_adds_sparse"]) == (64, 808, 22, 5480)
    assert max(hum["longest_lane_chain_per_level_fma"]) <= 30 and max(dog["longest_lane_chain_per_level_fma"]) <= 36
    assert hum["rel_err_factor"] < 1e-12 and dog["err_A"] < 1e-9
