"""Every `--scene imitate` arg file the reference ships parses, creates a context and steps (emulator build, no GPU).

Skipped when the reference checkout is not present (e.g. on the GPU box)."""
import glob
import os

import numpy as np
import pytest

from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv

REF = "/root/reference"


def _imitate_arg_files():
    out = []
    for f in sorted(glob.glob(os.path.join(REF, "args", "*.txt"))):
        p = model.ArgParser([])
        p.load_file(f)
        if p.str("scene", "") == "imitate":
            out.append(os.path.relpath(f, REF))
    return out


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "args")), reason="reference checkout not present")
def test_all_imitate_arg_files_run(emu_lib):
    files = _imitate_arg_files()
    assert len(files) == 44
    for f in files:
        t = model.load_scene_from_args(["--arg_file", f], data_root=REF)
        env = BatchEnv(t, 2, precision=64, lib_path=emu_lib)
        env.reset()
        out = env.step(None, 1.0 / 600, 2, open_loop=True)
        assert np.isfinite(out["state"]).all() and np.isfinite(out["reward"]).all(), f
        assert out["state"].shape == (2, t.state_dim) and env.A == t.action_dim, f
        env.close()
