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


def _amp_arg_files():
    single, dataset = [], []
    for f in sorted(glob.glob(os.path.join(REF, "args", "*.txt"))):
        p = model.ArgParser([])
        p.load_file(f)
        if p.str("scene", "") == "imitate_amp":
            (dataset if "datasets/" in p.str("motion_file", "") else single).append(os.path.relpath(f, REF))
    return single, dataset


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "args")), reason="reference checkout not present")
def test_imitate_amp_arg_files_run(emu_lib):
    """All 34 `--scene imitate_amp` arg files (every one is single-clip, `--kin_ctrl motion`) run.  Multi-clip datasets
    (`--kin_ctrl clips`, anim/ClipsController.cpp) only occur in the task scenes, which are refused with a clear error."""
    single, dataset = _amp_arg_files()
    assert len(single) == 34 and len(dataset) == 0
    for f in single:
        t = model.load_scene_from_args(["--arg_file", f], data_root=REF)
        assert t.cfg.scene == "imitate_amp"
        env = BatchEnv(t, 2, precision=64, lib_path=emu_lib)
        env.reset()
        out = env.step(None, 1.0 / 600, 2, open_loop=True, amp=True)
        assert out["amp_obs"].shape == (2, env.amp_size) and env.amp_size > 0 and np.isfinite(out["amp_obs"]).all(), f
        assert (out["reward"] == 0).all(), f
        ex = env.amp_expert(3)
        assert np.isfinite(ex).all(), f
        env.close()
    with pytest.raises(ValueError, match="accelerated path"):
        model.load_scene_from_args(["--arg_file", "args/train_amp_target_humanoid3d_locomotion_args.txt"], data_root=REF)
