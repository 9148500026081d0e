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
        model.load_scene_from_args(["--scene", "sim_char"], data_root=REF)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "args")), reason="reference checkout not present")
def test_goal_scene_arg_files_run(emu_lib):
    """`--scene heading_amp` / `target_amp` / `heading_amp_getup` / `strike_amp` / `dribble_amp`: every arg file whose dataset
    is complete in the reference checkout loads (multi-clip `--kin_ctrl clips`, enable_rand_rot_reset), creates a context, steps,
    and reports its goal vector.  (data/datasets/humanoid3d_clips_locomotion.txt and humanoid3d_clips_walk_punch.txt name clips
    under data/motions/{long,sie,amass} that the checkout does not ship: those arg files cannot load in the reference either.)"""
    ran, missing = [], []
    for f in sorted(glob.glob(os.path.join(REF, "args", "*.txt"))):
        p = model.ArgParser([]); p.load_file(f)
        if p.str("scene", "") not in model.GOAL_SCENES:
            continue
        rel = os.path.relpath(f, REF)
        try:
            t = model.load_scene_from_args(["--arg_file", rel], data_root=REF)
        except FileNotFoundError:
            missing.append(rel); continue
        assert t.goal_kind in (1, 2, 3, 4, 5) and t.cfg.kin_ctrl == "clips", rel    # (run_* files leave enable_rand_rot_reset off)
        env = BatchEnv(t, 2, precision=64, lib_path=emu_lib, seed=1)
        assert env.S == t.state_dim, rel
        assert env.G == t.goal_dim and env.amp_size > 0, rel
        env.reset()
        out = env.step(np.zeros((2, env.A), np.float32), 1.0 / 600, 2, amp=True)
        assert np.isfinite(out["goal"]).all() and out["goal"].shape == (2, t.goal_dim) and np.isfinite(out["amp_obs"]).all(), rel
        assert np.isfinite(env.amp_expert_clips(3)).all(), rel
        env.close(); ran.append(rel)
    # 6 heading / target files, the 2 heading_amp_getup files and the 2 dribble files over the zombie clip run; the locomotion-dataset
    # files (4 heading / target + 2 dribble) and the 2 strike files lack their clips
    assert len(ran) == 10 and len(missing) == 8, (ran, missing)
    assert sum("heading_getup" in r for r in ran) == 2 and sum("strike" in r for r in missing) == 2 and sum("dribble" in r for r in ran) == 2


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "args")), reason="reference checkout not present")
def test_kin_char_arg_files_play_back(monkeypatch):
    """`--scene kin_char` (the two play_motion arg files): the viewer's motion playback, served by the facade on the host -- no agent, the
    character's clock as scene time, the pose of the clip at that time"""
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepmimic_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from DeepMimicCore import DeepMimicCore
    monkeypatch.setenv("DM_DATA_ROOT", REF)
    for f in ("args/play_motion_humanoid3d_args.txt", "args/play_motion_dog3d_args.txt"):
        core = DeepMimicCore.cDeepMimicCore(False)
        core.ParseArgs(["--arg_file", f]); core.Init()
        assert core.GetName() == "Kinematic Char" and not core.IsRLScene() and core.GetNumAgents() == 0
        p0 = np.array(core.GetKinPose())
        for _ in range(30):
            core.Update(1.0 / 60)
        assert abs(core.GetTime() - 0.5) < 1e-12 and not core.IsEpisodeEnd() and core.CheckValidEpisode()
        p1 = np.array(core.GetKinPose())
        assert np.isfinite(p1).all() and np.abs(p1 - p0).max() > 1e-3 and abs(np.linalg.norm(p1[3:7]) - 1) < 1e-9
        core.Reset()
        assert core.GetTime() == 0.0 and np.array_equal(np.array(core.GetKinPose()), p0)
        with pytest.raises(RuntimeError, match="no agents"):
            core.RecordState(0)
        core.Shutdown()


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "util", "arg_parser.py")), reason="reference checkout not present")
def test_arg_parser_agrees_with_the_references_python_parser():
    """model.ArgParser against the reference's own util/arg_parser.py (imported from the checkout), on every shipped arg file and on token lists with
    comments, repeated keys (first wins), empty value lists and short dashes"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_arg_parser", os.path.join(REF, "util", "arg_parser.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    files = sorted(glob.glob(os.path.join(REF, "args", "*.txt")))
    assert len(files) == 98
    for f in files:
        r = mod.ArgParser(); assert r.load_file(f)
        p = model.ArgParser([]); assert p.load_file(f)
        assert p.table == r._table, f
    rng = np.random.default_rng(0)
    vocab = ["--scene", "--a", "--bb", "--", "-x", "--num_update_substeps", "#c", "# comment", "imitate", "1", "-2.5", "true", "T", "0", "--scene", "--k1", "--k2", "v"]
    for _ in range(300):
        toks = [vocab[i] for i in rng.integers(0, len(vocab), size=int(rng.integers(0, 14)))]
        r = mod.ArgParser(); r.load_args(toks)
        p = model.ArgParser(toks)
        assert p.table == r._table, toks
        for k in p.table:
            assert p.str(k, "d") == r.parse_string(k, "d") if p.table[k] else True
            if p.table[k]:
                assert p.bool(k, False) == r.parse_bool(k, False)
