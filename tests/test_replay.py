"""The Bullet lever, both halves: tools/replay_dump.py records a rollout through the drop-in `DeepMimicCore` on the SWIG surface,
tools/ref_replay.py -- which imports numpy and `DeepMimicCore` only -- replays it on whatever module has that name and prints the distance.  Here the
module is this repository's drop-in, so every difference must be EXACTLY 0: seed -> reference generator -> reset clip time, controller clock, actions,
state vector, reward, episode flags, through episode ends and resets.  Whoever owns a DeepMimicCore + Bullet 2.88 build runs the same file with
their module on PYTHONPATH; what it prints then is Bullet's rigid-body step against DM-physics (DeepMimicCore/Main.cpp:97-124, DeepMimic.py:62-80)."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "deepmimic_amd", "compat"))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden", "replay")


def _env(monkeypatch, lib, physics):
    monkeypatch.setenv("DM_HIP_LIB", lib); monkeypatch.setenv("DM_PHYSICS", str(physics)); monkeypatch.setenv("DM_RNG", "reference")
    monkeypatch.setenv("DM_PRECISION", "32"); monkeypatch.delenv("DM_FACADE_SHARED", raising=False); monkeypatch.delenv("DM_DATA_ROOT", raising=False)


def test_ref_replay_imports_nothing_from_this_repository():
    src = open(os.path.join(ROOT, "tools", "ref_replay.py")).read()
    imports = [l.split("#")[0].strip() for l in src.splitlines() if l.strip().startswith(("import ", "from "))]
    assert sorted(set(imports)) == sorted({"import argparse", "import json", "import sys", "import numpy as np", "from DeepMimicCore import DeepMimicCore"}), imports
    assert "deepmimic_amd" not in "".join(imports)


@pytest.mark.parametrize("arg_file,physics,stream,steps", [("args/run_humanoid3d_walk_args.txt", 1, "A1", 12), ("args/train_humanoid3d_spinkick_args.txt", 2, "A2", 18)])
def test_record_then_replay_is_identical_emulator(emu_lib, monkeypatch, tmp_path, arg_file, physics, stream, steps):
    import ref_replay
    import replay_dump
    from deepmimic_amd import model
    _env(monkeypatch, emu_lib, physics)
    t = model.load_asset(model.ARG_FILE_ASSETS[arg_file])
    path = str(tmp_path / "b.npz")
    b = replay_dump.record(["--arg_file", arg_file], t, steps, stream, 5, path, physics=physics, lib_path=emu_lib)
    if "spinkick" in arg_file:
        assert len(b["end_steps"]) >= 1 and b["end_flags"][0][1] <= 20      # the 0.5 s training episode limit of the arg file: an episode end + a reference-order reset inside the bundle
    rep = ref_replay.replay(ref_replay.load_bundle(path))
    assert rep["identical"], {k: rep[k] for k in rep if k not in ("steps", "ends")}
    assert rep["max_d_pose"] == 0 and rep["max_d_vel"] == 0 and rep["max_d_reward"] == 0 and rep["max_d_time"] == 0 and rep["first_differing_flag"] is None
    assert rep["episode_ends_here"] == rep["episode_ends_bundle"] == len(b["end_steps"])
    # a different seed starts elsewhere on the clip, and the tool says so from the first boundary on: it measures
    bb = ref_replay.load_bundle(path); bb["meta"]["seed"] = 6
    rep2 = ref_replay.replay(bb)
    assert not rep2["identical"] and rep2["steps"][0]["d_state"] > 1e-4
    # the other rigid-body step from the same seed: identical start, then a small, finite distance -- the shape of a Bullet owner's report
    monkeypatch.setenv("DM_PHYSICS", str(3 - physics))
    rep3 = ref_replay.replay(ref_replay.load_bundle(path))
    assert rep3["steps"][0]["d_state"] == 0 and rep3["steps"][0]["d_time"] == 0 and not rep3["identical"] and 0 < rep3["max_d_state"] < 10


def _golden():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def test_golden_bundles_are_the_documented_set():
    names = {os.path.basename(p) for p in _golden()}
    assert names == {"%s_v%d_%s.npz" % (n, v, s) for n in ("walk", "spinkick", "dog") for v in (1, 2) for s in ("A0", "A1", "A2")}
    import ref_replay
    for p in _golden():
        b = ref_replay.load_bundle(p)
        assert b["meta"]["format"] == "deepmimic-replay-2" and b["actions"].shape[0] == 60 and b["states"].shape[0] == 60
        assert b["meta"]["physics"] == "DM-physics v%s" % os.path.basename(p).split("_v")[1][0]


@pytest.mark.gpu
@pytest.mark.parametrize("path", _golden(), ids=lambda p: os.path.basename(p)[:-4])
def test_committed_bundles_replay_identically_on_the_hip_library(hip_lib, monkeypatch, path):
    """the committed bundles were recorded on libdm_hip.so (fp32 production kernels); the same kernels must reproduce them bit for bit (re-record with
    `python tools/replay_dump.py --golden tests/golden/replay` on an MI355X after a change that moves the kernels' rounding)"""
    import ref_replay
    b = ref_replay.load_bundle(path)
    _env(monkeypatch, hip_lib, int(b["meta"]["physics"][-1]))
    rep = ref_replay.replay(b)
    assert rep["identical"], {k: rep[k] for k in rep if k not in ("steps", "ends")}
