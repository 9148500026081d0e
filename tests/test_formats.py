"""On-disk formats (deepmimic_amd/formats.py) and the replay-bundle tool; device side = the CPU emulator build."""
import json
import os
import sys

import numpy as np
import pytest

from deepmimic_amd import formats, model, streams
from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_state_snapshot_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    p, v = rng.normal(size=43), rng.normal(size=43)
    f = str(tmp_path / "s.json")
    formats.write_state(f, p, v)
    txt = open(f).read()
    assert txt.startswith("{\n\"Pose\":[") and "\n\"Vel\":[" in txt and txt.endswith("\n}")      # BuildStateJson layout
    p2, v2 = formats.read_state(f, 43)
    assert np.array_equal(p, p2) and np.array_equal(v, v2)
    with pytest.raises(ValueError):
        formats.read_state(f, 44)
    open(f, "w").write('{"Pose": [1, 2]}')
    p3, v3 = formats.read_state(f)
    assert v3 is None and p3.tolist() == [1.0, 2.0]


def test_motion_clip_round_trip(tmp_path):
    t = model.load_asset("humanoid3d_walk")
    f = str(tmp_path / "m.txt")
    formats.write_motion(f, t.frames, t.loop)
    fr, loop = formats.read_motion(f)
    assert loop == t.loop and fr.shape == t.frames.shape
    assert np.array_equal(fr[:-1], t.frames[:-1]) and fr[-1, 0] == 0.0 and np.array_equal(fr[-1, 1:], t.frames[-1, 1:])
    d = json.load(open(f))
    assert d["Loop"] == "wrap" and d["EnableCycleSyncRootPos"] is True


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "data")), reason="reference checkout not present")
def test_motion_writer_reproduces_reference_clip_content(tmp_path):
    src = os.path.join(REF, "data", "motions", "humanoid3d_spinkick.txt")
    fr, loop = formats.read_motion(src)
    f = str(tmp_path / "m.txt")
    formats.write_motion(f, fr, loop)
    fr2, loop2 = formats.read_motion(f)
    assert loop2 == loop and np.array_equal(fr2[:-1], fr[:-1]) and np.array_equal(fr2[-1, 1:], fr[-1, 1:])


def test_pose_to_action_matches_oracle(oracle_built):
    t = model.load_asset("dog3d_pace")
    o = Oracle(t)
    for tt in (0.0, 0.31, 0.77):
        kp, _ = o.kin_eval(tt)
        assert np.abs(streams.pose_to_action(t, kp) - o.pose_to_action(kp)).max() < 1e-14


def test_replay_bundle_from_device_matches_oracle(emu_lib, tmp_path):
    """tools/replay_dump.py on the emulator build: actions recovered on the host reproduce the oracle's stream A1, the
    dumped states equal the oracle's states, and the bundle reads back."""
    import replay_dump
    t = model.load_asset("humanoid3d_walk")
    out = str(tmp_path / "bundle")
    acts, poses, vels, rew, term = replay_dump.run(t, ["--asset", "humanoid3d_walk"], 2, "A1", 64, out, emu_lib)
    o = Oracle(t); o.reset(0.0)
    for k in range(2):
        a = o.pose_to_action(o.kin_state()[0])
        assert np.abs(a.astype(np.float32) - acts[k]).max() < 1e-6
        o.set_action(acts[k])
        for u in range(20):
            o.update(1.0 / 600)
        p, v = o.sim_state()
        assert np.abs(p - poses[k + 1]).max() < 1e-9 and np.abs(v - vels[k + 1]).max() < 1e-7
        assert abs(o.calc_reward() - rew[k]) < 1e-6
    b = formats.read_replay_bundle(out)
    assert b["poses"].shape == (3, 43) and b["actions"].shape == (2, 28) and b["updates_per_step"] == 20
    assert np.array_equal(b["poses"], poses)
