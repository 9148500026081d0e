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


@pytest.mark.gpu
@pytest.mark.parametrize("asset,stream,steps", [("humanoid3d_walk", "A1", 40), ("humanoid3d_walk", "A2", 25), ("dog3d_pace", "A0", 12)])
def test_replay_bundle_from_hip_device_round_trip(hip_lib, tmp_path, asset, stream, steps):
    """tools/replay_dump.py on the real HIP library (production fp32 kernels): the bundle is written in the reference's own
    formats (WriteState JSON per step + the action list), read back, and REPLAYED by a consumer that only has the bundle -- here
    the oracle standing in for the DeepMimicCore + Bullet build the bundle is meant for: SetAction(actions[k]); 20 x Update;
    compare with state_%04d.json / rewards[k]."""
    import replay_dump
    t = model.load_asset(asset)
    out = str(tmp_path / "bundle")
    acts, poses, vels, rew, term = replay_dump.run(t, ["--asset", asset], steps, stream, 32, out, hip_lib)
    b = formats.read_replay_bundle(out)
    assert b["poses"].shape == (steps + 1, t.pose_dim) and b["actions"].shape == (steps, t.action_dim)
    assert np.array_equal(b["poses"], poses) and np.array_equal(b["vels"], vels) and b["updates_per_step"] == 20
    assert np.allclose(b["rewards"], rew) and list(b["terminate"]) == [int(x) for x in term]
    # replay from the bundle alone
    o = Oracle(t); o.reset(float(b["meta"]["t0"]))
    p0, _ = o.sim_state()
    assert np.abs(p0 - b["poses"][0]).max() < 2e-6           # state_0000.json is the state after Reset
    worst_r, alive = 0.0, 0
    for k in range(steps):
        o.set_sim_state(b["poses"][k], b["vels"][k])          # teacher-forced from the dumped state, as a --state_files replay would be
        o.set_action(b["actions"][k])
        for u in range(b["updates_per_step"]):
            o.update(b["timestep"])
        r = o.calc_reward()
        if r != 0.0:
            alive += 1
            worst_r = max(worst_r, abs(r - b["rewards"][k]))
        assert o.check_terminate() == b["terminate"][k] or r == 0.0
    assert alive >= 8 and worst_r < 5e-3, (alive, worst_r)    # fp32 kernels vs the fp64 replay: the fixed per-step bound of DESIGN.md section 7
