"""Kernel logic vs oracle on the CPU: the product's device + host sources compiled for the fiber emulator.

These run without a GPU (`-m "not gpu"`).  The same checks run against the real library in test_parity_gpu.py.
"""
import numpy as np
import pytest

import parity_common as pc
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv


@pytest.mark.parametrize("name", ["humanoid3d_walk", "dog3d_pace"])
def test_reset_query_fp64(emu_lib, name):
    pc.check_reset_and_query(name, 64, emu_lib, tol_state=1e-12, tol_reward=1e-6)


@pytest.mark.parametrize("name", ["humanoid3d_walk", "dog3d_pace"])
def test_dynamics_fp64(emu_lib, name):
    pc.check_dynamics(name, 64, emu_lib, rtol=1e-11)


def test_dynamics_fp32(emu_lib):
    pc.check_dynamics("humanoid3d_walk", 32, emu_lib, rtol=2e-5)


@pytest.mark.parametrize("name", ["humanoid3d_walk", "dog3d_pace"])
def test_spd_fp64(emu_lib, name):
    pc.check_spd(name, 64, emu_lib, rtol=1e-9)


def test_spd_fp32(emu_lib):
    pc.check_spd("humanoid3d_walk", 32, emu_lib, rtol=2e-3)


@pytest.mark.parametrize("name,lift", [("humanoid3d_walk", 0.0), ("humanoid3d_walk", -0.03), ("humanoid3d_walk", -0.3), ("dog3d_pace", 0.0)])
def test_substep_fp64(emu_lib, name, lift):
    nc = pc.check_substep(name, 64, emu_lib, tol_vel=1e-8, tol_pose=1e-10, lift=lift)
    if lift < 0:
        assert nc > 0      # the pushed-down batch must exercise the contact solver


def test_rollout_fp64_matches_oracle(emu_lib):
    dr, ds, ok = pc.rollout_compare("humanoid3d_walk", 64, emu_lib, steps=8)
    assert ok and dr.max() < 1e-6 and ds.max() < 1e-5      # outputs are float32 at the boundary


def test_rollout_fp32_reward_tolerance(emu_lib):
    # fp32 production kernel vs fp64 oracle: BASELINE tolerance is 1e-4 on the reward
    dr, ds, ok = pc.rollout_compare("humanoid3d_walk", 32, emu_lib, steps=12)
    assert ok and dr.max() < 1e-4


def test_open_loop_path_equals_action_path(emu_lib):
    dr, ds, ok = pc.rollout_compare("humanoid3d_walk", 64, emu_lib, steps=3, t0=0.4, open_loop_on_device=True)
    assert ok and dr.max() < 1e-6


def test_stepwise_resync_spinkick_fp64(emu_lib):
    dr, ds, ok = pc.rollout_compare("humanoid3d_spinkick", 64, emu_lib, steps=30, resync=True)
    assert ok and dr.max() < 1e-6 and ds.max() < 1e-4


# ---- two characters per wavefront (dm_device_duo.h)
def test_duo_rollout_fp64_matches_oracle(emu_lib):
    dr, ds, ok = pc.batch_rollout_compare("humanoid3d_walk", 64, emu_lib, steps=6, t0s=[0.0, 0.37, 0.8, 0.11], wave_packing=2)
    assert ok and dr.max() < 1e-6 and ds.max() < 1e-5


def test_duo_heavy_contact_fallback_fp64(emu_lib):
    """one character of each pair pressed 30 cm into the ground (> 32 constraint rows): the pair runs the 64-lane routine"""
    dr, ds, ok = pc.batch_rollout_compare("humanoid3d_walk", 64, emu_lib, steps=2, t0s=[0.0, 0.4, 0.2, 0.6], wave_packing=2,
                                          lifts=[-0.3, 0.0, 0.0, -0.25])
    assert dr.max() < 1e-6 and ds.max() < 1e-4


def test_duo_borrowed_lanes_fp64(emu_lib):
    """one character of each pair pressed 8-10 cm into the ground: 33..48 constraint rows beside a light partner -- the pair stays on the two-per-wave path with the heavy
    character's rows 32.. on lanes of the partner's half (DuoSim lane borrowing, round 6); heavy character in the lower half (pair 0) and in the upper half (pair 1)"""
    for lifts in ([-0.08, 0.0, 0.0, -0.08], [-0.1, 0.0, 0.0, -0.1]):
        st = {}
        dr, ds, ok = pc.batch_rollout_compare("humanoid3d_walk", 64, emu_lib, steps=2, t0s=[0.0, 0.4, 0.2, 0.6], wave_packing=2, lifts=lifts, stats=st)
        assert (st["borrowed"] > 0).all(), st
        assert ok and dr.max() < 1e-6 and ds.max() < 1e-5, (dr, ds)


def test_duo_matches_single_packing_fp32(emu_lib):
    d1 = pc.batch_rollout_compare("humanoid3d_walk", 32, emu_lib, steps=4, t0s=[0.05, 0.5], wave_packing=1)
    d2 = pc.batch_rollout_compare("humanoid3d_walk", 32, emu_lib, steps=4, t0s=[0.05, 0.5], wave_packing=2)
    assert d1[2] and d2[2] and d1[0].max() < 1e-4 and d2[0].max() < 1e-4


def test_root_heading_sync_dog_spin_fp64(emu_lib):
    """args/run_dog3d_spin_args.txt: sync_char_root_rot = true (SyncKinCharRoot at reset, SyncKinCharNewCycle on phase wrap)"""
    pc.check_reset_and_query("dog3d_spin", 64, emu_lib, tol_state=1e-12, tol_reward=1e-6)
    dr, ds, ok = pc.rollout_compare("dog3d_spin", 64, emu_lib, steps=30)      # clip = 0.73 s: one wrap
    assert ok and dr.max() < 1e-6 and ds.max() < 1e-4


# ---- self collision (capsule model)
@pytest.mark.parametrize("name,n", [("humanoid3d_backflip", 24), ("dog3d_spin", 8)])
def test_substep_with_self_contacts_fp64(emu_lib, name, n):
    """tucked backflip poses (wrist against knee / shin) and the dog's tail-thigh overlap: rows with two-link Jacobians"""
    pc.check_substep(name, 64, emu_lib, tol_vel=1e-8, tol_pose=1e-10, lift=1.0, n=n)
    assert pc.check_substep.self_contacts > 0


def test_self_collision_switch(emu_lib):
    t = model.load_asset("dog3d_pace")
    on = BatchEnv(t, 1, precision=64, lib_path=emu_lib); off = BatchEnv(t, 1, precision=64, lib_path=emu_lib, self_collision=False)
    for e in (on, off):
        e.reset(kin_times=[0.2], max_times=np.inf)
        e.step(None, pc.DT, 1, open_loop=True)
    assert np.abs(on.get_state()["vel"] - off.get_state()["vel"]).max() > 1e-3


@pytest.mark.parametrize("stream,steps", [("A0", 3), ("A2", 2)])
def test_action_streams_fp64(emu_lib, stream, steps):
    """explicit float32 actions through the exp-map -> PD-target path (a10): zeros and noisy mocap tracking"""
    dr, ds, ok, _ = pc.action_rollout_compare("humanoid3d_walk", 64, emu_lib, steps, stream, [0.0, 0.37])
    assert ok and dr.max() < 1e-6 and ds.max() < 1e-5, (dr, ds)


def test_action_stream_a2_duo_fp64(emu_lib):
    dr, ds, ok, _ = pc.action_rollout_compare("humanoid3d_walk", 64, emu_lib, 2, "A2", [0.0, 0.37], wave_packing=2)
    assert ok and dr.max() < 1e-6 and ds.max() < 1e-5, (dr, ds)


def test_auto_reset_mirrored_by_oracle(emu_lib):
    """rollout through auto-resets with the oracle replaying the device's reset draws: a finite episode timer (row a3) ends
    episodes every 6 control steps here, so several resets happen inside the window"""
    dr, ds, alive, resets, ok = pc.auto_reset_rollout_compare("humanoid3d_walk", 64, emu_lib, steps=14, n=2, seed=5, time_lim=0.2)
    assert ok and resets >= 4
    assert alive.all() and dr.max() < 1e-6 and ds.max() < 1e-5


@pytest.mark.parametrize("pack", [1, 2])
def test_episode_ends_at_the_update_of_the_fall(emu_lib, pack):
    """DM_END_EPISODE_EARLY (on with auto-reset): the open-loop walker falls in the middle of a control step; the device stops
    that env's updates there (one-per-wave: loop exit; two-per-wave: record stored, character parked, record reloaded), the
    oracle's control_step does the same, and the terminal reward / flags / next observation agree.  Both wave packings."""
    dr, ds, alive, resets, ok = pc.auto_reset_rollout_compare("humanoid3d_walk", 64, emu_lib, steps=36, n=2, seed=7, wave_packing=pack)
    assert ok and resets >= 2, resets
    assert dr.max() < 1e-6 and ds.max() < 1e-5, (dr.max(), ds.max())


def test_biped_tree_class_fp64(emu_lib, monkeypatch):
    """ClsBipedTree: humanoid3d one per wavefront with the branch-sparse, level-scheduled factor on its compiled dof tree (opt-in,
    DM_TREE_BIPED=1; profiles/r03_ab_biped_tree.json has the negative A/B): component and rollout parity as for the dense class"""
    monkeypatch.setenv("DM_TREE_BIPED", "1")
    pc.check_dynamics("humanoid3d_walk", 64, emu_lib, 1e-11)
    pc.check_spd("humanoid3d_walk", 64, emu_lib, 1e-9)
    pc.check_substep("humanoid3d_walk", 64, emu_lib, 1e-8, 1e-10, lift=-0.03)
    dr, ds, ok = pc.rollout_compare("humanoid3d_walk", 64, emu_lib, steps=6)
    assert ok and dr.max() < 1e-6 and ds.max() < 1e-5


def test_sampled_compare_from_device_states_emulator(emu_lib):
    """parity_common.sampled_compare (the checker of tests/test_parity_4096.py and of bench.py's `checks.parity`): oracles re-synchronised FROM
    the device envs before every control step (Oracle.set_full_state <- BatchEnv.get_state), two env groups, through auto-resets; fp64 build."""
    from deepmimic_amd import streams
    from deepmimic_amd.groups import EnvGroups
    t = model.load_asset("humanoid3d_walk")
    env = EnvGroups(t, 8, groups=2, seed=1234, precision=64, lib_path=emu_lib, test_mode=True)
    env.reset(kin_times=streams.reset_phase(np.arange(8), env.duration))
    step = lambda: env.step(None, pc.DT, 20, open_loop=True, auto_reset=True)
    for _ in range(24):
        step()
    dr, ds, alive, ok, ends = pc.sampled_compare(env.get_state, step, t, [0, 3, 5, 7], 12)
    assert ok and ends > 0 and alive.sum() > 30
    assert dr.max() < 1e-6 and np.nanmax(ds) < 1e-6
    env.close()
