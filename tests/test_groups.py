"""Env groups (deepmimic_amd/groups.py, round 4): a batch split into G contexts on their own streams must reproduce the single-context
rollout env by env -- bit for bit, both wave packings -- because every draw is keyed by the GLOBAL env id and the two-per-wave pairing
is (2b, 2b + 1) in every partition (dm_create refuses an odd env_id_offset with wave_packing 2)."""
import numpy as np
import pytest

from deepmimic_amd import model, streams
from deepmimic_amd.core import BatchEnv
from deepmimic_amd.groups import EnvGroups, split_even


def test_split_even():
    assert split_even(4096, 2) == [2048, 2048]
    assert split_even(4096, 1) == [4096]
    assert split_even(6, 2) == [6]            # 3 + 3 would put envs (2, 3) into different wavefront pairs
    assert split_even(8, 2) == [4, 4]
    assert split_even(7, 2) == [7]


def _compare(lib, n, groups, precision, packing, steps, scene="humanoid3d_walk", off=0):
    t = model.load_asset(scene)
    kt = streams.reset_phase(off + np.arange(n), 1.0)
    one = BatchEnv(t, n, seed=5, precision=precision, lib_path=lib, wave_packing=packing, env_id_offset=off, test_mode=True)
    grp = EnvGroups(t, n, groups=groups, seed=5, precision=precision, lib_path=lib, wave_packing=packing, env_id_offset=off, test_mode=True)
    assert grp.G == groups
    one.reset(kin_times=kt * one.duration, max_times=0.2); grp.reset(kin_times=kt * one.duration, max_times=0.2)     # short episodes: auto-resets inside the run
    ends = 0
    for k in range(steps):
        a = one.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
        b = grp.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
        for key in ("state", "reward", "terminate", "valid", "episode_end"):
            assert np.array_equal(a[key], b[key]), (k, key)
        ends += int(a["episode_end"].sum())
    assert ends > 0, "the rollout must cross episode ends (reset draws keyed by the global env id)"
    one.close(); grp.close()


@pytest.mark.parametrize("packing", [1, 2])
def test_groups_match_single_context_emulator(emu_lib, packing):
    _compare(emu_lib, 8, 2, 64, packing, 8)


def test_groups_with_offset_emulator(emu_lib):
    _compare(emu_lib, 4, 2, 32, 2, 8, off=6)


def test_odd_offset_refused_or_unpaired(emu_lib):
    t = model.load_asset("humanoid3d_walk")
    with pytest.raises(RuntimeError, match="even env_id_offset"):
        BatchEnv(t, 4, lib_path=emu_lib, wave_packing=2, env_id_offset=3)
    # the default packing falls back to one character per wavefront for such a shard: same trajectories as the one-per-wave run
    a = BatchEnv(t, 4, seed=2, precision=64, lib_path=emu_lib, wave_packing=0, env_id_offset=3)
    b = BatchEnv(t, 4, seed=2, precision=64, lib_path=emu_lib, wave_packing=1, env_id_offset=3)
    a.reset(); b.reset()
    for _ in range(2):
        oa = a.step(None, 1.0 / 600, 20, open_loop=True); ob = b.step(None, 1.0 / 600, 20, open_loop=True)
        assert np.array_equal(oa["state"], ob["state"]) and np.array_equal(oa["reward"], ob["reward"])


def test_group_device_step_offsets_the_amp_rows_emulator(emu_lib):
    """`step_group_device` takes WHOLE-BATCH arrays: the AMP observation pointer is moved to the group's rows like the others (emulator: "device" memory is
    host memory, so the raw-pointer interface runs here)."""
    t = model.load_asset("humanoid3d_walk"); t.cfg.scene = "imitate_amp"
    n = 4
    one = BatchEnv(t, n, seed=3, precision=32, lib_path=emu_lib, test_mode=True); grp = EnvGroups(t, n, groups=2, seed=3, precision=32, lib_path=emu_lib, test_mode=True)
    assert grp.G == 2 and grp.amp_size > 0
    kt = streams.reset_phase(np.arange(n), 1.0) * one.duration
    one.reset(kin_times=kt); grp.reset(kin_times=kt)
    ref = one.step(None, 1.0 / 600, 20, open_loop=True, amp=True)
    st = np.zeros((n, grp.S), np.float32); rw = np.zeros(n, np.float32); amp = np.zeros((n, grp.amp_size), np.float32)
    tm, vd, en = (np.zeros(n, np.int32) for _ in range(3))
    p = lambda a: a.ctypes.data
    for g in range(grp.G):
        grp.step_group_device(g, 0, p(st), p(rw), p(tm), p(vd), p(en), timestep=1.0 / 600, n_updates=20, open_loop=True, amp_ptr=p(amp))
    grp.synchronize()
    assert np.array_equal(st, ref["state"]) and np.array_equal(rw, ref["reward"]) and np.array_equal(amp, ref["amp_obs"])
    assert np.abs(amp[2:]).max() > 0
    one.close(); grp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene,packing", [("humanoid3d_walk", 0), ("dog3d_pace", 0)])
def test_groups_match_single_context_gpu(hip_lib, scene, packing):
    _compare(hip_lib, 64, 2, 32, packing, 12, scene=scene)


@pytest.mark.gpu
def test_groups_free_running_rollout_gpu(hip_lib):
    """the C rollout loops of two groups on two streams at once (what bench.py times) leave every env where the single context does"""
    t = model.load_asset("humanoid3d_walk")
    n = 256
    kt = streams.reset_phase(np.arange(n), 1.0)
    one = BatchEnv(t, n, seed=9, test_mode=True, lib_path=hip_lib)
    grp = EnvGroups(t, n, groups=2, seed=9, test_mode=True, lib_path=hip_lib)
    one.reset(kin_times=kt * one.duration); grp.reset(kin_times=kt * one.duration)
    one.bench_rollout(0, 40); ms = grp.bench_rollout(0, 40)
    assert ms > 0
    a = one.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True); b = grp.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
    assert np.array_equal(a["state"], b["state"]) and np.array_equal(a["reward"], b["reward"])
