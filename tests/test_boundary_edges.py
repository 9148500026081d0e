"""Edge cases of the C-ABI boundary (include/dm_hip.h) through the ctypes binding, on the emulator build of the same host code: empty and ragged
reset lists, single / odd batch sizes, zero-update steps, argument validation with a message in dm_last_error."""
import ctypes as C

import numpy as np
import pytest

from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv

DT = 1.0 / 600


def _env(emu_lib, n, **kw):
    return BatchEnv(model.load_asset("humanoid3d_walk"), n, precision=64, lib_path=emu_lib, seed=2, **kw)


def test_reset_of_an_empty_list_touches_nothing(emu_lib):
    env = _env(emu_lib, 4); env.reset()
    env.step(None, DT, 20, open_loop=True)
    before = env.get_state()
    env.reset(env_ids=[])                                           # NOT "all envs"
    after = env.get_state()
    for k in before:
        assert np.array_equal(before[k], after[k]), k


def test_reset_of_a_ragged_subset(emu_lib):
    env = _env(emu_lib, 5); env.reset()
    for _ in range(3):
        env.step(None, DT, 20, open_loop=True)
    before = env.get_state()
    env.reset(env_ids=[3, 1], kin_times=[0.25, 0.5], max_times=[1.0, 2.0])
    after = env.get_state()
    for e in (0, 2, 4):
        for k in before:
            assert np.array_equal(before[k][e], after[k][e]), (k, e)
    assert after["clocks"][3][0] == 0.25 and after["clocks"][1][0] == 0.5 and after["clocks"][3][4] == 1.0 and after["clocks"][1][4] == 2.0
    assert after["clocks"][3][3] == 0.0 and after["flags"][3][2] == before["flags"][3][2] + 1       # timer back to 0, next episode
    with pytest.raises(RuntimeError, match="env id out of range"):
        env.reset(env_ids=[5])
    with pytest.raises(RuntimeError, match="env id out of range"):
        env.reset(env_ids=[-1])


@pytest.mark.parametrize("n", [1, 3])
def test_single_and_odd_batches_run_one_character_per_wave(emu_lib, n):
    """the two-per-wave kernel needs an even batch: an odd one (and N = 1, the facade's case) takes the one-per-wave kernel, same results"""
    a = _env(emu_lib, n, wave_packing=2); b = _env(emu_lib, n, wave_packing=1)
    a.reset(); b.reset()
    for _ in range(3):
        oa = a.step(None, DT, 20, open_loop=True, auto_reset=True); ob = b.step(None, DT, 20, open_loop=True, auto_reset=True)
        assert np.array_equal(oa["state"], ob["state"]) and np.array_equal(oa["reward"], ob["reward"])
    # an even batch with the same seed: env e's trajectory does not depend on the batch size (global env id keys the draws)
    c = _env(emu_lib, n + 1, wave_packing=2); c.reset()
    for _ in range(3):
        oc = c.step(None, DT, 20, open_loop=True, auto_reset=True)
    assert np.abs(oc["state"][:n] - oa["state"]).max() < 1e-9


def test_zero_update_step_is_a_query(emu_lib):
    env = _env(emu_lib, 2); env.reset()
    st = env.get_state()
    out = env.step(None, DT, 0, open_loop=True)
    q = env.query()
    assert np.array_equal(out["state"], q["state"]) and np.array_equal(out["reward"], q["reward"])
    after = env.get_state()
    assert np.array_equal(st["pose"], after["pose"]) and np.array_equal(st["clocks"], after["clocks"])


def test_argument_validation_reports_through_last_error(emu_lib):
    t = model.load_asset("humanoid3d_walk")
    for kw, msg in ((dict(num_envs=0), "num_envs"), (dict(num_envs=2, precision=16), "precision"), (dict(num_envs=2, max_contacts=21), "max_contacts"),
                    (dict(num_envs=2, wave_packing=3), "wave_packing")):
        n = kw.pop("num_envs")
        with pytest.raises(RuntimeError, match=msg):
            BatchEnv(t, n, lib_path=emu_lib, **{"precision": 64, **kw})
    env = _env(emu_lib, 2)
    lib = env.lib
    lib.dm_last_error.restype = C.c_char_p
    assert lib.dm_reset(None, None, 0, None, None) != 0 and b"null" in lib.dm_last_error()
    assert lib.dm_set_action(env.h, None, 0) != 0 and b"null" in lib.dm_last_error()
    with pytest.raises(ValueError):
        env.set_action(np.zeros((2, env.A + 1), np.float32))         # wrong action width never reaches the library
    with pytest.raises(ValueError):
        env.step(np.zeros((3, env.A), np.float32), DT, 20)
    with pytest.raises(RuntimeError, match="no perturbation state"):
        env.get_perturb_state()
    with pytest.raises(RuntimeError, match="no free body"):
        env.get_obj_state()


def test_nan_actions_do_not_poison_other_envs(emu_lib):
    """a NaN action corrupts its own env only (every env is its own wavefront / half wavefront; nothing is shared but the model tables)"""
    env = _env(emu_lib, 4, wave_packing=2); ref = _env(emu_lib, 4, wave_packing=2)
    env.reset(); ref.reset()
    a = np.zeros((4, env.A), np.float32); bad = a.copy(); bad[1] = np.nan
    o1 = env.step(bad, DT, 20); o0 = ref.step(a, DT, 20)
    for e in (0, 2, 3):
        assert np.array_equal(o1["state"][e], o0["state"][e]) and o1["reward"][e] == o0["reward"][e]
