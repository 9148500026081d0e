"""Synthetic-input generators of the measurement contract (deepmimic_amd/streams.py)."""
import numpy as np

from deepmimic_amd import streams


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32-10 (Salmon et al.)."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for c, k, want in kat:
        assert [int(x) for x in streams.philox4x32_10(c, k)] == want


def test_noise_is_keyed_by_global_env_id_and_step():
    a = streams.normal_noise(np.arange(64), 5, 28)
    b = streams.normal_noise(np.arange(32, 64), 5, 28)          # second shard of a 2-rank job
    assert np.array_equal(a[32:], b)
    assert not np.array_equal(a, streams.normal_noise(np.arange(64), 6, 28))
    big = streams.normal_noise(np.arange(4096), 0, 28)
    assert abs(big.mean()) < 1e-3 and abs(big.std() - 0.05) < 1e-3 and np.isfinite(big).all()


def test_reset_phase_is_the_knuth_hash():
    ph = streams.reset_phase([0, 1, 2, 4095], 1.25)
    assert ph[0] == 0.0 and abs(ph[1] - 2654435761 / 2 ** 32 * 1.25) < 1e-15
    assert ((ph >= 0) & (ph < 1.25)).all()


def test_exponential_episode_timer(emu_lib):
    """`--timer_type exp` (util/Timer.cpp:64-67): max time = min(time_lim_min + Exp(mean time_lim_exp), time_lim_max), drawn ON THE DEVICE from the reset
    stream (seed, global env id, episode, stream 1) by explicit resets and -- round 4 -- by the in-kernel auto-reset of both wave packings;
    `model.draw_time_limit` is the host mirror"""
    import pytest
    from deepmimic_amd import model, streams
    from deepmimic_amd.core import BatchEnv
    t = model.load_asset("humanoid3d_walk")
    t.cfg.timer_type = "exp"; t.cfg.time_lim_min, t.cfg.time_lim_max, t.cfg.time_lim_exp = 0.5, 3.0, 0.8
    t.cfg.time_end_lim_min = t.cfg.time_end_lim_max = t.cfg.time_end_lim_exp = None
    n, seed = 64, 17
    env = BatchEnv(t, n, precision=64, lib_path=emu_lib, seed=seed, env_id_offset=100)
    lims = []
    for k in range(6):
        ep = env.get_state()["flags"][:, 2].copy()
        env.reset()
        mt = env.get_state()["clocks"][:, 4]
        want = [model.draw_time_limit("exp", 0.5, 3.0, 0.8, streams.reset_rand01(seed, 100 + e, int(ep[e]), 1)) for e in range(n)]
        assert np.abs(mt - np.array(want)).max() < 1e-12                                          # (device log1p vs numpy's: not bit-identical)
        lims.append(mt)
    lims = np.concatenate(lims)
    assert lims.min() >= 0.5 and lims.max() <= 3.0 and (lims == 3.0).mean() > 0.01              # the clipped tail: P = exp(-2.5 / 0.8) = 4.4 %
    assert abs(np.mean(np.minimum(lims, 2.9999) - 0.5) - 0.8 * (1 - np.exp(-2.5 / 0.8))) < 0.12  # mean of the truncated exponential
    # in-kernel auto-reset: run until every env has ended at least one episode (limits <= 3 s = 90 control steps); each new episode's limit is the
    # exponential draw of ITS episode counter, for one and two characters per wavefront
    import copy
    t2 = copy.deepcopy(t)
    t2.cfg.time_lim_min, t2.cfg.time_lim_max, t2.cfg.time_lim_exp = 0.1, 0.5, 0.15               # short episodes: <= 15 control steps
    for packing in (1, 2):
        e2 = BatchEnv(t2, 4, precision=64, lib_path=emu_lib, seed=seed, env_id_offset=100, wave_packing=packing)
        seen = 0
        for k in range(20):
            out = e2.step(None, 1 / 600, 20, open_loop=True, auto_reset=True)
            if out["episode_end"].any():
                st = e2.get_state()
                for e in np.nonzero(out["episode_end"])[0]:
                    ep_now = int(st["flags"][e, 2])                                              # the episode that has just begun
                    want1 = model.draw_time_limit("exp", 0.1, 0.5, 0.15, streams.reset_rand01(seed, 100 + int(e), ep_now - 1, 1))
                    assert abs(float(st["clocks"][e, 4]) - want1) < 1e-12 and float(st["clocks"][e, 3]) == 0.0, (packing, k, e)
                    seen += 1
        assert seen >= 4, seen
        e2.close()
    env.set_sample_count(0, test_mode=True)                                                      # test mode pins the limit (cRLSceneSimChar::ResetTimers)
    env.reset()
    assert (env.get_state()["clocks"][:, 4] == 3.0).all()
    t.cfg.timer_type = "gauss"
    with pytest.raises(ValueError, match="unsupported timer type"):
        BatchEnv(t, 2, precision=64, lib_path=emu_lib)
