"""Random perturbations (`--enable_rand_perturbs`, part of cSceneSimChar::Update): scenes/SceneSimChar.cpp:41-51, 92-99,
205-256, 618-626, 952-956; sim/Perturb.cpp; sim/PerturbManager.cpp; sim/World.cpp:93-96.

A force of random direction / magnitude / duration on a random body part every U[time_min, time_max] seconds, applied at the part's centre of
mass for whole scene updates.  Closed form first (a one-link character: dv = f / m dt exactly, the clocks of tPerturb and the re-arming), then
the device against the oracle through auto-resets on the humanoid (overlapping forces, both wave packings), then the arg-file keys."""
import numpy as np
import pytest

from deepmimic_amd import model, streams
from deepmimic_amd.core import BatchEnv
from oracle_lib import Oracle
from test_physics_validity import box_tables

DT = 1.0 / 600
G = 9.8


def perturbed(t, tmin=0.15, tmax=0.3, fmin=100.0, fmax=300.0, dmin=0.05, dmax=0.25, parts=None):
    c = t.cfg
    c.enable_rand_perturbs = True; c.perturb_time_min, c.perturb_time_max = tmin, tmax
    c.min_perturb, c.max_perturb = fmin, fmax; c.min_pertrub_duration, c.max_perturb_duration = dmin, dmax; c.perturb_part_ids = parts
    return t


def slot_row(timer=0.0, nxt=np.inf, draws=0.0, slots=()):
    r = np.zeros(16); r[0], r[1], r[2] = timer, nxt, draws
    for i, (part, f, dur, tm) in enumerate(slots):
        r[3 + 6 * i: 9 + 6 * i] = [part + 1, f[0], f[1], f[2], dur, tm]
    return r


def active(row):
    """the acting forces of a state row as a sorted list (the oracle keeps a list, the device two slots: the order is not part of the state)"""
    out = [tuple(np.round(row[3 + 6 * i: 9 + 6 * i], 12)) for i in range(2) if row[3 + 6 * i] > 0]
    return sorted(out)


# ---------------------------------------------------------------------------------------------------------------- closed form (oracle)
def test_oracle_force_on_a_falling_box_is_f_over_m(oracle_built):
    t = perturbed(box_tables(mass=10.0), tmin=100.0, tmax=100.0)
    o = Oracle(t); o.goal_rng(7, 0); o.set_perturb_state(slot_row()); o.reset(0.0)
    p, v = o.sim_state(); p[1] = 50.0; o.set_sim_state(p, v)
    f, dur = np.array([20.0, 5.0, -10.0]), 0.0492                        # acts for ceil(0.0492 / dt) = 30 updates
    o.set_perturb_state(slot_row(nxt=100.0, draws=1, slots=[(0, f, dur, 0.0)]))
    for k in range(1, 41):
        o.update(DT)
        _, v = o.sim_state()
        n = min(k, 30)                                                    # cPerturbManager: acts while mTime < mDuration, whole updates
        assert np.abs(v[:3] - (f / 10.0 * n * DT + np.array([0, -G, 0]) * k * DT)).max() < 1e-12, k
        assert np.abs(v[3:6]).max() < 1e-13                               # at the centre of mass: no torque (SimBodyLink.cpp:103-112, local_pos = 0)
    assert o.num_perturbs() == 0 and abs(o.perturb_state()[0] - 40 * DT) < 1e-12


def test_oracle_rearms_and_draws_in_range(oracle_built):
    t = perturbed(box_tables(), tmin=0.02, tmax=0.04, fmin=50, fmax=100, dmin=0.005, dmax=0.03)
    o = Oracle(t); o.goal_rng(11, 3); o.set_perturb_state(slot_row()); o.reset(0.0)
    p, v = o.sim_state(); p[1] = 1e4; o.set_sim_state(p, v)
    assert 0.02 <= o.perturb_state()[1] <= 0.04 and o.perturb_state()[2] == 1                     # ResetRandPertrub at the scene reset
    fired, mags, durs = 0, [], []
    for k in range(600):
        before = o.perturb_state()
        o.update(DT)
        row = o.perturb_state()
        if row[2] > before[2]:                                                                     # a perturbation fired: 7 draws, timer back to 0
            assert row[2] == before[2] + 7 and row[0] == 0.0 and 0.02 <= row[1] <= 0.04 and before[0] + DT >= before[1]
            new = [s for s in active(row) if s[5] == round(DT, 12)]                                 # it already acted during this update
            assert len(new) == 1
            mags.append(np.linalg.norm(new[0][1:4])); durs.append(new[0][4]); fired += 1
        assert o.num_perturbs() <= 2
    assert fired >= 20 and 50 <= min(mags) and max(mags) <= 100 + 1e-9 and 0.005 <= min(durs) and max(durs) <= 0.03
    o.reset(0.0)                                                                                   # cWorld::Reset clears the manager
    assert o.num_perturbs() == 0 and o.perturb_state()[0] == 0.0


# ---------------------------------------------------------------------------------------------------------------- device, closed form
def _device_box(lib, precision, tol):
    t = perturbed(box_tables(mass=10.0), tmin=100.0, tmax=100.0)
    env = BatchEnv(t, 2, precision=precision, lib_path=lib, wave_packing=1, seed=5)
    assert env.has_perturbs
    d0 = env.get_perturb_state()[:, 2]                                   # (dm_create resets every env once: one draw)
    env.reset(kin_times=[0.0] * 2, max_times=np.inf)
    rows = env.get_perturb_state()
    assert (rows[:, 0] == 0).all() and (rows[:, 1] == 100.0).all() and (rows[:, 2] == d0 + 1).all() and (rows[:, 3] == 0).all()
    st = env.get_state(); P = st["pose"].copy(); P[:, 1] = 50.0
    env.set_state(pose=P, vel=st["vel"])
    f = np.array([20.0, 5.0, -10.0])
    rows[0] = slot_row(nxt=100.0, draws=1, slots=[(0, f, 0.0492, 0.0)])
    rows[1] = slot_row(nxt=100.0, draws=1, slots=[(0, f, 0.0492, 0.0), (0, -0.5 * f, 0.0192, 0.0)])       # two at once: 12 updates of f/2, then f
    env.set_perturb_state(rows)
    env.step(None, DT, 40)
    v = env.get_state()["vel"]
    g = np.array([0, -G, 0]) * 40 * DT
    assert np.abs(v[0, :3] - (f / 10 * 30 * DT + g)).max() < tol, v[0, :3]
    assert np.abs(v[1, :3] - (f / 10 * (30 - 0.5 * 12) * DT + g)).max() < tol, v[1, :3]
    assert np.abs(v[:, 3:6]).max() < tol
    rows = env.get_perturb_state()
    assert (rows[:, 3] == 0).all() and (rows[:, 9] == 0).all() and np.abs(rows[:, 0] - 40 * DT).max() < 1e-12


def test_device_force_on_a_falling_box_emulator(emu_lib):
    _device_box(emu_lib, 64, 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [(64, 1e-11), (32, 2e-5)])
def test_device_force_on_a_falling_box_gpu(hip_lib, prec, tol):
    _device_box(hip_lib, prec, tol)


# ---------------------------------------------------------------------------------------------------------------- device vs oracle, humanoid
def rollout_with_perturbs(name, precision, lib_path, steps, n, seed, wave_packing=0, parts=None):
    """Open-loop rollout through auto-resets with perturbations on; the oracle mirrors the device's reset draws (streams 0 / 1) and makes its own
    perturbation draws (stream 5).  Returns per (step, env): |reward diff|, relative state diff, and whether the perturbation rows agreed."""
    t = perturbed(model.load_asset(name), parts=parts)
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing, seed=seed)
    d0 = env.get_perturb_state()[:, 2]                                   # draw counter before the reset the oracle mirrors
    env.reset()
    ep = env.get_state()["flags"][:, 2].astype(np.int64)
    oracles = []
    for e in range(n):
        o = Oracle(t); o.goal_rng(seed, e); o.set_perturb_state(slot_row(draws=d0[e]))
        o.reset(o.duration * streams.reset_rand01(seed, e, int(ep[e]) - 1, 0), np.inf)
        oracles.append(o)
    rows = env.get_perturb_state()
    for e, o in enumerate(oracles):
        assert np.abs(rows[e] - o.perturb_state()).max() < 1e-15, "perturbation reset draw mismatch"
    dr, ds = np.zeros((steps, n)), np.zeros((steps, n))
    hits, both, resets, rows_ok = 0, 0, 0, True
    for k in range(steps):
        out = env.step(None, DT, 20, open_loop=True, auto_reset=True)
        rows = env.get_perturb_state()
        for e, o in enumerate(oracles):
            kp, _, _ = o.kin_state()
            o.set_action(o.pose_to_action(kp))
            o.control_step(20, DT)
            dr[k, e] = abs(float(out["reward"][e]) - o.calc_reward())
            assert bool(out["episode_end"][e]) == o.is_episode_end() and int(out["terminate"][e]) == o.check_terminate(), (k, e)
            if o.is_episode_end():
                o.reset(o.duration * streams.reset_rand01(seed, e, int(ep[e]), 0), np.inf)
                ep[e] += 1; resets += 1
            so = o.record_state()
            ds[k, e] = np.abs(out["state"][e] - so).max() / max(1.0, np.abs(so).max())
            ro = o.perturb_state()
            rows_ok &= bool(np.abs(rows[e][:3] - ro[:3]).max() < 1e-9) and active(rows[e]) == active(ro)
            hits += len(active(ro)) > 0; both += len(active(ro)) == 2
    return dr, ds, rows_ok, hits, both, resets


def stepwise_with_perturbs(name, precision, lib_path, steps, n, seed, wave_packing=0):
    """Teacher-forced: the oracles free-run through their own resets and perturbation draws; before every control step the device envs are set to
    the oracles' states (perturbation rows included), so each of the steps x n comparisons checks ONE control step from identical inputs -- the
    contact decisions of a character being pushed over are chaotic in a free-running comparison.  Returns |reward diff|, relative state diff,
    rows equal, and how many env-steps had one / two forces acting."""
    t = perturbed(model.load_asset(name))
    env = BatchEnv(t, n, precision=precision, lib_path=lib_path, wave_packing=wave_packing, seed=seed)
    oracles, ep = [], np.zeros(n, dtype=np.int64)
    for e in range(n):
        o = Oracle(t); o.goal_rng(seed, e); o.set_perturb_state(slot_row())
        o.reset(o.duration * streams.reset_rand01(seed, e, 0, 0)); oracles.append(o)
    env.reset(kin_times=[o.kin_time() for o in oracles], max_times=np.inf)
    dr, ds = np.zeros((steps, n)), np.zeros((steps, n))
    rows_ok, hits, both = True, 0, 0
    for k in range(steps):
        P, V, T, K, CL, FL, R = [], [], [], [], [], [], []
        for o in oracles:
            kp, kv, ko = o.kin_state()
            o.set_action(o.pose_to_action(kp))
            p, v = o.sim_state()
            cm = int(sum(int(c) << j for j, c in enumerate(o.contacts())))
            P.append(p); V.append(v); T.append(o.tar_pose()); K.append(ko); R.append(o.perturb_state())
            CL.append([o.kin_time(), o.kin_time(), 0.0, o.time(), np.inf]); FL.append([int(o.need_new_action()), cm, 1, 1])
        env.set_state(pose=np.array(P), vel=np.array(V), tar=np.array(T), kin=np.array(K), clocks=np.array(CL), flags=np.array(FL, dtype=np.int32))
        env.set_perturb_state(np.array(R))
        out = env.step(None, DT, 20)
        rows = env.get_perturb_state()
        for e, o in enumerate(oracles):
            for u in range(20):
                o.update(DT)
            dr[k, e] = abs(float(out["reward"][e]) - o.calc_reward())
            so = o.record_state()
            ds[k, e] = np.abs(out["state"][e] - so).max() / max(1.0, np.abs(so).max())
            ro = o.perturb_state()
            rows_ok &= bool(np.abs(rows[e][:3] - ro[:3]).max() < 1e-9) and active(rows[e]) == active(ro)
            hits += len(active(ro)) > 0; both += len(active(ro)) == 2
            if o.check_terminate() != 0:
                ep[e] += 1
                o.reset(o.duration * streams.reset_rand01(seed, e, int(ep[e]), 0))
    return dr, ds, rows_ok, hits, both, int(ep.sum())


@pytest.mark.parametrize("packing", [1, 2])
def test_humanoid_steps_with_perturbs_emulator(emu_lib, packing):
    dr, ds, rows_ok, hits, both, resets = stepwise_with_perturbs("humanoid3d_walk", 64, emu_lib, 24, 4, 21, wave_packing=packing)
    assert rows_ok and hits > 30 and both >= 1 and resets >= 1, (hits, both, resets)      # forces acted most of the time, overlapped, and knocked someone over
    assert dr.max() < 1e-6 and ds.max() < 1e-6, (dr.max(), ds.max())                    # (the state vector is float32: 5e-8 is its floor)


@pytest.mark.parametrize("packing", [1, 2])
def test_humanoid_free_running_with_perturbs_emulator(emu_lib, packing):
    """through the kernel's own auto-resets (ResetRandPertrub inside the launch): clocks, draws and forces stay identical to the oracle's; the
    states agree except where a contact decision of a tumbling character flips (free-running chaos)"""
    dr, ds, rows_ok, hits, both, resets = rollout_with_perturbs("humanoid3d_walk", 64, emu_lib, 24, 4, 21, wave_packing=packing)
    assert rows_ok and hits > 30 and resets >= 1, (hits, both, resets)
    assert np.median(ds) < 1e-6 and (ds < 1e-5).mean() > 0.8, (np.median(ds), (ds < 1e-5).mean())


def test_part_ids_restrict_the_draw(emu_lib):
    with pytest.raises(ValueError, match="ascending"):      # the reference indexes the list as written; a repeat or a different order would name other parts here
        BatchEnv(perturbed(model.load_asset("humanoid3d_walk"), parts=[2, 5, 5, 11]), 4, precision=64, lib_path=emu_lib, seed=3)
    t = perturbed(model.load_asset("humanoid3d_walk"), parts=[2, 5, 11])
    env = BatchEnv(t, 4, precision=64, lib_path=emu_lib, seed=3)
    env.reset()
    seen = set()
    for _ in range(12):
        env.step(None, DT, 20, open_loop=True, auto_reset=True)
        rows = env.get_perturb_state()
        seen |= {int(r[3 + 6 * i]) - 1 for r in rows for i in range(2) if r[3 + 6 * i] > 0}
    assert seen and seen <= {2, 5, 11}, seen


def test_perturbs_off_is_bitwise_the_plain_scene(emu_lib):
    """enable_rand_perturbs with the default (infinite) interval never fires: same rollout as without the key"""
    t0 = model.load_asset("humanoid3d_walk")
    t1 = model.load_asset("humanoid3d_walk"); t1.cfg.enable_rand_perturbs = True
    outs = []
    for t in (t0, t1):
        env = BatchEnv(t, 2, precision=64, lib_path=emu_lib, seed=9)
        assert not env.has_perturbs
        env.reset()
        for _ in range(3):
            out = env.step(None, DT, 20, open_loop=True, auto_reset=True)
        outs.append(out["state"].copy())
    assert np.array_equal(outs[0], outs[1])


def test_create_rejects_bad_ranges(emu_lib):
    t = perturbed(model.load_asset("humanoid3d_walk"), tmin=0.1, tmax=0.2, dmin=0.1, dmax=0.5)      # three forces at once would be possible
    with pytest.raises(RuntimeError, match="perturbations would act at once"):
        BatchEnv(t, 2, precision=64, lib_path=emu_lib)
    t = perturbed(model.load_asset("humanoid3d_walk"), parts=[40])
    with pytest.raises((RuntimeError, ValueError, OverflowError)):
        BatchEnv(t, 2, precision=64, lib_path=emu_lib)


def test_snapshot_restores_perturbations(emu_lib):
    t = perturbed(model.load_asset("humanoid3d_walk"))
    env = BatchEnv(t, 2, precision=64, lib_path=emu_lib, seed=4)
    env.reset()
    for _ in range(5):
        env.step(None, DT, 20, open_loop=True, auto_reset=True)
    snap = env.snapshot()
    a = [env.step(None, DT, 20, open_loop=True, auto_reset=True)["state"].copy() for _ in range(4)]
    env.restore(snap)
    b = [env.step(None, DT, 20, open_loop=True, auto_reset=True)["state"].copy() for _ in range(4)]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_arg_keys_reach_the_scene():
    p = model.ArgParser(["--scene", "imitate", "--enable_rand_perturbs", "true", "--perturb_time_min", "1", "--perturb_time_max", "2", "--min_perturb", "10",
                         "--max_perturb", "20", "--min_pertrub_duration", "0.2", "--max_perturb_duration", "0.4", "--perturb_part_ids", "0", "3"])
    c = model.parse_scene_config(p)
    assert c.enable_rand_perturbs and (c.perturb_time_min, c.perturb_time_max, c.min_perturb, c.max_perturb) == (1.0, 2.0, 10.0, 20.0)
    assert (c.min_pertrub_duration, c.max_perturb_duration) == (0.2, 0.4) and c.perturb_part_ids == [0, 3]


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("packing", [1, 2])
def test_humanoid_steps_with_perturbs_gpu_fp64(hip_lib, packing):
    dr, ds, rows_ok, hits, both, resets = stepwise_with_perturbs("humanoid3d_walk", 64, hip_lib, 24, 4, 21, wave_packing=packing)
    assert rows_ok and hits > 30 and both >= 1, (hits, both, resets)
    assert dr.max() < 1e-6 and ds.max() < 1e-6, (dr.max(), ds.max())


@pytest.mark.gpu
@pytest.mark.parametrize("packing", [1, 2])
def test_humanoid_steps_with_perturbs_gpu_fp32(hip_lib, packing):
    """production precision, teacher-forced: the single-step fp32 bound of DESIGN.md section 7 holds with forces acting"""
    dr, ds, rows_ok, hits, both, resets = stepwise_with_perturbs("humanoid3d_walk", 32, hip_lib, 24, 8, 21, wave_packing=packing)
    assert rows_ok and hits > 60, (hits, both, resets)
    assert np.median(dr) < 2e-5 and (dr < 1e-3).mean() > 0.95, (np.median(dr), dr.max())


@pytest.mark.gpu
def test_humanoid_free_running_with_perturbs_gpu_fp64(hip_lib):
    dr, ds, rows_ok, hits, both, resets = rollout_with_perturbs("humanoid3d_walk", 64, hip_lib, 24, 4, 21, wave_packing=2)
    assert rows_ok and hits > 30 and resets >= 1, (hits, both, resets)
    assert np.median(ds) < 1e-6 and (ds < 1e-5).mean() > 0.8, (np.median(ds), (ds < 1e-5).mean())


@pytest.mark.gpu
def test_humanoid_perturbs_gpu_fp32_clocks_and_draws(hip_lib):
    """production precision: the perturbation rows (clocks, draws, forces: doubles) stay identical to the oracle's while no episode ends on one side
    only; rewards stay within the free-running fp32 bound of DESIGN.md section 7"""
    t = perturbed(model.load_asset("humanoid3d_walk"), fmin=20.0, fmax=40.0)
    n, seed = 8, 33
    env = BatchEnv(t, n, precision=32, lib_path=hip_lib, seed=seed)
    d0 = env.get_perturb_state()[:, 2]
    env.reset()
    ep = env.get_state()["flags"][:, 2].astype(np.int64)
    oracles = []
    for e in range(n):
        o = Oracle(t); o.goal_rng(seed, e); o.set_perturb_state(slot_row(draws=d0[e]))
        o.reset(o.duration * streams.reset_rand01(seed, e, int(ep[e]) - 1, 0), np.inf); oracles.append(o)
    live = np.ones(n, dtype=bool); worst = 0.0; compared = 0
    for k in range(15):
        out = env.step(None, DT, 20, open_loop=True, auto_reset=True)
        rows = env.get_perturb_state()
        for e, o in enumerate(oracles):
            if not live[e]:
                continue
            kp, _, _ = o.kin_state(); o.set_action(o.pose_to_action(kp)); o.control_step(20, DT)
            if bool(out["episode_end"][e]) or o.is_episode_end():
                live[e] = False; continue
            ro = o.perturb_state()
            assert np.abs(rows[e][:3] - ro[:3]).max() < 1e-9 and active(rows[e]) == active(ro), (k, e)
            worst = max(worst, abs(float(out["reward"][e]) - o.calc_reward())); compared += 1
    assert compared > 60 and worst < 2e-3, (compared, worst)


def test_facade_updates_with_perturbs_follow_the_oracle(emu_lib, monkeypatch):
    """the reference's call protocol (one Update per 1/600 s) through the SWIG-shaped facade with --enable_rand_perturbs on"""
    import sys, os
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepmimic_amd", "compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from DeepMimicCore import DeepMimicCore as mod
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64")
    monkeypatch.setenv("DM_RNG", "counter")        # the oracle replays the counter-based streams (the reference's generators through the draw tape: tests/test_ref_draw_order.py)
    t = perturbed(model.load_asset("humanoid3d_walk"), tmin=0.03, tmax=0.06, dmin=0.01, dmax=0.05)
    core = mod.cDeepMimicCore(False)
    core.SeedRand(5); core.LoadTables(t, num_update_substeps=10); core.Init()
    st = core._env.get_state()
    o = Oracle(t); o.goal_rng(core._seed, 0); o.reset(float(st["clocks"][0][0]))
    o.set_perturb_state(core._env.get_perturb_state()[0])
    rng = np.random.default_rng(1)
    fired = 0
    for u in range(100):
        if core.NeedNewAction(0):
            assert np.abs(np.array(core.RecordState(0)) - o.record_state()).max() < 2e-5 and abs(core.CalcReward(0) - o.calc_reward()) < 1e-5
            a = (0.1 * rng.normal(size=o.A)).astype(np.float32)
            core.SetAction(0, [float(x) for x in a]); o.set_action(a.astype(np.float64))
        core.Update(DT); o.update(DT)
        assert core.IsEpisodeEnd() == o.is_episode_end() and core.CheckTerminate(0) == o.check_terminate()
        fired += o.num_perturbs() > 0
        if core.IsEpisodeEnd():
            break
    ro = o.perturb_state(); rd = core._env.get_perturb_state()[0]
    assert fired > 20 and np.abs(rd[:3] - ro[:3]).max() < 1e-9 and active(rd) == active(ro)


def test_perturbation_draws_are_keyed_by_global_env_id(emu_lib):
    """shard invariance: envs 2..3 of a 4-env batch and a 2-env shard with env_id_offset = 2 see the same pushes and the same trajectory"""
    t = perturbed(model.load_asset("humanoid3d_walk"))
    full = BatchEnv(t, 4, precision=64, lib_path=emu_lib, seed=12)
    shard = BatchEnv(t, 2, precision=64, lib_path=emu_lib, seed=12, env_id_offset=2)
    full.reset(); shard.reset()
    for _ in range(8):
        a = full.step(None, DT, 20, open_loop=True, auto_reset=True); b = shard.step(None, DT, 20, open_loop=True, auto_reset=True)
        assert np.array_equal(a["state"][2:], b["state"]) and np.array_equal(a["reward"][2:], b["reward"])
    assert np.array_equal(full.get_perturb_state()[2:], shard.get_perturb_state()) and full.get_perturb_state()[:, 2].max() > 7
