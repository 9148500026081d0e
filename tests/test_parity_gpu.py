"""Parity tests proper: the HIP library on a real MI355X, through the C-ABI, against the oracle."""
import numpy as np
import pytest

import parity_common as pc
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["humanoid3d_walk", "humanoid3d_spinkick", "dog3d_pace"])
def test_reset_query_fp64(hip_lib, name):
    pc.check_reset_and_query(name, 64, hip_lib, tol_state=1e-12, tol_reward=1e-6)


@pytest.mark.parametrize("name", ["humanoid3d_walk", "dog3d_pace"])
def test_reset_query_fp32(hip_lib, name):
    pc.check_reset_and_query(name, 32, hip_lib, tol_state=2e-6, tol_reward=1e-5)


@pytest.mark.parametrize("name,prec,rtol", [("humanoid3d_walk", 64, 1e-11), ("dog3d_pace", 64, 1e-11),
                                            ("humanoid3d_walk", 32, 2e-5), ("dog3d_pace", 32, 5e-5)])
def test_dynamics(hip_lib, name, prec, rtol):
    pc.check_dynamics(name, prec, hip_lib, rtol=rtol)


@pytest.mark.parametrize("name,prec,rtol", [("humanoid3d_walk", 64, 1e-9), ("dog3d_pace", 64, 1e-9),
                                            ("humanoid3d_walk", 32, 2e-3), ("dog3d_pace", 32, 5e-3)])
def test_spd(hip_lib, name, prec, rtol):
    pc.check_spd(name, prec, hip_lib, rtol=rtol)


@pytest.mark.parametrize("name,lift", [("humanoid3d_walk", 0.0), ("humanoid3d_walk", -0.03), ("humanoid3d_walk", -0.3), ("dog3d_pace", -0.02)])
def test_substep_fp64(hip_lib, name, lift):
    nc = pc.check_substep(name, 64, hip_lib, tol_vel=1e-8, tol_pose=1e-10, lift=lift, n=16)
    if lift < 0:
        assert nc > 0


def test_substep_fp32(hip_lib):
    pc.check_substep("humanoid3d_walk", 32, hip_lib, tol_vel=5e-3, tol_pose=1e-5, lift=-0.03, n=16)


@pytest.mark.parametrize("name", ["humanoid3d_walk", "dog3d_pace"])
def test_rollout_fp64_300_steps(hip_lib, name):
    """config 1 of BASELINE.json: 300 control steps, open-loop mocap tracking; same algorithm, same precision."""
    dr, ds, ok = pc.rollout_compare(name, 64, hip_lib, steps=300)
    assert ok
    assert dr.max() < 1e-5, dr.max()          # rewards cross the boundary as float32


# ---- The BASELINE.json parity metric, measured so that it means something ------------------------------
# All bounds below are FIXED numbers (no bound derived from the float build of the oracle).  They are the tolerances DESIGN.md
# section 7 states; measured values (profiles/r02_parity_report.json) are quoted next to each.
#
# "live" = the oracle's reward is non-zero, i.e. the character has not fallen: a fallen character has reward 0 on both sides
# by definition, which would dilute every mean.

LIVE_CASES = [("humanoid3d_walk", 64, 0), ("humanoid3d_walk", 32, 2), ("humanoid3d_walk", 32, 1),
              ("humanoid3d_spinkick", 64, 0), ("humanoid3d_spinkick", 32, 2), ("dog3d_pace", 64, 0), ("dog3d_pace", 32, 0)]


@pytest.mark.parametrize("name,prec,pack", LIVE_CASES)
def test_rollout_300_steps_live_through_resets(hip_lib, name, prec, pack):
    """config 1 of BASELINE.json made informative: 300 control steps x 8 envs, free-running open-loop tracking THROUGH
    auto-resets which the oracle mirrors draw for draw (falls, motion end and the finite episode timers of the dog / spinkick
    arg files -- row a3), so > 95 % of the 2400 transitions are live.  Asserted: every terminate / episode-end / valid flag,
    reward MAE over live steps (the north-star metric; fp32 < 1e-4, measured 5.4e-5 walk / 2.5e-6 spinkick / 1.1e-5 dog), the
    90th percentile, and the state vector (mean relative error).  The free-running MAXIMUM is printed, not asserted: two correct
    contact simulations separate chaotically around a fall (the fp64 build shows 6e-4 there); the per-step maximum is asserted
    by the teacher-forced test below."""
    dr, ds, alive, resets, ok = pc.auto_reset_rollout_compare(name, prec, hip_lib, steps=300, n=8, seed=11, wave_packing=pack)
    live = dr[alive]
    print("%s fp%d pack%d: live %d/%d, resets %d, reward MAE %.2e p90 %.2e max %.2e; state mean %.2e max %.2e"
          % (name, prec, pack, alive.sum(), dr.size, resets, live.mean(), np.quantile(live, 0.9), live.max(), ds.mean(), ds.max()))
    assert ok and resets >= 50
    assert alive.mean() > 0.95
    if prec == 64:
        assert live.mean() < 1e-5 and np.quantile(live, 0.9) < 1e-6 and ds.mean() < 2e-3
    else:
        # fixed bounds one notch above profiles/r03_parity_report.json (walk: MAE 5.4e-5, p99 1.1e-3, state mean 3.4e-3, p99 4.3e-2)
        assert live.mean() < 1e-4 and np.quantile(live, 0.9) < 1e-4 and np.quantile(live, 0.99) < 3e-3 and ds.mean() < 1e-2 and np.quantile(ds, 0.99) < 0.1


@pytest.mark.parametrize("name,prec,pack", LIVE_CASES)
def test_stepwise_300_steps_live(hip_lib, name, prec, pack):
    """teacher-forced: each of 300 x 8 control steps (20 updates, 40 substeps) starts from the oracle's state; live steps only.
    This is the per-step precision of the kernels.  Fixed bounds -- fp32 production kernels: reward MAE < 1e-5 (measured
    3.4e-6), 99th percentile < 1e-4 (3.2e-5), fewer than 1 % of steps beyond 1e-4 (0.56 %), maximum < 5e-3 (1.1e-3 walk,
    3.5e-3 dog: states at which one control step of the contact problem amplifies its input by 1e4 ... 5e6 -- rows and contacts agree
    at every update of every such step, profiles/r03_fp32_outlier_diagnosis.json; the fp64 build has the same kind of step at 2.7e-4); state vector: mean relative error < 5e-3, 99th
    percentile < 5e-2.  fp64 algorithm build: MAE < 1e-6, 99th percentile < 1e-6, maximum < 1e-3."""
    dr, ds, alive, ok = pc.stepwise_live_compare(name, prec, hip_lib, steps=300, n=8, seed=12, wave_packing=pack)
    live, sl = dr[alive], ds[alive]
    print("%s fp%d pack%d: live %d/%d, reward MAE %.2e p99 %.2e max %.2e n>1e-4 %d; state mean %.2e p99 %.2e max %.2e"
          % (name, prec, pack, alive.sum(), dr.size, live.mean(), np.quantile(live, 0.99), live.max(), (live > 1e-4).sum(),
             sl.mean(), np.quantile(sl, 0.99), sl.max()))
    assert ok and alive.mean() > 0.9
    if prec == 64:
        assert live.mean() < 1e-6 and np.quantile(live, 0.99) < 1e-6 and live.max() < 1e-3
        assert np.quantile(sl, 0.99) < 1e-6
    else:
        assert live.mean() < 1e-5 and np.quantile(live, 0.99) < 1e-4 and (live > 1e-4).mean() < 0.01 and live.max() < 5e-3
        assert sl.mean() < 5e-3 and np.quantile(sl, 0.99) < 5e-2


def test_rollout_spinkick_free_running_prefix(hip_lib):
    """spinkick is chaotic once the swinging foot scuffs the ground (control step ~23: |ankle omega| jumps to 37 rad/s);
    after that even the fp64 kernel and the fp64 oracle (different libm / FMA contraction) separate.  Free-running
    parity is therefore asserted on the prefix, and step-wise parity on the whole rollout below."""
    dr, ds, ok = pc.rollout_compare("humanoid3d_spinkick", 64, hip_lib, steps=20)
    assert ok and dr.max() < 1e-5
    dr, ds, ok = pc.rollout_compare("humanoid3d_spinkick", 32, hip_lib, steps=20)
    assert ok and dr.max() < 1e-4


def _golden_cases():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_rollouts.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: "%s@%g" % (c["scene"], c["t0"]))
def test_rollout_against_committed_golden_vectors(hip_lib, case):
    """HIP path vs the committed fixtures (tests/golden/oracle_rollouts.json), all envs of one batch at once."""
    t = model.load_asset(case["scene"])
    for prec, tol_r, tol_s in ((64, 1e-5, 1e-3), (32, 1e-3, 5e-2)):       # fixed bounds; rewards cross the boundary as float32
        env = BatchEnv(t, 3, precision=prec)
        env.reset(kin_times=[case["t0"]] * 3, max_times=np.inf)
        q = env.query()
        assert np.abs(q["state"][1] - np.array(case["state0"])).max() < (1e-6 if prec == 64 else 2e-5)
        errs = []
        for k in range(case["steps"]):
            out = env.step(None, pc.DT, 20, open_loop=True)
            errs.append(np.abs(out["reward"] - case["rewards"][k]).max())
        errs = np.array(errs)
        # fp32: MAE inside 1e-4, every step inside 1e-3 (the stiff self contacts at the end of the 30-step dog case: measured 3e-4)
        assert errs.max() < tol_r and errs.mean() < tol_r / 10, (prec, errs.mean(), errs.max(), int(np.argmax(errs)))
        assert np.abs(out["state"][2] - np.array(case["final_state"])).max() < tol_s
        assert int(out["terminate"][0]) == case["terminate"]


def test_batch_invariance_and_shard_offset(hip_lib):
    """env i's trajectory does not depend on the batch size or on which shard holds it."""
    t = model.load_asset("humanoid3d_walk")
    big = BatchEnv(t, 64, seed=7)
    lo = BatchEnv(t, 16, seed=7, env_id_offset=0)
    hi = BatchEnv(t, 16, seed=7, env_id_offset=48)
    for env in (big, lo, hi):
        env.reset()                       # per-env RNG keyed by the global env id
    for _ in range(5):
        ob, ol, oh = (e.step(None, pc.DT, 20, open_loop=True, auto_reset=True) for e in (big, lo, hi))
    assert np.array_equal(ob["state"][:16], ol["state"]) and np.array_equal(ob["reward"][:16], ol["reward"])
    assert np.array_equal(ob["state"][48:], oh["state"]) and np.array_equal(ob["reward"][48:], oh["reward"])


def test_auto_reset_round_trip_4096(hip_lib):
    """full-size property check (config 2): 4096 envs, auto-reset; outputs stay finite, rewards in [0,1], quaternions unit."""
    t = model.load_asset("humanoid3d_walk")
    env = BatchEnv(t, 4096, seed=3)
    env.reset()
    ends = 0
    for _ in range(40):
        out = env.step(None, pc.DT, 20, open_loop=True, auto_reset=True)
        assert np.isfinite(out["state"]).all() and np.isfinite(out["reward"]).all()
        assert (out["reward"] >= 0).all() and (out["reward"] <= 1 + 1e-6).all()
        ends += int(out["episode_end"].sum())
    st = env.get_state()
    assert abs(np.linalg.norm(st["pose"][:, 3:7], axis=1) - 1).max() < 1e-5
    assert ends > 0                      # open-loop tracking falls within ~1 s, so resets must have happened
    assert (st["flags"][:, 2] >= 1).all()


def test_facade_protocol_matches_oracle(hip_lib):
    """cDeepMimicCore call protocol (DeepMimic.py:62-80): NeedNewAction -> RecordState/CalcReward -> SetAction -> Update."""
    t = model.load_asset("humanoid3d_walk")
    env = BatchEnv(t, 1, precision=64)
    o = Oracle(t)
    env.reset(kin_times=[0.1], max_times=np.inf); o.reset(0.1)
    rng = np.random.default_rng(0)
    n_actions = 0
    for u in range(45):
        q = env.query()
        assert bool(q["need_new_action"][0]) == o.need_new_action()
        if o.need_new_action():
            a = (0.3 * rng.normal(size=o.A)).astype(np.float32)
            env.set_action(a[None]); o.set_action(a.astype(np.float64)); n_actions += 1
            assert abs(float(q["reward"][0]) - o.calc_reward()) < 1e-6
        env.update(pc.DT, 1); o.update(pc.DT)
    assert n_actions == 3
    p, v = o.sim_state(); st = env.get_state()
    assert np.abs(st["pose"][0] - p).max() < 1e-9 and np.abs(st["vel"][0] - v).max() < 1e-7


# ---- two characters per wavefront (dm_device_duo.h): same checks through the batch entry point
@pytest.mark.parametrize("prec,tol_r,tol_s", [(64, 1e-6, 1e-5), (32, 1e-3, 0.5)])
def test_duo_rollout_matches_oracle(hip_lib, prec, tol_r, tol_s):
    """10 free-running steps of six envs (three pairs); fp32: every env inside 1e-3, all but one inside 1e-4 (the env started at
    0.11 crosses a control step over which the contact problem amplifies its input by > 1e4: profiles/r03_fp32_outlier_diagnosis.json --
    that env sets the maxima, reward 1e-3 / state 0.5), mean inside 1e-4; the other five envs hold the state vector to 5e-2."""
    t0s = [0.0, 0.37, 0.8, 0.11, 0.5, 0.9]
    dr, ds, ok = pc.batch_rollout_compare("humanoid3d_walk", prec, hip_lib, steps=10, t0s=t0s, wave_packing=2)
    assert ok and dr.max() < tol_r and ds.max() < tol_s, (dr, ds)
    if prec == 32:
        assert np.sort(dr)[-2] < 1e-4 and dr.mean() < 1e-4, dr
        assert np.sort(ds)[-2] < 5e-2, ds


def test_duo_heavy_contact_fallback(hip_lib):
    dr, ds, ok = pc.batch_rollout_compare("humanoid3d_walk", 64, hip_lib, steps=2, t0s=[0.0, 0.4, 0.2, 0.6], wave_packing=2,
                                          lifts=[-0.3, 0.0, 0.0, -0.25])
    assert dr.max() < 1e-6 and ds.max() < 1e-4


def test_duo_heavy_contact_fallback_fp32(hip_lib):
    """fp32: a character pushed 0.3 m into the ground needs > 32 rows, the pair falls back to the one-per-wave routine (narrow class by
    default; with -DDM_DUO_WIDE_FALLBACK=1 the wide class whose 64-row Gram matrix runs on the matrix core, wave_gram64).  One control step of a deep-penetration state is
    stiff (push-out velocities of tens of m/s), so the check is against the one-character-per-wave kernel (readlane Gram, HBM overflow
    rows: a different code path for the same arithmetic) and, loosely, against the fp64 oracle."""
    t0s, lifts = [0.0, 0.4, 0.2, 0.6], [-0.3, 0.0, 0.0, -0.25]
    dr2, ds2, _ = pc.batch_rollout_compare("humanoid3d_walk", 32, hip_lib, steps=1, t0s=t0s, wave_packing=2, lifts=lifts)
    dr1, ds1, _ = pc.batch_rollout_compare("humanoid3d_walk", 32, hip_lib, steps=1, t0s=t0s, wave_packing=1, lifts=lifts)
    assert dr2.max() < 1e-5 and dr1.max() < 1e-5, (dr2, dr1)           # measured 1e-7 / 4e-7
    assert ds2.max() < 1e-2 and ds1.max() < 1e-2, (ds2, ds1)           # measured 2e-3 / 3e-3 (velocities of a 0.3 m push-out)


@pytest.mark.parametrize("prec,tol_r,tol_s", [(64, 1e-6, 1e-5), (32, 1e-5, 1e-2)])
def test_duo_borrowed_lanes(hip_lib, prec, tol_r, tol_s):
    """a character with 33..48 constraint rows beside a light partner (pressed 8-10 cm into the ground): the pair stays on the two-per-wave path, the heavy character's
    rows 32.. on lanes borrowed from the partner's half (DuoSim lane borrowing, round 6: four MFMA Gram blocks, a sweep of up to 48 visits, Y in the overflow block
    meanwhile) -- heavy character in the lower half (pair 0) and in the upper half (pair 1); fp32 bounds as test_duo_heavy_contact_fallback_fp32"""
    for lifts in ([-0.08, 0.0, 0.0, -0.08], [-0.1, 0.0, 0.0, -0.1]):
        st = {}
        dr, ds, ok = pc.batch_rollout_compare("humanoid3d_walk", prec, hip_lib, steps=2, t0s=[0.0, 0.4, 0.2, 0.6], wave_packing=2, lifts=lifts, stats=st)
        print("fp%d lifts %s: borrowed %s fallback %s, reward diff %.2e state diff %.2e" % (prec, lifts, st["borrowed"], st["fallback"], dr.max(), ds.max()))
        assert (st["borrowed"] > 0).all(), st
        assert ok and dr.max() < tol_r and ds.max() < tol_s, (dr, ds)


def test_duo_spinkick_and_300_steps(hip_lib):
    dr, ds, ok = pc.batch_rollout_compare("humanoid3d_spinkick", 64, hip_lib, steps=20, t0s=[0.0, 0.3], wave_packing=2)
    assert ok and dr.max() < 1e-5
    dr, ds, ok = pc.batch_rollout_compare("humanoid3d_walk", 64, hip_lib, steps=300, t0s=[0.0, 0.21], wave_packing=2)
    assert ok and dr.max() < 1e-5


def test_duo_auto_reset_4096(hip_lib):
    t = model.load_asset("humanoid3d_walk")
    a = BatchEnv(t, 4096, seed=3, wave_packing=2); b = BatchEnv(t, 4096, seed=3, wave_packing=1)
    a.reset(); b.reset()
    ends = 0
    for _ in range(30):
        oa = a.step(None, pc.DT, 20, open_loop=True, auto_reset=True); ob = b.step(None, pc.DT, 20, open_loop=True, auto_reset=True)
        assert np.isfinite(oa["state"]).all() and (oa["reward"] >= 0).all() and (oa["reward"] <= 1 + 1e-6).all()
        ends += int(oa["episode_end"].sum())
    assert ends > 0
    # same physics, different summation order: episode statistics of the two packings agree
    assert abs(float(oa["reward"].mean()) - float(ob["reward"].mean())) < 0.05


def test_duo_fp32_300_steps_reward_tolerance(hip_lib):
    """the headline configuration of bench.py (wave_packing = 2, fp32), the literal config-1 rollout: 300 steps from t0 = 0, no
    reset.  Only the steps before the fall are live (the count is printed); reward and state vector are asserted on those."""
    t = model.load_asset("humanoid3d_walk")
    env = BatchEnv(t, 2, precision=32, wave_packing=2)
    env.reset(kin_times=[0.0, 0.0], max_times=np.inf)
    o = Oracle(t); o.reset(0.0)
    dr, ds, alive = [], [], []
    for k in range(300):
        out = env.step(None, pc.DT, 20, open_loop=True)
        kp, _, _ = o.kin_state(); o.set_action(o.pose_to_action(kp))
        for u in range(20):
            o.update(pc.DT)
        r = o.calc_reward(); so = o.record_state()
        dr.append(max(abs(float(out["reward"][e]) - r) for e in range(2)))
        ds.append(max(np.abs(out["state"][e] - so).max() for e in range(2)) / max(1.0, np.abs(so).max()))
        alive.append(r != 0.0)
        assert int(out["terminate"][0]) == o.check_terminate() == int(out["terminate"][1])
    dr, ds, alive = np.array(dr), np.array(ds), np.array(alive)
    print("live steps %d / 300: reward MAE %.2e max %.2e, state max %.2e; all 300: MAE %.2e" % (alive.sum(), dr[alive].mean(), dr[alive].max(), ds[alive].max(), dr.mean()))
    assert 20 <= alive.sum() <= 40                     # the open-loop walker falls after about one second
    assert dr[alive].mean() < 1e-4 and dr[alive].max() < 1e-3 and ds[alive].max() < 5e-2
    assert dr[~alive].max() < 1e-6                     # both sides report 0 for a fallen character


@pytest.mark.parametrize("prec,tol", [(64, 1e-5), (32, 1e-4)])
def test_root_heading_sync_dog_spin(hip_lib, prec, tol):
    """sync_char_root_rot = true (args/run_dog3d_spin_args.txt): 60 control steps = three phase wraps; the dog's tail / thigh and
    paw pairs are in permanent self contact on this clip"""
    if prec == 64:
        pc.check_reset_and_query("dog3d_spin", 64, hip_lib, tol_state=1e-12, tol_reward=1e-6)
    dr, ds, ok = pc.rollout_compare("dog3d_spin", prec, hip_lib, steps=60)
    print("dog3d_spin fp%d: reward MAE %.2e max %.2e, state max %.2e" % (prec, dr.mean(), dr.max(), ds.max()))
    assert ok and dr.mean() < tol / 5 and dr.max() < 10 * tol, (dr.mean(), dr.max())      # fixed: fp32 mean < 2e-5, max < 1e-3


@pytest.mark.parametrize("pack", [1, 2])
def test_action_stream_a0_collapse_and_fall(hip_lib, pack):
    """stream A0 (all-zero actions, the reference's own native driver input, Main.cpp:119-120): PD targets are identity
    rotations, the character collapses; fall termination must fire on the same control step as in the oracle.  The
    collapse is a chaotic many-contact motion (fp64 kernel and fp64 oracle differ in libm / FMA contraction), so the
    reward is held tight on the 15-step prefix and to 1e-3 over the whole fall; the flags must agree throughout."""
    dr, ds, ok, fallen = pc.action_rollout_compare("humanoid3d_walk", 64, hip_lib, 60, "A0", [0.0, 0.37, 0.6, 0.9], wave_packing=pack)
    assert fallen == 4, fallen
    assert ok and dr.max() < 1e-3, (dr, ds)
    dr, ds, ok, _ = pc.action_rollout_compare("humanoid3d_walk", 64, hip_lib, 15, "A0", [0.0, 0.37, 0.6, 0.9], wave_packing=pack)
    assert ok and dr.max() < 1e-5, (dr, ds)
    dr, ds, ok, _ = pc.action_rollout_compare("humanoid3d_walk", 32, hip_lib, 12, "A0", [0.0, 0.37], wave_packing=pack)
    assert ok and dr.max() < 1e-4, (dr, ds)


@pytest.mark.parametrize("name,pack", [("humanoid3d_walk", 1), ("humanoid3d_walk", 2), ("dog3d_pace", 0)])
def test_action_stream_a2_noisy_tracking(hip_lib, name, pack):
    """stream A2: mocap tracking + Philox N(0, 0.05^2) exploration noise, explicit float32 actions through the a10 path."""
    dr, ds, ok, _ = pc.action_rollout_compare(name, 64, hip_lib, 100, "A2", [0.0, 0.41, 0.77, 1.3], wave_packing=pack)
    assert ok and dr.max() < 1e-5, (dr, ds)
    dr, ds, ok, _ = pc.action_rollout_compare(name, 32, hip_lib, 10, "A2", [0.0, 0.41], wave_packing=pack)
    assert ok and dr.max() < 1e-4, (dr, ds)


def test_cabi_record_gather_through_rccl_one_rank(hip_lib):
    """The C-ABI collective (dm_comm_create with a unique id -> ncclCommInitRank, dm_gather_records -> ncclAllGather on the
    comm's own stream, ordered by HIP events) on the real RCCL with one rank; the step of control step k+1 is enqueued while
    the gather of step k is in flight."""
    import torch
    from deepmimic_amd.dist import CabiRecordExchange
    t = model.load_asset("humanoid3d_walk")
    n = 256
    env = BatchEnv(t, n, seed=5)
    env.reset()
    dev = torch.device("cuda", 0)
    ex = CabiRecordExchange(env, 1, 0, dev, depth=2, force_rccl=True)
    valid = torch.zeros(n, dtype=torch.int32, device=dev); ends = torch.zeros(n, dtype=torch.int32, device=dev)
    keep = []
    for k in range(6):
        slot = k & 1
        st, rw, tm = ex.begin(slot)
        env.step_device(0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), valid.data_ptr(), ends.data_ptr(), auto_reset=True, open_loop=True)
        ex.launch(slot)
        if k >= 1:
            S_, R_, T_ = ex.result((k - 1) & 1)
            keep.append((S_[0].clone(), R_[0].clone()))
    env.synchronize(); torch.cuda.synchronize()
    # reference: the same rollout without the exchange
    env2 = BatchEnv(t, n, seed=5)
    env2.reset()
    for k in range(5):
        out = env2.step(None, pc.DT, 20, open_loop=True, auto_reset=True)
        assert np.array_equal(keep[k][0].cpu().numpy(), out["state"]) and np.array_equal(keep[k][1].cpu().numpy(), out["reward"])
