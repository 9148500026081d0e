"""Known-answer tests that pin the CPU oracle (the reference has no tests).

Closed-form facts the reference's formulas imply (SURVEY 8c i-vi), the reference's shipped data files, and the committed
golden vectors under tests/golden/ (regression pins of the oracle itself, see make_golden.py)."""
import json
import os

import numpy as np
import pytest

from deepmimic_amd import model
from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def hum(oracle_built):
    t = model.load_asset("humanoid3d_walk")
    return t, Oracle(t)


@pytest.fixture(scope="module")
def dog(oracle_built):
    t = model.load_asset("dog3d_pace")
    return t, Oracle(t)


def test_dims_match_reference_checkpoints(hum, dog):
    """humanoid 15/43/28/227 and dog 23/83/58/347: shapes of the shipped policy checkpoints."""
    t, o = hum
    assert (o.J, o.P, o.A, o.S) == (15, 43, 28, 227) and t.state_dim == 227 and t.action_dim == 28
    t, o = dog
    assert (o.J, o.P, o.A, o.S) == (23, 83, 58, 347)


def test_total_mass_and_translational_block(hum, dog):
    for (t, o), mass in ((hum, 45.0), (dog, 29.25)):
        assert abs(t.body_defs[:, model.BD_MASS].sum() - mass) < 1e-12
        o.reset(0.3)
        p, v = o.sim_state()
        for which in (0, 1):
            H, _ = o.mass_bias(which, p, v)
            assert np.abs(H[:3, :3] - mass * np.eye(3)).max() < 1e-9          # CRBA: root translational block = m_total * I
            assert np.abs(H - H.T).max() < 1e-9
            live = np.abs(np.diag(H)) > 0                                       # dead 4th slots of the quaternions are exact zeros
            assert live.sum() == (34 if o.J == 15 else 64)
            assert np.linalg.eigvalsh(H[np.ix_(live, live)]).min() > 0


def test_reward_is_one_when_sim_equals_kin(hum, dog):
    for t, o in (hum, dog):
        for t0 in (0.0, 0.4, 1.7):
            o.reset(t0)
            r, terms = o.calc_reward_terms()
            lift = o.kin_state()[2][1]         # ResolveCharGroundIntersect raised both characters by `lift`; the kin heights are
            n_ee = int(t.joint_mat[:, model.JD_IS_EE].sum())      # measured from the kin origin (SceneImitate.cpp:58-66), the sim's from the ground
            assert np.abs(terms[[0, 1, 4]]).max() < 1e-9
            assert abs(terms[2] - n_ee * lift ** 2) < 1e-9 and abs(terms[3] - lift ** 2) < 1e-9
            if lift == 0:
                assert abs(r - 1.0) < 1e-9
            assert o.check_terminate() == 0 and o.check_valid_episode()


def test_record_state_layout_of_reset_pose(hum):
    t, o = hum
    o.reset(0.5)
    s = o.record_state()
    p, _ = o.sim_state()
    assert abs(s[0] - (0.5 / o.duration) % 1.0) < 1e-12        # phase (EnablePhaseInput, CtController.cpp:473-478)
    assert abs(s[1] - p[1]) < 1e-12                             # root height above the ground
    # link positions are relative to the root in the heading frame: the root link's own entry is its body offset
    assert np.abs(s[2:5]).max() < 0.2
    # unit normal / tangent columns of every link frame
    for j in range(o.J):
        n, tg = s[2 + 9 * j + 3:2 + 9 * j + 6], s[2 + 9 * j + 6:2 + 9 * j + 9]
        assert abs(np.linalg.norm(n) - 1) < 1e-9 and abs(np.linalg.norm(tg) - 1) < 1e-9 and abs(n @ tg) < 1e-9


def test_free_fall_com_acceleration(hum, dog):
    """No ground contact, no torque (the dog's reference poses do carry self contacts: internal forces, which must not move
    the COM).  From rest the COM gains exactly g*h in one substep; in motion the total linear momentum
    (COM velocity at the new pose and velocity) changes by m*g*h up to the O(h^2) error of the semi-implicit step."""
    h = 1.0 / 1200
    g = np.array([0, -9.8 * h, 0])
    for t, o in (hum, (dog[0], Oracle(dog[0], self_collision=0))):
        o.reset(0.2)
        p, v = o.sim_state(); p[1] += 2.0
        rng = np.random.default_rng(0); v = v + 0.5 * rng.normal(size=v.shape) * (v != 0)
        for vv, tol in ((np.zeros_like(v), 1e-15), (v, 5e-5)):
            o.set_sim_state(p, vv); o.set_tau(np.zeros(o.P))
            _, vc0 = o.calc_com(p, vv)
            o.substep(h)
            assert o.num_contacts() == 0
            p1, v1 = o.sim_state()
            _, vc1 = o.calc_com(p1, v1)
            assert np.abs((vc1 - vc0) - g).max() < tol


def test_self_contact_forces_are_internal(oracle_built):
    """Self contacts (tail against the rear thighs in the dog's reference pose) act with equal and opposite impulses: the
    COM of an airborne character still gains exactly g*h.  (erp is lowered so that the 2 cm overlap of the mocap pose does
    not drive a joint into the 100 rad/s coordinate-velocity clamp, which is the one thing that may leak momentum.)"""
    h = 1.0 / 1200
    t = model.load_asset("dog3d_pace")
    o = Oracle(t, erp=0.001)
    o.reset(0.2)
    p, v = o.sim_state(); p[1] += 2.0; v[:] = 0
    o.set_sim_state(p, v); o.set_tau(np.zeros(o.P))
    _, vc0 = o.calc_com(p, v)
    o.substep(h)
    assert o.num_self_contacts() >= 1 and o.num_contacts() == o.num_self_contacts()
    cl = o.contact_list()
    assert set((int(c[0]), int(c[1])) for c in cl) <= {(13, 21), (17, 21)} and (cl[:, 2] < 0).all()
    assert np.abs(np.linalg.norm(cl[:, 6:9], axis=1) - 1).max() < 1e-12
    p1, v1 = o.sim_state()
    assert np.abs(v1).max() < 50
    _, vc1 = o.calc_com(p, v1)                 # momentum at the substep's own pose: isolates the velocity update
    assert np.abs((vc1 - vc0) - np.array([0, -9.8 * h, 0])).max() < 1e-12
    # and with self collision switched off nothing touches
    o2 = Oracle(t, self_collision=0)
    o2.reset(0.2); o2.set_sim_state(p, v); o2.set_tau(np.zeros(o2.P)); o2.substep(h)
    assert o2.num_contacts() == 0


def test_mirror_symmetry_of_kinetic_energy(hum):
    """Reflecting the humanoid across its sagittal plane (swap left/right limbs, z -> -z) preserves q'^T H q' / 2."""
    t, o = hum
    o.reset(0.3)
    p, v = o.sim_state()
    rng = np.random.default_rng(3)
    v = v + rng.normal(size=v.shape) * (v != 0)
    off = lambda j: int(t.joint_mat[j, model.JD_PARAM_OFFSET])
    pm, vm = p.copy(), v.copy()
    mq = lambda q: np.array([q[0], -q[1], -q[2], q[3]])            # rotation conjugated by diag(1,1,-1)
    mw = lambda w: np.array([-w[0], -w[1], w[2]])                   # pseudo-vector
    pm[0:3] = [p[0], p[1], -p[2]]; pm[3:7] = mq(p[3:7]); vm[0:3] = [v[0], v[1], -v[2]]; vm[3:6] = mw(v[3:6])
    swap = {3: 9, 4: 10, 5: 11, 6: 12, 7: 13, 9: 3, 10: 4, 11: 5, 12: 6, 13: 7, 1: 1, 2: 2}
    for j, k in swap.items():
        ty = int(t.joint_mat[j, model.JD_TYPE])
        if ty == model.JT_SPHERICAL:
            pm[off(k):off(k) + 4] = mq(p[off(j):off(j) + 4]); vm[off(k):off(k) + 3] = mw(v[off(j):off(j) + 3])
        elif ty == model.JT_REVOLUTE:
            pm[off(k)] = p[off(j)]; vm[off(k)] = v[off(j)]
    H, _ = o.mass_bias(1, p, v); Hm, _ = o.mass_bias(1, pm, vm)
    assert abs(v @ H @ v - vm @ Hm @ vm) < 1e-9 * abs(v @ H @ v)


def test_action_exp_map_round_trip(hum):
    t, o = hum
    rng = np.random.default_rng(1)
    for _ in range(5):
        a = rng.uniform(-1.0, 1.0, size=o.A)
        o.set_action(a)
        assert np.abs(o.pose_to_action(o.tar_pose()) - a).max() < 1e-9
    q = o.tar_pose()
    for j in range(1, o.J):
        if int(t.joint_mat[j, model.JD_TYPE]) == model.JT_SPHERICAL:
            k = int(t.joint_mat[j, model.JD_PARAM_OFFSET])
            assert abs(np.linalg.norm(q[k:k + 4]) - 1) < 1e-12


def test_kin_sampling_hits_the_key_frames(hum):
    """cMotion::CalcFrame at a frame's start time returns that frame (joint part; the root is moved by the origin)."""
    t, o = hum
    o.reset(0.0)
    for f in (0, 5, 17, o.F - 2):
        fr, fv, ft = o.motion_frame(f)
        kp, kv = o.kin_eval(ft + 1e-12)
        assert np.abs(kp[7:] - fr[7:]).max() < 1e-9
    assert abs(o.motion_frame(o.F - 1)[2] - o.duration) < 1e-12


def test_control_latch_every_twentieth_update(hum):
    t, o = hum
    o.reset(0.123)
    fired = []
    for u in range(61):
        if o.need_new_action():
            fired.append(u); o.set_action(np.zeros(o.A))
        o.update(1.0 / 600)
    assert fired == [0, 20, 40, 60]


@pytest.mark.parametrize("case", json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_rollouts.json"))), ids=lambda c: "%s@%g" % (c["scene"], c["t0"]))
def test_oracle_matches_committed_golden_vectors(oracle_built, case):
    t = model.load_asset(case["scene"])
    o = Oracle(t)
    assert [o.J, o.P, o.A, o.S, o.F] == case["dims"] and abs(o.duration - case["duration"]) < 1e-12
    o.reset(case["t0"])
    assert np.abs(o.record_state() - np.array(case["state0"])).max() < 1e-9
    for k in range(case["steps"]):
        kp, _, _ = o.kin_state()
        o.set_action(o.pose_to_action(kp))
        for u in range(20):
            o.update(1.0 / 600)
        assert abs(o.calc_reward() - case["rewards"][k]) < 1e-7
        assert int(o.contacts().sum()) == case["links_in_contact"][k]
    p, v = o.sim_state()
    assert np.abs(p - np.array(case["final_pose"])).max() < 1e-6 and np.abs(o.record_state() - np.array(case["final_state"])).max() < 1e-5


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "data")), reason="reference checkout not present")
@pytest.mark.parametrize("name,argfile", [("humanoid3d_walk", "args/run_humanoid3d_walk_args.txt"),
                                          ("humanoid3d_spinkick", "args/run_humanoid3d_spinkick_args.txt"),
                                          ("dog3d_pace", "args/run_dog3d_pace_args.txt"),
                                          ("dog3d_spin", "args/run_dog3d_spin_args.txt")])
def test_compiled_assets_equal_the_reference_data_files(name, argfile):
    """deepmimic_amd/assets/*.json are byte-for-value compilations of the reference's data/ + args/ files."""
    a = model.load_asset(name)
    b = model.load_scene_from_args(["--arg_file", argfile], data_root=REF)
    assert np.array_equal(a.joint_mat, b.joint_mat) and np.array_equal(a.body_defs, b.body_defs)
    assert np.array_equal(a.pd_params, b.pd_params) and np.array_equal(a.frames, b.frames) and a.loop == b.loop
    assert np.array_equal(a.fall_mask(), b.fall_mask()) and a.cfg.num_sim_substeps == b.cfg.num_sim_substeps
    assert a.enable_phase_input == b.enable_phase_input and a.record_world_root_rot == b.record_world_root_rot
